"""Prototype check: bf16 split-product GEMM vs the fp32-MFMA product kernel (accuracy vs float64, time)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libproto.so'))
vp = ctypes.c_void_p
lib.proto_gemm_nt_bf16split.argtypes = [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    torch.manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in [(256, 128, 64), (40960, 768, 256), (40960, 768, 128), (40960, 256, 768), (4096, 4096, 4096)]:
        A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda') * 0.1
        bias = torch.randn(N, device='cuda')
        ref = (A.double() @ B.double().T + bias.double())
        C32 = torch.empty(M, N, device='cuda')
        XF.gemm_nt(A, B, C32, M, N, K, bias=bias)
        e32 = (C32.double() - ref).abs().max().item()
        t32 = timeit(lambda: XF.gemm_nt(A, B, C32, M, N, K, bias=bias))
        line = f'M={M} N={N} K={K}: fp32-mfma {t32:7.1f} us {2*M*N*K/t32/1e6:6.1f} TF err {e32:.2e}'
        for nprod, bk in ((3, 16), (3, 32), (3, 64), (1, 32)):
            C = torch.zeros(M, N, device='cuda')
            rc = lib.proto_gemm_nt_bf16split(A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr(), M, N, K, nprod, bk, st)
            assert rc == 0, rc
            torch.cuda.synchronize()
            err = (C.double() - ref).abs().max().item()
            t = timeit(lambda: lib.proto_gemm_nt_bf16split(A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr(), M, N, K, nprod, bk, st))
            line += f' | x{nprod}/bk{bk} {t:7.1f} us {2*M*N*K/t/1e6:6.1f} TF err {err:.2e}'
        print(line, f'(|ref| max {ref.abs().max().item():.1f})', flush=True)


main()
