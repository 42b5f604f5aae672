"""Diagnostic (-DXPS_GSTAMP build): per-wave cycle shares of the bf16 split-product k loop on the HBM-bound shapes of cfg 2.
build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DXPS_GSTAMP -o tools/libxps_gstamp.so cross_patient_speech_decoding_amd/csrc/*.hip"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['XPS_LIB_OVERRIDE'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libxps_gstamp.so')
import numpy as np, torch
from cross_patient_speech_decoding_amd import _lib
from cross_patient_speech_decoding_amd.nn_models import functional as XF
l = _lib.lib(); l.xps_debug_read_gstamps.argtypes = [C.c_void_p, C.c_int]
def report(name, nblocks, nkt, t_us):
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8192 * 4))(); l.xps_debug_read_gstamps(buf, 8192 * 4)
    nb = min(8192, nblocks * 4)
    a = np.array(buf[:], dtype=np.float64).reshape(8192, 4)[:nb] / nkt
    tot = np.median(a.sum(1)) * nkt
    print(f'{name}: {t_us:.1f} us; per k-tile cycles (median over waves): load-issue {np.median(a[:,0]):.0f}  frags+mfma {np.median(a[:,1]):.0f}  split+lds-store {np.median(a[:,2]):.0f}  barrier {np.median(a[:,3]):.0f}  | k loop per block {tot:.0f} cycles x {nkt} tiles (pure MFMA per k-tile: 12 x 32 = 384 (128-row) / 192 (64-row))')
def ev(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100
for name, M, N, K in [('nt proj L2 (one direction)', 40960, 384, 256), ('nt conv', 40960, 100, 640), ('nt sq 4096x4096x1024', 4096, 4096, 1024)]:
    A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); Cc = torch.empty(M, N, device='cuda')
    t = ev(lambda: XF.gemm_nt(A, B, Cc, M, N, K))
    mi = 64 if ((M + 127) // 128) * ((N + 127) // 128) < 2048 else 128
    report(name, ((M + mi - 1) // mi) * ((N + 127) // 128), K // 16, t)
for name, M, N, K in [('nn dx L2 (one direction)', 40960, 256, 384)]:
    A = torch.randn(M, K, device='cuda'); B = torch.randn(K, N, device='cuda'); Cc = torch.empty(M, N, device='cuda')
    t = ev(lambda: XF.gemm_nn(A, B, Cc, M, N, K))
    report(name, ((M + 63) // 64) * ((N + 127) // 128), K // 16, t)
