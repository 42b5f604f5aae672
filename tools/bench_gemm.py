"""Micro-benchmark of the fp32 MFMA GEMM entry points on the shapes of the training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF

def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

dev = 'cuda'
for name, M, N, K in [('nt sq', 4096, 4096, 4096), ('nt proj L1', 40960, 384, 256), ('nt proj L0', 40960, 384, 100), ('nt conv', 40960, 100, 640)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    t = timeit(lambda: XF.gemm_nt(A, B, C, M, N, K))
    print(f'{name:12s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF')
for name, M, N, K in [('nn dx L1', 40960, 256, 384), ('nn dx L0', 40960, 100, 384)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    t = timeit(lambda: XF.gemm_nn(A, B, C, M, N, K))
    print(f'{name:12s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF')
import ctypes as C
from cross_patient_speech_decoding_amd._lib import call, rowmap
from cross_patient_speech_decoding_amd.nn_models.functional import _ptr_array, _ptr, _stream
for name, M, N, K in [('nt-multi proj L1 x2', 40960, 384, 256), ('nt-multi proj L0 x2', 40960, 384, 100)]:
    A = torch.randn(M, K, device=dev); Bs = [torch.randn(N, K, device=dev) for _ in range(2)]
    bs = [torch.randn(N, device=dev) for _ in range(2)]; Cs = [torch.empty(M, N, device=dev) for _ in range(2)]
    ra, rb, rc = rowmap(K), rowmap(K), rowmap(N)
    t = timeit(lambda: call('xps_gemm_nt_multi_f32', _ptr(A), C.byref(ra), _ptr_array(Bs), C.byref(rb), _ptr_array(Cs), C.byref(rc), _ptr_array(bs), 2, M, N, K, _stream()))
    print(f'{name:20s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {4*M*N*K/t/1e12:6.1f} TF')
for name, M, N, K in [('nn2 dx L1', 40960, 256, 384), ('nn2 dx L0', 40960, 100, 384)]:
    A1 = torch.randn(M, K, device=dev); A2 = torch.randn(M, K, device=dev)
    B1 = torch.randn(K, N, device=dev); B2 = torch.randn(K, N, device=dev); Cc = torch.empty(M, N, device=dev)
    ra, rb, rc = rowmap(K), rowmap(N), rowmap(N)
    t = timeit(lambda: call('xps_gemm_nn2_f32', _ptr(A1), _ptr(B1), K, _ptr(A2), _ptr(B2), K, C.byref(ra), C.byref(rb), _ptr(Cc), C.byref(rc), M, N, 0, _stream()))
    print(f'{name:20s} M={M} N={N} K=2x{K}: {t*1e6:8.1f} us  {4*M*N*K/t/1e12:6.1f} TF')
for name, M, N, K in [('tn dWih L1', 384, 256, 40960), ('tn dWih L0', 384, 100, 40960), ('tn dWhh', 384, 128, 40960), ('tn conv', 100, 640, 40960)]:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    t = timeit(lambda: XF.gemm_tn(A, B, C, M, N, K))
    print(f'{name:12s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF (single, split-K + reduce)')
    bias = torch.empty(M, device=dev)
    t = timeit(lambda: XF.gemm_tn_grouped([XF.tn_problem(A, B, C, M, N, K, colsum_out=bias)], dev))
    print(f'{name:12s}   grouped(1 problem, +colsum): {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF')
