"""Micro-benchmark of the GEMM entry points on the shapes of the configs[3] step (H = 512, 2048 trials x 20 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF

def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

dev = 'cuda'
if os.environ.get('ZERO') == '1':      # data-dependent power: the same kernels on all-zero operands
    _randn = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **k)
R = 40960
for name, M, N, K in [('nt sq', 4096, 4096, 4096), ('nt proj L2', R, 1536, 1024), ('nt proj L1', R, 1536, 100), ('nt 8k', 8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    t = timeit(lambda: XF.gemm_nt(A, B, C, M, N, K))
    print(f'{name:12s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF  ({6*M*N*K/t/1e12/2500:.3f} of bf16 peak)')
for name, M, N, K in [('nn dx L2', R, 1024, 1536), ('nn sq', 4096, 4096, 4096)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    t = timeit(lambda: XF.gemm_nn(A, B, C, M, N, K))
    print(f'{name:12s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF  ({6*M*N*K/t/1e12/2500:.3f} of bf16 peak)')
for name, M, N, K in [('tn dWih L2', 1536, 1024, R), ('tn dWhh', 1536, 512, R), ('tn sq', 4096, 4096, 4096)]:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    bias = torch.empty(M, device=dev)
    t = timeit(lambda: XF.gemm_tn_grouped([XF.tn_problem(A, B, C, M, N, K, colsum_out=bias)], dev))
    print(f'{name:12s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF  ({6*M*N*K/t/1e12/2500:.3f} of bf16 peak)')
