"""Conv weight-gradient launch of the configs[3] / configs[1] step in isolation (grouped TN GEMM over window rows of x)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF
dev = 'cuda'
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, B, T, Cin, F, k, stride in [('configs3', 2048, 200, 30, 100, 10, 10), ('configs1', 2048, 200, 64, 100, 10, 10)]:
    Tp = (T - k) // stride + 1
    rows = Tp * B
    x = torch.randn(B, T, Cin, device=dev)
    dy = torch.randn(rows, F, device=dev)
    dw2 = torch.empty(F, k * Cin, device=dev)
    db = torch.empty(F, device=dev)
    prob = [XF.tn_problem(dy, x, dw2, F, k * Cin, rows, ra=XF.rowmap(F), rb=XF.rowmap(T * Cin, rpg=B, gs=stride * Cin), colsum_out=db)]
    us = timeit(lambda: XF.gemm_tn_grouped(prob, dev))
    by = (x.numel() + dy.numel()) * 4
    print(f'{name}: conv wgrad M={F} N={k * Cin} K={rows}: {us:7.1f} us  ({by / us / 1e3:.0f} GB/s of operands)')
    # the same contraction on a plain copy of the window matrix (no row map)
    win = x.unfold(1, k, stride).permute(1, 0, 3, 2).reshape(rows, k * Cin).contiguous()
    prob2 = [XF.tn_problem(dy, win, dw2, F, k * Cin, rows, colsum_out=db)]
    us2 = timeit(lambda: XF.gemm_tn_grouped(prob2, dev))
    print(f'{name}: ... on a materialised window matrix: {us2:7.1f} us')
