"""The weight-gradient group of configs[3]'s encoder layer 1 as the step launches it (six problems in ONE grouped launch + reduce:
per direction dW_hh r,z rows / dW_hh n rows against h_prev, dW_ih against the layer input), A/B in one process:
register-staged k loop (XPS_GEMM_DMA=0) against the LDS-DMA k loop; then the problems one by one.
XPS_HPREV_SPLIT=1: h_prev handed over as an XPS_FMT_SPLIT4 copy too (what a split copy of y would buy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd._lib import rowmap
from cross_patient_speech_decoding_amd.nn_models import functional as XF

T, B, H, ndir = 20, 2048, 512, 2
In = 2 * H
K = T * B
dev = 'cuda'
torch.manual_seed(0)
dgi = [XF.split4(torch.randn(K, 3 * H, device=dev) * 0.1) for _ in range(ndir)]
dghn = [XF.split4(torch.randn(K, H, device=dev) * 0.1) for _ in range(ndir)]
x = XF.split4(torch.randn(K, In, device=dev))
y_ext = torch.randn((T + 2) * B, ndir * H, device=dev)
y_s = XF.split4(y_ext)
hs = os.environ.get('XPS_HPREV_SPLIT') == '1'


def problems(which=None):
    probs = []
    for d in range(ndir):
        first = 0 if d == 0 else 2
        hp = (y_s if hs else y_ext).view(-1)[first * B * ndir * H + d * H:]
        dw = torch.empty(3 * H, H, device=dev); db = torch.empty(3 * H, device=dev)
        probs.append(('dW_hh rz', XF.tn_problem(dgi[d], hp, dw, 2 * H, H, K, ra=rowmap(3 * H, fmt=1), rb=rowmap(ndir * H, fmt=int(hs)), rc=rowmap(H), colsum_out=db)))
        probs.append(('dW_hh n', XF.tn_problem(dghn[d], hp, dw[2 * H:], H, H, K, ra=rowmap(H, fmt=1), rb=rowmap(ndir * H, fmt=int(hs)), rc=rowmap(H), colsum_out=db[2 * H:])))
    for d in range(ndir):
        dw = torch.empty(3 * H, In, device=dev); db = torch.empty(3 * H, device=dev)
        probs.append(('dW_ih', XF.tn_problem(dgi[d], x, dw, 3 * H, In, K, ra=rowmap(3 * H, fmt=1), rb=rowmap(In, fmt=1), colsum_out=db)))
    return [p for n, p in probs if which is None or n == which][:None if which is None else 1]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for which in (None, 'dW_ih', 'dW_hh rz', 'dW_hh n'):
    pr = problems(which)
    flop = sum(2.0 * p.M * p.N * p.K for p in pr)
    res = []
    for rnd in range(2):
        for dma in ('0', '1'):
            os.environ['XPS_GEMM_DMA'] = dma
            res.append((dma, timeit(lambda: XF.gemm_tn_grouped(pr, dev))))
    s = ', '.join(f'DMA={d}: {t:7.1f} us' for d, t in res)
    best = min(t for d, t in res if d == '1')
    print(f'{which or "whole group (6 problems)":26s} {s}   [{3 * flop / best / 1e6 / 2500e0:.3f} of the bf16 peak issued with DMA]', flush=True)
