"""Why does the isolated forward probe of bench.py (~590-610 us) differ from the same launch inside the step (~496 us)?  Times the
cluster forward launch (training form) of the configs[3] layer-1 shape behind different predecessors, HIP events around the
recurrence launch only (the predecessor is issued before the start event; the queue never runs dry: 3 launches are queued ahead)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF
from cross_patient_speech_decoding_amd._lib import call, rowmap
T, B, H, In = 20, 2048, 512, 1024
torch.manual_seed(0)
dev = torch.device('cuda')
x = torch.randn(T, B, In, device=dev) * 0.5
w_ih = [torch.randn(3 * H, In, device=dev) * In ** -0.5 for _ in range(2)]
b_ih = [torch.randn(3 * H, device=dev) * 0.1 for _ in range(2)]
w_hh = [torch.randn(3 * H, H, device=dev) * H ** -0.5 for _ in range(2)]
b_hh = [torch.randn(3 * H, device=dev) * 0.1 for _ in range(2)]
gi = torch.empty(2, T, B, 3 * H, device=dev)
xs = XF.split4(x); ws = [XF.split4(w) for w in w_ih]
def proj():
    ra, rb, rc = rowmap(In, fmt=1), rowmap(In, fmt=1), rowmap(3 * H)
    call('xps_gemm_nt_multi_f32', XF._ptr(xs), C.byref(ra), XF._ptr_array(ws), C.byref(rb), XF._ptr_array([gi[d] for d in range(2)]),
         C.byref(rc), XF._ptr_array(b_ih), 2, T * B, 3 * H, In, XF._stream())
proj(); torch.cuda.synchronize()
gi_src = gi.clone()
big = torch.empty(256 * 1024 * 1024, device=dev)     # 1 GiB scratch to flush caches
def rec():
    return XF.gru_forward_training_form(gi, w_hh, b_hh, T, B, H, 2)
def timed(pre, n=12):
    for _ in range(3):
        pre(); rec()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pre(); pre_done = None
        e0.record(); rec(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
    return v[len(v) // 2], v[0], v[-1]
cases = [('nothing (back-to-back launches on a stale gi)', lambda: None),
         ('gi.copy_(gi_src) (bench.py probe)', lambda: gi.copy_(gi_src)),
         ('the projection GEMM that writes gi in the step', proj),
         ('1-GiB fill (cold caches)', lambda: big.fill_(1.0))]
for name, pre in cases:
    med, lo, hi = timed(pre)
    print(f'{name:55s}: forward launch {med:7.1f} us (min {lo:.1f}, max {hi:.1f})', flush=True)
XF.check_gru_status()
