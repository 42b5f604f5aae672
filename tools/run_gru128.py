"""Counter / timing driver: the resident H = 128 recurrence (cfg-2 encoder layer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as xf
T, B, H, ndir = 20, 2048, 128, 2
torch.manual_seed(0)
gi = (torch.randn(ndir, T, B, 3 * H) * 0.5).cuda()
ws = [(torch.randn(3 * H, H) / H ** 0.5).cuda() for _ in range(ndir)]
bs = [(torch.randn(3 * H) * 0.1).cuda() for _ in range(ndir)]
dy = (torch.randn(T, B, ndir * H) * 0.1).cuda()
def ev(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for prec in ('bf16x3', 'fp32'):
    xf.set_gemm_precision(prec)
    y_ext, saved = xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)
    print(prec, 'fwd %.1f us  bwd (incl. transposes) %.1f us' % (ev(lambda: xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)),
          ev(lambda: xf._gru_backward(dy, None, y_ext, saved, ws, T, B, H, ndir, False))), flush=True)
