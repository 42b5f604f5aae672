"""One-shot driver for counter runs: a few launches of the H = 512 recurrence forward (and backward with BWD=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as xf
T, B, H, ndir = 20, 2048, 512, 2
torch.manual_seed(0)
gi = (torch.randn(ndir, T, B, 3 * H) * 0.5).cuda()
ws = [(torch.randn(3 * H, H) / H ** 0.5).cuda() for _ in range(ndir)]
bs = [(torch.randn(3 * H) * 0.1).cuda() for _ in range(ndir)]
dy = (torch.randn(T, B, ndir * H) * 0.1).cuda()
if os.environ.get('GRID'):          # 1: the 4 x 4 BPTT grid
    from cross_patient_speech_decoding_amd import _lib
    _lib.lib().xps_set_gru_bptt_grid(int(os.environ['GRID']))
for _ in range(5):
    y_ext, saved = xf.gru_forward_training_form(gi, ws, bs, T, B, H, ndir)       # (the form a training step launches)
    if os.environ.get('BWD'):
        xf._gru_backward(dy, None, y_ext, saved, ws, T, B, H, ndir, False, split4=xf.split4_wanted(T, B, H, ndir))   # (the output format of the step)
torch.cuda.synchronize()
xf.check_gru_status()
