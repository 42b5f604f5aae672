"""Launch the encoder GRU recurrence (bench shape) a few times: target for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF

which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
T, B, H = 20, 2048, 128
torch.manual_seed(0)
w_hh = [torch.randn(3 * H, H, device='cuda') * 0.08 for _ in range(2)]
b_hh = [torch.randn(3 * H, device='cuda') * 0.1 for _ in range(2)]
gi = torch.randn(2, T, B, 3 * H, device='cuda') * 0.5
y_ext, saved = XF._gru_forward(gi, w_hh, b_hh, None, T, B, H, 2, True)
dy = torch.randn(T + 2, B, 2 * H, device='cuda')
for _ in range(10):
    if which == 'fwd':
        XF._gru_forward(gi, w_hh, b_hh, None, T, B, H, 2, True)
    else:
        XF._gru_backward(dy[1:T + 1], None, y_ext, saved, w_hh, T, B, H, 2, False)
torch.cuda.synchronize()
print('done')
