"""Debug: per-gate consistency of the cluster forward kernel given its own h_{t-1}."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.nn_models import functional as xf

T, B, H = 3, 256, 512
torch.manual_seed(0)
xf.set_gemm_precision('fp32')
xf.set_gru_cluster_mode('steps')
for ndir, usebias in ((1, True), (2, False), (2, True)):
    gi = torch.randn(ndir, T, B, 3 * H).cuda()
    ws = [(torch.randn(3 * H, H) / 20).cuda() for _ in range(ndir)]
    bs = [(torch.randn(3 * H) * (0.3 if usebias else 0.0)).cuda() for _ in range(ndir)]
    y_ext, saved = xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)
    torch.cuda.synchronize()
    y = y_ext.cpu().double(); sv = saved.cpu().double(); g = gi.cpu().double()
    for d in range(ndir):
        wd = ws[d].cpu().double(); bd = bs[d].cpu().double()
        for s in range(T):
            t = s if d == 0 else T - 1 - s
            slot_prev = t if d == 0 else t + 2
            hprev = y[slot_prev][:, d * H:(d + 1) * H]
            gh = hprev @ wd.T + bd
            r = torch.sigmoid(g[d, t, :, :H] + gh[:, :H]); z = torch.sigmoid(g[d, t, :, H:2*H] + gh[:, H:2*H])
            q = gh[:, 2*H:]; n = torch.tanh(g[d, t, :, 2*H:] + r * q)
            h = n + z * (hprev - n)
            errs = [(sv[d, t, :, i*H:(i+1)*H] - ref).abs().max().item() for i, ref in enumerate((r, z, n, q))]
            eh = (y[t + 1][:, d * H:(d + 1) * H] - h).abs().max().item()
            print(f'ndir {ndir} bias {usebias} dir {d} step {s} (t={t}): r {errs[0]:.1e} z {errs[1]:.1e} n {errs[2]:.1e} q {errs[3]:.1e} h {eh:.1e}')
