"""Debug: cluster GRU recurrence vs torch CPU, error map by (mode, precision, t, dir, unit block, trial block)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.nn_models import functional as xf
from cross_patient_speech_decoding_amd._lib import lib

T, B, In, H, ndir = 4, 256, 24, int(os.environ.get('H', 512)), 2
torch.manual_seed(0)
gru = torch.nn.GRU(In, H, 1, bidirectional=True)
x = torch.randn(T, B, In)
y_ref, _ = gru(x)
y_ref = y_ref.detach().numpy()
ws = []
for d in range(ndir):
    sfx = '_l0' + ('_reverse' if d else '')
    ws += [getattr(gru, n + sfx).detach().clone().cuda() for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
for prec in ('fp32', 'bf16x3'):
    xf.set_gemm_precision(prec)
    for mode in ('steps', 'persistent'):
        xf.set_gru_cluster_mode(mode)
        with torch.no_grad():
            y, hn = xf.GRULayerFn.apply(x.cuda(), ndir, xf.HN_STACK, *ws)
        torch.cuda.synchronize()
        y = y.cpu().numpy()
        err = np.abs(y - y_ref)
        print(prec, mode, 'max err', err.max())
        for d in range(ndir):
            for t in range(T):
                e = err[t, :, d * H:(d + 1) * H]
                ub = e.reshape(B, H // 32, 32).max(axis=(0, 2))
                tb = e.reshape(B // 16, 16, H).max(axis=(1, 2))
                print(f'  dir {d} t {t}: max {e.max():.2e} | unit blocks(32) bad: {np.nonzero(ub > 1e-4)[0].tolist()} | trial tiles(16) bad: {np.nonzero(tb > 1e-4)[0].tolist()}')
