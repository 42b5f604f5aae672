"""Generate tests/golden/realtime_small.npz from the REFERENCE's RealtimeRNNModel + greedy_decode_batch.
Build container only.  lightning / torchaudio / torchmetrics are absent: in-process glue modules supply
LightningModule (= nn.Module + no-op hooks), edit_distance, Running and CharErrorRate stubs; they touch
logging/metrics only — forward() is genuine torch."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from weights import weights_from_seed          # noqa: E402


def _glue():
    L = types.ModuleType('lightning')

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            import inspect
            frame = inspect.currentframe().f_back
            args = {k: v for k, v in frame.f_locals.items() if k not in ('self', '__class__')}
            self.hparams = types.SimpleNamespace(**args)

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass
    L.LightningModule = LightningModule
    ta, taf = types.ModuleType('torchaudio'), types.ModuleType('torchaudio.functional')
    taf.edit_distance = lambda a, b: 0
    ta.functional = taf
    tm, tmw = types.ModuleType('torchmetrics'), types.ModuleType('torchmetrics.wrappers')
    tmw.Running = lambda m, window=100: m
    tm.CharErrorRate = lambda: None
    tm.wrappers = tmw
    sys.modules.update({'lightning': L, 'torchaudio': ta, 'torchaudio.functional': taf, 'torchmetrics': tm,
                        'torchmetrics.wrappers': tmw})


_glue()
sys.path.insert(0, '/root/reference/aligned_decoding')
from realtime_sim.realtime_nn_model import RealtimeRNNModel     # noqa: E402
from realtime_sim.ctc_decoder import greedy_decode_batch        # noqa: E402

if __name__ == '__main__':
    torch.manual_seed(0)
    C, win, stride, H, Lr, ncls = 6, 14, 4, 32, 2, 11
    m = RealtimeRNNModel(win * C, H, Lr, ncls, dropout=0.0, win_size=win, stride=stride)
    sd = weights_from_seed(m.state_dict(), 55)
    sd['h0'] = torch.from_numpy(np.random.default_rng(56).uniform(-0.3, 0.3, tuple(m.h0.shape)).astype(np.float32))
    m.load_state_dict(sd)
    m.eval()
    x = torch.from_numpy(np.random.default_rng(57).standard_normal((3, 62, C)).astype(np.float32))
    with torch.no_grad():
        logits = m(x)
        dec = greedy_decode_batch(torch.log_softmax(logits, -1), blank=0)
    out = dict(x=x.numpy(), logits=logits.numpy(), seed=55, cfg=np.array([C, win, stride, H, Lr, ncls]),
               h0=sd['h0'].numpy(), torch_version=np.array(torch.__version__))
    for i, d in enumerate(dec):
        out[f'dec{i}'] = d.numpy()
    np.savez_compressed(os.path.join(HERE, 'realtime_small.npz'), **out)
    print('realtime_small.npz', logits.shape, [len(d) for d in dec])
