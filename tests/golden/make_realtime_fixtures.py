"""Generate tests/golden/realtime_small.npz from the REFERENCE's RealtimeRNNModel + greedy_decode_batch.
Build container only.  lightning / torchaudio / torchmetrics are absent: in-process glue modules supply
LightningModule (= nn.Module + no-op hooks), edit_distance, Running and CharErrorRate stubs; they touch
logging/metrics only (edit_distance is a plain Levenshtein) — forward() is genuine torch."""
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

        def log(self, name, value, *a, **k):
            self.__dict__.setdefault('_logged', {})[name] = value

        def log_dict(self, *a, **k):
            pass
    L.LightningModule = LightningModule
    ta, taf = types.ModuleType('torchaudio'), types.ModuleType('torchaudio.functional')
    def _edit_distance(a, b):                      # Levenshtein, what torchaudio.functional.edit_distance computes
        a, b = [int(v) for v in a], [int(v) for v in b]
        prev = list(range(len(b) + 1))
        for i, x in enumerate(a, 1):
            cur = [i]
            for j, y in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
            prev = cur
        return prev[-1]
    taf.edit_distance = _edit_distance
    ta.functional = taf
    tm, tmw = types.ModuleType('torchmetrics'), types.ModuleType('torchmetrics.wrappers')
    tmw.Running = lambda m, window=100: m
    class _Metric:
        def update(self, *a, **k):
            pass
    tm.CharErrorRate = _Metric
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

    # ---- training step (CTC): loss and every parameter gradient from the reference's own training_step ----
    torch.manual_seed(1)
    m = RealtimeRNNModel(win * C, H, Lr, ncls, dropout=0.0, win_size=win, stride=stride)
    sd = weights_from_seed(m.state_dict(), 77)
    sd['h0'] = torch.from_numpy(np.random.default_rng(78).uniform(-0.3, 0.3, tuple(m.h0.shape)).astype(np.float32))
    m.load_state_dict(sd)
    m.train()
    rng = np.random.default_rng(79)
    x = torch.from_numpy(rng.standard_normal((5, 62, C)).astype(np.float32))
    targets = torch.from_numpy(rng.integers(1, ncls, (5, 3)).astype(np.int64))
    targets[1, 1] = targets[1, 0]                                   # a repeated label (needs the blank between)
    input_lengths = torch.tensor([62, 62, 50, 62, 30], dtype=torch.int64)
    target_lengths = torch.tensor([3, 3, 2, 1, 3], dtype=torch.int64)
    loss = m.training_step((x, targets, input_lengths, target_lengths), 0)
    loss.backward()
    out = dict(x=x.numpy(), targets=targets.numpy(), input_lengths=input_lengths.numpy(),
               target_lengths=target_lengths.numpy(), loss=loss.detach().numpy(), seed=77,
               cfg=np.array([C, win, stride, H, Lr, ncls]), h0=sd['h0'].numpy(), torch_version=np.array(torch.__version__))
    for k, p in m.named_parameters():
        out['grad.' + k] = p.grad.numpy()
    m.eval()
    with torch.no_grad():
        vloss = m.validation_step((x, targets, input_lengths, target_lengths), 0)
        # test_step (:283-286) hands the lengths to CTCLoss unadjusted: give it window counts
        win_lengths = ((input_lengths - win) // stride) + 1
        tloss = m.test_step((x, targets, win_lengths, target_lengths), 0)
    out['val_loss'], out['test_loss'] = vloss.numpy(), tloss.numpy()
    out['val_PER'] = np.asarray(float(m._logged['val_PER']))
    np.savez_compressed(os.path.join(HERE, 'realtime_train_small.npz'), **out)
    print('realtime_train_small.npz loss', float(loss), 'val', float(vloss), 'test', float(tloss), 'PER', float(out['val_PER']))
