"""Generate tests/golden/seq2seq_*.npz by running the REFERENCE's own Seq2SeqRNN on CPU.

Run in the build container only:   python tests/golden/make_seq2seq_fixtures.py

``nn_models/models.py`` imports ``lightning`` and ``torchmetrics``, which are not in the
image (ordinary ModuleNotFoundError).  This script registers two in-process
module objects for them before the import: ``lightning.LightningModule`` = a torch.nn.Module
with no-op ``log``/``log_dict``/``save_hyperparameters``, and
``torchmetrics.functional.classification.multiclass_confusion_matrix`` = a bincount.  They
replace logging glue only; every arithmetic op the fixtures record (Conv1d, BatchNorm1d,
GRU, Embedding, Linear, cross_entropy, AdamW, clip_grad_norm_) is the genuine torch code the
reference calls.  Nothing of this travels to the GPU box except the .npz files.

Weights are NOT stored: they are drawn from numpy's PCG64 (`weights_from_seed`, also used
by the tests) and loaded with load_state_dict, so a fixture holds seed + inputs + outputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from weights import weights_from_seed          # noqa: E402


def _register_glue():
    L = types.ModuleType('lightning')

    class LightningModule(torch.nn.Module):
        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

        def save_hyperparameters(self, *a, **k):
            pass
    L.LightningModule = LightningModule
    sys.modules['lightning'] = L
    tm = types.ModuleType('torchmetrics')
    tmf = types.ModuleType('torchmetrics.functional')
    tmc = types.ModuleType('torchmetrics.functional.classification')

    def multiclass_confusion_matrix(preds, target, num_classes):
        return torch.bincount(target * num_classes + preds,
                              minlength=num_classes ** 2).view(num_classes, num_classes)
    tmc.multiclass_confusion_matrix = multiclass_confusion_matrix
    tm.functional, tmf.classification = tmf, tmc
    sys.modules.update({'torchmetrics': tm, 'torchmetrics.functional': tmf,
                        'torchmetrics.functional.classification': tmc})


_register_glue()
sys.path.insert(0, '/root/reference/aligned_decoding')
from nn_models.models import Seq2SeqRNN, cmat_acc        # noqa: E402


def run_case(name, cfg, B, T, seed, store_grads):
    torch.set_num_threads(1)
    torch.manual_seed(seed)
    kw = dict(cfg)
    model = Seq2SeqRNN(kw.pop('in_channels'), kw.pop('n_filters'), kw.pop('hidden_size'), 9,
                       kw.pop('n_enc_layers'), kw.pop('n_dec_layers'), kw.pop('kernel_size'),
                       kw.pop('stride'), 0, 0.0, 0.0, 'gru', 1e-3, 1e-5, activation=kw.pop('activation'),
                       decay_iters=5)
    sd = weights_from_seed(model.state_dict(), seed)
    model.load_state_dict(sd)
    rng = np.random.default_rng(seed + 1)
    x = torch.from_numpy(rng.standard_normal((B, T, cfg['in_channels'])).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 9, (B, 3)))
    out = dict(seed=seed, x=x.numpy(), y=y.numpy(), cfg=np.array(repr(cfg)),
               torch_version=np.array(torch.__version__))
    # ---- eval-mode forward (running stats) : models.py:253-303 via validation path -------
    model.eval()
    with torch.no_grad():
        logits = model(x, y, teacher_forcing_ratio=0)
    out['eval_logits'] = logits.numpy()
    out['eval_argmax'] = logits.argmax(-1).numpy()
    out['eval_acc'] = cmat_acc(logits.view(-1, 9), y.view(-1), 9).numpy()
    # ---- one training step, teacher forcing always / never ---------------------------------
    for tag, ratio in (('tf1', 1.1), ('tf0', 0.0)):        # rand(1) < 1.1 always ; < 0 never
        model.load_state_dict(sd)
        model.train()
        opt = model.configure_optimizers()['optimizer']
        opt.zero_grad()
        y_hat = model(x, y, teacher_forcing_ratio=ratio)
        loss = model.criterion(y_hat.view(-1, 9), y.view(-1))
        loss.backward()
        out[f'{tag}_loss'] = loss.detach().numpy()
        out[f'{tag}_logits'] = y_hat.detach().numpy()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)   # Trainer(gradient_clip_val=0.5)
        out[f'{tag}_gnorm'] = gnorm.numpy()
        if store_grads:
            for k, p in model.named_parameters():
                out[f'{tag}_grad/{k}'] = p.grad.numpy().copy()           # clipped grads
        else:                                                            # big model: per-tensor norms only
            for k, p in model.named_parameters():
                out[f'{tag}_gradnorm/{k}'] = p.grad.norm().numpy()
        opt.step()
        if store_grads:
            for k, v in model.state_dict().items():
                out[f'{tag}_after/{k}'] = v.numpy().copy()
        else:
            out[f'{tag}_after_sum'] = np.array([v.double().sum().item() for v in model.state_dict().values()])
    np.savez_compressed(os.path.join(HERE, f'seq2seq_{name}.npz'), **out)
    print(name, os.path.getsize(os.path.join(HERE, f'seq2seq_{name}.npz')))


if __name__ == '__main__':
    tiny = dict(in_channels=6, n_filters=8, hidden_size=16, n_enc_layers=2, n_dec_layers=1,
                kernel_size=4, stride=4, activation=False)
    run_case('tiny', tiny, B=5, T=24, seed=101, store_grads=True)
    tiny_relu = dict(in_channels=5, n_filters=12, hidden_size=20, n_enc_layers=1, n_dec_layers=2,
                     kernel_size=5, stride=3, activation=True)
    run_case('tiny_relu_dec2', tiny_relu, B=7, T=26, seed=102, store_grads=True)
    cfg2 = dict(in_channels=64, n_filters=100, hidden_size=128, n_enc_layers=2, n_dec_layers=1,
                kernel_size=10, stride=10, activation=False)
    run_case('cfg2', cfg2, B=6, T=200, seed=103, store_grads=False)
