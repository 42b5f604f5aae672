"""Pin oracle/realtime_oracle.py to the golden vectors of the reference's RealtimeRNNModel.  CPU only."""
import os

import numpy as np
import torch

from oracle.realtime_oracle import RealtimeOracle, greedy_decode_batch
from weights import weights_from_seed


def build_oracle(g):
    C, win, stride, H, L, ncls = [int(v) for v in g['cfg']]
    o = RealtimeOracle(win * C, H, L, ncls, win, stride)
    ref_keys = {}
    for l in range(L):
        for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
            ref_keys[f'rnn.rnn.{n}_l{l}'] = getattr(o.gru, f'{n}_l{l}')
    # the fixture drew weights over the reference state_dict order: rnn.* (per layer), h0, classifier.fc.*
    order = {'h0': o.h0}
    sd_ref = {}
    for l in range(L):
        for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
            sd_ref[f'rnn.rnn.{n}_l{l}'] = getattr(o.gru, f'{n}_l{l}').detach()
    full = {'h0': o.h0.detach()}
    full.update(sd_ref)
    full.update({'classifier.fc.weight': o.fc.weight.detach(), 'classifier.fc.bias': o.fc.bias.detach()})
    sd = weights_from_seed(full, int(g['seed']))
    sd['h0'] = torch.from_numpy(g['h0'])
    o.load_reference_state(sd)
    return o, sd


def test_realtime_forward_and_greedy_decode(golden_dir):
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, 'realtime_small.npz'))
    o, _ = build_oracle(g)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        logits = o(x)
    np.testing.assert_allclose(logits.numpy(), g['logits'], atol=1e-6)
    dec = greedy_decode_batch(torch.log_softmax(logits, -1))
    for i, d in enumerate(dec):
        np.testing.assert_array_equal(d.numpy(), g[f'dec{i}'])


def test_ctc_training_step_loss_and_gradients(golden_dir):
    """training / validation / test step losses, PER and every parameter gradient of the reference's own steps."""
    from oracle.realtime_oracle import calc_per, ctc_step_loss
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, 'realtime_train_small.npz'))
    o, _ = build_oracle(g)
    batch = tuple(torch.from_numpy(g[k]) for k in ('x', 'targets', 'input_lengths', 'target_lengths'))
    loss = ctc_step_loss(o, batch)
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-6)
    np.testing.assert_allclose(float(loss), float(g['val_loss']), rtol=1e-6)
    loss.backward()
    names = {'h0': o.h0, 'classifier.fc.weight': o.fc.weight, 'classifier.fc.bias': o.fc.bias}
    for l in range(int(g['cfg'][4])):
        for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
            names[f'rnn.rnn.{n}_l{l}'] = getattr(o.gru, f'{n}_l{l}')
    for k, p in names.items():
        ref = g['grad.' + k]
        np.testing.assert_allclose(p.grad.numpy(), ref, atol=1e-6 * max(1.0, np.abs(ref).max()), err_msg=k)
    win, stride = int(g['cfg'][1]), int(g['cfg'][2])
    wl = ((batch[2] - win) // stride) + 1
    np.testing.assert_allclose(float(ctc_step_loss(o, (batch[0], batch[1], wl, batch[3]), adjust=False)),
                               float(g['test_loss']), rtol=1e-6)
    with torch.no_grad():
        dec = greedy_decode_batch(torch.log_softmax(o(batch[0]), -1))
    np.testing.assert_allclose(calc_per(dec, batch[1], batch[3]), float(g['val_PER']), rtol=1e-6)
