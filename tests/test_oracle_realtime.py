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
