"""Pin oracle/seq2seq_oracle.py to golden vectors produced by the reference's own
Seq2SeqRNN (tests/golden/make_seq2seq_fixtures.py).  CPU only."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle.seq2seq_oracle import Seq2SeqOracle, cmat_acc, phoneme_error_rate, train_step
from weights import weights_from_seed


def build(g):
    cfg = ast.literal_eval(str(g['cfg']))
    m = Seq2SeqOracle(cfg['in_channels'], cfg['n_filters'], cfg['hidden_size'], 9, cfg['n_enc_layers'],
                      cfg['n_dec_layers'], cfg['kernel_size'], cfg['stride'], 0, 0.0, 0.0,
                      learning_rate=1e-3, l2_reg=1e-5, activation=cfg['activation'], decay_iters=5)
    sd = weights_from_seed(m.state_dict(), int(g['seed']))
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize('name', ['tiny', 'tiny_relu_dec2', 'cfg2'])
def test_eval_forward_matches_reference(golden_dir, name):
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, f'seq2seq_{name}.npz'))
    m, _ = build(g)
    m.eval()
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    with torch.no_grad():
        logits = m(x, y, teacher_forcing_ratio=0)
    np.testing.assert_allclose(logits.numpy(), g['eval_logits'], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(logits.argmax(-1).numpy(), g['eval_argmax'])
    np.testing.assert_allclose(cmat_acc(logits.view(-1, 9), y.view(-1), 9).numpy(), g['eval_acc'])


@pytest.mark.parametrize('name', ['tiny', 'tiny_relu_dec2', 'cfg2'])
@pytest.mark.parametrize('tag,coin', [('tf1', True), ('tf0', False)])
def test_train_step_matches_reference(golden_dir, name, tag, coin):
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, f'seq2seq_{name}.npz'))
    m, sd = build(g)
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    opt, _ = m.make_optimizer()
    loss, logits = train_step(m, opt, x, y, coins=[coin] * 3, clip=0.5)
    np.testing.assert_allclose(loss.numpy(), g[f'{tag}_loss'], rtol=1e-6)
    np.testing.assert_allclose(logits.numpy(), g[f'{tag}_logits'], atol=1e-6)
    grads = dict(m.named_parameters())
    if f'{tag}_grad/decoder.fc_out.weight' in g:
        for k, p in grads.items():
            np.testing.assert_allclose(p.grad.numpy(), g[f'{tag}_grad/{k}'], rtol=1e-4, atol=1e-7)
        for k, v in m.state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f'{tag}_after/{k}'], rtol=1e-5, atol=1e-7)
    else:
        for k, p in grads.items():
            np.testing.assert_allclose(p.grad.norm().numpy(), g[f'{tag}_gradnorm/{k}'], rtol=1e-4)
        after = np.array([v.double().sum().item() for v in m.state_dict().values()])
        np.testing.assert_allclose(after, g[f'{tag}_after_sum'], rtol=1e-6, atol=1e-6)


def test_state_dict_keys_are_the_reference_keys():
    m = Seq2SeqOracle(6, 8, 16, 9, 2, 1, 4, 4)
    keys = set(m.state_dict().keys())
    for k in ['temporal_conv.conv.weight', 'temporal_conv.bn.running_var', 'temporal_conv.bn.num_batches_tracked',
              'encoder.rnn.weight_ih_l0', 'encoder.rnn.weight_hh_l1_reverse', 'encoder.rnn.bias_hh_l0_reverse',
              'decoder.embedding.weight', 'decoder.rnn.bias_ih_l0', 'decoder.fc_out.bias']:
        assert k in keys


def test_per_definition():
    assert phoneme_error_rate([[1, 2, 3]], [[1, 2, 3]]) == 0.0
    assert phoneme_error_rate([[1, 2, 3], [4, 5, 6]], [[1, 9, 3], [5, 6, 7]]) == pytest.approx(100 * 3 / 6)
