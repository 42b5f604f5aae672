"""End-to-end parity of the HIP Seq2SeqRNN against (a) the golden vectors produced by the
reference's own Seq2SeqRNN and (b) the CPU oracle on seeded inputs.
North-star bar: argmax indices bit-exact, logits within 1e-4 abs."""
import ast
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from weights import weights_from_seed  # noqa: E402


# The conv bias feeds a train-mode BatchNorm, which subtracts the batch mean: its gradient is
# analytically ZERO and numerically rounding noise (~1e-9) in the reference and here alike.  Adam
# divides by sqrt(v) of that noise, so the UPDATED bias is +-lr of noise on both sides: excluded.
NOISE_KEY = 'temporal_conv.conv.bias'


def build_hip(cfg, seed, dropout=0.0):
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    m = Seq2SeqRNN(cfg['in_channels'], cfg['n_filters'], cfg['hidden_size'], 9, cfg['n_enc_layers'],
                   cfg['n_dec_layers'], cfg['kernel_size'], cfg['stride'], 0, dropout, dropout, 'gru', 1e-3, 1e-5,
                   activation=cfg['activation'], decay_iters=5)
    m.load_state_dict(weights_from_seed(m.state_dict(), seed))
    return m.to('cuda')


@pytest.mark.parametrize('name', ['tiny', 'tiny_relu_dec2', 'cfg2'])
def test_eval_logits_and_argmax_match_reference_golden(golden_dir, name, gemm_precision):
    g = np.load(os.path.join(golden_dir, f'seq2seq_{name}.npz'))
    cfg = ast.literal_eval(str(g['cfg']))
    m = build_hip(cfg, int(g['seed'])).eval()
    x, y = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['y']).cuda()
    with torch.no_grad():
        logits = m(x, y, teacher_forcing_ratio=0)
    err = np.abs(logits.cpu().numpy() - g['eval_logits']).max()
    assert err <= 1e-4, err                                         # north-star tolerance
    np.testing.assert_array_equal(logits.argmax(-1).cpu().numpy(), g['eval_argmax'])   # bit-exact indices
    from cross_patient_speech_decoding_amd.nn_models import cmat_acc
    np.testing.assert_allclose(cmat_acc(logits.view(-1, 9), y.view(-1), 9).item(), float(g['eval_acc']))


@pytest.mark.parametrize('name', ['tiny', 'tiny_relu_dec2', 'cfg2'])
@pytest.mark.parametrize('tag,coin', [('tf1', True), ('tf0', False)])
def test_train_step_matches_reference_golden(golden_dir, name, tag, coin, gemm_precision):
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    g = np.load(os.path.join(golden_dir, f'seq2seq_{name}.npz'))
    cfg = ast.literal_eval(str(g['cfg']))
    m = build_hip(cfg, int(g['seed'])).train()
    x, y = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['y']).cuda()
    opt = FlatAdamW(m, lr=1e-3, weight_decay=1e-5, max_norm=0.5)
    logits = m(x, y, coins=[coin] * 3)
    loss = m.criterion(logits.view(-1, 9), y.view(-1))
    opt.zero_grad()
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g[f'{tag}_loss']), rtol=2e-5)
    assert np.abs(logits.detach().cpu().numpy() - g[f'{tag}_logits']).max() <= 1e-4
    opt.step()
    gnorm = opt.grad_norm()
    np.testing.assert_allclose(float(gnorm), float(g[f'{tag}_gnorm']), rtol=2e-4)
    params = dict(m.named_parameters())
    if f'{tag}_grad/decoder.fc_out.weight' in g:
        for k, p in params.items():          # clipped gradients, then updated weights
            ref = g[f'{tag}_grad/{k}']
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()),
                                       err_msg=k)
        for k, v in m.state_dict().items():
            if k == NOISE_KEY:
                continue
            got, ref = v.cpu().numpy(), g[f'{tag}_after/{k}']
            if gemm_precision == 'bf16x3' and f'{tag}_grad/{k}' in g:
                # Adam's first step moves a weight by lr * g / (|g| + eps) ~ lr * sign(g): elements whose gradient is at
                # the noise floor of the split products (1e-5 of the largest) can move the other way; compare the rest
                gr = np.abs(g[f'{tag}_grad/{k}'])
                keep = gr > 1e-3 * gr.max()
                got, ref = got[keep], ref[keep]
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5, err_msg=k)
    else:
        for k, p in params.items():
            if k == NOISE_KEY:
                continue
            np.testing.assert_allclose(p.grad.norm().item(), float(g[f'{tag}_gradnorm/{k}']), rtol=2e-3, err_msg=k)
        keys = list(m.state_dict().keys())
        after = np.array([v.double().sum().item() for v in m.state_dict().values()])
        keep = np.array([k != NOISE_KEY for k in keys])
        np.testing.assert_allclose(after[keep], g[f'{tag}_after_sum'][keep], rtol=1e-4, atol=2e-3)


def test_seeded_larger_batch_vs_oracle(gemm_precision):
    """cfg-2 architecture, B = 96 (partial 16-row tiles), random seeded weights: HIP vs the CPU oracle."""
    from oracle.seq2seq_oracle import Seq2SeqOracle
    torch.set_num_threads(8)
    cfg = dict(in_channels=64, n_filters=100, hidden_size=128, n_enc_layers=2, n_dec_layers=1, kernel_size=10,
               stride=10, activation=False)
    orc = Seq2SeqOracle(64, 100, 128, 9, 2, 1, 10, 10, 0, 0.0, 0.0, activation=False)
    sd = weights_from_seed(orc.state_dict(), 7)
    orc.load_state_dict(sd)
    m = build_hip(cfg, 7)
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.standard_normal((96, 200, 64)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 9, (96, 3)))
    orc.eval(); m.eval()
    with torch.no_grad():
        ref = orc(x, y, teacher_forcing_ratio=0)
        out = m(x.cuda(), y.cuda(), teacher_forcing_ratio=0)
    assert (out.cpu() - ref).abs().max().item() <= 1e-4
    assert torch.equal(out.argmax(-1).cpu(), ref.argmax(-1))
    # training mode with batch statistics, teacher forcing on: logits + loss
    orc.train(); m.train()
    ref = orc(x, y, coins=[True, False, True])
    out = m(x.cuda(), y.cuda(), coins=[True, False, True])
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-4


def test_full_bench_size_step_vs_oracle():
    """The bench workload itself (cfg 2: 2048 trials x 200 samples x 64 channels, H = 128): eval logits / argmax, training
    loss and the gradient of every parameter of one full-batch step against the CPU oracle (no dropout: deterministic)."""
    from oracle.seq2seq_oracle import Seq2SeqOracle
    torch.set_num_threads(16)
    cfg = dict(in_channels=64, n_filters=100, hidden_size=128, n_enc_layers=2, n_dec_layers=1, kernel_size=10,
               stride=10, activation=False)
    orc = Seq2SeqOracle(64, 100, 128, 9, 2, 1, 10, 10, 0, 0.0, 0.0, activation=False)
    orc.load_state_dict(weights_from_seed(orc.state_dict(), 21))
    m = build_hip(cfg, 21)
    rng = np.random.default_rng(22)
    B = 2048
    x = torch.from_numpy(rng.standard_normal((B, 200, 64)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 9, (B, 3)))
    orc.eval(); m.eval()
    with torch.no_grad():
        ref = orc(x, y, teacher_forcing_ratio=0)
        out = m(x.cuda(), y.cuda(), teacher_forcing_ratio=0)
    assert (out.cpu() - ref).abs().max().item() <= 1e-4
    assert torch.equal(out.argmax(-1).cpu(), ref.argmax(-1))                  # all 6144 argmax indices identical
    orc.train(); m.train()
    coins = [True, False, True]
    ref = orc(x, y, coins=coins)
    loss_ref = torch.nn.functional.cross_entropy(ref.reshape(-1, 9), y.reshape(-1))
    loss_ref.backward()
    out = m(x.cuda(), y.cuda(), coins=coins)
    loss = m.criterion(out.view(-1, 9), y.cuda().view(-1))
    loss.backward()
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-4
    np.testing.assert_allclose(float(loss.detach()), float(loss_ref.detach()), rtol=1e-5)
    refs = dict(orc.named_parameters())
    for k, p in m.named_parameters():
        if k == NOISE_KEY:
            continue
        g_ref = refs[k].grad
        tol = 1e-4 * max(float(g_ref.abs().max()), 1e-6)
        assert (p.grad.cpu() - g_ref).abs().max().item() <= tol, k


def test_reference_coin_sequence_is_reproduced():
    """Same seed -> the same teacher-forcing coins as the reference's torch.rand(1) draws (models.py:295)."""
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    m = Seq2SeqRNN(6, 8, 16, 9, 1, 1, 4, 4)
    torch.manual_seed(123)
    coins = m.draw_teacher_coins(torch.zeros(2, 3), 0.5)
    torch.manual_seed(123)
    ref = [torch.rand(1).item() < 0.5 for _ in range(3)]
    assert coins == ref
    torch.manual_seed(123)
    assert m.draw_teacher_coins(None, 0.5) == [False] * 3          # y is None: no draw at all
    a = torch.rand(1).item()
    torch.manual_seed(123)
    assert a == torch.rand(1).item()


@pytest.mark.parametrize('H,B', [(64, 37), (128, 50)])
def test_fused_decoder_equals_composed_path_and_oracle(H, B):
    """The one-launch decoder (xps_decoder_*) against the composed per-step path of the same model and
    against the CPU oracle: logits, argmax, and every parameter gradient, mixed teacher forcing."""
    from oracle.seq2seq_oracle import Seq2SeqOracle
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    cfg = dict(in_channels=10, n_filters=12, hidden_size=H, n_enc_layers=1, n_dec_layers=1, kernel_size=5,
               stride=5, activation=False)
    rng = np.random.default_rng(H)
    x = torch.from_numpy(rng.standard_normal((B, 30, 10)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 9, (B, 3)))
    coins = [True, False, True]
    orc = Seq2SeqOracle(10, 12, H, 9, 1, 1, 5, 5, 0, 0.0, 0.0, activation=False)
    orc.load_state_dict(weights_from_seed(orc.state_dict(), 31))
    orc.train()
    ref = orc(x, y, coins=coins)
    torch.nn.functional.cross_entropy(ref.reshape(-1, 9), y.reshape(-1)).backward()
    results = {}
    for fused in (True, False):
        m = build_hip(cfg, 31).train()
        orig = XF.decoder_supported
        if not fused:
            XF.decoder_supported = lambda *a: False
        try:
            out = m(x.cuda(), y.cuda(), coins=coins)
            m.criterion(out.view(-1, 9), y.cuda().view(-1)).backward()
        finally:
            XF.decoder_supported = orig
        results[fused] = (out.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()})
    for fused in (True, False):
        out, grads = results[fused]
        assert (out - ref.detach()).abs().max().item() <= 1e-4
        assert torch.equal(out.argmax(-1), ref.argmax(-1))
        for k, p in orc.named_parameters():
            if k == NOISE_KEY:
                continue
            tol = 2e-5 * max(1.0, p.grad.abs().max().item())
            assert (grads[k] - p.grad).abs().max().item() <= tol + 2e-3 * p.grad.abs().max().item(), (fused, k)
    # fused and composed agree with each other far below the oracle tolerance
    assert (results[True][0] - results[False][0]).abs().max().item() <= 1e-5


@pytest.mark.parametrize('H,B', [(512, 256), (500, 256), (512, 2048)])
def test_north_star_model_shape_vs_oracle(H, B, gemm_precision):
    """configs[3] model shape (aligned d = 30 input channels, F = 100, 2-layer bidirectional GRU encoder with H = 512 --
    and H = 500, the reference script's default, scripts/train_seq2seq.py:132 / nn_models/models.py:661-663 -- 1-layer
    decoder), B = 256 trials and B = 2048 = THE BENCH SHARD ITSELF (the per-GPU batch of `bench.py`'s headline: other
    cluster plans / round counts than B = 256): eval logits <= 1e-4 with identical argmax (all 6144 at B = 2048), then ONE
    full training step (teacher forcing mixed, no dropout): logits, loss (1e-5), every parameter gradient, clipped gradient
    norm and the AdamW-updated weights against the CPU oracle.  The encoder runs the cluster-persistent recurrence
    (csrc/xps_gru_cluster.hip), the layer GEMMs the 256-tile kernels at B = 2048."""
    from oracle.seq2seq_oracle import Seq2SeqOracle
    from cross_patient_speech_decoding_amd._lib import lib
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    C = 30
    assert lib().xps_gru_seq_status_offset(20, B, H, 2) >= 0          # this shape runs the cluster kernels
    cfg = dict(in_channels=C, n_filters=100, hidden_size=H, n_enc_layers=2, n_dec_layers=1, kernel_size=10,
               stride=10, activation=False)
    orc = Seq2SeqOracle(C, 100, H, 9, 2, 1, 10, 10, 0, 0.0, 0.0, learning_rate=1e-3, l2_reg=1e-5, activation=False)
    orc.load_state_dict(weights_from_seed(orc.state_dict(), 40 + H))
    m = build_hip(cfg, 40 + H)
    rng = np.random.default_rng(H)
    x = torch.from_numpy(rng.standard_normal((B, 200, C)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 9, (B, 3)))
    orc.eval(); m.eval()
    with torch.no_grad():
        ref = orc(x, y, teacher_forcing_ratio=0)
        out = m(x.cuda(), y.cuda(), teacher_forcing_ratio=0)
    assert (out.cpu() - ref).abs().max().item() <= 1e-4
    assert torch.equal(out.argmax(-1).cpu(), ref.argmax(-1))                  # all 768 argmax indices identical
    orc.train(); m.train()
    coins = [True, False, True]
    opt_ref, _ = orc.make_optimizer()
    opt_ref.zero_grad()
    ref = orc(x, y, coins=coins)
    loss_ref = torch.nn.functional.cross_entropy(ref.reshape(-1, 9), y.reshape(-1))
    loss_ref.backward()
    g_ref = {k: p.grad.detach().clone() for k, p in orc.named_parameters()}
    gn_ref = float(torch.nn.utils.clip_grad_norm_(orc.parameters(), 0.5))
    opt_ref.step()
    opt = FlatAdamW(m, lr=1e-3, weight_decay=1e-5, max_norm=0.5)
    opt.zero_grad()
    out = m(x.cuda(), y.cuda(), coins=coins)
    loss = m.criterion(out.view(-1, 9), y.cuda().view(-1))
    loss.backward()
    XF.check_gru_status()
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-4
    np.testing.assert_allclose(float(loss.detach()), float(loss_ref.detach()), rtol=1e-5)
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    for k, g in g_ref.items():
        if k == NOISE_KEY:
            continue
        tol = (1e-4 if gemm_precision == 'fp32' else 3e-4) * max(float(g.abs().max()), 1e-6)
        assert (grads[k] - g).abs().max().item() <= tol, (k, (grads[k] - g).abs().max().item(), tol)
    opt.step()
    np.testing.assert_allclose(float(opt.grad_norm()), gn_ref, rtol=2e-4)
    refs = dict(orc.named_parameters())
    for k, p in m.named_parameters():
        if k == NOISE_KEY:
            continue
        got, want = p.detach().cpu().numpy(), refs[k].detach().numpy()
        # Adam's first step moves a weight by lr * g / (|g| + eps) ~ lr * sign(g): an element whose gradient sits at the
        # rounding-noise floor (1e-3 of the tensor's largest and below) can move the other way; the rest is compared
        gr = g_ref[k].abs().numpy()
        keep = gr > 1e-3 * gr.max()
        got, want = got[keep], want[keep]
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-5, err_msg=k)


_per_oracle_cache = {}


def test_per_parity_at_configs1_shape_100_steps(gemm_precision):
    """north_star: phoneme error rate within +-0.5 % of the reference CPU path.  configs[1] shape itself (C = 64, F = 100,
    k = s = 10, H = 128, enc 2 x bi-GRU, dec 1 x GRU, 2048 trials x 200 samples), dropout 0 (deterministic), identical
    weights and teacher-forcing coins, 100 full-batch AdamW steps (clip 0.5) on the HIP path and on the CPU oracle; then
    PER by the reference's own formula (realtime_sim/realtime_nn_model.py:318-323: sum of edit distances / sum of target
    lengths x 100, `phoneme_error_rate`) and the token error 1 - acc (nn_models/models.py:875-889) of the eval-mode
    predictions of both models on the training trials and on 512 held-out trials."""
    from oracle.seq2seq_oracle import Seq2SeqOracle, phoneme_error_rate, train_step
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    STEPS, LR = 100, 2.5e-4              # (noise 4.0, 100 steps at this rate: the oracle ends at PER 16.4 % / 19.7 % -- mid-range, sensitive)
    X, yf = make_patient(0, 2560, T=200, C=64, noise=4.0)
    X, y = torch.from_numpy(X), torch.from_numpy(yf - 1)
    Xtr, ytr, Xte, yte = X[:2048], y[:2048], X[2048:], y[2048:]
    args = (64, 100, 128, 9, 2, 1, 10, 10, 0, 0.0, 0.0)
    coins = [[bool(c) for c in row] for row in np.random.default_rng(11).integers(0, 2, (STEPS, 3))]
    if 'ref' not in _per_oracle_cache:                      # the oracle trajectory does not depend on the HIP precision mode
        orc = Seq2SeqOracle(*args, learning_rate=LR, l2_reg=1e-5, activation=False)
        orc.load_state_dict(weights_from_seed(orc.state_dict(), 61))
        opt_o, _ = orc.make_optimizer()
        losses = [float(train_step(orc, opt_o, Xtr, ytr, coins=coins[i], clip=0.5)[0]) for i in range(STEPS)]
        orc.eval()
        with torch.no_grad():
            ptr = orc(Xtr, ytr, teacher_forcing_ratio=0).argmax(-1).numpy()
            pte = orc(Xte, yte, teacher_forcing_ratio=0).argmax(-1).numpy()
        _per_oracle_cache['ref'] = (losses, ptr, pte)
    losses_o, ptr_o, pte_o = _per_oracle_cache['ref']
    cfg = dict(in_channels=64, n_filters=100, hidden_size=128, n_enc_layers=2, n_dec_layers=1, kernel_size=10, stride=10,
               activation=False)
    hip = build_hip(cfg, 61)
    opt_h = FlatAdamW(hip, lr=LR, weight_decay=1e-5, max_norm=0.5)
    Xg, yg = Xtr.cuda(), ytr.cuda()
    hip.train()
    losses_h = []
    for i in range(STEPS):
        opt_h.zero_grad()
        logits = hip(Xg, yg, coins=coins[i])
        lh = hip.criterion(logits.view(-1, 9), yg.view(-1))
        lh.backward()
        opt_h.step()
        losses_h.append(lh)
    losses_h = [float(v) for v in losses_h]
    hip.eval()
    with torch.no_grad():
        ptr_h = hip(Xg, yg, teacher_forcing_ratio=0).argmax(-1).cpu().numpy()
        pte_h = hip(Xte.cuda(), yte.cuda(), teacher_forcing_ratio=0).argmax(-1).cpu().numpy()
    assert losses_o[-1] < 0.8 * losses_o[0], losses_o[::10]                    # the run learned something
    assert max(abs(a - b) for a, b in zip(losses_h, losses_o)) <= 5e-3, (losses_h[::10], losses_o[::10])
    for name, ph, po, yy in (('train', ptr_h, ptr_o, ytr.numpy()), ('held-out', pte_h, pte_o, yte.numpy())):
        per_h, per_o = phoneme_error_rate(ph, yy), phoneme_error_rate(po, yy)
        tok_h, tok_o = 100.0 * (ph != yy).mean(), 100.0 * (po != yy).mean()
        print(f'PER {name}: hip {per_h:.3f} oracle {per_o:.3f}; token error hip {tok_h:.3f} oracle {tok_o:.3f}; '
              f'argmax agreement {100.0 * (ph == po).mean():.3f} %')
        assert abs(per_h - per_o) <= 0.5, (name, per_h, per_o)                  # north-star tolerance: +-0.5 % absolute
        assert abs(tok_h - tok_o) <= 0.5, (name, tok_h, tok_o)
    assert phoneme_error_rate(ptr_o, ytr.numpy()) < 85.0                        # (chance ~ 89 %)


def test_per_parity_at_configs3_shape_50_steps(gemm_precision):
    """north_star's PER tolerance ON THE KERNELS THE HEADLINE RUNS: configs[3] model shape (aligned d = 30 input, F = 100,
    k = s = 10, 2 x bi-GRU H = 512 = cluster recurrence + 256-tile GEMMs + wide decoder, dec 1 x GRU), 512 training trials
    x 200 samples, dropout 0, identical weights and teacher-forcing coins, 50 full-batch AdamW steps (clip 0.5; noise 3.0 /
    lr 1.5e-4 so that the oracle ends mid-range: PER 13.5 % on the training trials, 25.5 % on 256 held-out trials) on the HIP
    path and on the CPU oracle (16 threads); then PER by the reference's own formula
    (realtime_sim/realtime_nn_model.py:318-323) and the token error 1 - acc (nn_models/models.py:875-889) of the eval-mode
    predictions: within +-0.5 % absolute on both sets, loss curves within 5e-3."""
    from oracle.seq2seq_oracle import Seq2SeqOracle, phoneme_error_rate, train_step
    from cross_patient_speech_decoding_amd._lib import lib
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    STEPS, LR, B = 50, 1.5e-4, 512
    assert lib().xps_gru_seq_status_offset(20, B, 512, 2) >= 0            # this shape trains through the cluster kernels
    X, yf = make_patient(0, B + 256, T=200, C=30, noise=3.0)
    X, y = torch.from_numpy(X), torch.from_numpy(yf - 1)
    Xtr, ytr, Xte, yte = X[:B], y[:B], X[B:], y[B:]
    args = (30, 100, 512, 9, 2, 1, 10, 10, 0, 0.0, 0.0)
    coins = [[bool(c) for c in row] for row in np.random.default_rng(12).integers(0, 2, (STEPS, 3))]
    if 'ref3' not in _per_oracle_cache:                     # the oracle trajectory does not depend on the HIP precision mode
        orc = Seq2SeqOracle(*args, learning_rate=LR, l2_reg=1e-5, activation=False)
        orc.load_state_dict(weights_from_seed(orc.state_dict(), 71))
        opt_o, _ = orc.make_optimizer()
        losses = [float(train_step(orc, opt_o, Xtr, ytr, coins=coins[i], clip=0.5)[0]) for i in range(STEPS)]
        orc.eval()
        with torch.no_grad():
            ptr = orc(Xtr, ytr, teacher_forcing_ratio=0).argmax(-1).numpy()
            pte = orc(Xte, yte, teacher_forcing_ratio=0).argmax(-1).numpy()
        _per_oracle_cache['ref3'] = (losses, ptr, pte)
    losses_o, ptr_o, pte_o = _per_oracle_cache['ref3']
    cfg = dict(in_channels=30, n_filters=100, hidden_size=512, n_enc_layers=2, n_dec_layers=1, kernel_size=10, stride=10,
               activation=False)
    hip = build_hip(cfg, 71)
    opt_h = FlatAdamW(hip, lr=LR, weight_decay=1e-5, max_norm=0.5)
    Xg, yg = Xtr.cuda(), ytr.cuda()
    hip.train()
    losses_h = []
    for i in range(STEPS):
        opt_h.zero_grad()
        logits = hip(Xg, yg, coins=coins[i])
        lh = hip.criterion(logits.view(-1, 9), yg.view(-1))
        lh.backward()
        opt_h.step()
        losses_h.append(lh.detach())
    XF.check_gru_status()
    losses_h = [float(v) for v in losses_h]
    hip.eval()
    with torch.no_grad():
        ptr_h = hip(Xg, yg, teacher_forcing_ratio=0).argmax(-1).cpu().numpy()
        pte_h = hip(Xte.cuda(), yte.cuda(), teacher_forcing_ratio=0).argmax(-1).cpu().numpy()
    assert losses_o[-1] < 0.5 * losses_o[0], losses_o[::10]
    assert max(abs(a - b) for a, b in zip(losses_h, losses_o)) <= 5e-3, (losses_h[::10], losses_o[::10])
    for name, ph, po, yy in (('train', ptr_h, ptr_o, ytr.numpy()), ('held-out', pte_h, pte_o, yte.numpy())):
        per_h, per_o = phoneme_error_rate(ph, yy), phoneme_error_rate(po, yy)
        tok_h, tok_o = 100.0 * (ph != yy).mean(), 100.0 * (po != yy).mean()
        print(f'H=512 PER {name}: hip {per_h:.3f} oracle {per_o:.3f}; token error hip {tok_h:.3f} oracle {tok_o:.3f}; '
              f'argmax agreement {100.0 * (ph == po).mean():.3f} %')
        assert abs(per_h - per_o) <= 0.5, (name, per_h, per_o)                  # north-star tolerance: +-0.5 % absolute
        assert abs(tok_h - tok_o) <= 0.5, (name, tok_h, tok_o)
    assert 5.0 < phoneme_error_rate(ptr_o, ytr.numpy()) < 60.0                  # mid-range: the comparison is sensitive
