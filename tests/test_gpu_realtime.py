"""Config 5: realtime CTC-RNN inference on the GPU — full-sequence forward against the reference golden,
hipGraph-captured per-step streaming against the full forward, greedy CTC decode."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from weights import weights_from_seed  # noqa: E402


def build(g):
    from cross_patient_speech_decoding_amd.realtime_sim import RealtimeRNNModel
    C, win, stride, H, L, ncls = [int(v) for v in g['cfg']]
    m = RealtimeRNNModel(win * C, H, L, ncls, dropout=0.0, win_size=win, stride=stride)
    sd = weights_from_seed(m.state_dict(), int(g['seed']))
    sd['h0'] = torch.from_numpy(g['h0'])
    m.load_state_dict(sd)
    return m.cuda().eval(), (C, win, stride, H, L, ncls)


def test_forward_matches_reference_golden(golden_dir, gemm_precision):
    from cross_patient_speech_decoding_amd.realtime_sim import greedy_decode_batch
    g = np.load(os.path.join(golden_dir, 'realtime_small.npz'))
    m, _ = build(g)
    x = torch.from_numpy(g['x']).cuda()
    with torch.no_grad():
        logits = m(x)
    assert np.abs(logits.cpu().numpy() - g['logits']).max() <= 1e-4
    np.testing.assert_array_equal(logits.argmax(-1).cpu().numpy(), g['logits'].argmax(-1))
    dec = greedy_decode_batch(torch.log_softmax(logits, -1), blank=0)
    for i, d in enumerate(dec):
        np.testing.assert_array_equal(d.cpu().numpy(), g[f'dec{i}'])
    w = m.reformat_time_windows(x)
    assert w.shape == (3, 13, 14 * 6)
    np.testing.assert_array_equal(w[1, 2].cpu().numpy(), g['x'][1, 8:22].reshape(-1))


@pytest.mark.parametrize('use_graph', [False, True])
def test_streaming_steps_equal_full_forward(golden_dir, use_graph):
    from cross_patient_speech_decoding_amd.realtime_sim import StreamingDecoder
    g = np.load(os.path.join(golden_dir, 'realtime_small.npz'))
    m, (C, win, stride, H, L, ncls) = build(g)
    x = torch.from_numpy(g['x']).cuda()
    full = m(x)                                         # (3, 13, 11)
    dec = StreamingDecoder(m, n_streams=1, use_graph=use_graph)
    for b in range(2):
        dec.reset()
        toks = []
        for w in range(full.shape[1]):
            window = x[b, w * stride:w * stride + win].reshape(1, -1)
            lg = dec.step(window)
            assert (lg[0] - full[b, w]).abs().max().item() <= 2e-5, (b, w)
            toks.append(int(dec.token[0]))
        assert toks == full[b].argmax(-1).tolist()
    # three streams at once (rows padded to 4)
    dec3 = StreamingDecoder(m, n_streams=3, use_graph=use_graph)
    for w in range(full.shape[1]):
        lg = dec3.step(x[:, w * stride:w * stride + win].reshape(3, -1))
        assert (lg - full[:, w]).abs().max().item() <= 2e-5


def test_config5_full_shape_graph_replay_full_forward_and_oracle_agree(gemm_precision):
    """BASELINE config 5 AT ITS OWN SHAPE (C = 128 channels -> 14 x 128 = 1792-wide windows, H = 128, L = 2, 11 classes;
    reference: realtime_sim/realtime_nn_model.py:153-199): (1) the full-sequence forward of the HIP model against the CPU
    oracle (oracle/realtime_oracle.py, pinned to the reference's goldens at the small shape): logits <= 1e-4, argmax
    identical at all 47 windows of every trial; (2) the hipGraph-replayed per-window step (what bench.py's `realtime`
    record times) against both: <= 1e-4 of the oracle, <= 2e-5 of the full forward, same tokens; (3) the greedy CTC
    decode of the two paths (ctc_decoder.py:172-189)."""
    from oracle.realtime_oracle import RealtimeOracle, greedy_decode_batch as greedy_ref
    from cross_patient_speech_decoding_amd.realtime_sim import RealtimeRNNModel, StreamingDecoder, greedy_decode_batch
    C, win, stride, H, L, ncls = 128, 14, 4, 128, 2, 11
    torch.set_num_threads(min(8, len(os.sched_getaffinity(0))))
    m = RealtimeRNNModel(win * C, H, L, ncls, dropout=0.0, win_size=win, stride=stride)
    sd = weights_from_seed(m.state_dict(), 505)
    # weights_from_seed draws 1-D tensors in +-0.1 and h0 (L, 1, H) like a matrix: keep h0 well away from zero
    sd['h0'] = torch.from_numpy(np.random.default_rng(506).uniform(-0.5, 0.5, (L, 1, H)).astype(np.float32))
    m.load_state_dict(sd)
    m = m.cuda().eval()
    orc = RealtimeOracle(win * C, H, L, ncls, win, stride)
    orc.load_reference_state(sd)
    orc.eval()
    rng = np.random.default_rng(507)
    B, T = 3, 200                                         # 200 samples -> 47 windows of 14 every 4
    x = torch.from_numpy(rng.standard_normal((B, T, C)).astype(np.float32))
    with torch.no_grad():
        ref = orc(x)
        full = m(x.cuda())
    nw = (T - win) // stride + 1
    assert tuple(ref.shape) == (B, nw, ncls) == tuple(full.shape)
    err = (full.cpu() - ref).abs().max().item()
    assert err <= 1e-4, err
    assert torch.equal(full.argmax(-1).cpu(), ref.argmax(-1))
    dec_h = greedy_decode_batch(torch.log_softmax(full, -1), blank=0)
    dec_o = greedy_ref(torch.log_softmax(ref, -1))
    for a, b in zip(dec_h, dec_o):
        np.testing.assert_array_equal(a.cpu().numpy(), b.numpy())
    dec = StreamingDecoder(m, n_streams=1, use_graph=True)
    xg = x.cuda()
    for b in range(B):
        dec.reset()
        toks = []
        for w in range(nw):
            lg = dec.step(xg[b, w * stride:w * stride + win].reshape(1, -1))
            assert (lg[0] - full[b, w]).abs().max().item() <= 2e-5, (b, w)
            assert (lg[0].cpu() - ref[b, w]).abs().max().item() <= 1e-4, (b, w)
            toks.append(int(dec.token[0]))
        assert toks == ref[b].argmax(-1).tolist()


def test_streaming_latency_budget():
    """hipGraph replay of one 20 ms step (C = 128 channels -> 1792-wide window, H = 128, L = 2, batch 1).  The
    reference reports 2.06 ms per prediction; the bar here is only that the graph path works at
    the config-5 shape and is well under that."""
    from cross_patient_speech_decoding_amd.realtime_sim import RealtimeRNNModel, StreamingDecoder
    torch.manual_seed(0)
    m = RealtimeRNNModel(14 * 128, 128, 2, 11, dropout=0.0).cuda().eval()
    dec = StreamingDecoder(m, n_streams=1, use_graph=True)
    win = torch.randn(1, 14 * 128, device='cuda')
    for _ in range(20):
        dec.step(win)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        dec.step(win)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print(f'streaming step latency (graph replay, incl. window upload): {us:.1f} us')
    assert us < 1000.0


# ----------------------------------------------------------------------------- CTC training
@pytest.mark.parametrize('T,B,C,L', [(13, 5, 11, 3), (47, 64, 11, 3), (20, 7, 5, 9), (6, 3, 70, 2), (30, 4, 11, 0)])
def test_ctc_kernel_vs_torch_cpu(T, B, C, L):
    """Fused log-softmax + CTC loss + gradient against torch's CPU nn.CTCLoss(zero_infinity=True): ragged input
    and target lengths, repeated labels, empty targets, inputs too short for their target (infinite -> zeroed)."""
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    g = torch.Generator().manual_seed(T * 100 + B)
    logits = torch.randn(T, B, C, generator=g) * 2
    targets = torch.randint(1, C, (B, max(L, 1)), generator=g)
    if L >= 2:
        targets[0, 1] = targets[0, 0]                                  # repeated label
    tl = torch.randint(0 if L == 0 else 1, L + 1, (B,), generator=g)
    tl[0] = L
    il = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g)
    il[0] = T
    if B > 2 and L >= 2:
        il[2], tl[2] = 2, L                                            # impossible alignment when 2 < L (+ repeats)
    ref_in = logits.clone().requires_grad_(True)
    ref = torch.nn.CTCLoss(blank=0, zero_infinity=True)(ref_in.log_softmax(2), targets, il, tl)
    ref.backward()
    x = logits.cuda().requires_grad_(True)
    loss = XF.ctc_loss(x, targets.cuda(), il, tl, blank=0, zero_infinity=True)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(ref.detach()), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref_in.grad.numpy(), atol=1e-5, rtol=1e-4)          # fp32 recursions
    # the module form takes log-probs like nn.CTCLoss
    from cross_patient_speech_decoding_amd.realtime_sim.realtime_nn_model import _HipCTCLoss
    l2 = _HipCTCLoss(blank=0, zero_infinity=True)(logits.cuda().log_softmax(2), targets.cuda(), il, tl)
    np.testing.assert_allclose(float(l2), float(ref), rtol=2e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        XF.ctc_loss(logits.cuda(), targets.cuda(), il + T, tl)


def test_ctc_training_step_matches_reference_golden(golden_dir, gemm_precision):
    """Loss, PER and every parameter gradient of the reference's training / validation / test steps."""
    g = np.load(os.path.join(golden_dir, 'realtime_train_small.npz'))
    m, (C, win, stride, H, L, ncls) = build(g)
    m.train()
    batch = tuple(torch.from_numpy(g[k]) for k in ('x', 'targets', 'input_lengths', 'target_lengths'))
    dbatch = (batch[0].cuda(), batch[1].cuda(), batch[2], batch[3])
    loss = m.training_step(dbatch, 0)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-5)
    for k, p in m.named_parameters():
        ref = g['grad.' + k]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)
    m.eval()
    m._xps_logged = {}
    with torch.no_grad():
        vloss = m.validation_step(dbatch, 0)
        wl = ((batch[2] - win) // stride) + 1
        tloss = m.test_step((dbatch[0], dbatch[1], wl, batch[3]), 0)
    np.testing.assert_allclose(float(vloss), float(g['val_loss']), rtol=1e-5)
    np.testing.assert_allclose(float(tloss), float(g['test_loss']), rtol=1e-5)
    np.testing.assert_allclose(float(m._xps_logged['val_PER']), float(g['val_PER']), rtol=1e-6)
    with pytest.raises(RuntimeError):                      # raw sample counts exceed the window count, as in torch
        m.test_step(dbatch, 0)


def test_ctc_model_trains_with_the_trainer():
    """RealtimeRNNModel under the HIP Trainer (4-tuple batches, ([opt], [sched]) optimiser form): the CTC loss of a
    learnable synthetic task falls and the phoneme error rate ends well below the untrained model's."""
    from torch.utils.data import DataLoader, TensorDataset
    from cross_patient_speech_decoding_amd.nn_models.trainer import Trainer, seed_everything
    from cross_patient_speech_decoding_amd.realtime_sim import RealtimeRNNModel
    seed_everything(0)
    rng = np.random.default_rng(0)
    N, T, C, ncls = 256, 110, 8, 6
    targets = rng.integers(1, ncls, (N, 3))
    proto = rng.standard_normal((ncls, C)).astype(np.float32) * 1.5
    x = rng.standard_normal((N, T, C)).astype(np.float32) * 0.3
    for i in range(N):                                     # three 30-sample segments carrying the phoneme patterns
        for j in range(3):
            x[i, 10 + 30 * j: 40 + 30 * j] += proto[targets[i, j]]
    ds = TensorDataset(torch.from_numpy(x), torch.from_numpy(targets), torch.full((N,), T), torch.full((N,), 3))
    model = RealtimeRNNModel(14 * C, 48, 2, ncls, dropout=0.1, learning_rate=1e-2, decay_steps=250)
    tr = Trainer(max_epochs=200, gradient_clip_val=1.0)
    loader = DataLoader(ds, batch_size=N, shuffle=False)
    model.cuda()
    before = tr.validate(model, loader)[0]
    tr.fit(model, loader, loader)
    after = tr.logged_metrics
    print('before', before, 'after', {k: float(v) for k, v in after.items()})
    assert after['train_loss'] < 0.5 * before['val_loss']
    assert after['val_PER'] < 0.5 * before['val_PER'] and after['val_PER'] < 40.0
