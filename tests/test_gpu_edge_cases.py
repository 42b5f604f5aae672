"""Edge cases of the HIP path against the CPU oracle: single trial, a single conv window (T' = 1), three
encoder layers, two decoder layers with the fused-decoder hidden size, ragged tiles, and ABI refusals."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from weights import weights_from_seed  # noqa: E402


def _pair(args, kw, seed):
    from oracle.seq2seq_oracle import Seq2SeqOracle
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    orc = Seq2SeqOracle(*args, **kw)
    sd = weights_from_seed(orc.state_dict(), seed)
    orc.load_state_dict(sd)
    hip = Seq2SeqRNN(*args[:11], 'gru', 1e-3, 1e-5, activation=kw.get('activation', True))
    hip.load_state_dict(sd)
    return orc, hip.cuda()


@pytest.mark.parametrize('B,T,args', [
    (1, 40, (6, 8, 16, 9, 2, 1, 4, 4, 0, 0.0, 0.0)),            # a single trial
    (9, 10, (6, 8, 16, 9, 1, 1, 10, 10, 0, 0.0, 0.0)),          # T' = 1: one conv window, one GRU step
    (17, 33, (5, 7, 64, 9, 3, 1, 3, 2, 0, 0.0, 0.0)),           # 3 encoder layers, overlapping windows, fused decoder H=64
    (33, 36, (4, 6, 128, 9, 1, 2, 6, 6, 0, 0.0, 0.0)),          # 2 decoder layers at H=128 -> composed decoder path
    (20, 30, (3, 5, 12, 4, 2, 1, 5, 5, 0, 0.0, 0.0)),           # 4 classes, H not a multiple of 16
])
def test_shapes_vs_oracle(B, T, args):
    torch.set_num_threads(4)
    nc = args[3]
    orc, hip = _pair(args, dict(activation=True), seed=B + T)
    rng = np.random.default_rng(B)
    x = torch.from_numpy(rng.standard_normal((B, T, args[0])).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, nc, (B, 3)))
    orc.eval(); hip.eval()
    with torch.no_grad():
        ref = orc(x, y, teacher_forcing_ratio=0)
        out = hip(x.cuda(), y.cuda(), teacher_forcing_ratio=0).cpu()
    assert out.shape == ref.shape == (B, 3, nc)
    assert (out - ref).abs().max().item() <= 1e-4
    assert torch.equal(out.argmax(-1), ref.argmax(-1))
    if B > 1:                                           # BatchNorm batch statistics need > 1 value per channel
        orc.train(); hip.train()
        coins = [True, True, False]
        r = orc(x, y, coins=coins)
        torch.nn.functional.cross_entropy(r.reshape(-1, nc), y.reshape(-1)).backward()
        o = hip(x.cuda(), y.cuda(), coins=coins)
        hip.criterion(o.view(-1, nc), y.cuda().view(-1)).backward()
        assert (o.detach().cpu() - r.detach()).abs().max().item() <= 1e-4
        for (k, p), q in zip(hip.named_parameters(), orc.parameters()):
            if k == 'temporal_conv.conv.bias':
                continue
            tol = 3e-5 * max(1.0, q.grad.abs().max().item()) + 3e-3 * q.grad.abs().max().item()
            assert (p.grad.cpu() - q.grad).abs().max().item() <= tol, k


def test_sequence_shorter_than_kernel_raises():
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    m = Seq2SeqRNN(4, 6, 16, 9, 1, 1, 10, 10).cuda().eval()
    with pytest.raises(ValueError, match='shorter than the kernel'):
        m(torch.zeros(2, 5, 4, device='cuda'))


def test_lstm_is_refused_like_the_broken_reference_branch():
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    with pytest.raises(NotImplementedError):
        Seq2SeqRNN(4, 6, 16, 9, 1, 1, 4, 4, model_type='lstm')
    with pytest.raises(ValueError, match='model_type must be one of'):
        from cross_patient_speech_decoding_amd.nn_models import EncoderRNN
        EncoderRNN(4, 8, 1, model_type='rnn')


def test_abi_refuses_unsupported_shapes():
    import ctypes as C
    from cross_patient_speech_decoding_amd import _lib
    lib = _lib.lib()
    assert lib.xps_decoder_supported(128, 9, 3) == 1
    assert lib.xps_decoder_supported(500, 9, 3) == 0 and lib.xps_decoder_supported(128, 40, 3) == 0
    t = torch.zeros(16, device='cuda')
    rc = lib.xps_gru_seq_fwd_f32(t.data_ptr(), None, None, None, t.data_ptr(), None, 1, 1, 4, 3, None, 0, None)
    assert rc == -1 and b'null' in lib.xps_last_error()
    with pytest.raises(_lib.XpsError, match='1..8 streams'):
        _lib.call('xps_gemv_f32', t.data_ptr(), t.data_ptr(), None, t.data_ptr(), 4, 4, 9, None)


def test_alignment_edge_cases():
    """One shared condition only; a condition with a single trial; 1-D integer labels; float64 input."""
    import cross_patient_speech_decoding_amd.alignment as A
    from oracle import align_oracle as ao
    rng = np.random.default_rng(4)
    Xa = rng.standard_normal((12, 30, 5)); ya = np.array([1] * 6 + [2] * 5 + [3])
    Xb = rng.standard_normal((10, 30, 4)); yb = np.array([3] * 4 + [7] * 6)          # only '3' is shared
    ref = ao.AlignCCAOracle().fit(Xa, Xb, ya, yb)
    al = A.AlignCCA(); al.fit(Xa, Xb, ya, yb)
    np.testing.assert_allclose(al.canon_corrs, ref.canon_corrs, atol=1e-8)
    out, exp = al.transform(Xb), ref.transform(Xb)
    assert np.abs(out - exp).max() <= 1e-6 * np.abs(exp).max()
    avg = A.cnd_avg(Xa, A.label2str(ya))
    np.testing.assert_array_equal(avg[2], Xa[11])                                    # single-trial condition
