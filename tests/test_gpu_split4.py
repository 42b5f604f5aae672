"""XPS_FMT_SPLIT4 operands (include/xps.h: xps_rowmap.fmt): a producer that owns the bf16 hi / lo split of a tensor that only
GEMMs read writes it ONCE (16-byte groups hi[0..3] | lo[0..3]); the tile kernels then stage it without conversion arithmetic.
The contract tested here: products are BIT-IDENTICAL to those from the fp32 operand (every GEMM form, small and 256-tile
kernels), a bias gradient folded from a split4 operand is within 2^-17 relative per element of the fp32 one, the BPTT kernels'
split4 outputs are exactly split4(fp32 outputs), and requests that cannot be served fail loudly."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cross_patient_speech_decoding_amd import _build  # noqa: E402
from cross_patient_speech_decoding_amd._lib import XpsError, call, lib, rowmap  # noqa: E402


@pytest.fixture(scope='module', autouse=True)
def _built():
    _build.build(verbose=False)
    assert torch.cuda.is_available(), 'gpu tests need the MI355X'


@pytest.fixture(autouse=True)
def _bf16x3():
    old = lib().xps_get_gemm_precision()
    assert lib().xps_set_gemm_precision(1) == 0
    yield
    lib().xps_set_gemm_precision(old)


def XF():
    from cross_patient_speech_decoding_amd.nn_models import functional
    return functional


def split4(x, drop_p=0.0, seed=0):
    out = torch.empty_like(x)
    call('xps_split4_f32', x.data_ptr(), out.data_ptr(), x.numel(), float(drop_p), int(seed), XF()._stream())
    return out


def split4_host(x):
    """numpy restatement of the format: groups of four fp32 -> four bf16 hi (round to nearest even) | four bf16 lo."""
    a = x.detach().cpu().numpy().astype(np.float32).reshape(-1, 4)

    def bf16_bits(v):
        u = v.view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)        # RNE (no NaN / inf in the test data)
        return r.astype(np.uint16)

    hi = bf16_bits(a)
    hi_f = (hi.astype(np.uint32) << 16).view(np.float32)
    lo = bf16_bits((a - hi_f).astype(np.float32))
    return np.concatenate([hi, lo], axis=1).reshape(-1).view(np.float32).reshape(x.shape)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def test_split4_kernel_matches_the_host_restatement_and_dropout_values():
    x = rnd(64, 260, seed=1) * torch.logspace(-6, 3, 260).cuda()
    got = split4(x).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), split4_host(x).view(np.uint32))
    # hi + lo reproduces x to 2^-17 relative
    g = got.reshape(-1, 4).view(np.uint16).reshape(-1, 8)
    rec = (g[:, :4].astype(np.uint32) << 16).view(np.float32) + (g[:, 4:].astype(np.uint32) << 16).view(np.float32)
    xr = x.cpu().numpy().reshape(-1, 4)
    assert np.max(np.abs(rec - xr) / np.abs(xr)) <= 2.0 ** -16
    # with dropout: split4 of exactly what xps_dropout_f32 writes
    dropped = torch.empty_like(x)
    call('xps_dropout_f32', x.data_ptr(), dropped.data_ptr(), None, x.numel(), 0.3, 1234, XF()._stream())
    np.testing.assert_array_equal(split4(x, 0.3, 1234).cpu().numpy().view(np.uint32), split4_host(dropped).view(np.uint32))
    assert (dropped == 0).float().mean().item() > 0.2


# (M, N, K): edge tiles, 64-row tiles, 128-row tiles, and shapes the 256 x 256 kernels take (>= 192 big tiles for nt / nn)
SHAPES = [(37, 72, 100), (300, 384, 256), (4096, 100, 128), (4096, 768, 256), (8192, 1536, 512)]


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_nt_nn_products_from_split4_operands_are_bit_identical(M, N, K):
    F = XF()
    A, Bt, Bn = rnd(M, K, seed=M), rnd(N, K, seed=N + 1), rnd(K, N, seed=K + 2)
    bias = rnd(N, seed=5)
    A4, Bt4, Bn4 = split4(A), split4(Bt), split4(Bn)
    ref = F.gemm_nt(A, Bt, torch.empty(M, N, device='cuda'), M, N, K, bias=bias)
    for fa, fb in [(1, 0), (0, 1), (1, 1)]:
        out = F.gemm_nt(A4 if fa else A, Bt4 if fb else Bt, torch.empty(M, N, device='cuda'), M, N, K, bias=bias,
                        ra=rowmap(K, fmt=fa), rb=rowmap(K, fmt=fb))
        assert torch.equal(out, ref), (fa, fb)
    ref = F.gemm_nn(A, Bn, torch.empty(M, N, device='cuda'), M, N, K)
    for fa, fb in [(1, 0), (0, 1), (1, 1)]:
        out = F.gemm_nn(A4 if fa else A, Bn4 if fb else Bn, torch.empty(M, N, device='cuda'), M, N, K,
                        ra=rowmap(K, fmt=fa), rb=rowmap(N, fmt=fb))
        assert torch.equal(out, ref), (fa, fb)


@pytest.mark.parametrize('M,N,K', [(2048, 256, 384), (40960, 1024, 1536)])
def test_nn2_and_nt_multi_from_split4_operands_are_bit_identical(M, N, K):
    F = XF()
    A1, A2 = rnd(M, K, seed=1), rnd(M, K, seed=2)
    B1, B2 = rnd(K, N, seed=3, scale=0.05), rnd(K, N, seed=4, scale=0.05)

    def nn2(a1, a2, b1, b2, fa, fb):
        out = torch.empty(M, N, device='cuda')
        ra, rb, rc = rowmap(K, fmt=fa), rowmap(N, fmt=fb), rowmap(N)
        call('xps_gemm_nn2_f32', a1.data_ptr(), b1.data_ptr(), K, a2.data_ptr(), b2.data_ptr(), K, C.byref(ra), C.byref(rb),
             out.data_ptr(), C.byref(rc), M, N, 0, F._stream())
        return out

    ref = nn2(A1, A2, B1, B2, 0, 0)
    assert torch.equal(nn2(split4(A1), split4(A2), B1, B2, 1, 0), ref)
    assert torch.equal(nn2(split4(A1), split4(A2), split4(B1), split4(B2), 1, 1), ref)
    del A2, B1, B2
    # nt_multi: one A against two weight matrices (the input projections of both directions)
    W = [rnd(N, K, seed=7, scale=0.05), rnd(N, K, seed=8, scale=0.05)]
    bs = [rnd(N, seed=9), rnd(N, seed=10)]

    def ntm(a, w, fa, fb):
        outs = [torch.empty(M, N, device='cuda') for _ in range(2)]
        ra, rb, rc = rowmap(K, fmt=fa), rowmap(K, fmt=fb), rowmap(N)
        call('xps_gemm_nt_multi_f32', a.data_ptr(), C.byref(ra), F._ptr_array(w), C.byref(rb), F._ptr_array(outs), C.byref(rc),
             F._ptr_array(bs), 2, M, N, K, F._stream())
        return outs

    ref = ntm(A1, W, 0, 0)
    for fa, fb in [(1, 0), (0, 1), (1, 1)]:
        got = ntm(split4(A1) if fa else A1, [split4(w) for w in W] if fb else W, fa, fb)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (fa, fb)


@pytest.mark.parametrize('Kr,M,N', [(4096, 384, 256), (4096, 100, 640), (8192, 1536, 1024), (6144, 512, 512)])
def test_grouped_weight_gradients_from_split4_operands(Kr, M, N):
    """products bit-identical (128-tile and 256-tile launches, with split-K slabs); the bias gradient sums hi + lo"""
    F = XF()
    A, B = rnd(Kr, M, seed=11), rnd(Kr, N, seed=12)

    def run(a, b, fa, fb):
        out, cs = torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda')
        F.gemm_tn_grouped([F.tn_problem(a, b, out, M, N, Kr, ra=rowmap(M, fmt=fa), rb=rowmap(N, fmt=fb), colsum_out=cs)], 'cuda')
        torch.cuda.synchronize()
        return out, cs

    ref, cs_ref = run(A, B, 0, 0)
    out, cs = run(A, split4(B), 0, 1)
    assert torch.equal(out, ref) and torch.equal(cs, cs_ref)
    for fb in (0, 1):
        out, cs = run(split4(A), split4(B) if fb else B, 1, fb)
        assert torch.equal(out, ref)
        bound = 2.0 ** -16 * A.abs().sum(0) + 1e-6                     # per element 2^-17 relative; fp32 summation noise on top
        assert bool(((cs - cs_ref).abs() <= bound).all())


def test_requests_that_cannot_be_served_fail_loudly():
    F = XF()
    A, B = rnd(64, 30, seed=1), rnd(32, 30, seed=2)                         # K = 30: no whole 16-byte groups along k
    with pytest.raises(XpsError):
        F.gemm_nt(A, B, torch.empty(64, 32, device='cuda'), 64, 32, 30, ra=rowmap(30, fmt=1))
    A, B = rnd(8, 32, seed=1), rnd(32, 32, seed=2)                          # few-row kernel: fp32 operands only
    with pytest.raises(XpsError):
        F.gemm_nt(A, B, torch.empty(8, 32, device='cuda'), 8, 32, 32, ra=rowmap(32, fmt=1))
    lib().xps_set_gemm_precision(0)                                         # fp32-MFMA mode has no split operands
    A, B = rnd(64, 32, seed=1), rnd(32, 32, seed=2)
    with pytest.raises(XpsError):
        F.gemm_nt(A, B, torch.empty(64, 32, device='cuda'), 64, 32, 32, ra=rowmap(32, fmt=1))
    assert not F.split4_supported(20, 2048, 128, 2)
    lib().xps_set_gemm_precision(1)
    assert F.split4_supported(20, 2048, 128, 2) and F.split4_supported(20, 2048, 512, 2)
    assert not F.split4_supported(20, 2048, 200, 2)                         # generic kernels: fp32 outputs only
    # policy: the BPTT kernels are asked for split outputs only where the 256-tile kernels read them
    assert F.split4_wanted(20, 2048, 512, 2) and not F.split4_wanted(20, 2048, 128, 2)


@pytest.mark.parametrize('T,B,H,ndir,drop', [(6, 300, 128, 2, None), (5, 64, 64, 1, None), (6, 300, 128, 2, (0.3, 77)),
                                             (4, 512, 512, 2, None), (3, 256, 500, 2, None)])
def test_bptt_split4_outputs_equal_the_split_of_the_fp32_outputs(T, B, H, ndir, drop):
    F = XF()
    gi = rnd(ndir, T, B, 3 * H, seed=1)
    w_hh = [rnd(3 * H, H, seed=2 + d, scale=H ** -0.5) for d in range(ndir)]
    b_hh = [rnd(3 * H, seed=5 + d, scale=0.1) for d in range(ndir)]
    y_ext, saved = F._gru_forward(gi, w_hh, b_hh, None, T, B, H, ndir, True)
    dy, dhn = rnd(T, B, ndir * H, seed=8), rnd(ndir, B, H, seed=9)
    assert F.split4_supported(T, B, H, ndir)
    dgi, dghn, dh0 = F._gru_backward(dy, dhn, y_ext, saved, w_hh, T, B, H, ndir, True, drop)
    sgi, sghn, sh0 = F._gru_backward(dy, dhn, y_ext, saved, w_hh, T, B, H, ndir, True, drop, split4=True)
    torch.cuda.synchronize()
    F.check_gru_status()
    assert torch.equal(sh0, dh0)
    np.testing.assert_array_equal(sgi.cpu().numpy().view(np.uint32), split4_host(dgi).view(np.uint32))
    np.testing.assert_array_equal(sghn.cpu().numpy().view(np.uint32), split4_host(dghn).view(np.uint32))


@pytest.mark.parametrize('mode', ['persistent', 'steps'])
@pytest.mark.parametrize('T,B,H,ndir,drop,save', [(4, 512, 512, 2, (0.3, 99), True), (3, 300, 448, 1, None, True),
                                                  (2, 256, 500, 2, (0.5, 7), False), (1, 2048, 512, 1, None, True)])
def test_forward_kernel_images_equal_the_split_passes_bitwise(T, B, H, ndir, drop, save, mode):
    """xps_gru_seq_fwd_images_f32: the XPS_FMT_SPLIT4 images the cluster forward kernel writes from its epilogue (y_ext, all T + 2
    slots; dropout(y)) equal xps_split4_f32 over the finished tensors bit for bit, and y_ext / saved equal the plain launch's."""
    F = XF()
    old = lib().xps_get_gru_cluster_mode()
    F.set_gru_cluster_mode(mode)
    try:
        assert lib().xps_gru_seq_fwd_images_supported(T, B, H, ndir)
        gi = rnd(ndir, T, B, 3 * H, seed=1)
        w_hh = [rnd(3 * H, H, seed=2 + d, scale=H ** -0.5) for d in range(ndir)]
        b_hh = [rnd(3 * H, seed=5 + d, scale=0.1) for d in range(ndir)]
        y_ref, saved_ref = F._gru_forward(gi, w_hh, b_hh, None, T, B, H, ndir, save)
        y_ext, saved, y_split, yd_split = F._gru_forward_images(gi, w_hh, b_hh, T, B, H, ndir, save, True, drop)
        torch.cuda.synchronize()
        F.check_gru_status()
        assert torch.equal(y_ext, y_ref)
        if save:
            assert torch.equal(saved, saved_ref)
            assert torch.equal(y_split.view(torch.int32), split4(y_ref).view(torch.int32))
        else:
            assert saved is None and y_split is None
        want = split4(y_ref[1:T + 1].contiguous(), *(drop or (0.0, 0)))
        assert torch.equal(yd_split.view(torch.int32), want.view(torch.int32))
    finally:
        lib().xps_set_gru_cluster_mode(old)


def test_forward_images_are_refused_off_the_cluster_path():
    F = XF()
    assert not lib().xps_gru_seq_fwd_images_supported(4, 512, 128, 2)              # register-resident kernels
    lib().xps_set_gemm_precision(0)
    assert not lib().xps_gru_seq_fwd_images_supported(4, 512, 512, 2)              # fp32 products: no split4 operands
    lib().xps_set_gemm_precision(1)
    gi = rnd(1, 2, 64, 3 * 128, seed=1)
    with pytest.raises(XpsError, match='images'):
        F._gru_forward_images(gi, [rnd(384, 128)], [rnd(384)], 2, 64, 128, 1, True, False, None)


def test_layer_with_forward_kernel_images_equals_the_split_passes(monkeypatch):
    """Two-layer bidirectional encoder, training mode: XPS_FWD_IMAGES=1 (images from the recurrence kernel) against =0 (one
    xps_split4_f32 pass per image): outputs and every gradient bit-identical."""
    from cross_patient_speech_decoding_amd.nn_models.models import EncoderRNN
    F = XF()
    T, B, H, In = 8, 512, 512, 64
    assert F.hprev_split_wanted(T, B, H, 2) and F.fwd_ysplit_wanted(T, B, H, 2)
    assert lib().xps_gru_seq_fwd_image_exchange_supported(T, B, H, 2) and not lib().xps_gru_seq_fwd_image_exchange_supported(T, 300, 448, 1)
    torch.manual_seed(3)
    enc = EncoderRNN(In, H, 2, dropout=0.3).cuda().train()
    x = rnd(T, B, In, seed=5)
    wy, wl = rnd(T, B, 2 * H, seed=6), rnd(B, H, seed=7)

    def run(flag):
        monkeypatch.setenv('XPS_FWD_IMAGES', flag)
        assert F.fwd_images_wanted(T, B, H, 2) == (flag == '1')
        F._DROP_COUNTER[0] = 4321
        enc.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y, last = enc.forward_tm_last(xg)
        ((y * wy).sum() + (last * wl).sum()).backward()
        torch.cuda.synchronize()
        F.check_gru_status()
        return [y.detach().clone(), last.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in enc.parameters()]

    base = run('0')                                      # (default: y_ext's image is the forward launch's exchange buffer)
    for a, b in zip(base, run('1')):
        assert torch.equal(a, b)
    monkeypatch.setenv('XPS_FWD_YSPLIT', '0')            # ring-buffer exchange + one xps_split4_f32 pass per image
    assert not F.fwd_ysplit_wanted(T, B, H, 2)
    for a, b in zip(base, run('0')):
        assert torch.equal(a, b)


@pytest.mark.parametrize('H,In', [(512, 100), (500, 32), (128, 64)])
def test_encoder_stack_with_split4_tensors_equals_the_fp32_tensor_path(H, In, monkeypatch):
    """Two-layer bidirectional encoder in TRAINING mode (inter-layer dropout on), XPS_SPLIT4=1 (dgi / dghn, the dropped
    layer output and the large W_ih travel as XPS_FMT_SPLIT4 operands) against XPS_SPLIT4=0 (fp32 tensors everywhere), same
    dropout seeds: outputs, input gradient and every weight gradient BIT-IDENTICAL; bias gradients (column sums of a split4
    operand: hi + lo) within 2^-16 of sum |dgi|."""
    from cross_patient_speech_decoding_amd.nn_models.models import EncoderRNN
    F = XF()
    T, B = 6, 4096
    torch.manual_seed(3)
    enc = EncoderRNN(In, H, 2, dropout=0.3).cuda().train()
    x = rnd(T, B, In, seed=5)
    wy, wl = rnd(T, B, 2 * H, seed=6), rnd(B, H, seed=7)

    def run(flag):
        monkeypatch.setenv('XPS_SPLIT4', flag)
        F._split4_ok.clear()
        F._DROP_COUNTER[0] = 12345
        enc.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y, last = enc.forward_tm_last(xg)
        ((y * wy).sum() + (last * wl).sum()).backward()
        torch.cuda.synchronize()
        F.check_gru_status()
        return y.detach().clone(), last.detach().clone(), xg.grad.clone(), {k: p.grad.clone() for k, p in enc.named_parameters()}

    y0, l0, dx0, g0 = run('0')
    y1, l1, dx1, g1 = run('1')
    F._split4_ok.clear()
    assert torch.equal(y1, y0) and torch.equal(l1, l0) and torch.equal(dx1, dx0)
    for k in g0:
        if 'bias' in k:
            scale = max(1.0, g0[k].abs().max().item())
            assert (g1[k] - g0[k]).abs().max().item() <= 2e-5 * scale + 2.0 ** -16 * T * B, k
        else:
            assert torch.equal(g1[k], g0[k]), k
    # the bias gradients are not just close, they are the exact column sums of hi + lo: within 1e-6 relative of the fp32 sums here
    rel = max(((g1[k] - g0[k]).abs().max() / g0[k].abs().max()).item() for k in g0 if 'bias' in k)
    assert rel <= 1e-5


@pytest.mark.parametrize('M,N,K,nprob', [(40960, 384, 256, 2), (40960, 384, 100, 2), (4100, 96, 64, 3), (8192, 256, 32, 1),
                                         (5000, 160, 200, 2)])
def test_weight_stationary_projection_equals_the_tile_kernels_bitwise(M, N, K, nprob, monkeypatch):
    """proj_ws_kernel (weights of a 256-column slice resident in registers, A streamed once per slice; csrc/xps_gemm.hip) issues
    the same products in the same order per accumulator as the tile kernels: identical bits, for fp32 and XPS_FMT_SPLIT4
    operands, ragged row counts, K that is not a multiple of 32, column slices that straddle problems or are partly idle."""
    F = XF()
    A = rnd(M, K, seed=1)
    W = [rnd(N, K, seed=10 + i, scale=K ** -0.5) for i in range(nprob)]
    bs = [rnd(N, seed=20 + i) for i in range(nprob)]

    def run(a, w, fa, fb):
        outs = [torch.full((M, N), float('nan'), device='cuda') for _ in range(nprob)]
        ra, rb, rc = rowmap(K, fmt=fa), rowmap(K, fmt=fb), rowmap(N)
        call('xps_gemm_nt_multi_f32', a.data_ptr(), C.byref(ra), F._ptr_array(w), C.byref(rb), F._ptr_array(outs), C.byref(rc),
             F._ptr_array(bs), nprob, M, N, K, F._stream())
        torch.cuda.synchronize()
        return outs

    monkeypatch.setenv('XPS_PROJ_WS', '0')
    ref = run(A, W, 0, 0)
    monkeypatch.setenv('XPS_PROJ_WS', '1')
    for fa, fb in [(0, 0), (1, 0), (1, 1)]:
        got = run(split4(A) if fa else A, [split4(w) for w in W] if fb else W, fa, fb)
        for i in range(nprob):
            assert torch.equal(got[i], ref[i]), (fa, fb, i)
    # and against fp64 (an indexing mistake common to both paths would pass the comparison above)
    ref64 = A.double() @ W[-1].double().T + bs[-1].double()
    bound = 1.6e-5 * (A.abs().double() @ W[-1].abs().double().T) + 1e-6
    assert bool(((ref[-1].double() - ref64).abs() <= bound).all())
