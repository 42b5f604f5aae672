"""GPU parity tests of the HIP kernels behind the seq2seq path, called through the C ABI
(ctypes) and compared with the CPU library calls the reference makes (torch.nn on CPU /
fp64 matmul).  Tolerances are stated per test; the north-star bar is logits <= 1e-4 abs."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cross_patient_speech_decoding_amd import _build  # noqa: E402
from cross_patient_speech_decoding_amd._lib import rowmap  # noqa: E402


@pytest.fixture(scope='module', autouse=True)
def _built():
    _build.build(verbose=False)
    assert torch.cuda.is_available(), 'gpu tests need the MI355X'


def XF():
    from cross_patient_speech_decoding_amd.nn_models import functional
    return functional


def dev(t):
    return t.to('cuda')


# ----------------------------------------------------------------------------- GEMM
# The tile kernels run in one of two product precisions (include/xps.h xps_set_gemm_precision): fp32 MFMA (exact fp32
# fma chains: a few ulp of the fp64 result) or bf16 split products (hi/lo split of both operands, three bf16 MFMAs:
# per product |error| <= ~3 * 2^-18 |a||b| = 1.2e-5 |a||b|).  Every GEMM test runs in both and bounds the error by the
# mode's constant times sum |a||b| — an indexing mistake would show up as an O(1) multiple of that sum.
TOL = {0: 1.0, 1: 40.0}           # multiplier on the fp32 tolerances below (4e-7 * 40 = 1.6e-5)


@pytest.fixture(params=[0, 1], ids=['fp32', 'bf16x3'])
def prec(request):
    from cross_patient_speech_decoding_amd._lib import lib
    old = lib().xps_get_gemm_precision()
    assert lib().xps_set_gemm_precision(request.param) == 0
    yield TOL[request.param]
    lib().xps_set_gemm_precision(old)


def test_gemm_precision_switch_rejects_unknown_modes():
    from cross_patient_speech_decoding_amd._lib import lib
    old = lib().xps_get_gemm_precision()
    assert old in (0, 1)
    assert lib().xps_set_gemm_precision(7) != 0
    assert lib().xps_get_gemm_precision() == old


@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (5, 9, 16), (37, 70, 100), (130, 129, 33), (300, 384, 256),
                                   (257, 100, 640), (64, 27, 6), (10, 384, 128), (16, 130, 77), (13, 70, 1031),
                                   (17, 64, 40)])
def test_gemm_nt_nn_tn_vs_fp64(M, N, K, prec):
    g = torch.Generator().manual_seed(M * 1000 + N * 10 + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g)          # asymmetric random data (guide: never symmetric B)
    bias = torch.randn(N, generator=g)
    ref = (A.double() @ Bm.double().T + bias.double())
    xf = XF()
    out = torch.empty(M, N, device='cuda')
    xf.gemm_nt(dev(A), dev(Bm), out, M, N, K, bias=dev(bias))
    scale = (A.abs().double() @ Bm.abs().double().T + bias.abs().double())
    assert ((out.cpu().double() - ref).abs() <= 4e-7 * prec * scale + 1e-30).all()
    # accumulate
    xf.gemm_nt(dev(A), dev(Bm), out, M, N, K, accumulate=True)
    ref2 = ref + A.double() @ Bm.double().T
    assert ((out.cpu().double() - ref2).abs() <= 1e-6 * prec * scale + 1e-30).all()
    # NN: C = A (M x K) @ B (K x N)
    Bk = Bm.T.contiguous()
    out2 = torch.empty(M, N, device='cuda')
    xf.gemm_nn(dev(A), dev(Bk), out2, M, N, K)
    ref_nn = A.double() @ Bk.double()
    assert ((out2.cpu().double() - ref_nn).abs() <= 4e-7 * prec * scale + 1e-30).all()
    # TN: C = At^T @ B, At (K x M), B (K x N)
    At = A.T.contiguous()
    out3 = torch.empty(M, N, device='cuda')
    xf.gemm_tn(dev(At), dev(Bk), out3, M, N, K)
    assert ((out3.cpu().double() - ref_nn).abs() <= 4e-7 * prec * scale + 1e-30).all()


@pytest.mark.parametrize('M,N,K1,K2', [(9, 100, 48, 48), (16, 65, 30, 7), (200, 100, 96, 96), (40, 33, 12, 50)])
def test_gemm_nn2_two_operand_pairs(M, N, K1, K2, prec):
    """C = A1 B1 + A2 B2 in one launch (both directions of a bidirectional layer's input gradient)."""
    import ctypes as C
    from cross_patient_speech_decoding_amd._lib import call
    g = torch.Generator().manual_seed(M + N + K1)
    A1, B1 = torch.randn(M, K1, generator=g), torch.randn(K1, N, generator=g)
    A2, B2 = torch.randn(M, K2, generator=g), torch.randn(K2, N, generator=g)
    if K1 != K2:                      # the two pairs share row maps: pad the leading dimensions to a common one
        ld = max(K1, K2)
        A1p, A2p = torch.zeros(M, ld), torch.zeros(M, ld)
        A1p[:, :K1], A2p[:, :K2] = A1, A2
    else:
        ld, A1p, A2p = K1, A1, A2
    a1, b1, a2, b2 = dev(A1p), dev(B1), dev(A2p), dev(B2)
    out = torch.full((M, N), 3.0, device='cuda')
    ra, rb, rc = rowmap(ld), rowmap(N), rowmap(N)
    call('xps_gemm_nn2_f32', a1.data_ptr(), b1.data_ptr(), K1, a2.data_ptr(), b2.data_ptr(), K2,
         C.byref(ra), C.byref(rb), out.data_ptr(), C.byref(rc), M, N, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = A1.double() @ B1.double() + A2.double() @ B2.double() + 3.0
    scale = A1.abs().double() @ B1.abs().double() + A2.abs().double() @ B2.abs().double() + 3.0
    assert ((out.cpu().double() - ref).abs() <= 1e-6 * prec * scale).all()


def test_gemm_tn_long_k_split(prec):
    g = torch.Generator().manual_seed(5)
    K, M, N = 20000, 48, 20
    At, Bk = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    out = torch.full((M, N), 7.0, device='cuda')
    XF().gemm_tn(dev(At), dev(Bk), out, M, N, K, accumulate=True)
    ref = At.double().T @ Bk.double() + 7.0
    np.testing.assert_allclose(out.cpu().double().numpy(), ref.numpy(), rtol=0, atol=(2e-4 * np.sqrt(K) / 100 + 1e-3) * (1 if prec == 1.0 else 4))
    # determinism: bitwise identical on a second run
    out_b = torch.full((M, N), 7.0, device='cuda')
    XF().gemm_tn(dev(At), dev(Bk), out_b, M, N, K, accumulate=True)
    assert torch.equal(out, out_b)


def test_gemm_rowmaps_conv_windows(prec):
    """NT GEMM over strided convolution windows with a time-major result."""
    g = torch.Generator().manual_seed(9)
    B, T, Cin, F, k, s = 3, 23, 5, 7, 4, 3
    x = torch.randn(B, T, Cin, generator=g)
    w = torch.randn(F, Cin, k, generator=g)
    bias = torch.randn(F, generator=g)
    Tp = (T - k) // s + 1
    ref = torch.nn.functional.conv1d(x.permute(0, 2, 1), w, bias, stride=s).permute(2, 0, 1)     # (T', B, F)
    w2 = w.permute(0, 2, 1).contiguous().view(F, k * Cin)
    y = torch.empty(Tp, B, F, device='cuda')
    XF().gemm_nt(dev(x), dev(w2), y, Tp * B, F, k * Cin, bias=dev(bias),
                 ra=rowmap(s * Cin, rpg=Tp, gs=T * Cin), rc=rowmap(B * F, rpg=Tp, gs=F))
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=2e-5 if prec == 1.0 else 2e-4)


def test_colsum_and_transpose():
    g = torch.Generator().manual_seed(2)
    X = torch.randn(1000, 77, generator=g)
    xf = XF()
    s = torch.empty(77, device='cuda'); s2 = torch.empty(77, device='cuda')
    xf.colsum(dev(X), 1000, 77, out=s, out_sq=s2)
    np.testing.assert_allclose(s.cpu().numpy(), X.double().sum(0).numpy(), atol=2e-4)
    np.testing.assert_allclose(s2.cpu().numpy(), (X.double() ** 2).sum(0).numpy(), rtol=1e-5)
    t = xf.transpose(dev(X), 1000, 77)
    assert torch.equal(t.cpu(), X.T)


# ----------------------------------------------------------------------------- GRU
def _cpu_gru(In, H, ndir, seed):
    torch.manual_seed(seed)
    return torch.nn.GRU(In, H, 1, batch_first=False, bidirectional=(ndir == 2))


def _weights(gru, ndir):
    out = []
    for d in range(ndir):
        sfx = '_l0' + ('_reverse' if d else '')
        out += [getattr(gru, n + sfx).detach().clone() for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    return out


@pytest.mark.parametrize('T,B,In,H,ndir', [(1, 5, 6, 16, 1), (7, 5, 6, 16, 2), (4, 33, 9, 20, 2), (20, 40, 100, 128, 2),
                                           (3, 17, 8, 6, 2), (5, 16, 12, 500, 1), (6, 21, 10, 64, 2), (9, 130, 24, 128, 1), (4, 150, 40, 256, 2), (3, 70, 24, 200, 2), (2, 33, 16, 132, 1)])
def test_gru_layer_forward_backward_vs_torch_cpu(T, B, In, H, ndir, gemm_precision):
    torch.set_num_threads(4)
    gru = _cpu_gru(In, H, ndir, seed=T * 100 + H)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, B, In, generator=g)
    x_ref = x.clone().requires_grad_(True)
    y_ref, hn_ref = gru(x_ref)
    wt = torch.randn(T, B, ndir * H, generator=g)
    (y_ref * wt).sum().backward()

    xf = XF()
    ws = [dev(w).requires_grad_(True) for w in _weights(gru, ndir)]
    xg = dev(x).requires_grad_(True)
    y, hn = xf.GRULayerFn.apply(xg, ndir, xf.HN_STACK, *ws)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(hn.detach().cpu().numpy(), hn_ref.detach().numpy(), atol=2e-5)
    (y * dev(wt)).sum().backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), x_ref.grad.numpy(), atol=5e-5, rtol=1e-4)
    names = []
    for d in range(ndir):
        sfx = '_l0' + ('_reverse' if d else '')
        names += [n + sfx for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    for w, n in zip(ws, names):
        ref = getattr(gru, n).grad.numpy()
        tol = 2e-4 * max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(w.grad.cpu().numpy(), ref, atol=tol, rtol=1e-3, err_msg=n)


def test_gru_layer_final_state_gradient_only():
    """Only h_n is consumed (the seq2seq encoder's top layer): the backward gets dhn and NO dy buffer."""
    torch.set_num_threads(4)
    T, B, In, H = 6, 19, 10, 64
    gru = _cpu_gru(In, H, 2, seed=11)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(T, B, In, generator=g)
    wt = torch.randn(2, B, H, generator=g)
    x_ref = x.clone().requires_grad_(True)
    _, hn_ref = gru(x_ref)
    (hn_ref * wt).sum().backward()
    xf = XF()
    ws = [dev(w).requires_grad_(True) for w in _weights(gru, 2)]
    xg = dev(x).requires_grad_(True)
    _, hn = xf.GRULayerFn.apply(xg, 2, xf.HN_STACK, *ws)
    (hn * dev(wt)).sum().backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), x_ref.grad.numpy(), atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(ws[1].grad.cpu().numpy(), gru.weight_hh_l0.grad.numpy(), atol=1e-4, rtol=1e-3)
    np.testing.assert_allclose(ws[4].grad.cpu().numpy(), gru.weight_ih_l0_reverse.grad.numpy(), atol=1e-4, rtol=1e-3)


def test_gru_layer_summed_final_state():
    """HN_SUM: h_fwd(T-1) + h_bwd(0) as one (B, H) tensor whose gradient reaches both directions."""
    torch.set_num_threads(4)
    T, B, In, H = 5, 21, 12, 64
    gru = _cpu_gru(In, H, 2, seed=5)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(T, B, In, generator=g)
    wt = torch.randn(B, H, generator=g)
    x_ref = x.clone().requires_grad_(True)
    _, hn_ref = gru(x_ref)
    ((hn_ref[0] + hn_ref[1]) * wt).sum().backward()
    xf = XF()
    ws = [dev(w).requires_grad_(True) for w in _weights(gru, 2)]
    xg = dev(x).requires_grad_(True)
    y, last = xf.GRULayerFn.apply(xg, 2, xf.HN_SUM, *ws)
    assert tuple(last.shape) == (B, H)
    np.testing.assert_allclose(last.detach().cpu().numpy(), (hn_ref[0] + hn_ref[1]).detach().numpy(), atol=2e-5)
    (last * dev(wt)).sum().backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), x_ref.grad.numpy(), atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(ws[1].grad.cpu().numpy(), gru.weight_hh_l0.grad.numpy(), atol=1e-4, rtol=1e-3)
    y2, none = xf.GRULayerFn.apply(xg, 2, xf.HN_NONE, *ws)
    assert none is None and torch.equal(y2, y)


def test_dropout_kernel_statistics_and_backward():
    xf = XF()
    x = torch.randn(64, 1000, device='cuda').requires_grad_(True)
    torch.manual_seed(3)
    out = xf.dropout(x, 0.3, True)
    keep = (out != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01
    nz = out != 0
    np.testing.assert_allclose(out[nz].detach().cpu().numpy(), (x[nz] / 0.7).detach().cpu().numpy(), rtol=1e-6)
    out.sum().backward()
    np.testing.assert_allclose(x.grad[nz].cpu().numpy(), 1 / 0.7, rtol=1e-6)
    assert (x.grad[~nz] == 0).all()
    out2 = xf.dropout(x, 0.3, True)                   # a new call draws a new mask
    assert ((out2 != 0) != nz).float().mean().item() > 0.2
    m = xf.dropout_mask((257, 33), 0.5, 'cuda')        # odd element count
    assert set(m.unique().tolist()) <= {0.0, 1.0} and abs(m.mean().item() - 0.5) < 0.02
    assert xf.dropout(x, 0.3, False) is x


def test_dropout_decisions_are_a_function_of_seed_and_index():
    """The backward pass regenerates the mask instead of storing it: the same (seed, index) must give the same decision
    whether the mask is materialised or fused, through the 16-byte and the 8-byte path (n % 4 != 0)."""
    from cross_patient_speech_decoding_amd._lib import call
    st = torch.cuda.current_stream().cuda_stream
    for n in (4096, 4098):
        x = torch.randn(n, device='cuda')
        mask = torch.empty(n, device='cuda')
        out = torch.empty(n, device='cuda')
        call('xps_dropout_f32', None, None, mask.data_ptr(), n, 0.4, 777, st)
        call('xps_dropout_f32', x.data_ptr(), out.data_ptr(), None, n, 0.4, 777, st)
        torch.cuda.synchronize()
        assert set(mask.unique().tolist()) <= {0.0, 1.0}
        np.testing.assert_array_equal(out.cpu().numpy(), (x * mask * np.float32(1.0 / (1.0 - np.float32(0.4)))).cpu().numpy())
    a = torch.empty(4096, device='cuda'); b = torch.empty(4098, device='cuda')
    call('xps_dropout_f32', None, None, a.data_ptr(), 4096, 0.4, 777, st)
    call('xps_dropout_f32', None, None, b.data_ptr(), 4098, 0.4, 777, st)
    assert torch.equal(a, b[:4096])                      # vector and pair paths agree element for element


def test_gru_recurrence_with_h0_and_dh0(gemm_precision):
    torch.set_num_threads(4)
    T, B, H = 1, 9, 24
    gru = _cpu_gru(H, H, 1, seed=3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(T, B, H, generator=g)
    h0 = torch.randn(1, B, H, generator=g)
    h0_ref = h0.clone().requires_grad_(True)
    y_ref, _ = gru(x, h0_ref)
    wt = torch.randn(T, B, H, generator=g)
    (y_ref * wt).sum().backward()
    xf = XF()
    w_ih, w_hh, b_ih, b_hh = [dev(w) for w in _weights(gru, 1)]
    gi = xf.linear(dev(x), w_ih, b_ih).view(1, T, B, 3 * H)
    h0g = dev(h0).requires_grad_(True)
    w_hh.requires_grad_(True)
    y_ext = xf.GRURecurFn.apply(gi, h0g, 1, w_hh, b_hh)
    np.testing.assert_allclose(y_ext[1].detach().cpu().numpy(), y_ref[0].detach().numpy(), atol=1e-5)
    (y_ext[1:2] * dev(wt)).sum().backward()
    np.testing.assert_allclose(h0g.grad.cpu().numpy(), h0_ref.grad.numpy(), atol=2e-5)
    np.testing.assert_allclose(w_hh.grad.cpu().numpy(), gru.weight_hh_l0.grad.numpy(), atol=5e-5)


# ----------------------------------------------------------------------------- conv + BN
@pytest.mark.parametrize('relu,training,stride,k', [(False, True, 10, 10), (True, True, 3, 5), (False, False, 4, 4),
                                                    (True, False, 2, 6)])
def test_temporal_conv_vs_torch_cpu(relu, training, stride, k):
    torch.manual_seed(0)
    B, T, Cin, F = 6, 50, 7, 11
    conv = torch.nn.Conv1d(Cin, F, k, stride=stride)
    bn = torch.nn.BatchNorm1d(F)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train(training)
    x = torch.randn(B, T, Cin)
    ref = bn(conv(x.permute(0, 2, 1)))
    if relu:
        ref = torch.relu(ref)
    wt = torch.randn_like(ref)
    if training:
        (ref * wt).sum().backward()
    xf = XF()
    params = [dev(p.detach().clone()).requires_grad_(True) for p in (conv.weight, conv.bias, bn.weight, bn.bias)]
    rm, rv = dev(rm0.clone()), dev(rv0.clone())
    out = xf.TemporalConvFn.apply(dev(x), *params, rm, rv, stride, training, relu, None, 1.0, 0.1, bn.eps, None)
    np.testing.assert_allclose(out.detach().permute(1, 2, 0).cpu().numpy(), ref.detach().numpy(), atol=3e-5)
    if training:
        np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.numpy(), atol=1e-6)
        np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.numpy(), atol=1e-5)
        (out * dev(wt.permute(2, 0, 1).contiguous())).sum().backward()
        for p, r in zip(params, (conv.weight, conv.bias, bn.weight, bn.bias)):
            np.testing.assert_allclose(p.grad.cpu().numpy(), r.grad.numpy(), atol=2e-4, rtol=1e-3)


# ----------------------------------------------------------------------------- glue / loss / optimiser
def test_gather_scatter_next_token():
    g = torch.Generator().manual_seed(6)
    table = torch.randn(10, 48, generator=g)
    idx = torch.randint(0, 10, (37,), generator=g)
    xf = XF()
    tb = dev(table).requires_grad_(True)
    out = xf.gather_rows(tb, dev(idx))
    assert torch.equal(out.detach().cpu(), table[idx])
    wt = torch.randn(37, 48, generator=g)
    (out * dev(wt)).sum().backward()
    ref = torch.zeros(10, 48).index_add_(0, idx, wt)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), ref.numpy(), atol=1e-5)
    logits = torch.randn(37, 9, generator=g)
    logits[3, 2] = logits[3, 7] = 99.0                      # tie -> first index, as torch.argmax
    teacher = torch.randint(0, 9, (37, 3), generator=g)
    nt = xf.next_token(dev(logits), dev(teacher)[:, 1], dev(torch.tensor([0], dtype=torch.int32)))
    assert torch.equal(nt.cpu(), logits.argmax(1)) and nt[3].item() == 2
    nt = xf.next_token(dev(logits), dev(teacher)[:, 1], dev(torch.tensor([1], dtype=torch.int32)))
    assert torch.equal(nt.cpu(), teacher[:, 1])


def test_cross_entropy_and_adamw_vs_torch():
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(300, 9, generator=g) * 3
    target = torch.randint(0, 9, (300,), generator=g)
    lr_ref = logits.clone().requires_grad_(True)
    loss_ref = torch.nn.functional.cross_entropy(lr_ref, target)
    loss_ref.backward()
    xf = XF()
    lg = dev(logits).requires_grad_(True)
    loss = xf.cross_entropy(lg, dev(target))
    loss.backward()
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-6)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), lr_ref.grad.numpy(), atol=1e-8, rtol=1e-5)
    # AdamW + clip: 3 steps against torch.optim.AdamW + clip_grad_norm_
    p0 = torch.randn(5000, generator=g)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=1e-2, weight_decay=1e-2)
    p = dev(p0.clone()); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step in range(1, 4):
        grad = torch.randn(5000, generator=g) * (0.01 if step == 2 else 1.0)
        p_ref.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([p_ref], 0.5)
        opt.step()
        gd = dev(grad.clone())
        ss = xf.grad_sumsq(gd)
        np.testing.assert_allclose(ss.item(), float((grad.double() ** 2).sum()), rtol=1e-6)
        xf.adamw_step(p, gd, m, v, ss, 0.5, 1e-2, 0.9, 0.999, 1e-8, 1e-2, step)
        np.testing.assert_allclose(gd.cpu().numpy(), p_ref.grad.numpy(), rtol=2e-6, atol=1e-9)   # clipped in place
        np.testing.assert_allclose(p.cpu().numpy(), p_ref.detach().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('H,ndir', [(128, 2), (64, 1), (96, 2)])
def test_inter_layer_dropout_fused_into_the_gru_kernels_same_bits(H, ndir, gemm_precision, monkeypatch):
    """torch.nn.GRU's inter-layer dropout (nn_models/models.py:661-663): the recurrence kernels of the resident shapes write
    the dropped output and re-make the decisions while they load dy.  Same seed -> the same bits as the separate
    xps_dropout_f32 passes (forward output, input gradient, every weight gradient); H = 96 has no fused path (both runs take
    the separate passes: the switch must be harmless there)."""
    xf = XF()
    T, B, In = 7, 50, 24
    torch.manual_seed(H + ndir)
    gru = torch.nn.GRU(In, H, 1, bidirectional=(ndir == 2))
    ws = []
    for d in range(ndir):
        sfx = '_l0' + ('_reverse' if d else '')
        ws += [getattr(gru, n + sfx).detach().clone().cuda() for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(T, B, In, generator=g).cuda()
    wt = torch.randn(T, B, ndir * H, generator=g).cuda()
    monkeypatch.setattr(xf, 'next_dropout_seed', lambda: 0x1234567)

    def run(fused):
        monkeypatch.setattr(xf, 'fused_dropout_supported', (lambda *a: bool(lib().xps_gru_seq_fused_dropout_supported(*a))) if fused else (lambda *a: False))
        xg = x.clone().requires_grad_(True)
        wl = [w.clone().requires_grad_(True) for w in ws]
        y, hn = xf.GRULayerDropFn.apply(xg, ndir, xf.HN_STACK, 0.3, *wl)
        ((y * wt).sum() + hn.sum()).backward()
        torch.cuda.synchronize()
        return [y.detach().clone(), hn.detach().clone(), xg.grad.clone()] + [w.grad.clone() for w in wl]

    from cross_patient_speech_decoding_amd._lib import lib
    assert bool(lib().xps_gru_seq_fused_dropout_supported(T, B, H, ndir)) == (H in (64, 128))
    sep = run(False)
    fus = run(True)
    for a, b in zip(sep, fus):
        assert torch.equal(a, b)
    y = sep[0]
    zero = float((y == 0).float().mean())
    assert 0.25 < zero < 0.35                                   # ~30 % of the outputs dropped
    y0, hn0 = xf.GRULayerFn.apply(x, ndir, xf.HN_STACK, *ws)    # no dropout: kept elements are y / 0.7, hn is undropped
    keep = y != 0
    np.testing.assert_allclose(y[keep].cpu().numpy(), (y0[keep] / 0.7).cpu().numpy(), rtol=1e-6)
    assert torch.equal(hn0, sep[1])


@pytest.mark.parametrize('B,H,C,use_teacher', [(130, 512, 9, 0), (67, 500, 9, 1), (5, 64, 16, 0), (300, 260, 3, 0)])
def test_decoder_select_kernel(B, H, C, use_teacher):
    """logits + next token (first maximum or teacher) + gather of the token's projection row in one launch
    (nn_models/models.py:285-301, 749-757) against torch."""
    from cross_patient_speech_decoding_amd._lib import call
    xf = XF()
    g = torch.Generator().manual_seed(B + H)
    h = torch.randn(B, H, generator=g).cuda()
    w = (torch.randn(C, H, generator=g) / H ** 0.5).cuda()
    bias = torch.randn(C, generator=g).cuda()
    ntok = C + 1
    table = torch.randn(ntok, 3 * H, generator=g).cuda()
    teacher = torch.randint(0, C, (B, 3), generator=g).cuda()
    flag = torch.tensor([use_teacher], dtype=torch.int32).cuda()
    logits = torch.empty(B, C, device='cuda')
    nxt = torch.empty(B, dtype=torch.int64, device='cuda')
    gi = torch.empty(B, 3 * H, device='cuda')
    call('xps_decoder_select_f32', h.data_ptr(), w.data_ptr(), bias.data_ptr(), logits.data_ptr(), teacher[:, 1].data_ptr(),
         teacher.stride(0), flag.data_ptr(), table.data_ptr(), nxt.data_ptr(), gi.data_ptr(), B, H, C, ntok, xf._stream())
    ref = (h.double() @ w.double().T + bias.double())
    np.testing.assert_allclose(logits.cpu().numpy(), ref.cpu().numpy(), atol=2e-6 * H ** 0.5, rtol=1e-5)
    exp = teacher[:, 1] if use_teacher else logits.argmax(dim=1)          # (argmax of the kernel's own logits: first maximum)
    assert torch.equal(nxt, exp)
    assert torch.equal(gi, table[nxt])
    logits2 = torch.empty(B, C, device='cuda')                            # last step: logits only
    call('xps_decoder_select_f32', h.data_ptr(), w.data_ptr(), bias.data_ptr(), logits2.data_ptr(), None, 0, None,
         table.data_ptr(), None, None, B, H, C, ntok, xf._stream())
    assert torch.equal(logits, logits2)
