"""GPU parity tests of the cluster-persistent GRU recurrence (csrc/xps_gru_cluster.hip; 128 < H <= 512, the
north-star shape H = 512 and the reference default H = 500, nn_models/models.py:661-699, scripts/train_seq2seq.py:132):
against torch.nn.GRU on the CPU (what the reference calls), and the three launch modes against each other --
'persistent' (in-kernel hand-off between the workgroups of a cluster) must equal 'steps' (the same kernels, one
launch per step: the kernel boundary is the hand-off) BIT FOR BIT, at full bench size, repeatedly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cross_patient_speech_decoding_amd import _build  # noqa: E402
from cross_patient_speech_decoding_amd._lib import lib  # noqa: E402


@pytest.fixture(scope='module', autouse=True)
def _built():
    _build.build(verbose=False)
    assert torch.cuda.is_available(), 'gpu tests need the MI355X'


def XF():
    from cross_patient_speech_decoding_amd.nn_models import functional
    return functional


@pytest.fixture
def cluster_mode():
    """Restores the launch mode of the recurrence after a test changed it."""
    old = lib().xps_get_gru_cluster_mode()
    yield XF().set_gru_cluster_mode
    lib().xps_set_gru_cluster_mode(old)


@pytest.fixture(params=['grid1d', 'grid2d'])
def bptt_grid(request):
    """Both BPTT kernels of the cluster path (include/xps.h xps_set_gru_bptt_grid): the 1-D cluster kernel (default) and the
    4 x 4 grid with partial-sum hand-off (taken for 384 < H <= 512 in bf16x3 mode; other shapes run the 1-D kernel either way)."""
    old = lib().xps_get_gru_bptt_grid()
    assert lib().xps_set_gru_bptt_grid(1 if request.param == 'grid2d' else 0) == 0
    yield request.param
    lib().xps_set_gru_bptt_grid(old)


def _weights(gru, ndir):
    out = []
    for d in range(ndir):
        sfx = '_l0' + ('_reverse' if d else '')
        out += [getattr(gru, n + sfx).detach().clone() for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    return out


def test_cluster_path_is_selected_for_the_north_star_shapes():
    """The shapes this file is about really run the cluster kernels (a status word exists only there)."""
    l = lib()
    assert l.xps_get_gru_cluster_mode() == 2
    for (T, B, H, ndir) in [(20, 2048, 512, 2), (20, 2048, 500, 2), (20, 256, 512, 2), (5, 150, 260, 2), (47, 2048, 384, 1)]:
        assert l.xps_gru_seq_status_offset(T, B, H, ndir) >= 0, (T, B, H, ndir)
        assert l.xps_gru_seq_fwd_f32_workspace(T, B, H, ndir) > 4096
    for (T, B, H, ndir) in [(20, 2048, 128, 2), (20, 64, 512, 2), (20, 2048, 130, 2), (20, 2048, 640, 2), (20, 2048, 256, 2)]:
        assert l.xps_gru_seq_status_offset(T, B, H, ndir) == -1, (T, B, H, ndir)


@pytest.mark.parametrize('mode', ['persistent', 'steps'])
@pytest.mark.parametrize('T,B,In,H,ndir', [(6, 256, 24, 512, 2), (5, 200, 16, 500, 2), (4, 130, 12, 320, 1), (3, 300, 10, 260, 2),
                                           (7, 160, 20, 448, 1), (2, 1030, 8, 388, 2)])
def test_layer_forward_backward_vs_torch_cpu(T, B, In, H, ndir, mode, gemm_precision, cluster_mode, bptt_grid):
    """Whole layer (input projection + recurrence + BPTT + weight gradients) against torch.nn.GRU on the CPU; same
    tolerances as the H <= 128 kernels (tests/test_gpu_nn_kernels.py)."""
    cluster_mode(mode)
    torch.set_num_threads(8)
    torch.manual_seed(T * 100 + H)
    gru = torch.nn.GRU(In, H, 1, batch_first=False, bidirectional=(ndir == 2))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, B, In, generator=g)
    x_ref = x.clone().requires_grad_(True)
    y_ref, hn_ref = gru(x_ref)
    wt = torch.randn(T, B, ndir * H, generator=g)
    (y_ref * wt).sum().backward()

    xf = XF()
    ws = [w.cuda().requires_grad_(True) for w in _weights(gru, ndir)]
    xg = x.cuda().requires_grad_(True)
    y, hn = xf.GRULayerFn.apply(xg, ndir, xf.HN_STACK, *ws)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(hn.detach().cpu().numpy(), hn_ref.detach().numpy(), atol=2e-5)
    (y * wt.cuda()).sum().backward()
    xf.check_gru_status()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), x_ref.grad.numpy(), atol=5e-5, rtol=1e-4)
    names = []
    for d in range(ndir):
        sfx = '_l0' + ('_reverse' if d else '')
        names += [n + sfx for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    for w, n in zip(ws, names):
        ref = getattr(gru, n).grad.numpy()
        tol = 2e-4 * max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(w.grad.cpu().numpy(), ref, atol=tol, rtol=1e-3, err_msg=n)


@pytest.mark.parametrize('H,ndir', [(512, 2), (288, 1)])
def test_recurrence_with_initial_state_and_its_gradient(H, ndir, gemm_precision, bptt_grid):
    """GRURecurFn with h0: exercises the h0 slots, the dhn-free start and the extra dh0 pass of the backward kernel."""
    torch.set_num_threads(8)
    T, B, In = 4, 192, 12
    torch.manual_seed(3)
    gru = torch.nn.GRU(In, H, 1, bidirectional=(ndir == 2))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, B, In, generator=g)
    h0 = (0.5 * torch.randn(ndir, B, H, generator=g)).requires_grad_(True)
    y_ref, hn_ref = gru(x, h0)
    wt = torch.randn(T, B, ndir * H, generator=g)
    (y_ref * wt).sum().backward()

    xf = XF()
    w = _weights(gru, ndir)
    gi = torch.stack([torch.nn.functional.linear(x, w[4 * d], w[4 * d + 2]) for d in range(ndir)]).cuda()
    h0g = h0.detach().cuda().requires_grad_(True)
    wb = [w[4 * d + 1].cuda().requires_grad_(True) for d in range(ndir)] + [w[4 * d + 3].cuda().requires_grad_(True) for d in range(ndir)]
    y_ext = xf.GRURecurFn.apply(gi, h0g, ndir, *wb)
    y = y_ext[1:T + 1]
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), atol=2e-5)
    (y * wt.cuda()).sum().backward()
    xf.check_gru_status()
    np.testing.assert_allclose(h0g.grad.cpu().numpy(), h0.grad.numpy(), atol=5e-5, rtol=1e-4)
    ref = gru.weight_hh_l0.grad.numpy()
    np.testing.assert_allclose(wb[0].grad.cpu().numpy(), ref, atol=2e-4 * max(1.0, float(np.abs(ref).max())), rtol=1e-3)


def _run_layer(xf, x, ws, wt, ndir):
    xg = x.clone().requires_grad_(True)
    wl = [w.clone().requires_grad_(True) for w in ws]
    y, hn = xf.GRULayerFn.apply(xg, ndir, xf.HN_STACK, *wl)
    (y * wt).sum().backward()
    torch.cuda.synchronize()
    xf.check_gru_status()
    return [y.detach().clone(), hn.detach().clone(), xg.grad.clone()] + [w.grad.clone() for w in wl]


@pytest.mark.parametrize('need_dh0', [False, True])
@pytest.mark.parametrize('T,B,H', [(6, 640, 512), (2, 512, 512), (5, 300, 448)])
def test_bptt_without_dy_equals_explicit_zero_dy(T, B, H, need_dh0, gemm_precision, cluster_mode, bptt_grid):
    """The top encoder layer of the seq2seq model gets NO dy (only the gradient of its last hidden state,
    nn_models/models.py:690-699): the BPTT kernels then read their inputs without a branch around any load (the dy load is
    redirected and dropped).  dy = None must give the bits of dy = 0, in the persistent form (interior fast paths of both
    grids, edges, the last step with and without the dh0 pass) and with one launch per step."""
    xf = XF()
    ndir = 2
    g = torch.Generator().manual_seed(T * 1000 + B + H)
    gi = (0.5 * torch.randn(ndir, T, B, 3 * H, generator=g)).cuda()
    ws = [(torch.randn(3 * H, H, generator=g) / H ** 0.5).cuda() for _ in range(ndir)]
    bs = [(0.1 * torch.randn(3 * H, generator=g)).cuda() for _ in range(ndir)]
    dhn = torch.randn(ndir, B, H, generator=g).cuda()
    outs = {}
    for mode in ('steps', 'persistent'):
        cluster_mode(mode)
        y_ext, saved = xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)
        for name, dy in (('none', None), ('zero', torch.zeros(T, B, ndir * H, device='cuda'))):
            dgi, dghn, dh0 = xf._gru_backward(dy, dhn, y_ext, saved, ws, T, B, H, ndir, need_dh0)
            torch.cuda.synchronize()
            xf.check_gru_status()
            outs[(mode, name)] = [dgi.clone(), dghn.clone()] + ([dh0.clone()] if need_dh0 else [])
    ref = outs[('steps', 'zero')]
    assert all(torch.isfinite(r).all() for r in ref) and float(ref[0].abs().max()) > 0
    for key, val in outs.items():
        for a, b, name in zip(val, ref, ['dgi', 'dghn', 'dh0']):
            assert torch.equal(a, b), f'{name} of {key} differs from steps / zero dy'


@pytest.mark.parametrize('H', [512, 500])
def test_persistent_equals_one_launch_per_step_bitwise_full_size(H, gemm_precision, cluster_mode, bptt_grid):
    """Bench-size layer (2048 trials x 20 steps, bidirectional: the per-GPU shard of configs[3]): the in-kernel
    hand-off ('persistent') and the kernel-boundary hand-off ('steps') run the same arithmetic, so EVERY output bit
    must agree -- a stale or early read of the exchange buffer shows up here.  Three launches of each (the second
    and third start with warm caches) and a memory-heavy kernel in between (uneven load)."""
    xf = XF()
    T, B, In, ndir = 20, 2048, 30, 2
    g = torch.Generator().manual_seed(17)
    x = torch.randn(T, B, In, generator=g).cuda()
    wt = torch.randn(T, B, ndir * H, generator=g).cuda()
    torch.manual_seed(4)
    gru = torch.nn.GRU(In, H, 1, bidirectional=True)
    ws = [w.cuda() for w in _weights(gru, ndir)]
    cluster_mode('steps')
    ref = _run_layer(xf, x, ws, wt, ndir)
    assert all(torch.isfinite(r).all() for r in ref)
    junk = torch.empty(64 << 20, device='cuda')
    cluster_mode('persistent')
    for rep in range(3):
        junk.normal_()                                  # streams 256 MB through L2 / Infinity Cache between the launches
        out = _run_layer(xf, x, ws, wt, ndir)
        for a, b, name in zip(out, ref, ['y', 'hn', 'dx'] + ['w%d' % i for i in range(8)]):
            assert torch.equal(a, b), f'{name} differs between persistent and per-step launches (repeat {rep})'
    cluster_mode('off')
    old = _run_layer(xf, x, ws, wt, ndir)               # the per-step GEMM kernels: same math, other summation order
    np.testing.assert_allclose(old[0].cpu().numpy(), ref[0].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(old[2].cpu().numpy(), ref[2].cpu().numpy(), atol=1e-4, rtol=1e-3)


def test_persistent_under_concurrent_load(cluster_mode):
    """The hand-off while another stream keeps the memory system busy and takes CUs away (uneven load): results
    still equal the per-step launches bit for bit."""
    xf = XF()
    T, B, In, H, ndir = 12, 1024, 16, 512, 2
    g = torch.Generator().manual_seed(23)
    x = torch.randn(T, B, In, generator=g).cuda()
    wt = torch.randn(T, B, ndir * H, generator=g).cuda()
    torch.manual_seed(9)
    gru = torch.nn.GRU(In, H, 1, bidirectional=True)
    ws = [w.cuda() for w in _weights(gru, ndir)]
    cluster_mode('steps')
    ref = _run_layer(xf, x, ws, wt, ndir)
    cluster_mode('persistent')
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device='cuda')
    big = torch.empty(32 << 20, device='cuda')
    for rep in range(3):
        with torch.cuda.stream(side):
            for _ in range(4):
                big.add_(1.0)
                a = torch.tanh(a @ a * 1e-3)
        out = _run_layer(xf, x, ws, wt, ndir)
        side.synchronize()
        for p, q in zip(out, ref):
            assert torch.equal(p, q), f'repeat {rep}'


@pytest.mark.parametrize('mode', ['persistent', 'steps'])
@pytest.mark.parametrize('T,B,ndir,need_dh0', [(4, 512, 2, False), (3, 256, 1, True), (1, 2048, 2, True)])
def test_bptt_with_outputs_as_exchange_rows_equals_the_ring_bitwise(T, B, ndir, need_dh0, mode, cluster_mode, monkeypatch):
    """XPS_GRU_XOUT=1 (gru_cluster_bwd_kernel<.., XOUT>: the operand images of step ps are moved from the dgi / dghn rows step ps - 1
    wrote, no exchange ring; H = 512, bf16x3, split4 outputs) gives the bits of the default kernel: dgi, dghn, dh0."""
    xf = XF()
    H = 512
    old = lib().xps_get_gemm_precision()
    lib().xps_set_gemm_precision(1)
    try:
        cluster_mode(mode)
        g = torch.Generator().manual_seed(T * 7 + B)
        gi = (torch.randn(ndir, T, B, 3 * H, generator=g) * 0.5).cuda()
        w_hh = [(torch.randn(3 * H, H, generator=g) * H ** -0.5).cuda() for _ in range(ndir)]
        b_hh = [(torch.randn(3 * H, generator=g) * 0.1).cuda() for _ in range(ndir)]
        dy, dhn = (torch.randn(T, B, ndir * H, generator=g) * 0.1).cuda(), (torch.randn(ndir, B, H, generator=g) * 0.1).cuda()
        y_ext, saved = xf._gru_forward(gi, w_hh, b_hh, None, T, B, H, ndir, True)
        outs = []
        for flag in ('0', '1'):
            monkeypatch.setenv('XPS_GRU_XOUT', flag)
            outs.append(xf._gru_backward(dy, dhn, y_ext, saved, w_hh, T, B, H, ndir, need_dh0, split4=True))
        torch.cuda.synchronize()
        xf.check_gru_status()
        for a, b in zip(*outs):
            if a is not None:
                assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    finally:
        lib().xps_set_gemm_precision(old)


def test_launch_form_change_between_forward_and_backward_is_refused(cluster_mode):
    """The saved gates are member-major on the cluster path and row-major off it: a backward under the other launch form would
    read them wrongly, so functional._gru_backward refuses (the forward stamps the buffer)."""
    xf = XF()
    T, B, H = 3, 256, 512
    torch.manual_seed(0)
    gi = torch.randn(1, T, B, 3 * H, device='cuda')
    h0 = torch.zeros(1, B, H, device='cuda', requires_grad=True)
    w = (0.05 * torch.randn(3 * H, H, device='cuda')).requires_grad_(True)
    b = torch.zeros(3 * H, device='cuda', requires_grad=True)
    cluster_mode('persistent')
    y = xf.GRURecurFn.apply(gi, h0, 1, w, b)
    cluster_mode('off')
    with pytest.raises(RuntimeError, match='must not change between a forward and its backward'):
        y.sum().backward()
    cluster_mode('persistent')
    y = xf.GRURecurFn.apply(gi, h0, 1, w, b)
    y.sum().backward()                                   # same form: accepted
    xf.check_gru_status()


def test_cluster_path_is_bounded_by_its_largest_32_bit_buffer():
    """The cluster kernels address y_ext, saved and dgi through raw buffer descriptors (32-bit sizes / offsets).  The shape
    guard must bound the LARGEST of them (saved: ndir * T * B * 4H floats), not y_ext alone: at B = 2048, H = 512, both
    directions, that is T <= 125; longer sequences must select the per-step path (status offset -1, 16-byte forward
    workspace) instead of wrapping store offsets (host logic only: no launch)."""
    from cross_patient_speech_decoding_amd._lib import lib
    L = lib()
    B, H = 2048, 512
    assert L.xps_gru_seq_status_offset(20, B, H, 2) >= 0
    for T in (120, 125):
        assert 2 * (T + 2) * B * 4 * H * 4 < 2 ** 32 and L.xps_gru_seq_status_offset(T, B, H, 2) >= 0, T
    for T in (126, 128, 200, 511):
        assert 2 * (T + 2) * B * 4 * H * 4 >= 2 ** 32
        assert L.xps_gru_seq_status_offset(T, B, H, 2) == -1, T
        assert L.xps_gru_seq_fwd_f32_workspace(T, B, H, 2) == 16
        assert L.xps_gru_seq_bwd_split4_supported(T, B, H, 2) == 0
    assert L.xps_gru_seq_status_offset(250, B, H, 1) >= 0 and L.xps_gru_seq_status_offset(254, B, H, 1) == -1
