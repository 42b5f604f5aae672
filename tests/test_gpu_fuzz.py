"""Randomised shape sweep of the GEMM entry points and the fused GRU layer against float64 CPU results:
odd sizes, unaligned leading dimensions (scalar load paths), row maps, accumulate, few-row and tail tiles."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _xf():
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    return XF


@pytest.fixture(params=[0, 1], ids=['fp32', 'bf16x3'])
def prec(request):
    """Both product precisions of the tile kernels; yields the tolerance multiplier (see test_gpu_nn_kernels.TOL)."""
    from cross_patient_speech_decoding_amd._lib import lib
    old = lib().xps_get_gemm_precision()
    assert lib().xps_set_gemm_precision(request.param) == 0
    yield (1.0, 16.0)[request.param]
    lib().xps_set_gemm_precision(old)


def test_gemm_random_shapes_and_strides(prec):
    from cross_patient_speech_decoding_amd._lib import rowmap
    XF = _xf()
    rng = np.random.default_rng(2024)
    for it in range(60):
        M = int(rng.choice([1, 3, 16, 17, 63, 64, 65, 127, 129, 200, 300, 513]))
        N = int(rng.choice([1, 5, 31, 32, 33, 100, 127, 128, 129, 257, 384]))
        K = int(rng.choice([1, 2, 7, 15, 16, 17, 33, 100, 130, 257, 640]))
        lda = K + int(rng.choice([0, 0, 1, 3, 4]))          # unaligned leading dimensions force the scalar paths
        ldb = K + int(rng.choice([0, 0, 2, 4]))
        ldc = N + int(rng.choice([0, 0, 1, 4]))
        A = torch.from_numpy(rng.standard_normal((M, lda)).astype(np.float32))
        Bm = torch.from_numpy(rng.standard_normal((N, ldb)).astype(np.float32))
        bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
        C0 = torch.from_numpy(rng.standard_normal((M, ldc)).astype(np.float32))
        acc = bool(rng.integers(0, 2))
        ref = A[:, :K].double() @ Bm[:, :K].double().T + bias.double() + (C0[:, :N].double() if acc else 0)
        scale = A[:, :K].abs().double() @ Bm[:, :K].abs().double().T + bias.abs().double() + C0[:, :N].abs().double()
        out = C0.clone().cuda()
        XF.gemm_nt(A.cuda(), Bm.cuda(), out, M, N, K, bias=bias.cuda(), ra=rowmap(lda), rb=rowmap(ldb), rc=rowmap(ldc),
                   accumulate=acc)
        got = out.cpu()
        assert ((got[:, :N].double() - ref).abs() <= 1e-6 * prec * scale + 1e-30).all(), ('nt', M, N, K, lda, ldb, ldc, acc)
        assert torch.equal(got[:, N:], C0[:, N:]), ('nt padding touched', M, N, K)
        # NN with the same data: B given as (K x N)
        Bk = torch.zeros(K, ldc)
        Bk[:, :N] = Bm[:, :K].T
        out2 = torch.full((M, ldc), 7.0).cuda()
        XF.gemm_nn(A.cuda(), Bk.cuda(), out2, M, N, K, ra=rowmap(lda), rb=rowmap(ldc), rc=rowmap(ldc))
        ref2 = A[:, :K].double() @ Bm[:, :K].double().T
        assert ((out2.cpu()[:, :N].double() - ref2).abs() <= 1e-6 * prec * scale + 1e-30).all(), ('nn', M, N, K, lda, ldc)
        # TN: out (M x N) = At^T B with At (K x M)
        At = torch.zeros(K, M + 3)
        At[:, :M] = A[:, :K].T
        out3 = torch.empty(M, N).cuda()
        XF.gemm_tn(At.cuda(), Bk.cuda(), out3, M, N, K, ra=rowmap(M + 3), rb=rowmap(ldc), rc=rowmap(N))
        assert ((out3.cpu().double() - ref2).abs() <= 2e-6 * prec * scale + 1e-30).all(), ('tn', M, N, K)


@pytest.mark.parametrize('seed', range(6))
def test_gru_layer_random_shapes(seed, gemm_precision):
    """GRULayerFn forward + backward vs torch.nn.GRU on the CPU for random (T, B, In, H, ndir)."""
    XF = _xf()
    rng = np.random.default_rng(100 + seed)
    T = int(rng.choice([1, 2, 5, 9, 20]))
    B = int(rng.choice([1, 3, 15, 16, 17, 33, 70]))
    In = int(rng.choice([1, 7, 30, 100]))
    H = int(rng.choice([8, 24, 64, 96, 128, 160]))
    ndir = int(rng.choice([1, 2]))
    torch.manual_seed(seed)
    gru = torch.nn.GRU(In, H, 1, bidirectional=(ndir == 2))
    x = torch.randn(T, B, In)
    wt = torch.randn(T, B, ndir * H)
    wh = torch.randn(ndir, B, H)
    x_ref = x.clone().requires_grad_(True)
    y_ref, hn_ref = gru(x_ref)
    ((y_ref * wt).sum() + (hn_ref * wh).sum()).backward()
    ws = []
    for d in range(ndir):
        sfx = '_l0' + ('_reverse' if d else '')
        ws += [getattr(gru, n + sfx).detach().clone().cuda().requires_grad_(True)
               for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    xg = x.cuda().requires_grad_(True)
    y, hn = XF.GRULayerFn.apply(xg, ndir, XF.HN_STACK, *ws)
    ((y * wt.cuda()).sum() + (hn * wh.cuda()).sum()).backward()
    tag = (T, B, In, H, ndir)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), atol=3e-5, err_msg=str(tag))
    np.testing.assert_allclose(hn.detach().cpu().numpy(), hn_ref.detach().numpy(), atol=3e-5, err_msg=str(tag))
    np.testing.assert_allclose(xg.grad.cpu().numpy(), x_ref.grad.numpy(), atol=1e-4, rtol=1e-3, err_msg=str(tag))
    names = []
    for d in range(ndir):
        sfx = '_l0' + ('_reverse' if d else '')
        names += [n + sfx for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    for w, n in zip(ws, names):
        ref = getattr(gru, n).grad.numpy()
        np.testing.assert_allclose(w.grad.cpu().numpy(), ref, atol=3e-4 * max(1.0, float(np.abs(ref).max())), rtol=2e-3,
                                   err_msg=f'{n} {tag}')
