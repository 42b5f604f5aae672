"""GPU tests of the 256 x 256-tile GEMM kernels (csrc/xps_gemm_big.h; the large layer products of configs[3]: input
projections nn_models/models.py:687 (x W_ih^T inside torch.nn.GRU), their input and weight gradients).  The products
whose k range is not split must equal the 128 x 128-tile kernels BIT FOR BIT (same split arithmetic, same order per
accumulator); every form is bounded against fp64 by the split-product error model of DESIGN.md 4.0."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cross_patient_speech_decoding_amd import _build  # noqa: E402
from cross_patient_speech_decoding_amd._lib import call, lib, rowmap  # noqa: E402


@pytest.fixture(scope='module', autouse=True)
def _built():
    _build.build(verbose=False)
    assert torch.cuda.is_available(), 'gpu tests need the MI355X'


def XF():
    from cross_patient_speech_decoding_amd.nn_models import functional
    return functional


@pytest.fixture
def tiles():
    """tiles(0 | 1): 128 x 128 tiles only / 256 x 256 where eligible; restores the setting, bf16x3 mode for the test."""
    l = lib()
    old_t, old_p = l.xps_get_gemm_big_tiles(), l.xps_get_gemm_precision()
    l.xps_set_gemm_precision(1)
    yield l.xps_set_gemm_big_tiles
    l.xps_set_gemm_big_tiles(old_t)
    l.xps_set_gemm_precision(old_p)


def _bound(a64, b64_t):
    """1.6e-5 * sum_k |a||b| for a (M, K) @ b_t (K, N)"""
    return 1.6e-5 * (a64.abs() @ b64_t.abs())


M0, N0 = 4096, 3072          # 16 x 12 = 192 tiles of 256 x 256: the smallest grid the dispatcher sends to the big kernels


@pytest.mark.parametrize('K,bias,acc', [(64, True, False), (208, False, True), (1024, True, True)])
def test_nt_equals_small_tiles_bitwise_and_fp64(K, bias, acc, tiles):
    xf = XF()
    g = torch.Generator().manual_seed(K)
    A = torch.randn(M0, K, generator=g).cuda(); B = torch.randn(N0, K, generator=g).cuda()
    bv = torch.randn(N0, generator=g).cuda() if bias else None
    C0 = torch.randn(M0, N0, generator=g).cuda()
    outs = []
    for t in (0, 1):
        tiles(t)
        Cc = C0.clone()
        xf.gemm_nt(A, B, Cc, M0, N0, K, bias=bv, accumulate=acc)
        outs.append(Cc)
    assert torch.equal(outs[0], outs[1])
    ref = A.double() @ B.double().T + (bv.double() if bias else 0) + (C0.double() if acc else 0)
    err = (outs[1].double() - ref).abs()
    assert bool((err <= _bound(A.double(), B.double().T) + 1e-6 * ref.abs() + 1e-6).all()), float(err.max())


def test_big_kernels_are_selected_for_the_north_star_products(tiles):
    """Guards the dispatch conditions: with the switch on, a qualifying product must NOT equal a run whose k order differs
    ... it cannot be observed from outside bitwise (same bits by design), so time both settings instead: the 256-tile
    launch of an 8192 x 4096 x 2048 product is well over 15 % faster."""
    xf = XF()
    M, N, K = 8192, 4096, 2048
    A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); Cc = torch.empty(M, N, device='cuda')
    times = []
    for t in (0, 1):
        tiles(t)
        for _ in range(3):
            xf.gemm_nt(A, B, Cc, M, N, K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            xf.gemm_nt(A, B, Cc, M, N, K)
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 10)
    assert times[1] < 0.85 * times[0], times


def test_nn_and_nn2_equal_small_tiles_bitwise(tiles):
    """input-gradient forms: A [m][k] times B [k][n]; nn2 sums two operand pairs (both directions) in registers"""
    xf = XF()
    g = torch.Generator().manual_seed(7)
    K1, K2 = 96, 160
    A1 = torch.randn(M0, K1, generator=g).cuda(); B1 = torch.randn(K1, N0, generator=g).cuda()
    A2 = torch.randn(M0, K1, generator=g).cuda(); B2 = torch.randn(K1, N0, generator=g).cuda()
    outs = []
    for t in (0, 1):
        tiles(t)
        c1 = torch.empty(M0, N0, device='cuda'); c2 = torch.empty(M0, N0, device='cuda')
        xf.gemm_nn(A1, B1, c1, M0, N0, K1)
        ra, rb, rc = rowmap(K1), rowmap(N0), rowmap(N0)
        call('xps_gemm_nn2_f32', xf._ptr(A1), xf._ptr(B1), K1, xf._ptr(A2), xf._ptr(B2), K1, C.byref(ra), C.byref(rb), xf._ptr(c2),
             C.byref(rc), M0, N0, 0, xf._stream())
        outs.append((c1, c2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = A1.double() @ B1.double() + A2.double() @ B2.double()
    bound = _bound(A1.double(), B1.double()) + _bound(A2.double(), B2.double())
    assert bool(((outs[1][1].double() - ref).abs() <= bound + 1e-6).all())
    del K2


def test_nt_multi_equals_small_tiles_bitwise(tiles):
    """the input projections of both directions in one launch (GRULayerFn.forward)"""
    xf = XF()
    g = torch.Generator().manual_seed(11)
    M, N, K = 4096, 1536, 112
    A = torch.randn(M, K, generator=g).cuda()
    Bs = [torch.randn(N, K, generator=g).cuda() for _ in range(2)]
    bs = [torch.randn(N, generator=g).cuda() for _ in range(2)]
    outs = []
    for t in (0, 1):
        tiles(t)
        Cs = [torch.empty(M, N, device='cuda') for _ in range(2)]
        ra, rb, rc = rowmap(K), rowmap(K), rowmap(N)
        call('xps_gemm_nt_multi_f32', xf._ptr(A), C.byref(ra), xf._ptr_array(Bs), C.byref(rb), xf._ptr_array(Cs), C.byref(rc),
             xf._ptr_array(bs), 2, M, N, K, xf._stream())
        outs.append(Cs)
    for d in range(2):
        assert torch.equal(outs[0][d], outs[1][d])
        ref = A.double() @ Bs[d].double().T + bs[d].double()
        assert bool(((outs[1][d].double() - ref).abs() <= _bound(A.double(), Bs[d].double().T) + 1e-6).all())


def test_weight_gradient_group_mixes_both_tile_shapes(tiles):
    """One grouped launch with problems for the 256-tile kernel (M, N multiples of 256), for the 128-tile kernel (an edge
    shape) and column sums / accumulate flags on both: every output against fp64, and against the all-small-tile run."""
    xf = XF()
    g = torch.Generator().manual_seed(13)
    K = 8192
    ld = 1024
    dgi = torch.randn(K, 1536, generator=g).cuda()
    x = torch.randn(K, ld, generator=g).cuda()
    xs = torch.randn(K, 100, generator=g).cuda()

    def run():
        dw_a = torch.zeros(1024, 512, device='cuda'); db_a = torch.zeros(1536, device='cuda')      # r,z rows of a dW_hh + its bias
        dw_b = torch.ones(1536, 1024, device='cuda'); db_b = torch.ones(1536, device='cuda')       # dW_ih, accumulate
        dw_c = torch.zeros(1536, 100, device='cuda')                                               # edge shape: small tiles
        probs = [xf.tn_problem(dgi, x[:, 512:], dw_a, 1024, 512, K, ra=rowmap(1536), rb=rowmap(ld), rc=rowmap(512), colsum_out=db_a),
                 xf.tn_problem(dgi, x, dw_b, 1536, 1024, K, colsum_out=db_b, accumulate=True),
                 xf.tn_problem(dgi, xs, dw_c, 1536, 100, K)]
        keep = xf.gemm_tn_grouped(probs, dgi.device)
        torch.cuda.synchronize()
        del keep
        return dw_a, db_a, dw_b, db_b, dw_c

    tiles(0); small = run()
    tiles(1); big = run()
    d64, x64 = dgi.double(), x.double()
    refs = [d64[:, :1024].T @ x64[:, 512:], d64.sum(0)[:1024], 1 + d64.T @ x64, 1 + d64.sum(0), d64.T @ xs.double()]
    bounds = [_bound(d64[:, :1024].T, x64[:, 512:]), 1e-6 * d64.abs().sum(0)[:1024], _bound(d64.T, x64), 1e-6 * d64.abs().sum(0),
              _bound(d64.T, xs.double())]
    for name, b, s, r, bd in zip(['dw_a', 'db_a', 'dw_b', 'db_b', 'dw_c'], big, small, refs, bounds):
        if name == 'db_a':
            b, s = b[:1024], s[:1024]
        err = (b.double() - r).abs()
        assert bool((err <= bd + 1e-5).all()), (name, float(err.max()))
        assert float((b - s).abs().max()) <= 2 * float(bd.max()) + 1e-5, name
    again = run()
    for b, a in zip(big, again):
        assert torch.equal(b, a)                       # deterministic


@pytest.mark.parametrize('K,M,N,ldb,boff,cs,acc', [(8192, 1536, 1024, 1024, 0, True, False),      # dW_ih of configs[3] layer 1 (short K)
                                                   (40960, 1024, 512, 1024, 512, True, True),    # the r, z rows of a dW_hh at full K: B a column window
                                                   (5120, 512, 256, 256, 0, False, False),       # splits with a ragged last chunk
                                                   (4096, 256, 256, 256, 0, True, False)])
def test_weight_gradient_lds_dma_loop_equals_register_staged_loop_bitwise(K, M, N, ldb, boff, cs, acc, tiles, monkeypatch):
    """The LDS-DMA k loop of the 256-tile weight-gradient kernel (csrc/xps_gemm_dma.h: XPS_FMT_SPLIT4 operands read in place, no
    staging registers / split arithmetic / ds_write) against the register-staged loop (XPS_GEMM_DMA=0) and against the
    128-tile kernels on the SAME split4 operands: products bit for bit (same MFMA order per accumulator), against fp64 within
    the split-product bound; the column sums (on the matrix pipe in the DMA loop: another summation order) within the split4
    rule 2^-16 sum |a| of both the fp64 sums and the register-staged loop's; deterministic."""
    xf = XF()
    g = torch.Generator().manual_seed(K + M)
    A = torch.randn(K, M, generator=g).cuda()
    Bfull = (torch.randn(K, ldb, generator=g) * 0.5).cuda()
    A4, B4 = xf.split4(A), xf.split4(Bfull)
    Bv = B4.view(-1)[boff:]
    C0 = torch.randn(M, N, generator=g).cuda()

    def run():
        out = C0.clone() if acc else torch.empty(M, N, device='cuda')
        db = torch.zeros(M, device='cuda') if cs else None
        keep = xf.gemm_tn_grouped([xf.tn_problem(A4, Bv, out, M, N, K, ra=rowmap(M, fmt=1), rb=rowmap(ldb, fmt=1), rc=rowmap(N),
                                                 colsum_out=db, accumulate=acc)], 'cuda')
        torch.cuda.synchronize()
        del keep
        return out, db

    tiles(1)
    monkeypatch.setenv('XPS_GEMM_DMA', '1'); dma, dma_cs = run()
    again, again_cs = run()
    monkeypatch.setenv('XPS_GEMM_DMA', '0'); reg, reg_cs = run()
    tiles(0); small, small_cs = run()
    assert torch.equal(dma, reg) and torch.equal(dma, again)
    a64, b64 = A.double(), Bfull.double()[:, boff:boff + N]
    ref = a64.T @ b64 + (C0.double() if acc else 0)
    assert bool(((dma.double() - ref).abs() <= _bound(a64.T, b64) + 1e-5).all())
    # a 256-tile launch and a 128-tile launch choose other k-splits: equal up to the order of the fp32 slab sums
    assert float((dma - small).abs().max()) <= 2 * float(_bound(a64.T, b64).max()) + 1e-5
    if cs:
        assert torch.equal(dma_cs, again_cs)
        bound = 2.0 ** -16 * a64.abs().sum(0) + 1e-6
        assert bool(((dma_cs.double() - a64.sum(0)).abs() <= bound).all())
        assert bool(((dma_cs.double() - reg_cs.double()).abs() <= bound).all())


@pytest.mark.parametrize('K', [32, 96, 1024])
def test_direct_forms_lds_dma_loop_equals_register_staged_loop_bitwise(K, tiles, monkeypatch):
    """The LDS-DMA k loop of the direct 256-tile forms (csrc/xps_gemm_dma.h: direct_dma_pipeline; XPS_FMT_SPLIT4 operands, [x][k]
    rows moved as whole cache lines with a source-side bank swizzle, group pairs regrouped in registers; the [k][x] operand of
    the NN form through transposing reads) against the register-staged loop (XPS_GEMM_DMA=0) and the 128-tile kernels: NT, NN,
    the two-operand NN of the bidirectional input gradient and the multi-B projection -- bit for bit, bias and accumulate
    epilogues included; against fp64 within the split-product bound."""
    xf = XF()
    g = torch.Generator().manual_seed(100 + K)
    A = torch.randn(M0, K, generator=g).cuda()
    A2 = torch.randn(M0, K, generator=g).cuda()
    B = (torch.randn(N0, K, generator=g) * 0.5).cuda()                       # NT: [n][k]
    Bt = (torch.randn(K, N0, generator=g) * 0.5).cuda()                      # NN: [k][n]
    Bt2 = (torch.randn(K, N0, generator=g) * 0.5).cuda()
    bias = torch.randn(N0, generator=g).cuda()
    C0 = torch.randn(M0, N0, generator=g).cuda()
    A4, A24, B4, Bt4, Bt24 = (xf.split4(t) for t in (A, A2, B, Bt, Bt2))
    ra, rbk, rbn, rc = rowmap(K, fmt=1), rowmap(K, fmt=1), rowmap(N0, fmt=1), rowmap(N0)

    def run():
        c_nt = C0.clone()
        xf.gemm_nt(A4, B4, c_nt, M0, N0, K, bias=bias, accumulate=True, ra=ra, rb=rbk)
        c_nn = torch.empty(M0, N0, device='cuda')
        xf.gemm_nn(A4, Bt4, c_nn, M0, N0, K, ra=ra, rb=rbn)
        c_nn2 = torch.empty(M0, N0, device='cuda')
        call('xps_gemm_nn2_f32', xf._ptr(A4), xf._ptr(Bt4), K, xf._ptr(A24), xf._ptr(Bt24), K, C.byref(ra), C.byref(rbn), xf._ptr(c_nn2),
             C.byref(rc), M0, N0, 0, xf._stream())
        cs = [torch.empty(M0, N0, device='cuda') for _ in range(2)]
        call('xps_gemm_nt_multi_f32', xf._ptr(A4), C.byref(ra), xf._ptr_array([B4, B4]), C.byref(rbk), xf._ptr_array(cs), C.byref(rc),
             xf._ptr_array([bias, bias]), 2, M0, N0, K, xf._stream())
        torch.cuda.synchronize()
        return c_nt, c_nn, c_nn2, cs[0], cs[1]

    tiles(1)
    monkeypatch.setenv('XPS_GEMM_DMA', '1'); dma = run()
    monkeypatch.setenv('XPS_GEMM_DMA', '0'); reg = run()
    tiles(0); small = run()
    for a, b, c in zip(dma, reg, small):
        assert torch.equal(a, b) and torch.equal(a, c)
    a64 = A.double()
    ref_nt = a64 @ B.double().T + bias.double() + C0.double()
    assert bool(((dma[0].double() - ref_nt).abs() <= _bound(a64, B.double().T) + 1e-5).all())
    ref_nn2 = a64 @ Bt.double() + A2.double() @ Bt2.double()
    assert bool(((dma[2].double() - ref_nn2).abs() <= _bound(a64, Bt.double()) + _bound(A2.double(), Bt2.double()) + 1e-5).all())


@pytest.mark.parametrize('N', [100, 64, 200])
def test_skinny_input_gradient_on_the_lds_dma_loop_equals_edge_tiles_bitwise(N, tiles, monkeypatch):
    """configs[3] layer 0: dx = dgi_fwd W_fwd + dgi_bwd W_bwd with In = 100 input channels (40960 x 100 x 2 * 1536; here 24576 rows).
    The product is bound by the read of dgi: with W handed over as a zero-padded 256-column split4 image (xps_split4_pad_f32) it
    runs ONE 256-wide tile per 256 rows on the LDS-DMA loop (gemm_big_kernel<.., 5>: waves beyond column N idle, masked C
    store) -- bit for bit what the 64-row edge tiles give on the unpadded operands, within the split-product bound of fp64."""
    xf = XF()
    g = torch.Generator().manual_seed(N)
    M, K = 96 * 256, 1536
    A1 = (torch.randn(M, K, generator=g) * 0.1).cuda()
    A2 = (torch.randn(M, K, generator=g) * 0.1).cuda()
    W1 = (torch.randn(K, N, generator=g) * 0.3).cuda()
    W2 = (torch.randn(K, N, generator=g) * 0.3).cuda()
    A14, A24 = xf.split4(A1), xf.split4(A2)
    ra, rc = rowmap(K, fmt=1), rowmap(N)

    def nn2(b1, b2, rb):
        out = torch.full((M, N), 7.0, device='cuda')
        call('xps_gemm_nn2_f32', xf._ptr(A14), xf._ptr(b1), K, xf._ptr(A24), xf._ptr(b2), K, C.byref(ra), C.byref(rb), xf._ptr(out),
             C.byref(rc), M, N, 0, xf._stream())
        torch.cuda.synchronize()
        return out

    tiles(1)
    monkeypatch.setenv('XPS_GEMM_DMA', '1')
    P1, P2 = xf.split4_pad(W1, 256), xf.split4_pad(W2, 256)
    assert P1.shape == (K, 256) and float(P1[:, N:].abs().max()) == 0.0
    dma = nn2(P1, P2, rowmap(256, fmt=1))
    edge = nn2(xf.split4(W1), xf.split4(W2), rowmap(N, fmt=1))
    plain = nn2(W1, W2, rowmap(N))
    assert torch.equal(dma, edge) and torch.equal(dma, plain)
    ref = A1.double() @ W1.double() + A2.double() @ W2.double()
    assert bool(((dma.double() - ref).abs() <= _bound(A1.double(), W1.double()) + _bound(A2.double(), W2.double()) + 1e-5).all())
    monkeypatch.setenv('XPS_SKINNY_DX', '1')                                   # (opt-in: measured slower than the edge tiles)
    assert xf.skinny_dx_wanted(40960, 100, 1536) and not xf.skinny_dx_wanted(40960, 1024, 1536)


def test_opt_in_32_deep_stages_same_bits():
    """XPS_GEMM_BIG_DEEP=1 (32-deep LDS stages with swizzled [x][k] images, read once per process): the direct forms must
    still equal the 128-tile kernels bit for bit.  Own process because the switch is read at first use."""
    import os
    import subprocess
    import sys
    code = (
        "import torch, ctypes as C\n"
        "from cross_patient_speech_decoding_amd._lib import lib, call, rowmap\n"
        "from cross_patient_speech_decoding_amd.nn_models import functional as xf\n"
        "l = lib(); l.xps_set_gemm_precision(1)\n"
        "g = torch.Generator().manual_seed(5)\n"
        "M, N, K = 4096, 3072, 160\n"
        "A = torch.randn(M, K, generator=g).cuda(); B = torch.randn(N, K, generator=g).cuda(); Bt = torch.randn(K, N, generator=g).cuda()\n"
        "outs = []\n"
        "for t in (0, 1):\n"
        "    l.xps_set_gemm_big_tiles(t)\n"
        "    c1 = torch.empty(M, N, device='cuda'); c2 = torch.empty(M, N, device='cuda')\n"
        "    xf.gemm_nt(A, B, c1, M, N, K); xf.gemm_nn(A, Bt, c2, M, N, K)\n"
        "    outs.append((c1, c2))\n"
        "assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])\n"
        "print('DEEP_OK')\n")
    env = dict(os.environ, XPS_GEMM_BIG_DEEP='1')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert 'DEEP_OK' in r.stdout, r.stdout + r.stderr
