"""Device linear algebra for the aligners, on top of the C ABI (include/xps.h).

Everything heavy — per-condition means, centred Gram / cross-covariance accumulation on the
f64 MFMA, Jacobi eigen/singular decompositions, batched transform apply — runs in libxps.so.
Host numpy only sorts / normalises / sign-fixes the small (d x d) results and builds index
lists; there is no CPU fallback for the kernels.
"""
import os

import numpy as np
import torch

from .._lib import call, lib

F64 = torch.float64
EPS = float(np.finfo(np.float64).eps)


def device():
    if not torch.cuda.is_available():
        raise RuntimeError('cross_patient_speech_decoding_amd.alignment needs the MI355X: the HIP path has no '
                           'CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_current_device = torch._C._cuda_getDevice if hasattr(torch._C, '_cuda_getDevice') else torch.cuda.current_device


def _stream():
    """Raw handle of torch's current stream (the C-level getter: ~0.3 us instead of ~10 us for the Stream object)."""
    if _raw_stream is not None:
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


def _ws(nbytes):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device())


def to_device(x):
    """numpy / torch (float32 or float64; anything else -> float64) -> contiguous device tensor."""
    if isinstance(x, torch.Tensor):
        t = x
    else:
        x = np.asarray(x)
        if x.dtype not in (np.float32, np.float64):
            x = x.astype(np.float64)
        t = torch.from_numpy(np.ascontiguousarray(x))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(F64)
    return t.to(device()).contiguous()


def like_input(out, X):
    """Transforms return what they were given: a DEVICE tensor for a device-resident input (no PCIe round trip: the next
    stage -- pooling, the DataModule, the model -- consumes it in HBM), otherwise the reference's float64 ndarray."""
    if isinstance(X, torch.Tensor) and X.is_cuda:
        return out
    return out.cpu().numpy()


def _is32(t):
    return int(t.dtype == torch.float32)


# ------------------------------------------------------------------------------------ k1
def condition_index(keys):
    """Sorted unique string keys + CSR (order, start) of the trials of each condition, trials in
    original order inside a condition (numpy boolean-mask order)."""
    uniq, inv = np.unique(np.asarray(keys), return_inverse=True)
    order = np.argsort(inv, kind='stable').astype(np.int32)
    start = np.concatenate([[0], np.cumsum(np.bincount(inv, minlength=len(uniq)))]).astype(np.int32)
    return uniq, order, start


def cnd_avg_device(data_d, order, start):
    """data_d (N, ...) float32/float64 device tensor -> (n_cond, ...) float64 device tensor."""
    n_cond = len(start) - 1
    row_len = int(np.prod(data_d.shape[1:])) if data_d.dim() > 1 else 1
    out = torch.empty((n_cond,) + tuple(data_d.shape[1:]), dtype=F64, device=data_d.device)
    o = torch.from_numpy(order).to(data_d.device)
    s = torch.from_numpy(start).to(data_d.device)
    fn = 'xps_cnd_avg_f32' if data_d.dtype == torch.float32 else 'xps_cnd_avg_f64'
    call(fn, data_d.data_ptr(), o.data_ptr(), s.data_ptr(), out.data_ptr(), n_cond, row_len, _stream())
    return out


# ------------------------------------------------------------------------------------ k2
def col_mean(X):
    """Column means (float64) of an (n, d) device matrix."""
    n, d = X.shape
    out = torch.empty(d, dtype=F64, device=X.device)
    nb = lib().xps_colsum_f64_workspace(n, d)
    ws = _ws(nb)
    call('xps_colsum_f64', X.data_ptr(), _is32(X), X.stride(0), n, d, out.data_ptr(), ws.data_ptr(), nb, _stream())
    return out / n


def xcov(A, B=None, mean_a=None, mean_b=None):
    """(A - mean_a)^T (B - mean_b) in float64 on the f64 MFMA.  A (n, da), B (n, db) device."""
    B = A if B is None else B
    if B is A and mean_b is None:
        mean_b = mean_a
    n, da = A.shape
    db = B.shape[1]
    C = torch.empty(da, db, dtype=F64, device=A.device)
    nb = lib().xps_xcov_f64_workspace(n, da, db)
    ws = _ws(nb)
    call('xps_xcov_f64', A.data_ptr(), _is32(A), A.stride(0), None if mean_a is None else mean_a.data_ptr(),
         B.data_ptr(), _is32(B), B.stride(0), None if mean_b is None else mean_b.data_ptr(),
         C.data_ptr(), db, n, da, db, ws.data_ptr(), nb, _stream())
    return C


def dgemm(A, B, ta=False, tb=False):
    """op(A) @ op(B) for small float64 device matrices."""
    A, B = A.contiguous(), B.contiguous()
    M = A.shape[1] if ta else A.shape[0]
    K = A.shape[0] if ta else A.shape[1]
    N = B.shape[0] if tb else B.shape[1]
    C = torch.empty(M, N, dtype=F64, device=A.device)
    if K >= 512 and ((M + 63) // 64) * ((N + 63) // 64) <= 64:
        # few output tiles, long contraction (the skinny products of the subspace iteration): split-K slabs + one reduce
        nb = lib().xps_dgemm_splitk_workspace(M, N, K)
        ws = _ws(nb)
        call('xps_dgemm_splitk', A.data_ptr(), A.stride(0), int(ta), B.data_ptr(), B.stride(0), int(tb), C.data_ptr(), N,
             M, N, K, ws.data_ptr(), nb, _stream())
        return C
    call('xps_dgemm_small', A.data_ptr(), A.stride(0), int(ta), B.data_ptr(), B.stride(0), int(tb), C.data_ptr(), N,
         M, N, K, _stream())
    return C


def cheb_filter(Cd, A, deg, c, e, sigma1):
    """Y_deg of the scaled Chebyshev recurrence on the block A (n x m) with the symmetric operator Cd -- `deg` products
    enqueued by ONE library call (xps_cheb_filter_f64: per product a split-K launch + a reduce that applies the three-term update)."""
    A = A.contiguous()
    n, m = A.shape
    out = torch.empty_like(A)
    nb = lib().xps_cheb_filter_f64_workspace(n, m)
    ws = _ws(nb)
    call('xps_cheb_filter_f64', Cd.data_ptr(), Cd.stride(0), n, A.data_ptr(), m, int(deg), float(c), float(e), float(sigma1),
         out.data_ptr(), ws.data_ptr(), nb, _stream())
    return out


def apply(X, W, mean=None, out_f32=False):
    """(X - mean) @ W over all rows of X (..., d_in) -> (..., d_out).  W float64 (d_in, d_out)."""
    W = W.contiguous()
    X2 = X.reshape(-1, X.shape[-1])
    n, d_in = X2.shape
    d_out = W.shape[1]
    Y = torch.empty(n, d_out, dtype=torch.float32 if out_f32 else F64, device=X.device)
    call('xps_apply_f64', X2.data_ptr(), _is32(X2), X2.stride(0), None if mean is None else mean.data_ptr(),
         W.data_ptr(), W.stride(0), Y.data_ptr(), int(out_f32), d_out, n, d_in, d_out, _stream())
    return Y.view(*X.shape[:-1], d_out)


# ------------------------------------------------------------------------------------ k3 / k5
def _jacobi(Wc, n_cols, m_rows, max_sweeps=40, tol=None, want_v=True):
    """Wc: (n_cols, m_rows) float64 device tensor = column-major (m x n) matrix.  Rotates in place;
    returns V (n_cols x n_cols as rows = columns of V, i.e. V^T in row-major terms), or None when
    ``want_v`` is False (rotations not accumulated: half the work).

    Small problems (n <= 128 and the matrix within one workgroup's LDS) are decomposed by ONE launch that
    runs every sweep and the convergence test on the device; larger ones by one launch per round."""
    if tol is None:
        # converged = every pair orthogonal to rounding level; the largest |cos| over n^2/2 pairs of length-m
        # columns cannot be pushed below a few eps * sqrt(m), so the bound grows slowly with the size
        tol = max(1e-14, 4.0 * EPS * np.sqrt(max(m_rows, n_cols)) * np.log2(max(n_cols, 2)))
    Vc = torch.empty(n_cols, n_cols, dtype=F64, device=Wc.device) if want_v else None
    if lib().xps_jacobi_small_supported(m_rows, n_cols, int(want_v)):
        call('xps_jacobi_small_f64', Wc.data_ptr(), Wc.stride(0), 0, None if Vc is None else Vc.data_ptr(), n_cols, 0,
             m_rows, n_cols, 1, max_sweeps, tol, None, None, _stream())
        return Vc
    if want_v:
        Vc.copy_(torch.eye(n_cols, dtype=F64, device=Wc.device))
    off = torch.zeros(1, dtype=F64, device=Wc.device)
    done = 0
    while done < max_sweeps:
        k = 5 if done == 0 else 1
        call('xps_jacobi_sweeps_f64', Wc.data_ptr(), Wc.stride(0), None if Vc is None else Vc.data_ptr(), n_cols, m_rows,
             n_cols, k, off.data_ptr(), None, 0, _stream())
        done += k
        if off.item() <= tol:
            break
    return Vc


def jacobi_batched(Wb, want_v=True, max_sweeps=40, tol=None):
    """Wb: (batch, n, m) float64 device tensor, each [b] a column-major m x n matrix; all decomposed by one
    launch (one workgroup per matrix).  Returns Vb (batch, n, n) or None."""
    batch, n, m = Wb.shape
    if tol is None:
        tol = max(1e-14, 4.0 * EPS * np.sqrt(max(m, n)) * np.log2(max(n, 2)))
    if not lib().xps_jacobi_small_supported(m, n, int(want_v)):
        raise ValueError(f'jacobi_batched: {m} x {n} does not fit the single-workgroup kernel')
    Vb = torch.empty(batch, n, n, dtype=F64, device=Wb.device) if want_v else None
    call('xps_jacobi_small_f64', Wb.data_ptr(), Wb.stride(1), Wb.stride(0), None if Vb is None else Vb.data_ptr(), n,
         n * n, m, n, batch, max_sweeps, tol, None, None, _stream())
    return Vb


def chol_whiten_blocks(Gd, offs, scale, shift, LHS=None):
    """Upper-triangular whitening factors S_b = L_b^-T of the diagonal blocks A_b = scale * G_bb + shift * I = L_b L_b^T of the
    device matrix Gd (D x D), written into a block-diagonal D x D device matrix S (S^T blockdiag(A_b) S = I); A_b is also written
    over the diagonal blocks of ``LHS`` when given.  One launch when the blocks have one size.  Returns S, or None when a block is
    too large for the LDS-resident kernel or is not positive definite (the caller then takes the eigendecomposition route)."""
    D = Gd.shape[0]
    sizes = [int(offs[b + 1] - offs[b]) for b in range(len(offs) - 1)]
    if not all(lib().xps_chol_whiten_supported(n) for n in sizes):
        return None
    S = torch.zeros_like(Gd)
    info = torch.zeros(len(sizes), dtype=torch.int32, device=Gd.device)
    ld = Gd.stride(0)
    groups = [(0, len(sizes))] if len(set(sizes)) == 1 else [(b, 1) for b in range(len(sizes))]
    for b0, cnt in groups:
        n, o = sizes[b0], int(offs[b0])
        step = n * (ld + 1) * 8
        call('xps_chol_whiten_f64', Gd.data_ptr() + o * (ld + 1) * 8, ld, n * (ld + 1), float(scale), float(shift),
             None if LHS is None else LHS.data_ptr() + o * (LHS.stride(0) + 1) * 8, 0 if LHS is None else LHS.stride(0),
             0 if LHS is None else n * (LHS.stride(0) + 1), S.data_ptr() + o * (S.stride(0) + 1) * 8, S.stride(0),
             n * (S.stride(0) + 1), n, cnt, info.data_ptr() + 4 * b0, _stream())
    if int(info.abs().max().item()) != 0:
        return None
    return S


def svd(A):
    """Thin SVD of a small float64 device matrix by one-sided Jacobi: A = U diag(s) Vt, s descending.
    Returns numpy (U, s, Vt)."""
    A = A.to(F64)
    m, n = A.shape
    if m < n:
        U, s, Vt = svd(A.t().contiguous())
        return Vt.T, s, U.T
    Wc = A.t().contiguous().clone()                 # row j = column j of A
    Vc = _jacobi(Wc, n, m)
    W = Wc.cpu().numpy()                            # (n, m): row j = u_j * s_j
    V = Vc.cpu().numpy()                            # (n, n): row j = v_j
    s = np.linalg.norm(W, axis=1)
    order = np.argsort(-s, kind='stable')
    s, W, V = s[order], W[order], V[order]
    U = np.zeros_like(W)
    nz = s > 0
    U[nz] = W[nz] / s[nz, None]
    return U.T, s, V


# Eigenvectors without accumulated rotations.  After the one-sided Jacobi W = C V = V diag(w), so for a
# positive-definite C the eigenvectors are the normalised columns of W (half the work, and for n = 128 the
# whole problem then fits one workgroup's LDS).  The direction of column j carries a relative error of about
# eps * w_max / w_j, so every PSD matrix is first shifted by sigma = 2^-10 * (Gershgorin bound of w_max):
# same eigenvectors, spectrum within a factor 2^10 -> direction error <= ~2e-13, and the eigenvalues w - sigma
# keep the absolute accuracy eps * w_max of a LAPACK eigensolver (the reference's numpy / sklearn calls).
_SHIFT = 2.0 ** -10


def _eig_from_w(W, sigma):
    w = np.linalg.norm(W, axis=1)
    if len(w) == 0 or not np.all(w > 0.5 * _SHIFT * w.max()):
        return None
    order = np.argsort(-w, kind='stable')
    return np.maximum(w[order] - sigma, 0.0), (W[order] / w[order, None]).T


def _shift_of(C):
    g = float(C.abs().sum(dim=-1).max().item())
    return g * _SHIFT if g > 0 else 1.0


def eigh_psd(C, well_conditioned=False):
    """Eigendecomposition of a symmetric PSD float64 device matrix (Jacobi): returns numpy
    (w descending, V with eigenvectors in columns).  ``well_conditioned``: the caller guarantees a spectrum
    bounded away from zero (regularised / already shifted matrices): no extra shift is applied."""
    C = C.to(F64)
    n = C.shape[0]
    sigma = 0.0 if well_conditioned else _shift_of(C)
    Wc = C.contiguous().clone()
    if sigma:
        Wc.diagonal().add_(sigma)
    _jacobi(Wc, n, n, want_v=False)
    res = _eig_from_w(Wc.cpu().numpy(), sigma)
    if res is not None:
        return res
    if well_conditioned:                             # the caller's promise did not hold: shift after all
        return eigh_psd(C, False)
    Wc = C.contiguous().clone()                      # not PSD (negative eigenvalue beyond the shift): accumulate V
    Vc = _jacobi(Wc, n, n)
    W = Wc.cpu().numpy()
    V = Vc.cpu().numpy()
    w = np.linalg.norm(W, axis=1)
    order = np.argsort(-w, kind='stable')
    return w[order], V[order].T


def eigh_psd_batched(Cs, well_conditioned=False):
    """eigh_psd of a list of equally sized small PSD matrices in one launch."""
    n = Cs[0].shape[0]
    if len({tuple(c.shape) for c in Cs}) != 1 or not lib().xps_jacobi_small_supported(n, n, 0):
        return [eigh_psd(to_device(c), well_conditioned) for c in Cs]
    Cb = torch.stack([to_device(c).to(F64) for c in Cs]).contiguous()
    sig = [0.0 if well_conditioned else _shift_of(Cb[i]) for i in range(len(Cs))]
    Wb = Cb.clone()
    for i, sg in enumerate(sig):
        if sg:
            Wb[i].diagonal().add_(sg)
    jacobi_batched(Wb, want_v=False)
    res = [_eig_from_w(W, sg) for W, sg in zip(Wb.cpu().numpy(), sig)]
    return [r if r is not None else eigh_psd(Cb[i], well_conditioned) for i, r in enumerate(res)]


def _eigh_sym_top_full(C, k):
    """All eigenpairs by one-sided Jacobi on the shifted matrix, the first k returned (small n, or k close to n)."""
    n = C.shape[0]
    mu = float(C.abs().sum(dim=1).max().item())
    Cs = C + (2.0 * mu) * torch.eye(n, dtype=F64, device=C.device)
    w, V = eigh_psd(Cs, well_conditioned=True)
    return (w - 2.0 * mu)[:k], V[:, :k]


_SUBSPACE_START = {}


def _subspace_start(n, m, dev):
    """Deterministic full-rank start block (the same for every call of a shape: fits are reproducible run to run)."""
    key = (n, m, str(dev))
    if key not in _SUBSPACE_START:
        if len(_SUBSPACE_START) > 16:
            _SUBSPACE_START.clear()
        _SUBSPACE_START[key] = torch.from_numpy(np.random.default_rng(20240229).standard_normal((n, m))).to(dev)
    return _SUBSPACE_START[key]


def _lanczos_bounds(C, steps=24):
    """(lo, hi) enclosing the spectrum of the symmetric device matrix C from `steps` Lanczos steps (no
    reorthogonalisation: only the two extreme Ritz values are used): extreme Ritz value -/+ its residual bound
    |beta_j s_j|, widened by 2 % of the width.  The recurrence runs on the device without synchronising; the
    steps x steps tridiagonal matrix is diagonalised by the small device Jacobi."""
    C = C.contiguous()
    n = C.shape[0]
    steps = min(steps, n)
    dev = C.device
    v = torch.sin(0.7 * torch.arange(1, n + 1, dtype=F64, device=dev)) + 0.01
    v = (v / torch.linalg.vector_norm(v)).contiguous()
    ab_d = torch.empty(2 * steps, dtype=F64, device=dev)
    nb = lib().xps_lanczos_f64_workspace(n)
    ws = _ws(nb)
    # (the whole recurrence is enqueued by one library call: 2 launches per step; the Python loop paid ~10 launches per step)
    call('xps_lanczos_f64', C.data_ptr(), C.stride(0), n, steps, v.data_ptr(), ab_d.data_ptr(), ab_d.data_ptr() + 8 * steps,
         ws.data_ptr(), nb, _stream())
    ab = ab_d.cpu().numpy()
    a, b = ab[:steps], ab[steps:]
    if not np.isfinite(ab).all():
        return None
    T = np.diag(a) + np.diag(b[:-1], 1) + np.diag(b[:-1], -1)
    shift = float(np.abs(T).sum(axis=1).max()) * 1.5 + 1e-300           # Gershgorin: T + shift I is positive definite
    th, S = eigh_psd(to_device(T + shift * np.eye(steps)), well_conditioned=True)
    th = th - shift                                                      # descending
    hi = th[0] + abs(b[-1] * S[-1, 0])
    lo = th[-1] - abs(b[-1] * S[-1, -1])
    pad = 0.02 * (hi - lo)
    return lo - pad, hi + pad


def _orthonormal_columns_cholqr3(Y):
    """Orthonormal basis of the column span of Y (n x m, m << n) by SHIFTED CHOLESKY QR in three passes (Fukaya, Kannan,
    Nakatsukasa, Yamamoto, Yanagisawa 2020) on the column-normalised block: pass 1 factors G + s I with
    s = 11 (n m + m (m + 1)) eps ||G|| (a factor exists whatever the condition number; it brings the block to a condition of
    ~sqrt(1 / eps) at worst), passes 2 and 3 are plain Cholesky QR.  Per pass: Gram matrix and the triangular solve as f64 MFMA
    products on the device, the m x m (<= 48 x 48) factor on the host.  ~0.4 ms against ~1.3 ms for the one-sided Jacobi on the
    tall block (three workgroups busy).  None when a factorisation fails or the result is not orthonormal (-> the Jacobi route)."""
    n, m = Y.shape
    s = torch.linalg.vector_norm(Y, dim=0)
    smin, smax = (float(x) for x in torch.stack([s.min(), s.max()]).cpu())
    if not np.isfinite(smax) or smin <= 0.0:
        return None
    A = (Y / s).contiguous()
    for p in range(3):
        G = dgemm(A, A, ta=True).cpu().numpy()
        G = 0.5 * (G + G.T)
        if not np.isfinite(G).all():
            return None
        if p == 0:
            G = G + (11.0 * (n * m + m * (m + 1)) * EPS * float(np.linalg.norm(G, 2))) * np.eye(m)
        try:
            Linv = np.linalg.inv(np.linalg.cholesky(G))
        except np.linalg.LinAlgError:
            return None
        A = dgemm(A, to_device(np.ascontiguousarray(Linv.T)))
    G = dgemm(A, A, ta=True).cpu().numpy()
    if not np.isfinite(G).all() or np.abs(G - np.eye(m)).max() > 1e-12:
        return None
    return A


def _orthonormal_columns(Y):
    """Orthonormal basis of the column span of Y (n x m device matrix, m << n).  Shifted Cholesky QR (above) where it
    succeeds; else -- and always with XPS_TOPK_ORTHO=jacobi -- one-sided Jacobi on Y itself (no Gram matrix: a filtered block
    whose columns differ in size by 1e7 keeps its small directions).  None if rank was lost."""
    if os.environ.get('XPS_TOPK_ORTHO', 'chol') != 'jacobi':
        Q = _orthonormal_columns_cholqr3(Y)
        if Q is not None:
            return Q
    Wt = Y.t().contiguous()
    m, n = Wt.shape
    _jacobi(Wt, m, n, want_v=False)
    s = torch.linalg.vector_norm(Wt, dim=1)
    smin, smax = (float(x) for x in torch.stack([s.min(), s.max()]).cpu())
    if not np.isfinite(smax) or smin <= smax * 1e-13:
        return None
    return (Wt / s[:, None]).t().contiguous()


def _reorthonormalise(A):
    """Orthonormal basis of the span of an ALREADY nearly orthonormal block (the purge of the locked directions moved it by
    rounding errors only): Cholesky QR -- Gram matrix on the device (f64 MFMA), its 45 x 45 Cholesky factor on the host, one
    product -- instead of a second one-sided Jacobi on the tall block (~1 ms each).  The Gram matrix of such a block is
    within rounding of the identity, so squaring the condition number costs nothing; anything else takes the robust route."""
    G = dgemm(A, A, ta=True).cpu().numpy()
    G = 0.5 * (G + G.T)
    if not np.isfinite(G).all() or np.abs(G - np.eye(G.shape[0])).max() > 1e-3:
        return _orthonormal_columns(A)
    Linv = np.linalg.inv(np.linalg.cholesky(G))                    # G = L L^T  ->  A L^-T is orthonormal
    return dgemm(A, to_device(np.ascontiguousarray(Linv.T)))


def eigh_sym_top(C, k, tol=2e-14, max_outer=40, max_degree=40, stats=None):
    """Top-k (algebraically largest) eigenpairs of a symmetric, possibly indefinite, float64 device matrix:
    numpy (w descending, V with eigenvectors in columns).

    The MCCA fit (AlignMCCA.py) needs n_components (10-30) of D = 512-1024 pairs; a full Jacobi diagonalisation
    spends its time on the D - k pairs nobody reads (76 of the 130 ms of an 8-view fit).  Here: Chebyshev-filtered
    subspace iteration with locking on a block of m = k + buffer vectors.  Per outer step the active block is
    multiplied `degree` times by the (deflated) matrix -- f64 MFMA GEMMs; the three-term recurrence damps the interval
    [lower spectrum bound, smallest Ritz value of the block] and is scaled so that the largest active Ritz value maps
    to 1 (Zhou & Saad) -- then orthonormalised (one-sided Jacobi on the tall block) and Rayleigh-Ritz-projected (small
    device Jacobi).  Leading pairs whose residual ||C v - theta v|| <= tol * ||C|| are locked: they leave the block
    and the operator moves their eigenvalues to the lower bound (C - L (Theta - lo) L^T), so that a well separated
    group of large eigenvalues (the shared latents of MCCA: 15 against a bulk below 2.1) does not cap the degree that
    the pairs inside the bulk need.  The degree adapts to an amplification of <= 2e7 per outer step across the block.
    Small problems (n <= 640) and wide requests (block over 128 columns or over n / 3) take the full decomposition,
    and so does any run that loses rank or does not converge.  ``stats``: receives iteration counts."""
    C = C.to(F64).contiguous()
    n = C.shape[0]
    k = min(k, n)
    m = k + max(8, (k + 1) // 2)
    # the full Jacobi costs ~73 ms x (n / 1024)^3, the subspace iteration 9-35 ms almost independent of n: worth it above ~640
    if n <= 640 or m > 128 or 3 * m > n or not lib().xps_jacobi_small_supported(m, m, 0):
        return _eigh_sym_top_full(C, k)
    dev = C.device
    bounds = _lanczos_bounds(C)
    if bounds is None or not bounds[1] > bounds[0]:
        return _eigh_sym_top_full(C, k)
    lo, hi = bounds
    scale = max(abs(lo), abs(hi))
    L, tl = None, np.zeros(0)                          # locked vectors (n x l device) and their eigenvalues
    Cd = C                                             # deflated operator C - L (Theta - lo) L^T, rebuilt when pairs lock

    def op(X):
        return dgemm(Cd, X)

    def purge(X):                                      # remove what rounding re-introduced of the locked directions
        return X if L is None else X - dgemm(L, dgemm(L, X, ta=True))

    A = _orthonormal_columns(op(op(_subspace_start(n, m, dev))))
    theta, nprod = None, 2
    for outer in range(max_outer):
        if A is None:
            break
        if theta is not None:
            b = float(theta[-1])
            e, c = 0.5 * (b - lo), 0.5 * (b + lo)
            xtop = (float(theta[0]) - c) / e
            if not (e > 0 and xtop > 1.0):
                break
            deg = int(np.log(2e7) / np.arccosh(xtop))
            deg = max(2, min(max_degree, deg))
            sigma1 = e / (float(theta[0]) - c)
            Y = cheb_filter(Cd, A, deg, c, e, sigma1)       # (the whole recurrence: one library call, 2 launches per product)
            nprod += deg
            if outer > 12 or nprod > 260:              # a spectrum this method is not made for: stop paying for it
                break
            A = _orthonormal_columns(purge(Y))
            if A is None:
                break
            if L is not None:
                A = _reorthonormalise(purge(A))
                if A is None:
                    break
        # Rayleigh-Ritz of the (deflated) operator on the active block; H + shift is positive definite
        CA = op(A)
        nprod += 1
        Hm = dgemm(A, CA, ta=True)
        shift = (hi - lo) - lo
        w, Z = eigh_psd(0.5 * (Hm + Hm.t()) + shift * torch.eye(A.shape[1], dtype=F64, device=dev), well_conditioned=True)
        theta = w - shift
        Zd = to_device(np.ascontiguousarray(Z))
        A = dgemm(A, Zd)
        R = dgemm(CA, Zd) - A * torch.from_numpy(theta).to(dev)
        res = torch.linalg.vector_norm(R, dim=0).cpu().numpy()
        if theta[-1] < lo:                             # the lower bound was not one: widen it (the filter stays safe)
            lo = float(theta[-1]) - 0.05 * (hi - lo)
        have = 0 if L is None else L.shape[1]
        nl = 0
        while nl < len(theta) and have + nl < k and res[nl] <= tol * scale:
            nl += 1
        if stats is not None:
            stats.update(outer=outer + 1, products=nprod, locked=have + nl)
        if nl:
            Ln = A[:, :nl].contiguous()
            Cd = Cd - dgemm(Ln * torch.from_numpy(theta[:nl] - lo).to(dev), Ln, tb=True)
            L = Ln if L is None else torch.cat([L, Ln], dim=1).contiguous()
            tl = np.concatenate([tl, theta[:nl]])
            A, theta = A[:, nl:].contiguous(), theta[nl:]
        if L is not None and L.shape[1] >= k:
            order = np.argsort(-tl, kind='stable')     # (locking is in Ritz order; equal within rounding at worst)
            return tl[order][:k], L.cpu().numpy()[:, order][:, :k]
        if A.shape[1] < 2:
            break
    if stats is not None:
        stats['fallback'] = True
    return _eigh_sym_top_full(C, k)                    # rank loss / no convergence: the safe path


def svd_tall_device(A, mean=None):
    """Thin SVD of a tall (n x d, n >= d) device matrix by one-sided Jacobi ON THE MATRIX ITSELF (no Gram matrix: singular
    values keep their relative accuracy, so the numerical rank can follow LAPACK's rule and ill-conditioned data do not lose
    half their digits).  ``mean``: column means subtracted first (apply kernel).  Returns (Wt, s, V): Wt (d x n) device
    tensor whose row j is u_j * s_j, s (numpy, descending) and V (numpy, columns = right singular vectors), rows / columns
    in the same (descending) order."""
    A = A if A.dim() == 2 else A.reshape(-1, A.shape[-1])
    n, d = A.shape
    eye = torch.eye(d, dtype=F64, device=A.device)
    Ac = apply(A, eye, mean)                          # centred float64 copy: (A - mean) I on the apply kernel
    Wt = Ac.t().contiguous()                          # row j = column j
    Vc = _jacobi(Wt, d, n)
    s = torch.linalg.vector_norm(Wt, dim=1).cpu().numpy()
    order = np.argsort(-s, kind='stable')
    idx = torch.as_tensor(order, device=A.device)
    return Wt[idx].contiguous(), s[order], Vc.cpu().numpy()[order].T


def rank_from_singular_values(s, shape):
    """numpy.linalg.matrix_rank semantics (the reference's AlignCCA.py:263-264): s > s_max * max(shape) * eps."""
    if len(s) == 0 or s[0] <= 0:
        return 0
    return int((s > s[0] * max(shape) * EPS).sum())


def rank_from_gram_eigs(w, n_rows):
    """Numerical rank of an (n_rows x d) matrix from the eigenvalues of its Gram matrix.

    LAPACK's matrix_rank (used by the reference, AlignCCA.py:263-264) thresholds singular values
    at s_max * max(shape) * eps.  A Gram-based method cannot resolve singular values below
    sqrt(eps)-ish of s_max, so the threshold here is on the eigenvalues: w > w_max * max(shape) *
    eps (i.e. s > s_max * sqrt(max(shape) * eps)).  The two rules agree unless the condition number
    of the data lies between ~1e5 and ~1e10 (documented in DESIGN.md)."""
    if len(w) == 0 or w[0] <= 0:
        return 0
    tol = w[0] * max(n_rows, len(w)) * EPS
    return int((w > tol).sum())


def pinv_small(M):
    """numpy.linalg.pinv semantics (rcond = 1e-15 on singular values) through the Jacobi SVD."""
    U, s, Vt = svd(to_device(M))
    cutoff = 1e-15 * (s.max() if len(s) else 0.0)
    inv = np.where(s > cutoff, 1.0 / np.where(s > cutoff, s, 1.0), 0.0)
    return dgemm(to_device(Vt.T * inv), to_device(U), tb=True).cpu().numpy()
