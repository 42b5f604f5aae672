"""Multiview CCA alignment on the MI355X — drop-in surface of the reference's
``alignment/AlignMCCA.py`` (AlignMCCA :13-138, get_MCCA_transforms :140, n_components_var :156).

The reference delegates the arithmetic to ``mvlearn.embed.MCCA`` (:9, :152-153), which is not
vendored, pinned or installed here: PARITY WITH MVLEARN IS UNPINNED.  This module
implements the published regularised SUMCOR-MCCA generalised eigenproblem on the device:

  G   = Zc^T Zc                 centred Gram of the concatenated views    xps_xcov_f64 (f64 MFMA)
  R_b = (1 - r) G_bb + r I      per-view regularised covariance
  LHS = G with diagonal blocks R_b ;  RHS = blockdiag(R_b)
  R_b = L L^T    ->  S = blockdiag(L^-T)   (regs > 0)                     xps_chol_whiten_f64, all views in one launch
  R_b = V L V^T  ->  S = blockdiag(V L^-1/2 V^T)  (regs None / 0: pseudo-inverse of a singular block)   xps_jacobi_* per view
  C   = S^T LHS S ,  top-k eigenpairs of C (Chebyshev-filtered subspace iteration)   xps_cheb_filter_f64 / xps_jacobi_*
  loadings = rows of S V_k per view; sign rule on the normalised common scores
  transform_view(X, i) = (X - mean_i) @ loadings_i                        xps_apply_f64
"""
import os

import numpy as np

from . import _linalg as LA
from .alignment_utils import _group_conditions_device


class DeviceMCCA:
    """The object kept in ``AlignMCCA.mcca``: exposes what the reference uses of mvlearn's MCCA —
    ``loadings_`` (list of (d_b, k) float64 arrays), ``transform_view(X2d, i)`` — plus ``means_``
    and ``evals_``."""

    def __init__(self, n_components=10, regs=0.5, signal_ranks=None):
        self.n_components, self.regs, self.signal_ranks = n_components, regs, signal_ranks

    def fit(self, views, group=None):
        """views: list of (n, d_b) arrays / device tensors (one row per condition-time sample).

        ``group`` (a torch.distributed group with more than one rank): the fit is SHARDED BY PATIENT: view p is
        owned by rank p % world and only its owner needs to hold it (other entries may be None).  The owners broadcast their
        views (n x d_p float64: 13 MB at the north-star shape), every rank computes the block rows
        C_{p,.} = (L_p - mean_p)^T [L_1 - mean_1 ... L_P - mean_P] of ITS views on its own GPU (xps_xcov_f64) and the block rows
        are exchanged (D x D float64 in all: 8 MB at D = 1024); the eigensolve is replicated (deterministic, no broadcast).
        The single-process fit computes the same block rows with the same launches, so both give the same bits."""
        world, rank = _group_world_rank(group)
        P = len(views)
        Vd = [None] * P
        for i, v in enumerate(views):
            if i % world == rank:
                t = LA.to_device(v)
                Vd[i] = t.reshape(-1, t.shape[-1]).to(LA.F64).contiguous()
        if world > 1:
            Vd = [_bcast_matrix(Vd[i], i % world, group) for i in range(P)]
        dims = [v.shape[1] for v in Vd]
        offs = np.concatenate([[0], np.cumsum(dims)])
        Z = LA.torch.cat(Vd, dim=1).contiguous()
        mean = LA.torch.cat([LA.col_mean(v) for v in Vd])
        # centred Gram of the concatenated views, one block row per view (f64 MFMA); sharded: own rows only, then exchanged
        rows = self.own_block_rows(Vd, Z, mean, offs, world, rank)
        if world > 1:
            rows = [_bcast_matrix(rows[i], i % world, group) for i in range(P)]
        Gd = LA.torch.cat(rows, dim=0).contiguous()           # (D, D), stays on the device for the eigensolve
        self._gram_d = Gd                                     # centred Gram of the concatenated views (before any rank reduction)
        self.block_rows_computed_ = [i for i in range(P) if i % world == rank]
        self.means_ = [mean[offs[i]:offs[i + 1]].cpu().numpy() for i in range(len(Vd))]
        self._means_d = [mean[offs[i]:offs[i + 1]].contiguous() for i in range(len(Vd))]
        bases = None
        if self.signal_ranks is not None:
            # per-view rank-k PCA basis = top eigenvectors of G_bb; the problem is solved on the scores
            bases = []
            for i, r in enumerate(self.signal_ranks):
                _, Vb = LA.eigh_psd(Gd[int(offs[i]):int(offs[i + 1]), int(offs[i]):int(offs[i + 1])].contiguous())
                bases.append(Vb[:, :max(int(r), 1)])
            Bm = LA.to_device(_block_diag(bases))
            Gd = LA.dgemm(LA.dgemm(Bm, Gd, ta=True), Bm)
            dims = [b.shape[1] for b in bases]
            offs = np.concatenate([[0], np.cumsum(dims)])
        load_red, self.evals_ = _gevp(Gd, offs, self.n_components, self.regs)
        if bases is not None:
            load_red = [LA.dgemm(LA.to_device(b), LA.to_device(l)).cpu().numpy() for b, l in zip(bases, load_red)]
        # sign rule: entry of largest magnitude of each normalised common-score column is positive
        Vfull = LA.to_device(np.vstack(load_red))
        common = LA.apply(Z, Vfull, mean).cpu().numpy()
        common = common / np.linalg.norm(common, axis=0)
        rows = np.argmax(np.abs(common), axis=0)
        signs = np.sign(common[rows, np.arange(common.shape[1])])
        signs[signs == 0] = 1
        self.loadings_ = [l * signs for l in load_red]
        self._load_d = [LA.to_device(np.ascontiguousarray(l)) for l in self.loadings_]
        self.n_views_ = len(Vd)
        return self

    @property
    def gram_(self):
        """The centred Gram matrix of the concatenated views as a host array (copied from the device on demand)."""
        return self._gram_d.cpu().numpy()

    @staticmethod
    def own_block_rows(Vd, Z, mean, offs, world, rank):
        """Block rows C_{p,.} = (L_p - mean_p)^T [L_1 - mean_1 ... L_P - mean_P] of the views rank `rank` of `world` owns
        (p % world == rank; None for the others): one xps_xcov_f64 launch group per view -- the SAME launches whatever the
        world size, which is what makes the sharded fit bit-identical to the single-process one."""
        return [LA.xcov(Vd[i], Z, mean[offs[i]:offs[i + 1]].contiguous(), mean) if i % world == rank else None
                for i in range(len(Vd))]

    def transform_view(self, X, view):
        return LA.like_input(LA.apply(LA.to_device(X), self._load_d[view], self._means_d[view]), X)

    def transform(self, Xs):
        return [self.transform_view(x, i) for i, x in enumerate(Xs)]


def _block_diag(blocks):
    R, C = sum(b.shape[0] for b in blocks), sum(b.shape[1] for b in blocks)
    out = np.zeros((R, C))
    r = c = 0
    for b in blocks:
        out[r:r + b.shape[0], c:c + b.shape[1]] = b
        r += b.shape[0]
        c += b.shape[1]
    return out


def _group_world_rank(group):
    if group is None:
        return 1, 0
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _bcast_matrix(t, owner, group):
    """Broadcast a 2-D float64 device matrix from rank `owner` of `group` (its shape first); returns it on every rank."""
    import torch.distributed as dist
    dev = LA.device()
    src = dist.get_global_rank(group, owner) if hasattr(dist, 'get_global_rank') else owner
    shape = LA.torch.tensor(list(t.shape) if t is not None else [0, 0], dtype=LA.torch.int64, device=dev)
    dist.broadcast(shape, src=src, group=group)
    if t is None:
        t = LA.torch.empty(int(shape[0]), int(shape[1]), dtype=LA.F64, device=dev)
    t = t.contiguous()
    dist.broadcast(t, src=src, group=group)
    return t


def _gevp(Gd, offs, n_components, regs):
    """Top generalised eigenpairs of (LHS, RHS) built from the Gram matrix Gd (DEVICE tensor, D x D); every matrix of size
    D stays on the device (round 4: the 8-MB Gram / LHS / S matrices used to cross PCIe four times through pageable host memory:
    3-30 ms of an 8-view fit, varying from call to call); the host sees the small per-view spectra and the k eigenpairs only.
    Returns per-view loadings (host, float64) and the eigenvalues."""
    torch = LA.torch
    P = len(offs) - 1
    D = Gd.shape[0]
    LHS = Gd.clone()
    S = None
    if regs and os.environ.get('XPS_MCCA_WHITEN', 'chol') != 'eig':
        # regularised blocks R_b = (1 - r) G_bb + r I are positive definite with condition <= (lambda_max + r) / r: whitened by
        # their Cholesky factors, S = blockdiag(L_b^-T), in ONE launch (the symmetric R_b^-1/2 from eight 128 x 128 Jacobi
        # eigendecompositions was 5 of the 17.5 ms of this function; the eigenvectors S u of the pencil are the same)
        S = LA.chol_whiten_blocks(Gd, offs, 1.0 - regs, regs, LHS)
    if S is None:
        S = torch.zeros_like(Gd)
        blocks = []
        for b in range(P):
            sl = slice(int(offs[b]), int(offs[b + 1]))
            Rb = Gd[sl, sl].contiguous()
            if regs is not None:
                Rb = Rb.mul(1.0 - regs)
                Rb.diagonal().add_(regs)
            LHS[sl, sl] = Rb
            blocks.append(Rb)
        # all diagonal blocks in one launch when they have one size
        eigs = LA.eigh_psd_batched(blocks)
        for b, (w, V) in enumerate(eigs):
            sl = slice(int(offs[b]), int(offs[b + 1]))
            keep = w > w[0] * max(blocks[b].shape[0], 1) * LA.EPS   # guards a singular unregularised block
            Vk = V[:, keep] / np.sqrt(w[keep])
            S[sl, sl] = LA.dgemm(LA.to_device(Vk), LA.to_device(V[:, keep]), tb=True)       # R_b^-1/2 (symmetric)
    Cm = LA.dgemm(LA.dgemm(S, LHS, ta=True), S)                 # S^T LHS S
    Cm = 0.5 * (Cm + Cm.t())
    k = min(n_components, D)
    # Cm = S G S + blockdiag(I - R_b^-1/2 G_bb R_b^-1/2) is indefinite once regs > 0 (eigenvalues down to
    # -regs / (1 - regs)): eigh_sym_top shifts it positive definite before the (sign-blind) one-sided Jacobi
    w, Vc = LA.eigh_sym_top(Cm, k)
    Vg = LA.dgemm(S, LA.to_device(np.ascontiguousarray(Vc))).cpu().numpy()      # RHS-orthonormal
    return [Vg[int(offs[b]):int(offs[b + 1])] for b in range(P)], w


class AlignMCCA:
    """MCCA-based alignment of multiple neural datasets into a shared space.

    Attributes:
        n_components (int), regs (float), pca_var (float): as in the reference (:27).
        mcca: fitted DeviceMCCA (set after fit) with ``loadings_`` and ``transform_view``.
    """

    def __init__(self, n_components=10, regs=0.5, pca_var=1):
        self.n_components = n_components
        self.regs = regs
        self.pca_var = pca_var

    def get_params(self, deep=True):
        return {'n_components': self.n_components, 'regs': self.regs, 'pca_var': self.pca_var}

    def set_params(self, **params):
        for k, v in params.items():
            setattr(self, k, v)
        return self

    def fit(self, X, y, group=None):
        """``group``: shard the fit by patient over a torch.distributed group (view p owned by rank p % world; X[p] may be
        None on the other ranks, y must hold the labels of EVERY view on every rank: they decide the shared conditions)."""
        self.mcca = get_MCCA_transforms(X, y, n_components=self.n_components, regs=self.regs,
                                        pca_var=self.pca_var, group=group)

    def transform(self, X, idx=-1):
        if not self._check_fit():
            raise RuntimeError('Must call fit() before transforming data.')
        if idx == -1:
            return self._transform_multiple(X)
        if idx >= len(self.mcca.loadings_):
            raise IndexError('Input idx is greater than the number of learned '
                             'transforms. For transformation of data from a '
                             'specific session, provide the input idx as the '
                             'index of the session in the input list. If '
                             'transforming multiple sessions, set idx=-1 '
                             '(default).')
        return self._transform_single(X, idx)

    def fit_transform(self, X, y):
        self.fit(X, y)
        return self.transform(X)

    def _transform_multiple(self, X):
        return (*[self._transform_single(x, i) for i, x in enumerate(X)],)

    def _transform_single(self, X, idx):
        shape = tuple(X.shape)
        out = self.mcca.transform_view(X.reshape(-1, shape[-1]), idx)
        return out.reshape(shape[:-1] + (-1,))

    def _check_fit(self):
        try:
            self.mcca
        except AttributeError:
            return False
        return True


def get_MCCA_transforms(features, labels, n_components=10, regs=0.5, pca_var=1, group=None):
    """Condition averages of the shared conditions -> flattened views -> optional per-view signal
    ranks from the PCA variance of the RAW data (:146-150) -> MCCA fit.  ``group``: sharded by patient (see DeviceMCCA.fit):
    a rank reads the raw trials of ITS views only (condition means, signal ranks)."""
    world, rank = _group_world_rank(group)
    own = [i % world == rank for i in range(len(features))]
    avgs = _group_conditions_device(features, labels, own=own)
    avgs = [None if a is None else a.reshape(-1, a.shape[-1]) for a in avgs]
    ranks = None
    if pca_var > 0 and pca_var < 1:
        ranks = [min(n_components, n_components_var(x, pca_var)) if o else 0 for x, o in zip(features, own)]
        if world > 1:
            import torch.distributed as dist
            t = LA.torch.tensor(ranks, dtype=LA.torch.int64, device=LA.device())
            dist.all_reduce(t, group=group)                  # (every entry is non-zero on exactly one rank)
            ranks = [int(v) for v in t.cpu()]
    return DeviceMCCA(n_components=n_components, regs=regs, signal_ranks=ranks).fit(avgs, group=group)


def n_components_var(X, var):
    """Index of the first component at which the cumulative variance of the UNCENTRED data exceeds
    ``var`` (reference :156-174).  NB this is ``argmax`` of a boolean array — a 0-based index, one
    less than the component count; kept bug-compatible.  Squared singular values = eigenvalues of
    X^T X (device Gram + Jacobi)."""
    Xd = LA.to_device(X)
    Xd = Xd.reshape(-1, Xd.shape[-1])
    w, _ = LA.eigh_psd(LA.xcov(Xd))
    s = w / np.sum(w)
    return int(np.argmax(np.cumsum(s) > var))
