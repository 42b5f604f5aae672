"""Oracle: restatement of alignment/AlignMCCA.py around a regularised MCCA.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED for the third-party call.  The reference delegates the
arithmetic to ``mvlearn.embed.MCCA`` (alignment/AlignMCCA.py:9,152-153,
transform_view :110,125, loadings_ :78).  mvlearn is not vendored, not pinned in
environment.yml / requirements.txt, not installed in this image, and the
reference holds no test or golden vector at that boundary.  What is restated
here is the published regularised SUMCOR-MCCA generalised eigenproblem as
mvlearn 0.5.x implements it to the best of the author's recollection:

  * every view is mean-centred; the mean is kept for transform_view;
  * RHS_b = (1 - r) X_b^T X_b + r I ;  LHS[a][b] = X_a^T X_b (a != b),
    LHS[b][b] = RHS_b ;  scipy.linalg.eigh(LHS, blockdiag(RHS)) -> the
    n_components largest generalised eigenpairs (eigenvectors RHS-orthonormal);
  * loadings_b = the view's block of rows of the eigenvectors;
  * sign rule: common scores = sum_b X_b loadings_b, columns normalised; each
    component is flipped so the entry of largest magnitude of its normalised
    common score is positive;
  * with signal_ranks: each centred view is first replaced by its rank-k PCA
    scores X_b V_b (thin SVD), the same problem is solved on the scores and the
    loadings are mapped back with V_b.

Everything AROUND that call follows alignment/AlignMCCA.py:140-174 line by line
(including n_components_var's off-by-one, :174, which is kept).  The test-suite
pins this oracle only by self-consistency properties (generalised-eigen
residual, RHS-orthonormality, invariances), not by mvlearn output.
"""
import numpy as np
import scipy.linalg

from .align_oracle import extract_group_conditions


def n_components_var(X, var):
    """alignment/AlignMCCA.py:156-174.  NB ``argmax`` of the boolean cumsum is a
    0-based index, i.e. one LESS than the number of components needed — kept."""
    s = np.linalg.svd(np.asarray(X, dtype=np.float64), compute_uv=False) ** 2
    s = s / s.sum()
    return int(np.argmax(np.cumsum(s) > var))


def mcca_gevp_blocks(views, regs):
    """Block matrices of the generalised eigenproblem (see module header)."""
    P = len(views)
    rhs = []
    for X in views:
        G = X.T @ X
        rhs.append(G if regs is None else (1.0 - regs) * G + regs * np.eye(G.shape[0]))
    lhs = [[None] * P for _ in range(P)]
    for a in range(P):
        for b in range(P):
            lhs[a][b] = rhs[a] if a == b else views[a].T @ views[b]
    return np.block(lhs), scipy.linalg.block_diag(*rhs)


def mcca_gevp(views, n_components, regs):
    """Top-``n_components`` generalised eigenpairs, split per view, sign-fixed."""
    dims = [v.shape[1] for v in views]
    LHS, RHS = mcca_gevp_blocks(views, regs)
    D = LHS.shape[0]
    k = min(n_components, D)
    w, V = scipy.linalg.eigh(LHS, RHS, subset_by_index=[D - k, D - 1])
    order = np.argsort(-w)
    w, V = w[order], V[:, order]
    offs = np.concatenate([[0], np.cumsum(dims)])
    loadings = [V[offs[i]:offs[i + 1]] for i in range(len(views))]
    common = sum(X @ L for X, L in zip(views, loadings))
    common = common / np.linalg.norm(common, axis=0)
    rows = np.argmax(np.abs(common), axis=0)
    signs = np.sign(common[rows, np.arange(common.shape[1])])
    signs[signs == 0] = 1
    return [L * signs for L in loadings], w


class MCCAOracle:
    """Stand-in for the object the reference keeps in ``AlignMCCA.mcca``:
    ``loadings_`` (list), ``means_`` and ``transform_view(X2d, i)``."""

    def __init__(self, n_components=10, regs=0.5, signal_ranks=None):
        self.n_components, self.regs, self.signal_ranks = n_components, regs, signal_ranks

    def fit(self, views):
        views = [np.asarray(v, dtype=np.float64) for v in views]
        self.means_ = [v.mean(axis=0) for v in views]
        cent = [v - m for v, m in zip(views, self.means_)]
        if self.signal_ranks is None:
            self.loadings_, self.evals_ = mcca_gevp(cent, self.n_components, self.regs)
        else:
            bases, scores = [], []
            for X, r in zip(cent, self.signal_ranks):
                _, _, Vt = np.linalg.svd(X, full_matrices=False)
                Vr = Vt[:max(int(r), 1)].T
                bases.append(Vr)
                scores.append(X @ Vr)
            red, self.evals_ = mcca_gevp(scores, self.n_components, self.regs)
            self.loadings_ = [B @ L for B, L in zip(bases, red)]
        return self

    def transform_view(self, X, view):
        return (np.asarray(X, dtype=np.float64) - self.means_[view]) @ self.loadings_[view]


def get_mcca_transforms(features, labels, n_components=10, regs=0.5, pca_var=1):
    """alignment/AlignMCCA.py:140-154."""
    avgs = [d.reshape(-1, d.shape[-1]) for d in extract_group_conditions(features, labels)]
    ranks = None
    if 0 < pca_var < 1:
        ranks = [min(n_components, n_components_var(np.asarray(x).reshape(-1, x.shape[-1]), pca_var))
                 for x in features]
    return MCCAOracle(n_components, regs, ranks).fit(avgs)


def mcca_transform(mcca, X, idx):
    """alignment/AlignMCCA.py:110-111,125-126."""
    X = np.asarray(X)
    out = mcca.transform_view(X.reshape(-1, X.shape[-1]), idx)
    return out.reshape(X.shape[:-1] + (-1,))
