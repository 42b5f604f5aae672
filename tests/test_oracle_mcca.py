"""oracle/mcca_oracle.py is PARITY-UNPINNED against mvlearn (absent); these tests pin
it by the self-consistency properties of the published formulation.  CPU only."""
import numpy as np
import scipy.linalg

from oracle import mcca_oracle as mo


def _views(seed=3, P=3, n=120):
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((n, 4))
    return [Z @ rng.standard_normal((4, d)) + 0.4 * rng.standard_normal((n, d)) for d in (7, 5, 6)[:P]]


def test_gevp_residual_and_orthonormality():
    views = [v - v.mean(0) for v in _views()]
    loadings, w = mo.mcca_gevp(views, 4, 0.5)
    LHS, RHS = mo.mcca_gevp_blocks(views, 0.5)
    V = np.vstack(loadings)
    np.testing.assert_allclose(LHS @ V, RHS @ V * w, atol=1e-8 * np.abs(LHS).max())
    np.testing.assert_allclose(V.T @ RHS @ V, np.eye(4), atol=1e-10)
    assert np.all(np.diff(w) <= 1e-12)                   # descending
    full = scipy.linalg.eigh(LHS, RHS, eigvals_only=True)
    np.testing.assert_allclose(w, full[::-1][:4], rtol=1e-10)


def test_sign_rule_and_transform_view():
    views = _views(seed=4)
    m = mo.MCCAOracle(3, 0.5).fit(views)
    common = sum(m.transform_view(v, i) for i, v in enumerate(views))
    common /= np.linalg.norm(common, axis=0)
    rows = np.argmax(np.abs(common), axis=0)
    assert np.all(common[rows, np.arange(3)] > 0)
    # transform_view subtracts the stored mean
    np.testing.assert_allclose(m.transform_view(views[1], 1).mean(0), 0, atol=1e-10)


def test_n_components_var_off_by_one():
    X = np.diag([3.0, 2.0, 1.0, 0.5]) @ np.eye(4)
    s2 = np.array([9, 4, 1, 0.25]); c = np.cumsum(s2 / s2.sum())
    k = mo.n_components_var(X, 0.9)
    assert k == int(np.argmax(c > 0.9))                  # 0-based index, one less than the count
    assert c[k] > 0.9 and (k == 0 or c[k - 1] <= 0.9)


def test_get_mcca_transforms_shapes_and_ranks():
    rng = np.random.default_rng(9)
    seqs = np.array([[a, b, 2] for a in (1, 2, 3) for b in (1, 2)])
    Z = np.cumsum(rng.standard_normal((6, 10, 3)), axis=1)
    feats, labs = [], []
    for n, C in ((25, 6), (30, 8), (22, 5)):
        c = np.concatenate([np.arange(6), rng.integers(0, 6, n - 6)])
        feats.append(Z[c] @ rng.standard_normal((3, C)) + 0.2 * rng.standard_normal((n, 10, C)))
        labs.append(seqs[c])
    m = mo.get_mcca_transforms(feats, labs, n_components=3, regs=0.5, pca_var=1)
    assert [L.shape for L in m.loadings_] == [(6, 3), (8, 3), (5, 3)]
    t = mo.mcca_transform(m, feats[1], 1)
    assert t.shape == (30, 10, 3)
    m2 = mo.get_mcca_transforms(feats, labs, n_components=3, regs=0.5, pca_var=0.8)
    assert [L.shape for L in m2.loadings_] == [(6, 3), (8, 3), (5, 3)] or all(L.shape[0] in (6, 8, 5) for L in m2.loadings_)
