"""Pin oracle/align_oracle.py to the golden vectors produced by the reference
(tests/golden/make_align_fixtures.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import align_oracle as ao


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_label_keys_and_cnd_avg(golden_dir):
    g = _load(golden_dir, 'align_labels.npz')
    assert list(ao.label_keys(g['y1'])) == list(g['s1'])
    assert list(ao.label_keys(g['y3'])) == list(g['s3'])
    a1 = ao.cnd_avg(g['X1'], ao.label_keys(g['y1']))
    a3 = ao.cnd_avg(g['X3'], ao.label_keys(g['y3']))
    assert a1.dtype == np.float64 and a3.dtype == np.float64
    np.testing.assert_array_equal(a1, g['avg1'])          # same arithmetic -> bit-equal
    np.testing.assert_array_equal(a3, g['avg3'])          # float32 mean then float64 store
    # string order, not numeric: '10' < '11' < '2' < '21' < '3'
    uniq, _ = ao.condition_index(ao.label_keys(g['y1']))
    assert list(uniq) == ['1', '10', '11', '2', '21', '3']


def test_extract_group_conditions(golden_dir):
    g = _load(golden_dir, 'align_labels.npz')
    out = ao.extract_group_conditions([g['gXa'], g['gXb'], g['gXc']], [g['gya'], g['gyb'], g['gyc']])
    for i in range(3):
        np.testing.assert_array_equal(out[i], g[f'g{i}'])
    assert out[0].shape[0] == out[1].shape[0] == out[2].shape[0] == 3   # 5 conds, one dropped in b, one absent in c


@pytest.mark.parametrize('case', ['full', 'uneq', 'rdef'])
def test_cca_against_reference(golden_dir, case):
    g = _load(golden_dir, 'align_cca.npz')
    Xa, ya, Xb, yb = (g[f'{case}_{k}'] for k in ('Xa', 'ya', 'Xb', 'yb'))
    al = ao.AlignCCAOracle('shared').fit(Xa, Xb, ya, yb)
    np.testing.assert_allclose(al.canon_corrs, g[f'{case}_S'], rtol=0, atol=1e-12)
    assert al.M_a.shape == g[f'{case}_M_a'].shape and al.M_b.shape == g[f'{case}_M_b'].shape
    np.testing.assert_allclose(al.M_a, g[f'{case}_M_a'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(al.M_b, g[f'{case}_M_b'], rtol=1e-9, atol=1e-12)
    ta, tb = al.transform([Xa, Xb])
    np.testing.assert_allclose(ta, g[f'{case}_shared_ta'], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(tb, g[f'{case}_shared_tb'], rtol=1e-9, atol=1e-11)
    al.return_space = 'b_to_a'
    np.testing.assert_allclose(al.transform(Xb), g[f'{case}_b_to_a_t'], rtol=1e-9, atol=1e-10)
    al.return_space = 'a_to_b'
    np.testing.assert_allclose(al.transform(Xa), g[f'{case}_a_to_b_t'], rtol=1e-9, atol=1e-10)
    if case == 'rdef':
        assert al.M_b.shape == (7, 6)          # rank truncation d = min(rank_a, rank_b) = 6


def test_cca_raw_and_properties(golden_dir):
    g = _load(golden_dir, 'align_cca.npz')
    La, Lb = g['raw_La'].copy(), g['raw_Lb'].copy()
    Ma, Mb, S = ao.cca_align(La, Lb)
    np.testing.assert_array_equal(La, g['raw_La'])          # oracle does not mutate
    np.testing.assert_allclose(S, g['raw_S'], atol=1e-12)
    np.testing.assert_allclose(Ma, g['raw_Ma'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(Mb, g['raw_Mb'], rtol=1e-9, atol=1e-12)
    # property: canonical variates are orthonormal and correlate with canon_corrs
    A = (La - La.mean(1, keepdims=True)).T @ Ma
    B = (Lb - Lb.mean(1, keepdims=True)).T @ Mb
    np.testing.assert_allclose(A.T @ A, np.eye(A.shape[1]), atol=1e-10)
    np.testing.assert_allclose(np.diag(A.T @ B), S, atol=1e-10)


def test_not_fitted_raises():
    with pytest.raises(RuntimeError, match='Must call fit'):
        ao.AlignCCAOracle().transform(np.zeros((2, 3, 4)))


def test_joint_pca(golden_dir):
    g = _load(golden_dir, 'align_jointpca.npz')
    Xs, ys = [g[f'X{i}'] for i in range(3)], [g[f'y{i}'] for i in range(3)]
    for exact in (False, True):
        W = ao.joint_pca_transforms(Xs, ys, n_components=4, exact=exact)
        for i in range(3):
            np.testing.assert_allclose(W[i], g[f'W{i}'], rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(ao.joint_pca_apply(Xs[i], W[i]), g[f't{i}'], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(ao.joint_pca_apply(Xs[1], W[1]), g['t1_single'], rtol=1e-7, atol=1e-8)


def test_pca_exact_matches_sklearn():
    rng = np.random.default_rng(5)
    X = (rng.standard_normal((4000, 6)) @ rng.standard_normal((6, 12))
         + 0.3 * rng.standard_normal((4000, 12))).astype(np.float64)
    p, Z = ao.pca_fit(X, 0.95)
    mean, comps, var = ao.pca_exact(X, 0.95)
    assert comps.shape == p.components_.shape
    np.testing.assert_allclose(comps, p.components_, atol=1e-9)
    np.testing.assert_allclose(var, p.explained_variance_, rtol=1e-10)
    np.testing.assert_allclose((X - mean) @ comps.T, Z, atol=1e-9)


def test_process_aligner_composition():
    rng = np.random.default_rng(7)
    seqs = np.array([[a, b, 1] for a in (1, 2, 3) for b in (1, 2, 3)])
    Z = np.cumsum(rng.standard_normal((9, 12, 4)), axis=1)
    def view(n, C):
        c = np.concatenate([np.arange(9), rng.integers(0, 9, n - 9)])
        return (Z[c] @ rng.standard_normal((4, C)) + 0.3 * rng.standard_normal((n, 12, C))).astype(np.float32), seqs[c]
    Xt, yt = view(30, 10)
    pool = [view(28, 8) + (None,), view(33, 9) + (None,)]
    pool = [(x, y, y) for x, y, _ in pool]
    Xp, yp, tar = ao.process_aligner(Xt, yt, yt, pool)
    assert Xp.dtype == np.float32 and Xp.shape[0] == 30 + 28 + 33 and Xp.shape[1] == 12
    assert Xp.shape[2] == tar.n_components_
    assert yp.shape == (91, 3)
