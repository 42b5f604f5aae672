"""BASELINE config 1 (cross-patient SVM decode): the decoder wrappers with the MI355X reduction / alignment
against the reference's own crossPtDecoder_* outputs (tests/golden/decoders_cfg1.npz): pooled training features,
test features and the bagged linear-SVM predictions."""
import os

import numpy as np
import pytest
from sklearn.base import clone
from sklearn.ensemble import BaggingClassifier
from sklearn.svm import SVC

pytestmark = pytest.mark.gpu


def svm(device=False):
    """The reference's decoder (scripts/aligned_decode_svm.py:262-263): ten bagged linear SVMs.  device=True: the same bagging around
    the HIP SVC (decoders/svm.py) instead of sklearn's libsvm wrapper -- sklearn draws the same bootstrap samples either way."""
    if device:
        from cross_patient_speech_decoding_amd.decoders import SVC as DeviceSVC
        return BaggingClassifier(DeviceSVC(kernel='linear'), n_estimators=10, random_state=0)
    return BaggingClassifier(SVC(kernel='linear'), n_estimators=10, random_state=0)


def data():
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    pats = [make_patient(p, 72 - 6 * p, T=14, C=12 + 2 * p, n_cond=9, noise=2.0) for p in range(3)]
    pats = [(x.astype(np.float64), y) for x, y in pats]
    Xt, yt = pats[0]
    return Xt, yt, [(x, y[:, 0], y) for x, y in pats[1:]]


def _close_up_to_sign_per_latent(X, ref, T, tol):
    """Pooled features are (trials, T*d) flattened from (trials, T, d); PCA components are sign-fixed identically
    on both sides, so a plain comparison is expected to hold; tolerance relative to the feature scale."""
    assert X.shape == ref.shape
    assert np.abs(X - ref).max() <= tol * np.abs(ref).max()


@pytest.mark.parametrize('device_svm', [False, True], ids=['sklearn-svm', 'hip-svm'])
@pytest.mark.parametrize('name', ['sepAlign', 'sepDimRed', 'jointDimRed'])
def test_config1_decoders_match_reference(golden_dir, name, device_svm):
    import cross_patient_speech_decoding_amd.alignment as A
    from cross_patient_speech_decoding_amd.decoders import (crossPtDecoder_jointDimRed, crossPtDecoder_sepAlign,
                                                            crossPtDecoder_sepDimRed)
    g = np.load(os.path.join(golden_dir, 'decoders_cfg1.npz'))
    Xt, yt, cross = data()
    y1 = yt[:, 0]
    tr, te = g['train_idx'], g['test_idx']
    if name == 'sepAlign':
        dec = crossPtDecoder_sepAlign(cross, svm(device_svm), A.AlignCCA, n_comp=0.9)
    elif name == 'sepDimRed':
        dec = crossPtDecoder_sepDimRed(cross, svm(device_svm), n_comp=0.9)
    else:
        dec = crossPtDecoder_jointDimRed(cross, svm(device_svm), A.JointPCA, n_comp=6)
    clone(dec)                                                    # constructor args stored verbatim
    kw = {} if name == 'sepDimRed' else {'y_align': yt[tr]}
    dec.fit(Xt[tr], y1[tr], **kw)
    X_p, y_p = dec.preprocess_train(Xt[tr], y1[tr], **kw)
    np.testing.assert_array_equal(y_p, g[f'{name}_ypool'])
    _close_up_to_sign_per_latent(X_p, g[f'{name}_Xpool'], 20, 1e-6)
    X_te = dec.preprocess_test(Xt[te])
    _close_up_to_sign_per_latent(X_te, g[f'{name}_Xtest'], 20, 1e-6)
    pred = dec.predict(Xt[te])
    assert np.mean(pred == g[f'{name}_pred']) >= 0.97            # identical up to SVM ties on near-equal features
    assert abs(dec.score(Xt[te], y1[te]) - float(g[f'{name}_acc'])) <= 0.04


def test_mcca_decoder_and_nocenter_pca():
    import cross_patient_speech_decoding_amd.alignment as A
    from cross_patient_speech_decoding_amd.decoders import crossPtDecoder_mcca
    from cross_patient_speech_decoding_amd.decomposition import DimRedReshape, NoCenterPCA
    Xt, yt, cross = data()
    dec = crossPtDecoder_mcca(cross, svm(), A.AlignMCCA, n_comp=5, regs=0.5)
    dec.fit(Xt[:48], yt[:48, 0], y_align=yt[:48])
    assert dec.predict(Xt[48:]).shape == (24,)
    assert isinstance(dec.aligner, A.AlignMCCA)                  # the reference replaces the class by the instance too
    # NoCenterPCA vs the SVD definition (reference NoCenterPCA.py:41-58)
    X = Xt.reshape(-1, Xt.shape[-1])
    p = NoCenterPCA(n_components=0.9).fit(X)
    _, S, Vt = np.linalg.svd(X, full_matrices=False)
    k = int(np.argmax(np.cumsum(S ** 2) / np.sum(S ** 2) >= 0.9) + 1)
    assert p.components_.shape == (X.shape[1], k)
    np.testing.assert_allclose(p.explained_variance_, S ** 2, rtol=1e-9)
    np.testing.assert_allclose(np.abs(p.components_.T @ Vt[:k].T), np.eye(k), atol=1e-8)
    np.testing.assert_allclose(np.abs(p.transform(X)), np.abs(X @ Vt[:k].T), atol=1e-8 * np.abs(X).max() * 10)
    with pytest.raises(ValueError, match='PCA must be fit'):
        NoCenterPCA(3).transform(X)
    z = DimRedReshape(A.PCA, n_components=4).fit_transform(Xt)
    assert z.shape == (Xt.shape[0], 4)


@pytest.mark.parametrize('n,d,k,C', [(120, 10, 2, 1.0), (300, 40, 5, 1.0), (500, 140, 9, 0.3), (64, 6, 3, 10.0)])
def test_hip_svc_equals_libsvm(n, d, k, C):
    """The HIP SVC against sklearn's SVC(kernel='linear') (libsvm, what the reference calls): same one-vs-one decision values to the
    solvers' tolerance (both stop at a maximal KKT violation of tol = 1e-3), identical predictions wherever the vote is not decided
    by a decision value inside that tolerance, same accuracy; overlapping classes (bounded and free support vectors), imbalanced
    class sizes, labels that are not 0..k-1."""
    from cross_patient_speech_decoding_amd.decoders import SVC as DeviceSVC
    rng = np.random.default_rng(n + d)
    centers = rng.standard_normal((k, d)) * 1.2
    sizes = rng.multinomial(n - 4 * k, np.ones(k) / k) + 4
    X = np.vstack([centers[c] + rng.standard_normal((m, d)) for c, m in enumerate(sizes)])
    y = np.repeat(np.arange(k) * 3 + 1, sizes)
    perm = rng.permutation(len(y))
    X, y = X[perm], y[perm]
    Xte = np.vstack([centers[c] + rng.standard_normal((20, d)) for c in range(k)])
    ref = SVC(kernel='linear', C=C, decision_function_shape='ovo').fit(X, y)
    dev = DeviceSVC(kernel='linear', C=C, decision_function_shape='ovo').fit(X, y)
    np.testing.assert_array_equal(dev.classes_, ref.classes_)
    d_ref = ref.decision_function(Xte)
    d_dev = dev.decision_function(Xte)
    scale = max(1.0, float(np.abs(d_ref).max()))
    assert d_dev.shape == d_ref.shape
    assert np.abs(d_dev - d_ref).max() <= 2e-2 * scale, np.abs(d_dev - d_ref).max()       # two tol = 1e-3 solutions of one problem
    p_ref, p_dev = ref.predict(Xte), dev.predict(Xte)
    sure = np.abs(d_ref if d_ref.ndim == 2 else d_ref[:, None]).min(axis=1) > 5e-2 * scale
    np.testing.assert_array_equal(p_dev[sure], p_ref[sure])
    assert np.mean(p_dev == p_ref) >= 0.97
    assert abs(dev.score(Xte, np.repeat(np.arange(k) * 3 + 1, 20)) - ref.score(Xte, np.repeat(np.arange(k) * 3 + 1, 20))) <= 0.03
    # bagging / cloning / parameter plumbing as the reference's scripts use it
    bag = BaggingClassifier(clone(DeviceSVC(kernel='linear', C=C)), n_estimators=3, random_state=1).fit(X, y)
    assert bag.predict(Xte).shape == (20 * k,)
    assert DeviceSVC().set_params(C=2.0).get_params()['C'] == 2.0
    with pytest.raises(NotImplementedError):
        DeviceSVC(kernel='poly').fit(X, y)
    if k == 2:                                     # sklearn flips coef_ / intercept_ of a binary problem: so does the HIP SVC
        np.testing.assert_allclose(dev.coef_, ref.coef_, atol=2e-2 * max(1.0, np.abs(ref.coef_).max()))
        np.testing.assert_allclose(dev.intercept_, ref.intercept_, atol=2e-2 * scale)


@pytest.mark.parametrize('n,d,k,C,gamma,cw', [(160, 12, 2, 1.0, 'scale', 'balanced'), (300, 40, 5, 1.0, 'scale', 'balanced'),
                                             (420, 90, 9, 3.0, 'auto', None), (90, 6, 3, 0.5, 0.2, {1: 2.0, 4: 0.5})])
def test_hip_svc_rbf_and_class_weights_equal_libsvm(n, d, k, C, gamma, cw):
    """SVC(kernel='rbf', class_weight='balanced') -- the decoder of scripts/aligned_decode_svm_ncv.py:313-317 and of the four
    *_subsample.py scripts (:249-261) -- against sklearn's libsvm: gamma ('scale' / 'auto' / a number), the class weights, the
    one-vs-one decision values to the solvers' tolerance, sklearn's 'ovr' scores (the DEFAULT decision_function_shape: votes +
    squashed confidences, now computed instead of silently returning the one-vs-one matrix), predictions and accuracy; imbalanced
    classes, labels that are not 0..k-1, bootstrap multiplicities as sample weights."""
    from cross_patient_speech_decoding_amd.decoders import SVC as DeviceSVC
    rng = np.random.default_rng(n * 7 + d)
    centers = rng.standard_normal((k, d)) * 1.1
    sizes = rng.multinomial(n - 5 * k, np.linspace(1, 3, k) / np.linspace(1, 3, k).sum()) + 5         # imbalanced
    X = np.vstack([centers[c] + rng.standard_normal((m, d)) for c, m in enumerate(sizes)])
    labels = np.arange(k) * 3 + 1
    y = np.repeat(labels, sizes)
    perm = rng.permutation(len(y))
    X, y = X[perm], y[perm]
    Xte = np.vstack([centers[c] + rng.standard_normal((20, d)) for c in range(k)])
    yte = np.repeat(labels, 20)
    sw = rng.integers(0, 3, len(y)).astype(np.float64)                        # (zeros: dropped before training)
    for weights in (None, sw):
        ref_o = SVC(kernel='rbf', C=C, gamma=gamma, class_weight=cw, decision_function_shape='ovo').fit(X, y, sample_weight=weights)
        ref_r = SVC(kernel='rbf', C=C, gamma=gamma, class_weight=cw).fit(X, y, sample_weight=weights)
        dev_o = DeviceSVC(kernel='rbf', C=C, gamma=gamma, class_weight=cw, decision_function_shape='ovo').fit(X, y, sample_weight=weights)
        dev_r = DeviceSVC(kernel='rbf', C=C, gamma=gamma, class_weight=cw).fit(X, y, sample_weight=weights)
        np.testing.assert_array_equal(dev_o.classes_, ref_o.classes_)
        np.testing.assert_allclose(dev_o._gamma, ref_o._gamma, rtol=1e-12)
        np.testing.assert_allclose(dev_o.class_weight_, ref_o.class_weight_, rtol=1e-12)
        d_ref, d_dev = ref_o.decision_function(Xte), dev_o.decision_function(Xte)
        scale = max(1.0, float(np.abs(d_ref).max()))
        assert d_dev.shape == d_ref.shape
        assert np.abs(d_dev - d_ref).max() <= 2e-2 * scale, np.abs(d_dev - d_ref).max()
        r_ref, r_dev = ref_r.decision_function(Xte), dev_r.decision_function(Xte)
        assert r_dev.shape == r_ref.shape == ((len(yte),) if k == 2 else (len(yte), k))
        sure = np.abs(d_ref if d_ref.ndim == 2 else d_ref[:, None]).min(axis=1) > 5e-2 * scale
        if sure.any():
            assert np.abs(r_dev - r_ref)[sure].max() <= 2e-2 * scale             # (a vote flips only where a pair value is ~0)
        if k > 2:                                                                # the 'ovr' rule itself, on identical pair values
            from cross_patient_speech_decoding_amd.decoders.svm import _ovr_from_ovo
            np.testing.assert_allclose(_ovr_from_ovo(d_ref, k), r_ref, rtol=0, atol=1e-12)
        p_ref, p_dev = ref_r.predict(Xte), dev_r.predict(Xte)
        np.testing.assert_array_equal(p_dev[sure], p_ref[sure])
        assert np.mean(p_dev == p_ref) >= 0.97
        assert abs(dev_r.score(Xte, yte) - ref_r.score(Xte, yte)) <= 0.03
    assert clone(DeviceSVC(kernel='rbf', class_weight='balanced')).get_params()['class_weight'] == 'balanced'


def test_rbf_decoder_pipeline_matches_reference_golden(golden_dir):
    """The reference's own nested-CV decoder (scripts/aligned_decode_svm_ncv.py:313-321: make_pipeline(DimRedReshape(dim_red),
    SVC(kernel='rbf', class_weight='balanced')) inside crossPtDecoder_sepAlign) run by the reference package on the config-1
    synthetic patients (tests/golden/make_decoder_fixtures.py): the same pipeline with the device aligner, device PCA and the HIP
    SVC reproduces its gamma, class weights, 'ovr' decision values and predictions."""
    import cross_patient_speech_decoding_amd.alignment as A
    from sklearn.pipeline import make_pipeline
    from cross_patient_speech_decoding_amd.decoders import SVC as DeviceSVC
    from cross_patient_speech_decoding_amd.decoders import crossPtDecoder_sepAlign
    from cross_patient_speech_decoding_amd.decomposition import DimRedReshape
    g = np.load(os.path.join(golden_dir, 'decoders_cfg1.npz'))
    Xt, yt, cross = data()
    y1 = yt[:, 0]
    tr, te = g['train_idx'], g['test_idx']
    clf = make_pipeline(DimRedReshape(A.PCA), DeviceSVC(kernel='rbf', class_weight='balanced'))
    dec = crossPtDecoder_sepAlign(cross, clf, A.AlignCCA, n_comp=0.9)
    dec.fit(Xt[tr], y1[tr], y_align=yt[tr])
    svc = dec.decoder[-1]
    np.testing.assert_allclose(svc.class_weight_, g['rbf_class_weight'], rtol=1e-12)
    np.testing.assert_allclose(svc._gamma, float(g['rbf_gamma']), rtol=1e-6)          # (1 / (d * var) of the PCA scores: rotation-invariant)
    X_te = dec.preprocess_test(Xt[te])
    _close_up_to_sign_per_latent(X_te, g['rbf_Xtest'], 20, 1e-6)
    ovr = dec.decoder.decision_function(X_te)
    assert ovr.shape == g['rbf_dec_ovr'].shape
    assert np.abs(ovr - g['rbf_dec_ovr']).max() <= 2e-2
    np.testing.assert_array_equal(dec.predict(Xt[te]), g['rbf_pred'])
    assert abs(dec.score(Xt[te], y1[te]) - float(g['rbf_acc'])) <= 1e-12
    # the decision values of the SVC alone on the golden pooled features (no aligner / PCA differences in between)
    alone = DeviceSVC(kernel='rbf', class_weight='balanced').fit(g['rbf_Xpool'], g['rbf_ypool'])
    assert np.abs(alone.decision_function(g['rbf_Xtest']) - g['rbf_dec_ovr']).max() <= 2e-2


def test_smo_launches_at_the_lds_resident_limit():
    """xps_svm_smo_f64_max_points() is derived from the same formula as the launch's LDS request (5813 / 5814 points
    passed the argument check and then failed AT LAUNCH): a binary problem of exactly that many points launches and returns, one
    more is refused by the argument check."""
    import ctypes as C
    import torch
    from cross_patient_speech_decoding_amd._lib import XpsError, call, lib
    from cross_patient_speech_decoding_amd.alignment import _linalg as LA
    n = int(lib().xps_svm_smo_f64_max_points())
    assert 5000 < n < 5814 and n * 28 + 64 + 128 <= 160 * 1024 - 1024
    dev = LA.device()
    g = torch.Generator().manual_seed(0)
    X = torch.randn(n, 8, generator=g, dtype=torch.float64)
    X[: n // 2, 0] += 6.0                                    # separable: a handful of iterations
    Xd = X.to(dev)
    K = LA.dgemm(Xd, Xd, tb=True)
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    off = torch.tensor([0, n], dtype=torch.int32, device=dev)
    npos = torch.tensor([n // 2], dtype=torch.int32, device=dev)
    cb = torch.ones(n, dtype=torch.float64, device=dev)
    alpha = torch.empty(n, dtype=torch.float64, device=dev)
    rho = torch.empty(1, dtype=torch.float64, device=dev)
    iters = torch.empty(1, dtype=torch.int32, device=dev)
    args = (K.data_ptr(), K.stride(0), idx.data_ptr(), off.data_ptr(), npos.data_ptr(), 1)
    tail = (cb.data_ptr(), 1e-3, 200, alpha.data_ptr(), rho.data_ptr(), iters.data_ptr(), LA._stream())
    call('xps_svm_smo_f64', *args, n, *tail)
    torch.cuda.synchronize()
    assert 0 < int(iters[0]) <= 200 and bool(torch.isfinite(alpha).all()) and float(alpha.sum()) > 0
    with pytest.raises(XpsError):
        call('xps_svm_smo_f64', *args, n + 1, *tail)
