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


def svm():
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


@pytest.mark.parametrize('name', ['sepAlign', 'sepDimRed', 'jointDimRed'])
def test_config1_decoders_match_reference(golden_dir, name):
    import cross_patient_speech_decoding_amd.alignment as A
    from cross_patient_speech_decoding_amd.decoders import (crossPtDecoder_jointDimRed, crossPtDecoder_sepAlign,
                                                            crossPtDecoder_sepDimRed)
    g = np.load(os.path.join(golden_dir, 'decoders_cfg1.npz'))
    Xt, yt, cross = data()
    y1 = yt[:, 0]
    tr, te = g['train_idx'], g['test_idx']
    if name == 'sepAlign':
        dec = crossPtDecoder_sepAlign(cross, svm(), A.AlignCCA, n_comp=0.9)
    elif name == 'sepDimRed':
        dec = crossPtDecoder_sepDimRed(cross, svm(), n_comp=0.9)
    else:
        dec = crossPtDecoder_jointDimRed(cross, svm(), A.JointPCA, n_comp=6)
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
