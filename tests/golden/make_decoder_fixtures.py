"""tests/golden/decoders_cfg1.npz: BASELINE config 1 on the REFERENCE — crossPtDecoder_sepAlign(AlignCCA) and
crossPtDecoder_jointDimRed(JointPCA) around a bagged linear SVM (scripts/aligned_decode_svm.py:262-263) with the
random_state injected by this harness; pooled features and test predictions are stored.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, '/root/reference/aligned_decoding')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from sklearn.ensemble import BaggingClassifier          # noqa: E402
from sklearn.svm import SVC                             # noqa: E402
from alignment.AlignCCA import AlignCCA                 # noqa: E402
from alignment.JointPCA import JointPCA                 # noqa: E402
from decoders.cross_pt_decoders import crossPtDecoder_jointDimRed, crossPtDecoder_sepAlign, crossPtDecoder_sepDimRed   # noqa: E402
from cross_patient_speech_decoding_amd.utils.synthetic import make_patient   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def svm():
    return BaggingClassifier(SVC(kernel='linear'), n_estimators=10, random_state=0)


if __name__ == '__main__':
    pats = [make_patient(p, 72 - 6 * p, T=14, C=12 + 2 * p, n_cond=9, noise=2.0) for p in range(3)]
    pats = [(x.astype(np.float64), y) for x, y in pats]
    Xt, yt = pats[0]
    y1 = yt[:, 0]
    cross = [(x, y[:, 0], y) for x, y in pats[1:]]
    tr, te = np.arange(0, 48), np.arange(48, 72)
    out = dict(train_idx=tr, test_idx=te)
    for name, dec in (('sepAlign', crossPtDecoder_sepAlign(cross, svm(), AlignCCA, n_comp=0.9)),
                      ('sepDimRed', crossPtDecoder_sepDimRed(cross, svm(), n_comp=0.9)),
                      ('jointDimRed', crossPtDecoder_jointDimRed(cross, svm(), JointPCA, n_comp=6))):
        if name == 'sepDimRed':
            X_p, y_p = dec.preprocess_train(Xt[tr], y1[tr])
            dec.decoder.fit(X_p, y_p)
        else:
            X_p, y_p = dec.preprocess_train(Xt[tr], y1[tr], y_align=yt[tr])
            dec.decoder.fit(X_p, y_p)
        out[f'{name}_Xpool'] = X_p
        out[f'{name}_ypool'] = y_p
        out[f'{name}_Xtest'] = dec.preprocess_test(Xt[te])
        out[f'{name}_pred'] = dec.decoder.predict(out[f'{name}_Xtest'])
        out[f'{name}_acc'] = np.mean(out[f'{name}_pred'] == y1[te])
        print(name, X_p.shape, out[f'{name}_acc'])
    # the nested-CV / sub-sampling scripts' decoder (scripts/aligned_decode_svm_ncv.py:313-321): an RBF SVC with balanced class
    # weights behind DimRedReshape, inside crossPtDecoder_sepAlign; deterministic (no bagging): decision values are stored too
    from sklearn.decomposition import PCA                   # noqa: E402
    from sklearn.pipeline import make_pipeline              # noqa: E402
    from decomposition.DimRedReshape import DimRedReshape   # noqa: E402
    clf = make_pipeline(DimRedReshape(PCA), SVC(kernel='rbf', class_weight='balanced'))
    dec = crossPtDecoder_sepAlign(cross, clf, AlignCCA, n_comp=0.9)
    X_p, y_p = dec.preprocess_train(Xt[tr], y1[tr], y_align=yt[tr])
    dec.decoder.fit(X_p, y_p)
    X_te = dec.preprocess_test(Xt[te])
    out['rbf_Xpool'], out['rbf_ypool'], out['rbf_Xtest'] = X_p, y_p, X_te
    out['rbf_pred'] = dec.decoder.predict(X_te)
    out['rbf_dec_ovr'] = dec.decoder.decision_function(X_te)
    svc = dec.decoder[-1]
    out['rbf_gamma'] = np.float64(svc._gamma)
    out['rbf_class_weight'] = svc.class_weight_
    out['rbf_acc'] = np.mean(out['rbf_pred'] == y1[te])
    print('rbf', X_p.shape, out['rbf_acc'], float(svc._gamma), svc.class_weight_)
    np.savez_compressed(os.path.join(HERE, 'decoders_cfg1.npz'), **out)
    print(os.path.getsize(os.path.join(HERE, 'decoders_cfg1.npz')))
