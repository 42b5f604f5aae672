"""Cross-patient decoder wrappers (BASELINE config 1) — counterpart of the reference's
decoders/cross_pt_decoders.py (crossPtDecoder :11, _sepDimRed :89, _sepAlign :183, _jointDimRed :288,
_mcca :367): sklearn ``BaseEstimator``s that reduce / align / pool several patients' trials and hand the
flattened (T * d) features to any sklearn decoder (SVC, bagged linear SVM ...).

Constructor arguments are stored verbatim (``clone`` / ``set_params`` / BayesSearchCV keep working); the
dimensionality reduction and alignment steps run on the MI355X through this package's ``PCA`` /
``AlignCCA`` / ``AlignMCCA`` / ``JointPCA``; the decoder itself is whatever sklearn estimator the caller
passes (CPU, as in the reference)."""
import numpy as np
from sklearn.base import BaseEstimator

from ..alignment.pca import PCA


def _flat(x):
    return x.reshape(x.shape[0], -1)


class crossPtDecoder(BaseEstimator):
    """fit / predict / score around ``preprocess_train`` / ``preprocess_test`` of the subclasses."""

    def preprocess_train(self, X, y=None):
        pass

    def preprocess_test(self, X, y=None):
        pass

    def fit(self, X, y, **kwargs):
        X_p, y_p = self.preprocess_train(X, y, **kwargs)
        return self.decoder.fit(X_p, y_p)

    def predict(self, X):
        return self.decoder.predict(self.preprocess_test(X))

    def score(self, X, y, **kwargs):
        return self.decoder.score(self.preprocess_test(X), y, **kwargs)

    # shared pooling rule of every subclass
    def _pool(self, X_tar, X_cross, y):
        ys = [yc for _, yc, _ in self.cross_pt_data]
        if self.tar_in_train:
            return np.vstack([X_tar] + X_cross), np.hstack([y] + ys)
        return np.vstack(X_cross), np.hstack(ys)

    def _reduce_each(self, X):
        """Independent reduction of the target and of every pooled patient over (trial*time, channel) rows.
        Returns the target (3-D), the pooled patients (3-D) and keeps the target model for the test set."""
        out = []
        for x, _, _ in self.cross_pt_data:
            z = self.dim_red(n_components=self.n_comp).fit_transform(x.reshape(-1, x.shape[-1]))
            out.append(z.reshape(x.shape[0], -1, z.shape[-1]))
        self.tar_dr = self.dim_red(n_components=self.n_comp)
        z = self.tar_dr.fit_transform(X.reshape(-1, X.shape[-1]))
        return z.reshape(X.shape[0], -1, z.shape[-1]), out


class crossPtDecoder_sepDimRed(crossPtDecoder):
    """Separate reductions truncated to the smallest latent size, pooled without alignment."""

    def __init__(self, cross_pt_data, decoder, dim_red=PCA, n_comp=0.8, tar_in_train=True):
        self.cross_pt_data = cross_pt_data
        self.decoder = decoder
        self.dim_red = dim_red
        self.n_comp = n_comp
        self.tar_in_train = tar_in_train

    def preprocess_train(self, X, y, **kwargs):
        X_tar, X_cross = self._reduce_each(X)
        self.common_dim = min([X_tar.shape[-1]] + [x.shape[-1] for x in X_cross])
        d = self.common_dim
        return self._pool(_flat(X_tar[..., :d]), [_flat(x[..., :d]) for x in X_cross], y)

    def preprocess_test(self, X):
        z = self.tar_dr.transform(X.reshape(-1, X.shape[-1]))[:, :self.common_dim]
        return z.reshape(X.shape[0], -1)


class crossPtDecoder_sepAlign(crossPtDecoder):
    """Separate reductions, every pooled patient aligned to the target with ``aligner()`` (pairwise API)."""

    def __init__(self, cross_pt_data, decoder, aligner, dim_red=PCA, n_comp=0.8, tar_in_train=True):
        self.cross_pt_data = cross_pt_data
        self.decoder = decoder
        self.dim_red = dim_red
        self.n_comp = n_comp
        self.aligner = aligner
        self.tar_in_train = tar_in_train

    def preprocess_train(self, X, y, y_align=None):
        X_tar, X_cross = self._reduce_each(X)
        if y_align is None:
            y_align = y
        self.algns = [self.aligner() for _ in self.cross_pt_data]
        aligned = []
        for algn, x, (_, _, ya) in zip(self.algns, X_cross, self.cross_pt_data):
            algn.fit(X_tar, x, y_align, ya)
            aligned.append(_flat(algn.transform(x)))
        return self._pool(_flat(X_tar), aligned, y)

    def preprocess_test(self, X):
        return self.tar_dr.transform(X.reshape(-1, X.shape[-1])).reshape(X.shape[0], -1)


class crossPtDecoder_jointDimRed(crossPtDecoder):
    """One joint reduction (e.g. JointPCA) over all patients' condition averages."""

    def __init__(self, cross_pt_data, decoder, joint_dr_method, n_comp=0.8, tar_in_train=True):
        self.cross_pt_data = cross_pt_data
        self.decoder = decoder
        self.joint_dr_method = joint_dr_method
        self.n_comp = n_comp
        self.tar_in_train = tar_in_train

    def preprocess_train(self, X, y, y_align=None):
        if y_align is None:
            y_align = y
        self.joint_dr = self.joint_dr_method(n_components=self.n_comp)
        out = self.joint_dr.fit_transform([X] + [x for x, _, _ in self.cross_pt_data],
                                          [y_align] + [ya for _, _, ya in self.cross_pt_data])
        return self._pool(_flat(out[0]), [_flat(x) for x in out[1:]], y)

    def preprocess_test(self, X):
        return _flat(self.joint_dr.transform(X, idx=0))


class crossPtDecoder_mcca(crossPtDecoder):
    """Multiview CCA over all patients.  As in the reference (:416-417) ``self.aligner`` is replaced by the
    fitted instance in ``preprocess_train``."""

    def __init__(self, cross_pt_data, decoder, aligner, n_comp=10, regs=0.5, pca_var=1, tar_in_train=True):
        self.cross_pt_data = cross_pt_data
        self.decoder = decoder
        self.aligner = aligner
        self.n_comp = n_comp
        self.regs = regs
        self.pca_var = pca_var
        self.tar_in_train = tar_in_train

    def preprocess_train(self, X, y, y_align=None):
        if y_align is None:
            y_align = y
        self.aligner = self.aligner(n_components=self.n_comp, regs=self.regs, pca_var=self.pca_var)
        out = self.aligner.fit_transform([X] + [x for x, _, _ in self.cross_pt_data],
                                         [y_align] + [ya for _, _, ya in self.cross_pt_data])
        return self._pool(_flat(out[0]), [_flat(x) for x in out[1:]], y)

    def preprocess_test(self, X):
        return _flat(self.aligner.transform(X, idx=0))
