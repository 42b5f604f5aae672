"""PCA on the MI355X with scikit-learn's conventions (the reference calls
``sklearn.decomposition.PCA`` at nn_models/data_utils/datamodules.py:542-548 and
alignment/JointPCA.py:199).

covariance (xps_xcov_f64, f64 MFMA) -> Jacobi eigendecomposition (xps_jacobi_*) -> sklearn's sign
rule (largest-|.| entry of each component positive) and component-count rule for a variance
fraction (``searchsorted(cumsum(ratio), n, side='right') + 1``) -> transform = (X - mean) W on the
device (xps_apply_f64).  This is exactly what sklearn's 'covariance_eigh' solver computes; where
sklearn would pick its unseeded randomized solver this class returns the exact answer instead.
"""
import numpy as np

from . import _linalg as LA


class PCA:
    def __init__(self, n_components=None, **_ignored):
        self.n_components = n_components

    def get_params(self, deep=True):
        return {'n_components': self.n_components}

    def set_params(self, **params):
        for k, v in params.items():
            setattr(self, k, v)
        return self

    def fit(self, X, y=None):
        Xd = LA.to_device(X)
        Xd = Xd.reshape(-1, Xd.shape[-1])
        n, d = Xd.shape
        mean = LA.col_mean(Xd)
        cov = LA.xcov(Xd, None, mean) / (n - 1)
        w, V = LA.eigh_psd(cov)
        w = np.clip(w, 0.0, None)
        comps = V.T.copy()
        idx = np.argmax(np.abs(comps), axis=1)
        signs = np.sign(comps[np.arange(comps.shape[0]), idx])
        signs[signs == 0] = 1
        comps *= signs[:, None]
        total = w.sum()
        ratio = w / total if total > 0 else np.zeros_like(w)
        nc = self.n_components
        if nc is None:
            k = min(n, d)
        elif 0 < nc < 1:
            k = int(np.searchsorted(np.cumsum(ratio), nc, side='right') + 1)
        else:
            k = int(nc)
        k = max(1, min(k, d))
        self.mean_ = mean.cpu().numpy()
        self.components_ = comps[:k]
        self.explained_variance_ = w[:k]
        self.explained_variance_ratio_ = ratio[:k]
        self.singular_values_ = np.sqrt(w[:k] * (n - 1))
        self.n_components_ = k
        self.n_features_in_ = d
        self.n_samples_ = n
        self._mean_d = mean
        self._W_d = LA.to_device(np.ascontiguousarray(self.components_.T))
        return self

    def _check(self):
        if not hasattr(self, 'components_'):
            raise RuntimeError('This PCA instance is not fitted yet.')

    def transform_device(self, Xd, out_f32=False):
        self._check()
        return LA.apply(Xd, self._W_d, self._mean_d, out_f32=out_f32)

    def transform(self, X):
        self._check()
        return LA.like_input(self.transform_device(LA.to_device(X)), X)

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)

    def inverse_transform(self, Z):
        self._check()
        Zd = LA.to_device(Z)
        W = LA.to_device(np.ascontiguousarray(self.components_))
        return (LA.apply(Zd, W) + self._mean_d).cpu().numpy()
