"""PCA without mean-centring — counterpart of the reference's decomposition/NoCenterPCA.py:13-113.
The reference takes the SVD of the raw data matrix; here the squared singular values and right singular
vectors come from the UNCENTRED Gram matrix X^T X (f64 MFMA) and its Jacobi eigendecomposition on the
MI355X, and ``transform`` is the device apply kernel.  Component signs are fixed so that the entry of
largest magnitude of every component is positive (an SVD's signs are arbitrary)."""
import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin

from ..alignment import _linalg as LA


class NoCenterPCA(BaseEstimator, TransformerMixin):
    def __init__(self, n_components=None):
        self.n_components = n_components
        self._fit = False

    def fit(self, X, y=None):
        Xd = LA.to_device(X)
        w, V = LA.eigh_psd(LA.xcov(Xd))                     # w = S**2, V columns = right singular vectors
        w = np.clip(w, 0.0, None)
        idx = np.argmax(np.abs(V), axis=0)
        signs = np.sign(V[idx, np.arange(V.shape[1])])
        signs[signs == 0] = 1
        V = V * signs
        k = self._get_components(X, np.sqrt(w))
        self.components_ = V[:, :k]
        self.explained_variance_ = w[:min(X.shape)]
        self._W_d = LA.to_device(np.ascontiguousarray(self.components_))
        self._fit = True
        return self

    def transform(self, X):
        if not self._fit:
            raise ValueError("PCA must be fit before transforming data.")      # message of the reference (NoCenterPCA.py:112-113)
        return LA.apply(LA.to_device(X), self._W_d).cpu().numpy()

    def fit_transform(self, X, y=None):
        return self.fit(X, y).transform(X)

    def _get_components(self, X, S):
        """How many components to keep (semantics of the reference's NoCenterPCA.py:86-104): everything when n_components is
        unset or not below min(X.shape) (with the same notice), the smallest count whose cumulative share of the squared
        singular values reaches a fractional n_components, else n_components itself."""
        limit = min(X.shape)
        want = self.n_components
        if want is None or want >= limit:
            print("n_components is None or greater than the number of features"
                  "/samples. Using n_components = min(X.shape)")
            return limit
        if want >= 1:
            return int(want)
        energy = np.square(np.asarray(S, dtype=np.float64))
        share = np.cumsum(energy) / energy.sum()
        return int(np.searchsorted(share, want, side='left')) + 1
