"""Joint-PCA ("LFADS stitching") alignment on the MI355X — drop-in surface of the reference's
``alignment/JointPCA.py`` (JointPCA :13-163, get_joint_PCA_transforms :165-211).

condition averages (xps_cnd_avg_*) -> channel-concatenated matrix -> PCA scores (covariance on
the f64 MFMA + Jacobi, see pca.py) -> per-patient read-in matrix  pinv(A_p) @ latent  solved on
the Gram side:  A_p^T A_p = V L V^T (Jacobi),  W_p = V L^-1 V^T (A_p^T latent).
``dim_red`` stays a pluggable sklearn-style class (:27,199): only the default is accelerated; any
other class is called on the host exactly as the reference would.
"""
import numpy as np

from . import _linalg as LA
from .alignment_utils import _group_conditions_device
from .pca import PCA


class JointPCA:
    def __init__(self, n_components=40, dim_red=PCA):
        self.n_components = n_components
        self.dim_red = dim_red

    def get_params(self, deep=True):
        return {'n_components': self.n_components, 'dim_red': self.dim_red}

    def set_params(self, **params):
        for k, v in params.items():
            setattr(self, k, v)
        return self

    def fit(self, X, y):
        self.transforms = get_joint_PCA_transforms(X, y, n_components=self.n_components, dim_red=self.dim_red)
        self._W_d = [LA.to_device(np.ascontiguousarray(w)) for w in self.transforms]

    def transform(self, X, idx=-1):
        if not self._check_fit():
            raise RuntimeError('Must call fit() before transforming data.')
        if idx == -1:
            return self._transform_multiple(X)
        if idx >= len(self.transforms):
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

    def _w(self, i):
        if not hasattr(self, '_W_d') or len(self._W_d) != len(self.transforms):
            self._W_d = [LA.to_device(np.ascontiguousarray(w)) for w in self.transforms]
        return self._W_d[i]

    def _transform_multiple(self, X):
        return (*[self._transform_single(x, i) for i, x in enumerate(X)],)

    def _transform_single(self, X, idx):
        return LA.like_input(LA.apply(LA.to_device(X), self._w(idx)), X)

    def _check_fit(self):
        try:
            self.transforms
        except AttributeError:
            return False
        return True


def _is_default_pca(dim_red):
    if dim_red is PCA:
        return True
    return getattr(dim_red, '__name__', '') == 'PCA' and getattr(dim_red, '__module__', '').startswith('sklearn.')


def get_joint_PCA_transforms(features, labels, n_components=40, dim_red=PCA):
    avgs = _group_conditions_device(features, labels)                       # [(n_c, T, C_p)] float64, device
    flats = [a.reshape(-1, a.shape[-1]) for a in avgs]
    cat = LA.torch.cat(flats, dim=1).contiguous()                            # (n_c*T, sum C)
    if _is_default_pca(dim_red):
        latent = PCA(n_components=n_components).fit(cat).transform_device(cat)      # device (n_s, k)
    else:
        latent = LA.to_device(dim_red(n_components=n_components).fit_transform(cat.cpu().numpy()))
    out = []
    for A in flats:
        n = A.shape[0]
        w, V = LA.eigh_psd(LA.xcov(A))                                       # A^T A (uncentred)
        keep = w > w[0] * max(n, len(w)) * LA.EPS
        P = LA.dgemm(LA.to_device(V[:, keep] / w[keep]), LA.to_device(V[:, keep]), tb=True)   # (A^T A)^+
        out.append(LA.dgemm(P, LA.xcov(A, latent)).cpu().numpy())            # pinv(A) @ latent
    return (*out,)
