"""CCA alignment of two patients' latent dynamics on the MI355X — drop-in surface of the
reference's ``alignment/AlignCCA.py`` (AlignCCA :11-119, reshape_latent_dynamics :122,
extract_latent_dynamics_by_class :156, extract_latent_dynamics_by_trial_subselect :186,
shared_trial_subselect :205, CCA_align :235).

Same constructor, ``fit(X_a, X_b, y_a, y_b)`` / ``transform(X)``, numpy float64 attributes
``M_a``, ``M_b``, ``canon_corrs`` and error strings.  Arithmetic (HIP, libxps.so):

  condition means      xps_cnd_avg_*   (np.mean semantics, bit for bit)
  covariances          xps_xcov_f64    C_aa, C_bb, C_ab centred, f64 MFMA
  whitening + SVD      xps_jacobi_*    C_aa = V L V^T  ->  W_a = V L^-1/2 (rank-truncated);
                                       K = W_a^T C_ab W_b = U S V^T
  directions           M_a = W_a U[:, :d], M_b = W_b V[:, :d], d = min(rank_a, rank_b)
  transform            xps_apply_f64   X @ (M_b pinv(M_a))

This is the reference's QR+SVD CCA (:269-277) written on the covariance side: with L^T = Q R,
R^-1 = W_a O for an orthogonal O, so pinv(R_a) U_ref = W_a U; M_a / M_b agree with the reference
up to a common sign per canonical pair, and ``transform`` outputs agree exactly (signs cancel).
"""
import numpy as np

from . import _linalg as LA
from .alignment_utils import _cnd_avg_device, label2str


class AlignCCA:
    """CCA-based alignment of two neural datasets into a shared latent space.

    Attributes:
        type (str): 'class' (condition averages) or 'trial' (matched random trials).
        return_space (str): 'b_to_a', 'a_to_b' or 'shared'.
        M_a, M_b (ndarray float64): manifold directions (set by fit).
        canon_corrs (ndarray float64): canonical correlations (set by fit).
    """

    def __init__(self, type='class', return_space='b_to_a'):
        self.type = type
        self.return_space = return_space

    # sklearn-style parameter access so the class can sit inside clone()-able wrappers
    def get_params(self, deep=True):
        return {'type': self.type, 'return_space': self.return_space}

    def set_params(self, **params):
        for k, v in params.items():
            setattr(self, k, v)
        return self

    def fit(self, X_a, X_b, y_a, y_b):
        L_a, L_b = _latent_dynamics_device(X_a, X_b, y_a, y_b, self.type)
        M_a, M_b, S = _cca_device(L_a, L_b)
        self.M_a, self.M_b, self.canon_corrs = M_a, M_b, S
        self._maps = {}
        return None

    def transform(self, X):
        if not self._check_fit():
            raise RuntimeError('Must call fit() before transforming data.')
        if self.return_space in ['b_to_a', 'a_to_b']:
            return self._transform_single(X)
        return self._transform_shared(X)

    def _map(self, key):
        # M_b pinv(M_a) (or the reverse) is formed once per fit, on the device
        maps = self.__dict__.setdefault('_maps', {})
        if key not in maps:
            src, dst = (self.M_b, self.M_a) if key == 'b_to_a' else (self.M_a, self.M_b)
            maps[key] = LA.dgemm(LA.to_device(src), LA.to_device(LA.pinv_small(dst)))
        return maps[key]

    def _transform_single(self, X):
        key = 'b_to_a' if self.return_space == 'b_to_a' else 'a_to_b'
        return _apply_host(X, self._map(key))

    def _transform_shared(self, X):
        return _apply_host(X[0], LA.to_device(self.M_a)), _apply_host(X[1], LA.to_device(self.M_b))

    def _check_fit(self):
        try:
            self.M_a
            self.M_b
        except AttributeError:
            return False
        return True


def _apply_host(X, W_dev):
    """X (..., d_in) ndarray or tensor -> float64 (..., d_out) = X @ W on the device (ndarray; a device tensor stays one)."""
    return LA.like_input(LA.apply(LA.to_device(X), W_dev), X)


def _latent_dynamics_device(X_a, X_b, y_a, y_b, type='class'):
    if type == 'class':
        k_a, k_b = label2str(y_a), label2str(y_b)
        u_a, A = _cnd_avg_device(X_a, k_a)
        u_b, B = _cnd_avg_device(X_b, k_b)
        _, i_a, i_b = np.intersect1d(u_a, u_b, assume_unique=True, return_indices=True)
        A = A[LA.torch.from_numpy(i_a).to(A.device)]
        B = B[LA.torch.from_numpy(i_b).to(B.device)]
    elif type == 'trial':
        A, B = extract_latent_dynamics_by_trial_subselect(X_a, X_b, y_a, y_b)
        A, B = LA.to_device(A), LA.to_device(B)
    else:
        raise ValueError('type must be "class" or "trial".')
    return A.reshape(-1, A.shape[-1]), B.reshape(-1, B.shape[-1])


def _cca_device(La, Lb):
    """La (n, d_a), Lb (n, d_b) device matrices (rows = samples) -> numpy M_a, M_b, S.

    Reference (AlignCCA.py:259-283): centre, d = min(matrix_rank), thin QR of both, SVD of Q_a^T Q_b, M = pinv(R) [U | V][:, :d].
    Here the orthonormal bases come from one-sided Jacobi SVDs of the centred data themselves (L = U S V^T, computed on the
    tall matrix, not on its Gram matrix): the numerical rank follows LAPACK's rule on the singular values
    (s > s_max * max(shape) * eps), the condition number is not squared, and with Q = U[:, :r] the canonical variates
    L M = Q [U_k | V_k] are the reference's (any orthonormal basis of the same column space gives the same variates and the
    same M for full-rank data)."""
    n = La.shape[0]
    Wa, s_a, V_a = LA.svd_tall_device(La, LA.col_mean(La))
    Wb, s_b, V_b = LA.svd_tall_device(Lb, LA.col_mean(Lb))
    r_a = LA.rank_from_singular_values(s_a, (n, La.shape[1]))
    r_b = LA.rank_from_singular_values(s_b, (n, Lb.shape[1]))
    d = min(r_a, r_b)
    # K = Q_a^T Q_b with Q = W^T diag(1 / s) restricted to the numerical range
    G = LA.dgemm(Wa[:r_a].contiguous(), Wb[:r_b].contiguous(), tb=True).cpu().numpy()
    K = G / s_a[:r_a, None] / s_b[None, :r_b]
    U, S, Vt = LA.svd(LA.to_device(np.ascontiguousarray(K)))
    M_a = (V_a[:, :r_a] / s_a[:r_a]) @ U[:, :d]
    M_b = (V_b[:, :r_b] / s_b[:r_b]) @ Vt.T[:, :d]
    S = S[:d].copy()
    S[S < 0] = 0
    S[S >= 1] = 1
    return M_a, M_b, S


def reshape_latent_dynamics(X_a, X_b, y_a, y_b, type='class'):
    """Shared-condition latent dynamics of both datasets with time folded into rows,
    (n_shared * T, d_a), (n_shared * T, d_b) float64."""
    L_a, L_b = _latent_dynamics_device(X_a, X_b, y_a, y_b, type)
    return L_a.cpu().numpy(), L_b.cpu().numpy()


def extract_latent_dynamics_by_class(X_a, X_b, y_a, y_b):
    k_a, k_b = label2str(y_a), label2str(y_b)
    u_a, A = _cnd_avg_device(X_a, k_a)
    u_b, B = _cnd_avg_device(X_b, k_b)
    _, i_a, i_b = np.intersect1d(u_a, u_b, assume_unique=True, return_indices=True)
    return A.cpu().numpy()[i_a], B.cpu().numpy()[i_b]


def extract_latent_dynamics_by_trial_subselect(X_a, X_b, y_a, y_b):
    y_a, y_b = label2str(y_a), label2str(y_b)
    return shared_trial_subselect(X_a, X_b, y_a, y_b)


def shared_trial_subselect(X_a, X_b, y_a, y_b):
    """Equal numbers of randomly chosen trials per shared class (host indexing; the draws come
    from numpy's global RNG exactly as in the reference, :225-226)."""
    X_a, X_b = np.asarray(X_a), np.asarray(X_b)
    L_a, L_b = [], []
    for c in np.intersect1d(y_a, y_b):
        curr_a = np.random.permutation(np.where(y_a == c)[0])
        curr_b = np.random.permutation(np.where(y_b == c)[0])
        k = min(curr_a.shape[0], curr_b.shape[0])
        L_a.append(X_a[curr_a[:k]])
        L_b.append(X_b[curr_b[:k]])
    return np.vstack(L_a), np.vstack(L_b)


def CCA_align(L_a, L_b):
    """CCA between two (d, n_samples) latent matrices -> (M_a, M_b, S).

    Like the reference (:259-260) the inputs are mean-centred IN PLACE along the sample axis when
    they are float ndarrays; the means themselves come from the device reduction."""
    A = LA.to_device(np.ascontiguousarray(np.asarray(L_a).T))
    B = LA.to_device(np.ascontiguousarray(np.asarray(L_b).T))
    if isinstance(L_a, np.ndarray) and np.issubdtype(L_a.dtype, np.floating):
        L_a -= LA.col_mean(A).cpu().numpy()[:, None].astype(L_a.dtype)
    if isinstance(L_b, np.ndarray) and np.issubdtype(L_b.dtype, np.floating):
        L_b -= LA.col_mean(B).cpu().numpy()[:, None].astype(L_b.dtype)
    return _cca_device(A, B)
