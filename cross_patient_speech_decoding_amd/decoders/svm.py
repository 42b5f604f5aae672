"""Support-vector classifier on the MI355X -- the decoder of BASELINE config 1 (cross-patient SVM decode).

The reference hands the pooled, aligned features to ``sklearn.svm.SVC``: ``SVC(kernel='linear')`` inside a ``BaggingClassifier``
(scripts/aligned_decode_svm.py:262-263) and ``SVC(kernel='rbf', class_weight='balanced')`` behind ``DimRedReshape`` in the
nested-CV and sub-sampling scripts (scripts/aligned_decode_svm_ncv.py:313-321, scripts/aligned_decode_*_subsample.py:249-261);
decoders/cross_pt_decoders.py:29-38 only call ``decoder.fit / predict / score``.  ``SVC`` below is a drop-in for that estimator
(constructor arguments, ``fit`` / ``predict`` / ``decision_function`` / ``score`` / ``classes_`` / ``get_params`` / ``set_params`` /
``clone``), so it can be bagged, grid-searched, put into a pipeline and wrapped by ``crossPtDecoder_*`` unchanged.  It solves what
libsvm solves: one C-SVC per class pair (classes ascending, the lower class positive), SMO with the second-order working-set rule,
tolerance ``tol`` on the maximal KKT violation, libsvm's ``rho``; per-point upper bound C * class_weight * sample_weight; the
prediction is libsvm's one-vs-one vote (ties: the lowest class).  On the device: the Gram matrix X X^T (f64 MFMA), for
``kernel='rbf'`` exp(-gamma ||x - y||^2) from it (csrc/xps_svm.hip, libsvm's formula), ALL pair problems in one launch, the
decision values as f64 GEMMs.  There is no CPU fallback."""
import numpy as np
import torch
from sklearn.base import BaseEstimator, ClassifierMixin

from .._lib import call, lib
from ..alignment import _linalg as LA


def _ovr_from_ovo(dec, n_classes):
    """sklearn.utils.multiclass._ovr_decision_function(dec < 0, -dec, n_classes): votes plus the monotonically squashed sum of
    the pairwise confidences -- what SVC.decision_function returns for decision_function_shape='ovr'."""
    predictions, confidences = dec < 0, -dec
    m = dec.shape[0]
    votes = np.zeros((m, n_classes))
    sums = np.zeros((m, n_classes))
    k = 0
    for i in range(n_classes):
        for j in range(i + 1, n_classes):
            sums[:, i] -= confidences[:, k]
            sums[:, j] += confidences[:, k]
            votes[predictions[:, k] == 0, i] += 1
            votes[predictions[:, k] == 1, j] += 1
            k += 1
    return votes + sums / (3 * (np.abs(sums) + 1))


class SVC(ClassifierMixin, BaseEstimator):
    """C-support vector classification, ``kernel='linear'`` or ``'rbf'`` (sklearn.svm.SVC semantics for these)."""

    def __init__(self, C=1.0, kernel='rbf', gamma='scale', tol=1e-3, max_iter=-1, decision_function_shape='ovr', break_ties=False,
                 class_weight=None, random_state=None):
        self.C = C
        self.kernel = kernel
        self.gamma = gamma
        self.tol = tol
        self.max_iter = max_iter
        self.decision_function_shape = decision_function_shape
        self.break_ties = break_ties
        self.class_weight = class_weight
        self.random_state = random_state

    # ------------------------------------------------------------------ kernels
    def _gamma_value(self, X):
        if self.kernel != 'rbf':
            return 0.0
        if isinstance(self.gamma, str):
            if self.gamma == 'scale':                       # sklearn: 1 / (n_features * X.var()) of the X handed to fit
                var = float(X.var())
                return 1.0 / (X.shape[1] * var) if var != 0 else 1.0
            if self.gamma == 'auto':
                return 1.0 / X.shape[1]
            raise ValueError(f"When 'gamma' is a string, it should be either 'scale' or 'auto'. Got '{self.gamma}' instead.")
        if self.gamma < 0:
            raise ValueError('gamma must be non-negative')
        return float(self.gamma)

    @staticmethod
    def _row_sq_norms(Ad, chunk=2048):
        """|a_i|^2 as the diagonal of the Gram matrix of the rows (f64 MFMA GEMM, in chunks of rows; no element-wise host math)."""
        return torch.cat([torch.diagonal(LA.dgemm(Ad[i:i + chunk], Ad[i:i + chunk], tb=True)) for i in range(0, Ad.shape[0], chunk)]).contiguous()

    def _kernel_matrix(self, Ad, na, Bd, nb):
        """K(A, B) on the device: Gram matrix by the f64 MFMA GEMM; rbf: exp(-gamma (|a|^2 + |b|^2 - 2 a.b)) from it."""
        G = LA.dgemm(Ad, Bd, tb=True)
        if self.kernel == 'linear':
            return G
        K = torch.empty_like(G)
        call('xps_rbf_from_gram_f64', G.data_ptr(), G.stride(0), na.data_ptr(), nb.data_ptr(), G.shape[0], G.shape[1], float(self._gamma),
             K.data_ptr(), K.stride(0), LA._stream())
        return K

    # ------------------------------------------------------------------ fit
    def fit(self, X, y, sample_weight=None):
        if self.kernel not in ('linear', 'rbf'):
            raise NotImplementedError("the HIP SVC implements kernel='linear' and kernel='rbf' (what the reference's decoders use)")
        if self.break_ties:
            raise NotImplementedError('break_ties is not implemented on the HIP path')
        if self.decision_function_shape not in ('ovr', 'ovo'):
            raise ValueError("decision_function_shape must be 'ovr' or 'ovo'")
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        y = np.asarray(y)
        if X.ndim != 2 or X.shape[0] != y.shape[0]:
            raise ValueError('X must be (n_samples, n_features) and y (n_samples,)')
        self._gamma = self._gamma_value(X)
        classes, yi_all = np.unique(y, return_inverse=True)
        # class weights (sklearn.utils.class_weight.compute_class_weight on the y handed to fit)
        if self.class_weight is None:
            cw = np.ones(len(classes))
        elif isinstance(self.class_weight, str):
            if self.class_weight != 'balanced':
                raise ValueError("class_weight must be 'balanced', a dict or None")
            cw = len(y) / (len(classes) * np.bincount(yi_all, minlength=len(classes)).astype(np.float64))
        else:
            cw = np.array([float(self.class_weight.get(c, 1.0)) for c in classes])
        self.class_weight_ = cw
        # sample weights (BaggingClassifier passes the bootstrap multiplicities): per-point bound C * class weight * w; zero-weight
        # points are dropped before training, as sklearn's libsvm does
        w = np.ones(X.shape[0]) if sample_weight is None else np.asarray(sample_weight, dtype=np.float64)
        if w.shape != (X.shape[0],) or (w < 0).any():
            raise ValueError('sample_weight must be a non-negative (n_samples,) vector')
        keep = w > 0
        X, y, w = X[keep], y[keep], w[keep]
        self.classes_, yi = np.unique(y, return_inverse=True)
        if len(self.classes_) != len(classes):               # a class lost all its weight: its weight entry goes with it
            cw = cw[np.isin(classes, self.classes_)]
        k = len(self.classes_)
        if k < 2:
            raise ValueError('The number of classes has to be greater than one; got 1 class')
        n = X.shape[0]
        members = [np.flatnonzero(yi == c).astype(np.int32) for c in range(k)]       # original order inside a class (libsvm groups so)
        idx, off, npos, pairs = [], [0], [], []
        for a in range(k):
            for b in range(a + 1, k):
                idx += [members[a], members[b]]
                off.append(off[-1] + len(members[a]) + len(members[b]))
                npos.append(len(members[a]))
                pairs.append((a, b))
        idx = np.concatenate(idx)
        max_pts = int(max(np.diff(off)))
        if max_pts > lib().xps_svm_smo_f64_max_points():
            raise ValueError(f'a class pair has {max_pts} samples; the LDS-resident solver takes {lib().xps_svm_smo_f64_max_points()}')
        dev = LA.device()
        Xd = torch.from_numpy(X).to(dev)
        self._sq = self._row_sq_norms(Xd) if self.kernel == 'rbf' else None
        K = self._kernel_matrix(Xd, self._sq, Xd, self._sq)
        idx_d = torch.from_numpy(idx).to(dev)
        off_d = torch.tensor(off, dtype=torch.int32, device=dev)
        npos_d = torch.tensor(npos, dtype=torch.int32, device=dev)
        P = len(pairs)
        alpha = torch.empty(len(idx), dtype=torch.float64, device=dev)
        rho = torch.empty(P, dtype=torch.float64, device=dev)
        iters = torch.empty(P, dtype=torch.int32, device=dev)
        max_iter = int(self.max_iter) if self.max_iter and self.max_iter > 0 else max(10_000_000, 100 * max_pts)
        cb = torch.from_numpy(float(self.C) * cw[yi[idx]] * w[idx]).to(dev)
        call('xps_svm_smo_f64', K.data_ptr(), K.stride(0), idx_d.data_ptr(), off_d.data_ptr(), npos_d.data_ptr(), P, max_pts,
             cb.data_ptr(), float(self.tol), max_iter, alpha.data_ptr(), rho.data_ptr(), iters.data_ptr(), LA._stream())
        # signed dual coefficients of every pair scattered into a dense (P, n) matrix
        coef = torch.zeros(P, n, dtype=torch.float64, device=dev)
        sign = torch.ones(len(idx), dtype=torch.float64, device=dev)
        for p_, (o0, o1, npp) in enumerate(zip(off[:-1], off[1:], npos)):
            sign[o0 + npp:o1] = -1.0
        rows = torch.repeat_interleave(torch.arange(P, device=dev), torch.from_numpy(np.diff(off)).to(dev))
        coef[rows, idx_d.long()] = alpha * sign
        self._Xd = Xd
        self._rho = rho
        self._pairs = pairs
        self.n_iter_ = iters.cpu().numpy()
        self.dual_coef_pairs_ = coef                                # (kept on the device; sklearn's dual_coef_ packs it differently)
        self.n_features_in_ = X.shape[1]
        flip = -1.0 if k == 2 else 1.0                              # sklearn flips coef_ / intercept_ / the decision of a binary problem
        if self.kernel == 'linear':
            self._W = LA.dgemm(coef, Xd)                            # (P, d) weight vectors: W = coef X (f64 GEMM)
            self.coef_ = flip * self._W.cpu().numpy()
        self.intercept_ = flip * -rho.cpu().numpy()
        return self

    # ------------------------------------------------------------------ decisions
    def _pair_decisions(self, X):
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        Xd = torch.from_numpy(X).to(LA.device())
        if self.kernel == 'linear':
            return (LA.dgemm(Xd, self._W, tb=True) - self._rho[None, :]).cpu().numpy()     # (m, P)
        Kx = self._kernel_matrix(Xd, self._row_sq_norms(Xd), self._Xd, self._sq)                     # (m, n)
        return (LA.dgemm(Kx, self.dual_coef_pairs_, tb=True) - self._rho[None, :]).cpu().numpy()

    def decision_function(self, X):
        """sklearn's layout: two classes: a vector, positive for ``classes_[1]``; more: ``decision_function_shape='ovo'`` the
        one-vs-one values (m, P), columns in libsvm's pair order (0 v 1, 0 v 2, ..., k-2 v k-1), ``'ovr'`` (the default) the (m, k)
        vote-plus-confidence scores sklearn derives from them."""
        dec = self._pair_decisions(X)
        k = len(self.classes_)
        if k == 2:
            return -dec[:, 0]
        if self.decision_function_shape == 'ovr':
            return _ovr_from_ovo(dec, k)
        return dec

    def predict(self, X):
        dec = self._pair_decisions(X)
        k = len(self.classes_)
        votes = np.zeros((dec.shape[0], k), dtype=np.int64)
        for p_, (a, b) in enumerate(self._pairs):
            pos = dec[:, p_] > 0
            votes[pos, a] += 1
            votes[~pos, b] += 1
        return self.classes_[np.argmax(votes, axis=1)]             # first maximum: libsvm's tie rule
