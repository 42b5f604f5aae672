"""Linear support-vector classifier on the MI355X -- the decoder of BASELINE config 1 (cross-patient SVM decode).

The reference hands the pooled, aligned features to ``sklearn.svm.SVC(kernel='linear')`` inside a ``BaggingClassifier``
(scripts/aligned_decode_svm.py:262-263; decoders/cross_pt_decoders.py:29-38 only call ``decoder.fit / predict / score``).
``SVC`` below is a drop-in for that estimator (same constructor arguments for the linear case, ``fit`` / ``predict`` /
``decision_function`` / ``score`` / ``classes_`` / ``get_params`` / ``set_params`` / ``clone``), so it can be bagged, grid-searched
and wrapped by ``crossPtDecoder_*`` unchanged.  It solves what libsvm solves: one C-SVC per class pair (classes ascending, the lower
class positive), SMO with the second-order working-set rule, tolerance ``tol`` on the maximal KKT violation, libsvm's ``rho``; the
prediction is libsvm's one-vs-one vote (ties: the lowest class).  On the device: the Gram matrix X X^T (f64 MFMA), ALL pair
problems in one launch (csrc/xps_svm.hip), the weight vectors and the decision values (f64 GEMM).  There is no CPU fallback."""
import ctypes as C

import numpy as np
import torch
from sklearn.base import BaseEstimator, ClassifierMixin

from .._lib import call, lib
from ..alignment import _linalg as LA


class SVC(ClassifierMixin, BaseEstimator):
    """C-support vector classification with a linear kernel (sklearn.svm.SVC(kernel='linear') semantics)."""

    def __init__(self, C=1.0, kernel='linear', tol=1e-3, max_iter=-1, decision_function_shape='ovr', break_ties=False,
                 class_weight=None, random_state=None):
        self.C = C
        self.kernel = kernel
        self.tol = tol
        self.max_iter = max_iter
        self.decision_function_shape = decision_function_shape
        self.break_ties = break_ties
        self.class_weight = class_weight
        self.random_state = random_state

    def fit(self, X, y, sample_weight=None):
        if self.kernel != 'linear':
            raise NotImplementedError("the HIP SVC implements kernel='linear' (what the reference's decoders use)")
        if self.class_weight is not None or self.break_ties:
            raise NotImplementedError('class_weight / break_ties are not implemented on the HIP path')
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        y = np.asarray(y)
        if X.ndim != 2 or X.shape[0] != y.shape[0]:
            raise ValueError('X must be (n_samples, n_features) and y (n_samples,)')
        # sample weights (BaggingClassifier passes the bootstrap multiplicities): per-point bound C * w; zero-weight points are
        # dropped before training, as sklearn's libsvm does
        w = np.ones(X.shape[0]) if sample_weight is None else np.asarray(sample_weight, dtype=np.float64)
        if w.shape != (X.shape[0],) or (w < 0).any():
            raise ValueError('sample_weight must be a non-negative (n_samples,) vector')
        keep = w > 0
        X, y, w = X[keep], y[keep], w[keep]
        self.classes_, yi = np.unique(y, return_inverse=True)
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
        Xt = Xd.t().contiguous()                                   # (d, n): K = (X^T)^T (X^T) on the f64 MFMA
        K = LA.xcov(Xt, Xt)
        idx_d = torch.from_numpy(idx).to(dev)
        off_d = torch.tensor(off, dtype=torch.int32, device=dev)
        npos_d = torch.tensor(npos, dtype=torch.int32, device=dev)
        P = len(pairs)
        alpha = torch.empty(len(idx), dtype=torch.float64, device=dev)
        rho = torch.empty(P, dtype=torch.float64, device=dev)
        iters = torch.empty(P, dtype=torch.int32, device=dev)
        max_iter = int(self.max_iter) if self.max_iter and self.max_iter > 0 else max(10_000_000, 100 * max_pts)
        cb = torch.from_numpy(float(self.C) * w[idx]).to(dev)
        call('xps_svm_smo_f64', K.data_ptr(), K.stride(0), idx_d.data_ptr(), off_d.data_ptr(), npos_d.data_ptr(), P, max_pts,
             cb.data_ptr(), float(self.tol), max_iter, alpha.data_ptr(), rho.data_ptr(), iters.data_ptr(), LA._stream())
        # signed dual coefficients of every pair scattered into a dense (P, n) matrix -> weight vectors W = coef X (f64 GEMM)
        coef = torch.zeros(P, n, dtype=torch.float64, device=dev)
        sign = torch.ones(len(idx), dtype=torch.float64, device=dev)
        for p_, (o0, o1, npp) in enumerate(zip(off[:-1], off[1:], npos)):
            sign[o0 + npp:o1] = -1.0
        rows = torch.repeat_interleave(torch.arange(P, device=dev), torch.from_numpy(np.diff(off)).to(dev))
        coef[rows, idx_d.long()] = alpha * sign
        self._W = LA.dgemm(coef, Xd)                                # (P, d)
        self._rho = rho
        self._pairs = pairs
        self.n_iter_ = iters.cpu().numpy()
        self.dual_coef_pairs_ = coef                                # (kept on the device; sklearn's dual_coef_ packs it differently)
        self.coef_ = self._W.cpu().numpy()
        self.intercept_ = -rho.cpu().numpy()
        self.n_features_in_ = X.shape[1]
        return self

    def _pair_decisions(self, X):
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        Xd = torch.from_numpy(X).to(LA.device())
        return (LA.dgemm(Xd, self._W, tb=True) - self._rho[None, :]).cpu().numpy()     # (m, P)

    def decision_function(self, X):
        """One-vs-one decision values (m, P), columns in libsvm's pair order (0 v 1, 0 v 2, ..., k-2 v k-1); for two classes a
        vector, positive for ``classes_[1]`` as sklearn reports it."""
        dec = self._pair_decisions(X)
        if len(self.classes_) == 2:
            return -dec[:, 0]
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
