"""k-fold DataModules for single-patient and cross-patient (aligned) decoding — counterpart of the
reference's nn_models/data_utils/datamodules.py (SimpleMicroDataModule :21, AlignedMicroDataModule
:211, AlignedMicroValDataModule :442, process_aligner :515).

Same constructor arguments and methods (setup / set_fold / train|val|test_dataloader /
get_data_shape / select_cv).  Differences by design:
  * the per-fold PCA (0.95 variance) and CCA alignment run on the MI355X (alignment.PCA /
    alignment.AlignCCA: f64-MFMA covariances + Jacobi + batched apply);
  * fold caches live in memory (and optionally as ``fold_data/fold_{k}.npz`` with the reference's
    dataset names train_data, train_labels, val_data, val_labels, test_data, test_labels) instead
    of being re-read from HDF5 on every *_dataloader() call (h5py is not in the image);
  * ``process_aligner_multiview`` adds the MCCA branch the reference lacks (its DataModules only
    accept the pairwise fit(X_a, X_b, y_a, y_b) signature, datamodules.py:561-565).
"""
import os
from pathlib import Path

import numpy as np
import torch
from sklearn.model_selection import KFold, StratifiedKFold, train_test_split
from torch.utils.data import DataLoader, TensorDataset

from ...alignment import PCA


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def process_aligner(X, y, y_align, pool_data, algner, n_components=0.95):
    """PCA-reduce every patient, align each pooled patient to the target with ``algner()``
    (pairwise fit/transform API), pool.  Returns (X_pool float32 tensor, y_pool long tensor,
    fitted target PCA) like the reference (:574)."""
    Xn = _np(X)
    cross = [_np(x) for x, _, _ in pool_data]
    cross_dr = []
    for x in cross:
        z = PCA(n_components=n_components).fit_transform(x.reshape(-1, x.shape[-1]))
        cross_dr.append(z.reshape(x.shape[0], -1, z.shape[-1]))
    tar_dr = PCA(n_components=n_components)
    z = tar_dr.fit_transform(Xn.reshape(-1, Xn.shape[-1]))
    X_tar = z.reshape(Xn.shape[0], -1, z.shape[-1])
    if y_align is None:
        y_align = y
    aligned = []
    for x_dr, (_, _, ya_c) in zip(cross_dr, pool_data):
        al = algner()
        al.fit(X_tar, x_dr, _np(y_align), _np(ya_c))
        aligned.append(al.transform(x_dr))
    X_pool = np.vstack([X_tar] + aligned)
    ys = [_np(y)] + [_np(yy) for _, yy, _ in pool_data]
    try:                                   # (N, L) labels with equal N hstack to (N, L*P): reference quirk kept
        y_pool = np.hstack(ys)
    except ValueError:
        y_pool = np.vstack(ys)
    y_pool = torch.Tensor(y_pool).long()
    if y_pool.dim() > 1 and y_pool.shape[1] == 1:
        y_pool = y_pool.squeeze(1)
    return torch.Tensor(X_pool), y_pool, tar_dr


def process_aligner_sharded(X, y, y_align, pool_data, algner, n_components=0.95, group=None):
    """``process_aligner`` with the PATIENTS sharded over the ranks of ``group`` (P patients <-> P GPUs).

    Rank 0 owns the target, pooled patient i (0-based) is owned by rank (i + 1) % world.  Every owner reduces its
    patients with PCA on its own GPU; the target's latent trials are broadcast once (2048 x 200 x d floats: 49 MB at
    d = 30); every owner fits / applies its CCA maps; the aligned trials of each patient are broadcast from their owner,
    so all ranks end with the pooled training set of the single-process function (same order, same values: every
    decomposition is deterministic).  Exchanges: 1 + P broadcasts of (N, T, d) float32 blocks and their shapes; no other
    collective (the eigen-decompositions are tiny and never split).  Returns what ``process_aligner`` returns; the
    fitted target PCA lives on every rank (each fits it: 9 ms, cheaper than shipping it)."""
    import torch.distributed as dist
    if group is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return process_aligner(X, y, y_align, pool_data, algner, n_components)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    Xn = _np(X)
    if y_align is None:
        y_align = y
    tar_dr = PCA(n_components=n_components)
    z = tar_dr.fit_transform(Xn.reshape(-1, Xn.shape[-1]))
    X_tar = z.reshape(Xn.shape[0], -1, z.shape[-1])
    # the target PCA is deterministic, so every rank already holds the same X_tar: no broadcast needed for it
    aligned = []
    for i, (x, _, ya_c) in enumerate(pool_data):
        owner = (i + 1) % world
        src = dist.get_global_rank(group, owner) if hasattr(dist, 'get_global_rank') else owner
        if rank == owner:
            xn = _np(x)
            zz = PCA(n_components=n_components).fit_transform(xn.reshape(-1, xn.shape[-1]))
            x_dr = zz.reshape(xn.shape[0], -1, zz.shape[-1])
            al = algner()
            al.fit(X_tar, x_dr, _np(y_align), _np(ya_c))
            out = np.ascontiguousarray(al.transform(x_dr), dtype=np.float32)
            shape = torch.tensor(out.shape, dtype=torch.int64, device=dev)
        else:
            shape = torch.zeros(3, dtype=torch.int64, device=dev)
        dist.broadcast(shape, src=src, group=group)
        buf = torch.from_numpy(out).to(dev) if rank == owner else torch.empty(tuple(int(v) for v in shape), dtype=torch.float32,
                                                                             device=dev)
        dist.broadcast(buf, src=src, group=group)
        aligned.append(buf.cpu().numpy())
    X_pool = np.vstack([X_tar] + aligned)
    ys = [_np(y)] + [_np(yy) for _, yy, _ in pool_data]
    try:
        y_pool = np.hstack(ys)
    except ValueError:
        y_pool = np.vstack(ys)
    y_pool = torch.Tensor(y_pool).long()
    if y_pool.dim() > 1 and y_pool.shape[1] == 1:
        y_pool = y_pool.squeeze(1)
    return torch.Tensor(X_pool), y_pool, tar_dr


def process_aligner_multiview_sharded(X, y, y_align, pool_data, algner, n_components=0.95, group=None):
    """``process_aligner_multiview`` with the PATIENTS sharded over the ranks of ``group`` (P patients
    <-> P GPUs; view 0 = the target).  View p is owned by rank p % world: its owner runs its PCA, its condition means and its
    block row of the MCCA cross-covariance on its own GPU (``AlignMCCA.fit(..., group=)``: views and block rows are exchanged,
    the eigensolve is replicated), maps its own trials into the shared space and broadcasts them (N x T x k float32).  Every
    rank ends with the pooled set of the single-process function, bit for bit.  ``algner()`` must accept ``fit(Xs, ys,
    group=)`` (AlignMCCA).  The target map needs the target's PCA on every rank: each fits it (deterministic, 9 ms)."""
    import torch.distributed as dist
    if group is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return process_aligner_multiview(X, y, y_align, pool_data, algner, n_components)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device('cuda', torch.cuda.current_device())
    raw = [X] + [x for x, _, _ in pool_data]
    pcas, views = [None] * len(raw), [None] * len(raw)
    for i, x in enumerate(raw):
        if i % world == rank or i == 0:
            xn = _np(x)
            pcas[i] = PCA(n_components=n_components)
            z = pcas[i].fit_transform(xn.reshape(-1, xn.shape[-1]))
            views[i] = z.reshape(xn.shape[0], -1, z.shape[-1])
    if y_align is None:
        y_align = y
    labs = [_np(y_align)] + [_np(ya) for _, _, ya in pool_data]
    al = algner()
    al.fit([v if i % world == rank else None for i, v in enumerate(views)], labs, group=group)
    shared = []
    for i in range(len(raw)):
        owner = i % world
        src = dist.get_global_rank(group, owner) if hasattr(dist, 'get_global_rank') else owner
        if rank == owner:
            out = torch.from_numpy(np.ascontiguousarray(al.transform(views[i], idx=i), dtype=np.float32)).to(dev)
            shape = torch.tensor(out.shape, dtype=torch.int64, device=dev)
        else:
            shape = torch.zeros(3, dtype=torch.int64, device=dev)
        dist.broadcast(shape, src=src, group=group)
        if rank != owner:
            out = torch.empty(tuple(int(v) for v in shape), dtype=torch.float32, device=dev)
        dist.broadcast(out, src=src, group=group)
        shared.append(out.cpu().numpy())
    X_pool = np.vstack(shared)
    ys = [_np(y)] + [_np(yy) for _, yy, _ in pool_data]
    y_pool = torch.Tensor(np.vstack(ys) if ys[0].ndim > 1 else np.hstack(ys)).long()

    class _TargetMap:
        def transform(self, X2d):
            return al.transform(pcas[0].transform(X2d), idx=0)
    return torch.Tensor(X_pool), y_pool, _TargetMap()


def process_aligner_multiview(X, y, y_align, pool_data, algner, n_components=0.95):
    """Multiview (MCCA / joint-PCA) counterpart: PCA per patient, ONE ``algner()`` fitted on all
    views ([target] + pooled; fit(Xs, ys) / transform(X, idx) API), every view mapped into the shared
    space and pooled.  Returns (X_pool, y_pool, transform_target) where ``transform_target(X2d)``
    maps raw target data (val / test folds) through the target PCA and the target's view map."""
    Xn = _np(X)
    views_raw = [Xn] + [_np(x) for x, _, _ in pool_data]
    pcas, views = [], []
    for x in views_raw:
        p = PCA(n_components=n_components)
        z = p.fit_transform(x.reshape(-1, x.shape[-1]))
        pcas.append(p)
        views.append(z.reshape(x.shape[0], -1, z.shape[-1]))
    if y_align is None:
        y_align = y
    labs = [_np(y_align)] + [_np(ya) for _, _, ya in pool_data]
    al = algner()
    al.fit(views, labs)
    shared = [np.ascontiguousarray(al.transform(v, idx=i), dtype=np.float32) for i, v in enumerate(views)]
    X_pool = np.vstack(shared)
    ys = [_np(y)] + [_np(yy) for _, yy, _ in pool_data]
    y_pool = torch.Tensor(np.vstack(ys) if ys[0].ndim > 1 else np.hstack(ys)).long()

    class _TargetMap:
        def transform(self, X2d):
            z = pcas[0].transform(X2d)
            return al.transform(z, idx=0)
    return torch.Tensor(X_pool), y_pool, _TargetMap()


class _FoldModule:
    """Shared k-fold machinery: in-memory fold cache + loaders."""

    def __init__(self, data, labels, batch_size, folds, val_size, augmentations, data_path, save_folds=False):
        self.data, self.labels = data, labels
        self.batch_size, self.folds, self.val_size = batch_size, folds, val_size
        self.augmentations = augmentations if augmentations else []
        self.current_fold = 0
        self.data_path = Path(os.getcwd() if data_path is None else data_path)
        self.save_folds = save_folds
        self._folds = {}

    # -- reference API ---------------------------------------------------------------------
    def set_fold(self, fold):
        assert 0 <= fold < self.folds, "Fold index out of range"
        self.current_fold = fold

    def select_cv(self, folds):
        cv_labels = self.labels[:, 0] if len(self.labels.shape) > 1 else self.labels
        class_counts = torch.bincount(torch.as_tensor(cv_labels))
        if (class_counts < folds).any():
            return KFold(n_splits=folds, shuffle=True)
        return StratifiedKFold(n_splits=folds, shuffle=True)

    def get_data_shape(self):
        return tuple(self._folds[self.current_fold]['train_data'].shape)

    def _store(self, k, **arrays):
        self._folds[k] = {n: (None if a is None else torch.as_tensor(_np(a))) for n, a in arrays.items()}
        if self.save_folds:
            os.makedirs(self.data_path / 'fold_data', exist_ok=True)
            np.savez(self.data_path / 'fold_data' / f'fold_{k}.npz',
                     **{n: _np(a) for n, a in arrays.items() if a is not None})

    FOLD_KEYS = ('train_data', 'train_labels', 'val_data', 'val_labels', 'test_data', 'test_labels')

    def load_folds(self):
        """Re-populate the in-memory fold cache from ``fold_data/fold_{k}.npz`` (written by ``save_folds=True``): the
        datasets carry the reference's names (datamodules.py:506-512); a missing validation split stays None."""
        for k in range(self.folds):
            path = self.data_path / 'fold_data' / f'fold_{k}.npz'
            with np.load(path) as f:
                self._folds[k] = {n: (torch.as_tensor(f[n]) if n in f.files else None) for n in self.FOLD_KEYS}
        return self

    def _loader(self, which, shuffle):
        f = self._folds[self.current_fold]
        d, l = f[f'{which}_data'], f[f'{which}_labels']
        if d is None:
            return None
        n = len(d) if self.batch_size == -1 else self.batch_size
        return DataLoader(TensorDataset(d.float(), l.long()), batch_size=n, shuffle=shuffle)

    def train_dataloader(self):
        return self._loader('train', True)

    def val_dataloader(self):
        return self._loader('val', False)

    def test_dataloader(self):
        return self._loader('test', False)

    # -- helpers ----------------------------------------------------------------------------
    def _split_val(self, train_data, train_labels, *extra):
        n_classes = len(torch.unique(torch.as_tensor(train_labels)))
        if self.val_size * len(train_data) < n_classes:
            strat = None
        elif len(train_labels.shape) > 1:
            strat = train_labels[:, 0]
        else:
            strat = train_labels
        return train_test_split(train_data, train_labels, *extra, test_size=self.val_size, stratify=strat)

    def _augment(self, data, *label_sets):
        out_d = [data]
        out_l = [[l] for l in label_sets]
        for aug in self.augmentations:
            out_d.append(aug(data))
            for o, l in zip(out_l, label_sets):
                o.append(l)
        return (torch.cat(out_d),) + tuple(torch.cat(o) for o in out_l)


class SimpleMicroDataModule(_FoldModule):
    """Single-patient k-fold CV (reference :21-208)."""

    def __init__(self, data, labels, batch_size=128, folds=20, val_size=0.2, augmentations=None, data_path=None,
                 save_folds=False):
        super().__init__(data, labels, batch_size, folds, val_size, augmentations, data_path, save_folds)

    def setup(self, stage=None):
        cv = self.select_cv(self.folds)
        for k, (tr, te) in enumerate(cv.split(self.data, self.labels if len(self.labels.shape) == 1 else self.labels[:, 0])):
            trd, ted, trl, tel = self.data[tr], self.data[te], self.labels[tr], self.labels[te]
            vd = vl = None
            if self.val_size > 0:
                trd, vd, trl, vl = self._split_val(trd, trl)
            ad, al = self._augment(trd, trl)
            self._store(k, train_data=ad, train_labels=al, val_data=vd, val_labels=vl, test_data=ted, test_labels=tel)


class AlignedMicroDataModule(_FoldModule):
    """Cross-patient pooled + aligned k-fold CV; alignment AFTER the train/val split
    (reference :211-440)."""

    align_before_split = False

    def __init__(self, data, labels, align_labels, pool_data, algner, batch_size=128, folds=20, val_size=0.2,
                 augmentations=None, data_path=None, save_folds=False, multiview=False, process_group=None):
        super().__init__(data, labels, batch_size, folds, val_size, augmentations, data_path, save_folds)
        self.align_labels, self.pool_data, self.algner = align_labels, pool_data, algner
        self.multiview = multiview
        self.process_group = process_group           # data parallel: the pooled patients are aligned one per rank

    def _align(self, X, y, y_align, pool):
        if self.multiview:
            return process_aligner_multiview(X, y, y_align, pool, self.algner)
        return process_aligner_sharded(X, y, y_align, pool, self.algner, group=self.process_group)

    def _project(self, dim_red, X):
        shp = X.shape
        z = dim_red.transform(_np(X).reshape(-1, shp[-1]))
        return torch.Tensor(z.reshape(shp[0], shp[1], -1))

    def setup(self, stage=None):
        cv = self.select_cv(self.folds)
        lab1 = self.labels.squeeze(1) if (len(self.labels.shape) > 1 and self.labels.shape[1] == 1) else self.labels
        strat = lab1 if len(lab1.shape) == 1 else lab1[:, 0]
        for k, (tr, te) in enumerate(cv.split(self.data, strat)):
            trd, ted = self.data[tr], self.data[te]
            trl, tel = self.labels[tr], self.labels[te]
            alg = self.align_labels[tr]
            vd = vl = None
            if self.align_before_split:
                trd, trl, dim_red = self._align(trd, trl, alg, self.pool_data)
                if self.val_size > 0:
                    trd, vd, trl, vl = self._split_val(trd, trl)
                ad, al = self._augment(trd, trl)
            else:
                if self.val_size > 0:
                    trd, vd, trl, vl, alg, _ = self._split_val(trd, trl, alg)
                ad, al, aa = self._augment(trd, trl, alg)
                pool = []
                for (x, y, ya) in self.pool_data:
                    px, py, pa = self._augment(x, y, ya)
                    pool.append((px, py, pa))
                ad, al, dim_red = self._align(ad, al, aa, pool)
                if vd is not None:
                    vd = self._project(dim_red, vd)
            ted = self._project(dim_red, ted)
            self._store(k, train_data=ad, train_labels=al, val_data=vd, val_labels=vl, test_data=ted, test_labels=tel)


class AlignedMicroValDataModule(AlignedMicroDataModule):
    """Variant that aligns BEFORE the train/val split so validation data is aligned too
    (reference :442-512; the one scripts/train_seq2seq.py uses, :111-113)."""

    align_before_split = True
