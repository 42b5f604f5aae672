"""Alignment utilities — drop-in surface of the reference's ``alignment/alignment_utils.py``
(extract_group_conditions :12, cnd_avg :42, label2str :64, label_seq2str :83, save/load_pkl
:102/:113, decoding_data_from_dict :127, get_features_labels :160, phon_to_artic_seq :187).

``cnd_avg`` runs on the MI355X (segmented mean kernel, xps_cnd_avg_*): trials are summed in trial
order in the input dtype and stored as float64, which reproduces the reference's
``np.mean(data[labels == seq], axis=0)`` into a float64 array bit for bit.
"""
import pickle
from functools import reduce

import numpy as np

from . import _linalg as LA


def label_seq2str(labels):
    """(n_trials, L) label sequences -> (n_trials,) strings, e.g. [1, 2, 3] -> '123'."""
    labels = np.asarray(labels)
    if labels.ndim == 2 and labels.dtype.kind in 'iub' and labels.shape[0] > 64:
        # the same strings, built once per DISTINCT sequence (a patient has tens of conditions and thousands of trials: the
        # per-trial Python join was 3-10 ms per view, more than the device work of an 8-view MCCA fit)
        lo, hi = int(labels.min()), int(labels.max())
        if lo >= 0 and (hi + 1) ** labels.shape[1] < 2 ** 62:       # one integer key per row: a 1-D unique instead of a row sort
            key = np.zeros(labels.shape[0], dtype=np.int64)
            for j in range(labels.shape[1]):
                key = key * (hi + 1) + labels[:, j]
            _, first, inv = np.unique(key, return_index=True, return_inverse=True)
            uniq = labels[first]
        else:
            uniq, inv = np.unique(labels, axis=0, return_inverse=True)
        return np.array([''.join(str(v) for v in row) for row in uniq])[np.asarray(inv).reshape(-1)]
    return np.array([''.join(str(v) for v in row) for row in labels])


def label2str(labels):
    """1-D labels -> astype(str); 2-D label sequences -> label_seq2str.  The string form matters:
    conditions are ordered lexicographically ('10' < '2'), as in the reference."""
    if hasattr(labels, 'detach'):
        labels = labels.detach().cpu().numpy()
    labels = np.asarray(labels)
    if len(labels.shape) > 1:
        return label_seq2str(labels)
    return labels.astype(str)


def _cnd_avg_device(data, labels):
    """-> (sorted unique labels, float64 device tensor (n_cond, ...))."""
    uniq, order, start = LA.condition_index(labels)
    return uniq, LA.cnd_avg_device(LA.to_device(data), order, start)


def cnd_avg(data, labels):
    """Mean over the trials of each condition (conditions in sorted-label order):
    (n_trials, ...) -> float64 ndarray (n_conditions, ...)."""
    return _cnd_avg_device(data, np.asarray(labels))[1].cpu().numpy()


def _group_conditions_device(Xs, ys, own=None):
    """own[i] False: view i belongs to another rank (patient-sharded fits): its data is not touched and its entry is None;
    the labels of EVERY view still decide the shared conditions."""
    keys = [label2str(y) for y in ys]
    own = [True] * len(Xs) if own is None else own
    avgs = [_cnd_avg_device(x, k) if o else None for x, k, o in zip(Xs, keys, own)]
    shared = reduce(np.intersect1d, keys)
    out = []
    for k, av in zip(keys, avgs):
        if av is None:
            out.append(None)
            continue
        uniq, a = av
        keep = np.flatnonzero(np.isin(uniq, shared, assume_unique=True))
        out.append(a[LA.torch.from_numpy(keep).to(a.device)])
    return out


def extract_group_conditions(Xs, ys):
    """Condition averages of every dataset, restricted to the conditions present in ALL of them."""
    return [a.cpu().numpy() for a in _group_conditions_device(Xs, ys)]


def save_pkl(data, filename):
    with open(filename, 'wb+') as f:
        pickle.dump(data, f, protocol=-1)


def load_pkl(filename):
    with open(filename, 'rb') as f:
        return pickle.load(f)


def decoding_data_from_dict(data_dict, pt, p_ind, lab_type='phon', algn_type='phon_seq'):
    """((D_tar, lab_tar, lab_tar_full), [(D, lab, lab_full) per pre-training patient])."""
    tar = get_features_labels(data_dict[pt], p_ind, lab_type, algn_type)
    pre = [get_features_labels(data_dict[p], p_ind, lab_type, algn_type) for p in data_dict[pt]['pre_pts']]
    return tar, pre


def get_features_labels(data, p_ind, lab_type, algn_type):
    lab_full = data['y_full_' + algn_type[:-4]]
    if p_ind == -1:
        D = data['X_collapsed']
        lab = data['y_' + lab_type + '_collapsed']
        lab_full = np.tile(lab_full, (3, 1))
    else:
        D = data['X' + str(p_ind)]
        lab = data['y' + str(p_ind)]
    if lab_type == 'artic':
        lab = phon_to_artic_seq(lab)
    return D, lab, lab_full


_PHON_TO_ARTIC = {1: 1, 2: 1, 3: 2, 4: 2, 5: 3, 6: 3, 7: 3, 8: 4, 9: 4}


def phon_to_artic(phon_idx, phon_to_artic_conv):
    return phon_to_artic_conv[phon_idx]


def phon_to_artic_seq(phon_seq):
    flat = np.asarray(phon_seq).flatten()
    return np.array([phon_to_artic(int(p), _PHON_TO_ARTIC) for p in flat]).reshape(np.shape(phon_seq))
