"""SURVEY 8f rank 1, the data interface: the dict schema of ``decoding_data_from_dict`` / ``get_features_labels`` /
``phon_to_artic_seq`` (reference alignment/alignment_utils.py:127-215) against golden vectors produced by the reference's
own functions (tests/golden/make_data_fixtures.py), the pickle round trip, and the fold cache (dataset names of
nn_models/data_utils/datamodules.py:506-512) written and read back."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data_interface.npz')


def _dict_from_fixture(g):
    d = {}
    for key in g.files:
        if not key.startswith('in/'):
            continue
        _, pt, name = key.split('/')
        d.setdefault(pt, {})[name] = list(g[key]) if name == 'pre_pts' else g[key]
    return d


@pytest.mark.parametrize('p_ind', [1, 3, -1])
@pytest.mark.parametrize('lab_type', ['phon', 'artic'])
def test_decoding_data_from_dict_matches_reference(p_ind, lab_type):
    from cross_patient_speech_decoding_amd.alignment.alignment_utils import decoding_data_from_dict
    g = np.load(GOLD)
    d = _dict_from_fixture(g)
    (D, lab, lab_full), pre = decoding_data_from_dict(d, 'S2', p_ind, lab_type, 'phon_seq')
    tag = f'out/S2/p{p_ind}/{lab_type}'
    np.testing.assert_array_equal(D, g[f'{tag}/D'])
    np.testing.assert_array_equal(lab, g[f'{tag}/lab'])
    np.testing.assert_array_equal(lab_full, g[f'{tag}/lab_full'])
    assert len(pre) == 2                                         # pre_pts of S2: S1, S3 in the stored order
    for i, (Dp, lp, lfp) in enumerate(pre):
        np.testing.assert_array_equal(Dp, g[f'{tag}/pre{i}/D'])
        np.testing.assert_array_equal(lp, g[f'{tag}/pre{i}/lab'])
        np.testing.assert_array_equal(lfp, g[f'{tag}/pre{i}/lab_full'])


def test_phon_to_artic_seq_and_missing_key():
    from cross_patient_speech_decoding_amd.alignment.alignment_utils import decoding_data_from_dict, phon_to_artic_seq
    g = np.load(GOLD)
    out = phon_to_artic_seq(g['artic/in'])
    np.testing.assert_array_equal(out, g['artic/out'])
    assert out.shape == g['artic/in'].shape
    d = _dict_from_fixture(g)
    with pytest.raises(KeyError):                                # same failure mode as the reference: plain dict lookups
        decoding_data_from_dict(d, 'S9', 1)
    with pytest.raises(KeyError):
        decoding_data_from_dict(d, 'S2', 4)


def test_pkl_round_trip(tmp_path):
    from cross_patient_speech_decoding_amd.alignment.alignment_utils import decoding_data_from_dict, load_pkl, save_pkl
    g = np.load(GOLD)
    d = _dict_from_fixture(g)
    path = tmp_path / 'pt_decoding_data.pkl'
    save_pkl(d, path)
    back = load_pkl(path)
    assert sorted(back) == sorted(d)
    (D, lab, lab_full), pre = decoding_data_from_dict(back, 'S2', 1)
    np.testing.assert_array_equal(D, g['out/S2/p1/phon/D'])
    np.testing.assert_array_equal(pre[1][2], g['out/S2/p1/phon/pre1/lab_full'])


def test_fold_cache_round_trip(tmp_path):
    """``save_folds=True`` writes fold_data/fold_{k}.npz with the reference's six dataset names; a fresh module reads them
    back and serves identical loaders (no augmentation, no alignment: host-side only)."""
    from cross_patient_speech_decoding_amd.nn_models.data_utils.datamodules import SimpleMicroDataModule
    rng = np.random.default_rng(0)
    X = torch.from_numpy(rng.standard_normal((60, 12, 5)).astype(np.float32))
    y = torch.from_numpy(np.repeat(np.arange(3), 20))
    torch.manual_seed(0)
    np.random.seed(0)
    dm = SimpleMicroDataModule(X, y, batch_size=-1, folds=3, val_size=0.25, data_path=tmp_path, save_folds=True)
    dm.setup()
    for k in range(3):
        with np.load(tmp_path / 'fold_data' / f'fold_{k}.npz') as f:
            assert sorted(f.files) == sorted(SimpleMicroDataModule.FOLD_KEYS)
            assert f['train_data'].shape[1:] == (12, 5) and f['train_data'].dtype == np.float32
            n = len(f['train_labels']) + len(f['val_labels']) + len(f['test_labels'])
            assert n == 60                                       # the three splits partition the trials
    dm2 = SimpleMicroDataModule(X, y, batch_size=-1, folds=3, val_size=0.25, data_path=tmp_path).load_folds()
    for k in range(3):
        dm.set_fold(k); dm2.set_fold(k)
        assert dm.get_data_shape() == dm2.get_data_shape()
        for which in ('val', 'test'):
            (a, la), = list(getattr(dm, f'{which}_dataloader')())
            (b, lb), = list(getattr(dm2, f'{which}_dataloader')())
            assert torch.equal(a, b) and torch.equal(la, lb)
        a, la = dm._folds[k]['train_data'], dm._folds[k]['train_labels']
        b, lb = dm2._folds[k]['train_data'], dm2._folds[k]['train_labels']
        assert torch.equal(a, b) and torch.equal(la, lb)
    with pytest.raises(AssertionError):
        dm2.set_fold(3)
