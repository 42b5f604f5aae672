"""Generate tests/golden/data_interface.npz and augmentations.npz by running the REFERENCE's own code:
alignment/alignment_utils.py (decoding_data_from_dict / get_features_labels / phon_to_artic_seq, :127-215) and
nn_models/data_utils/augmentations.py (:13-90).

Run in the build container only (it imports /root/reference, which does not exist on the GPU box):
    python tests/golden/make_data_fixtures.py
The reference's augmentations import torchvision (absent here: an ordinary ModuleNotFoundError).  torchvision 0.22.1
(environment.yml:219) implements ``Resize(size)`` of a tensor as
``torch.nn.functional.interpolate(img, size, mode='bilinear', align_corners=False, antialias=True)`` over the last two
dimensions; the in-process stand-in below is exactly that call.  Every recorded number comes from the reference's own
functions; inputs, the RNG seeds and outputs are stored.
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/aligned_decoding'
HERE = os.path.dirname(os.path.abspath(__file__))


def install_torchvision_standin():
    tv = types.ModuleType('torchvision')
    tr = types.ModuleType('torchvision.transforms')

    class Resize:
        def __init__(self, size):
            self.size = [int(s) for s in size]

        def __call__(self, img):
            x = img.unsqueeze(0) if img.dim() == 3 else img           # (1, C, H, W): torchvision adds the batch dimension
            y = torch.nn.functional.interpolate(x, size=self.size, mode='bilinear', align_corners=False, antialias=True)
            return y.squeeze(0) if img.dim() == 3 else y
    tr.Resize = Resize
    tv.transforms = tr
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tr


def synthetic_dict(rng):
    """Dictionary with the reference's keys (alignment_utils.py:127-184): X{p}, y{p}, X_collapsed, y_phon_collapsed,
    y_artic_collapsed, y_full_phon, pre_pts."""
    d = {}
    for name, (n, c) in {'S1': (11, 5), 'S2': (9, 7), 'S3': (13, 4)}.items():
        pt = {'y_full_phon': rng.integers(1, 10, (n, 3))}
        for p in (1, 2, 3):
            pt[f'X{p}'] = rng.standard_normal((n, 20, c)).astype(np.float32)
            pt[f'y{p}'] = pt['y_full_phon'][:, p - 1].copy()
        pt['X_collapsed'] = np.concatenate([pt['X1'], pt['X2'], pt['X3']])
        pt['y_phon_collapsed'] = np.concatenate([pt['y1'], pt['y2'], pt['y3']])
        pt['y_artic_collapsed'] = pt['y_phon_collapsed'] % 4 + 1
        pt['pre_pts'] = [q for q in ('S1', 'S2', 'S3') if q != name]
        d[name] = pt
    return d


def main():
    sys.path.insert(0, REF)
    from alignment import alignment_utils as ref_utils
    out = {'versions': np.array([f'numpy {np.__version__}', f'torch {torch.__version__}'])}
    rng = np.random.default_rng(11)
    d = synthetic_dict(rng)
    for name, pt in d.items():
        for k, v in pt.items():
            if k != 'pre_pts':
                out[f'in/{name}/{k}'] = v
        out[f'in/{name}/pre_pts'] = np.array(pt['pre_pts'])
    for p_ind in (1, 3, -1):
        for lab_type in ('phon', 'artic'):
            (D, lab, lab_full), pre = ref_utils.decoding_data_from_dict(d, 'S2', p_ind, lab_type, 'phon_seq')
            tag = f'out/S2/p{p_ind}/{lab_type}'
            out[f'{tag}/D'], out[f'{tag}/lab'], out[f'{tag}/lab_full'] = D, lab, lab_full
            for i, (Dp, lp, lfp) in enumerate(pre):
                out[f'{tag}/pre{i}/D'], out[f'{tag}/pre{i}/lab'], out[f'{tag}/pre{i}/lab_full'] = Dp, lp, lfp
    seq = rng.integers(1, 10, (6, 3))
    out['artic/in'], out['artic/out'] = seq, ref_utils.phon_to_artic_seq(seq)
    np.savez_compressed(os.path.join(HERE, 'data_interface.npz'), **out)

    # ---- augmentations (reference module with the torchvision stand-in) ----
    install_torchvision_standin()
    sys.path.insert(0, os.path.join(REF, 'nn_models', 'data_utils'))
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_augmentations', os.path.join(REF, 'nn_models', 'data_utils', 'augmentations.py'))
    aug = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(aug)
    a = {'versions': out['versions']}
    x = torch.from_numpy(np.random.default_rng(5).standard_normal((7, 40, 6)).astype(np.float32))
    a['x'] = x.numpy()
    cases = [('time_shifting', {}), ('time_shifting', {'shift_max': 7}), ('time_masking', {}), ('time_masking', {'mask_ratio': 0.3}),
             ('scaling', {}), ('scaling', {'scale_range': (0.5, 2.0)}), ('noise_jitter', {}), ('noise_jitter', {'noise_level': 0.2}),
             ('time_warping', {}), ('time_warping', {'factor_range': (1.05, 1.2)}), ('time_warping', {'factor_range': (0.8, 0.95)})]
    names = []
    for i, (fn, kw) in enumerate(cases):
        seed = 100 + i
        np.random.seed(seed)
        torch.manual_seed(seed)
        y = getattr(aug, fn)(x.clone(), **kw)
        a[f'case{i}/out'] = y.numpy()
        a[f'case{i}/seed'] = np.array(seed)
        names.append(f'{fn}|{repr(kw)}')
    a['cases'] = np.array(names)
    # the three augmentations of scripts/train_seq2seq.py:111-113 chained in one seeded stream, as DataModule.setup() applies them
    np.random.seed(7)
    torch.manual_seed(7)
    a['chain/out'] = np.stack([aug.time_shifting(x.clone()).numpy(), aug.noise_jitter(x.clone()).numpy(), aug.scaling(x.clone()).numpy()])
    np.savez_compressed(os.path.join(HERE, 'augmentations.npz'), **a)
    print('wrote data_interface.npz, augmentations.npz')


if __name__ == '__main__':
    main()
