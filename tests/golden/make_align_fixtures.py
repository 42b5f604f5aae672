"""Generate tests/golden/align_*.npz by running the REFERENCE's own alignment code.

Run in the build container only (it imports /root/reference, which does not exist on
the GPU box):   python tests/golden/make_align_fixtures.py
Every fixture stores inputs AND the reference's outputs; tests load the .npz only.
Library versions are recorded in each file (reference pins numpy 2.2.5 / scipy 1.15.3 /
scikit-learn 1.6.1, environment.yml:106,154,155).
"""
import os
import sys

import numpy as np

REF = '/root/reference/aligned_decoding'
sys.path.insert(0, REF)
from alignment.AlignCCA import AlignCCA, CCA_align                       # noqa: E402
from alignment.JointPCA import JointPCA                                  # noqa: E402
from alignment.alignment_utils import (cnd_avg, extract_group_conditions,   # noqa: E402
                                       label2str)

HERE = os.path.dirname(os.path.abspath(__file__))


def versions():
    import scipy
    import sklearn
    return np.array([f'numpy {np.__version__}', f'scipy {scipy.__version__}',
                     f'sklearn {sklearn.__version__}'])


def smooth_latents(rng, n_cond, T, k):
    z = np.cumsum(rng.standard_normal((n_cond, T, k)), axis=1)
    return (z - z.mean(axis=1, keepdims=True)) / z.std(axis=1, keepdims=True)


def make_view(rng, Z, seqs, n_trials, C, noise=0.5, dtype=np.float64, drop=None):
    """Trials of a synthetic patient: X[n] = Z[cond(n)] @ A + noise; labels = seqs[cond]."""
    k = Z.shape[-1]
    A = rng.standard_normal((k, C)) * 0.5
    cond = rng.integers(0, len(seqs), n_trials)
    if drop is not None:
        cond[cond == drop] = (drop + 1) % len(seqs)
    cond[:len(seqs)] = np.arange(len(seqs))
    if drop is not None:
        cond[drop] = (drop + 1) % len(seqs)
    X = Z[cond] @ A + noise * rng.standard_normal((n_trials, Z.shape[1], C))
    return X.astype(dtype), seqs[cond]


def fx_labels():
    rng = np.random.default_rng(11)
    y1 = np.array([2, 10, 11, 3, 10, 2, 1, 11, 3, 3, 21, 1])      # multi-digit: string order != numeric
    y3 = rng.integers(1, 12, (40, 3))
    X1 = rng.standard_normal((12, 5, 4))
    X3 = rng.standard_normal((40, 6, 3)).astype(np.float32)
    out = dict(versions=versions(), y1=y1, y3=y3, X1=X1, X3=X3,
               s1=label2str(y1), s3=label2str(y3),
               avg1=cnd_avg(X1, label2str(y1)), avg3=cnd_avg(X3, label2str(y3)))
    # group conditions: view 1 lacks one condition of view 0, view 2 has an extra one
    seqs = np.array([[1, 2, 3], [2, 2, 1], [3, 1, 1], [1, 10, 2], [4, 4, 4]])
    Z = smooth_latents(rng, 5, 7, 3)
    Xa, ya = make_view(rng, Z, seqs, 30, 6)
    Xb, yb = make_view(rng, Z, seqs, 26, 4, drop=2)
    Xc, yc = make_view(rng, Z[:4], seqs[:4], 22, 5, dtype=np.float32)
    g = extract_group_conditions([Xa, Xb, Xc], [ya, yb, yc])
    out.update(gXa=Xa, gya=ya, gXb=Xb, gyb=yb, gXc=Xc, gyc=yc, g0=g[0], g1=g[1], g2=g[2])
    np.savez_compressed(os.path.join(HERE, 'align_labels.npz'), **out)


def fx_cca():
    rng = np.random.default_rng(12)
    out = dict(versions=versions())
    seqs = np.array([[a, b, c] for a in (1, 2, 3) for b in (1, 2) for c in (1, 2, 3)])   # 18 conditions
    Z = smooth_latents(rng, len(seqs), 10, 5)
    cases = {
        'full': dict(Ca=8, Cb=8, deficient=False),
        'uneq': dict(Ca=9, Cb=6, deficient=False),
        'rdef': dict(Ca=8, Cb=7, deficient=True),
    }
    for name, c in cases.items():
        Xa, ya = make_view(rng, Z, seqs, 48, c["Ca"])
        Xb, yb = make_view(rng, Z, seqs, 40, c["Cb"], drop=4)
        if c['deficient']:                      # channel 6 of B = exact copy-combination: rank 6 of 7
            Xb[..., 6] = Xb[..., 0] - 2.0 * Xb[..., 3]
        out.update({f'{name}_Xa': Xa, f'{name}_ya': ya, f'{name}_Xb': Xb, f'{name}_yb': yb})
        for space in ('b_to_a', 'a_to_b', 'shared'):
            al = AlignCCA(return_space=space)
            al.fit(Xa, Xb, ya, yb)
            if space == 'shared':
                ta, tb = al.transform([Xa, Xb])
                out[f'{name}_{space}_ta'], out[f'{name}_{space}_tb'] = ta, tb
            else:
                out[f'{name}_{space}_t'] = al.transform(Xb if space == 'b_to_a' else Xa)
        out.update({f'{name}_M_a': al.M_a, f'{name}_M_b': al.M_b, f'{name}_S': al.canon_corrs})
    # module-level CCA_align on raw (d, n) inputs (mutates its inputs: pass copies)
    La, Lb = rng.standard_normal((5, 90)), rng.standard_normal((4, 90))
    Lb[3] = 0.5 * La[1] + 0.1 * rng.standard_normal(90)
    Ma, Mb, S = CCA_align(La.copy(), Lb.copy())
    out.update(raw_La=La, raw_Lb=Lb, raw_Ma=Ma, raw_Mb=Mb, raw_S=S)
    # ---- round 2 (appended: the draws above are unchanged) --------------------------------------------------------------
    # ill-conditioned but FULL-RANK views (channel scales 10^0 .. 10^-7 / 10^-9): LAPACK's rank rule on singular values
    # (AlignCCA.py:263-264) keeps every channel; a Gram-eigenvalue rule would not.  The reference's answer is well defined here
    # (uncertainty ~ eps * cond).
    for name, span in (('illc7', 7.0), ('illc9', 9.0)):
        Xa, ya = make_view(rng, Z, seqs, 48, 8)
        Xb, yb = make_view(rng, Z, seqs, 40, 7, drop=4)
        Xa = Xa * 10.0 ** (-span * np.arange(8) / 7.0)
        Xb = Xb * 10.0 ** (-span * np.arange(7) / 6.0)
        out.update({f'{name}_Xa': Xa, f'{name}_ya': ya, f'{name}_Xb': Xb, f'{name}_yb': yb})
        for space in ('b_to_a', 'a_to_b'):
            al = AlignCCA(return_space=space)
            al.fit(Xa, Xb, ya, yb)
            out[f'{name}_{space}_t'] = al.transform(Xb if space == 'b_to_a' else Xa)
        out.update({f'{name}_M_a': al.M_a, f'{name}_M_b': al.M_b, f'{name}_S': al.canon_corrs})
    # The rank-deficient case is NOT a function of its input in the reference: Householder QR completes Q with a direction made of
    # rounding noise.  Evidence kept with the fixture: the reference re-run on the 'rdef' input perturbed by RELATIVE noise of
    # 1e-16 (below one ulp of most entries) -- its own canonical correlations / transforms move by the amounts stored here.
    Xa, ya, Xb, yb = out['rdef_Xa'], out['rdef_ya'], out['rdef_Xb'], out['rdef_yb']
    prng = np.random.default_rng(99)
    dS, dT = [], []
    for _ in range(8):
        Xa2 = Xa * (1.0 + 1e-16 * prng.standard_normal(Xa.shape))
        Xb2 = Xb * (1.0 + 1e-16 * prng.standard_normal(Xb.shape))
        al = AlignCCA(return_space='b_to_a')
        al.fit(Xa2, Xb2, ya, yb)
        if al.canon_corrs.shape != out['rdef_S'].shape:
            continue
        dS.append(np.abs(al.canon_corrs - out['rdef_S']).max())
        t = al.transform(Xb)
        dT.append(np.abs(t - out['rdef_b_to_a_t']).max() / np.abs(out['rdef_b_to_a_t']).max())
    out['rdef_ref_selfdiff_S'] = np.array(dS)
    out['rdef_ref_selfdiff_T'] = np.array(dT)
    np.savez_compressed(os.path.join(HERE, 'align_cca.npz'), **out)


def fx_jointpca():
    rng = np.random.default_rng(13)
    seqs = np.array([[a, b, 1] for a in (1, 2, 3, 4) for b in (1, 2, 3)])               # 12 conditions
    Z = smooth_latents(rng, len(seqs), 15, 4)
    views = [make_view(rng, Z, seqs, n, C) for n, C in ((40, 7), (36, 5), (44, 6))]
    Xs, ys = [v[0] for v in views], [v[1] for v in views]
    jp = JointPCA(n_components=4)           # 180 x 18 matrix -> sklearn 'full' solver (deterministic)
    t = jp.fit_transform(Xs, ys)
    out = dict(versions=versions())
    for i in range(3):
        out.update({f'X{i}': Xs[i], f'y{i}': ys[i], f'W{i}': jp.transforms[i], f't{i}': t[i]})
    out['t1_single'] = jp.transform(Xs[1], idx=1)
    np.savez_compressed(os.path.join(HERE, 'align_jointpca.npz'), **out)


if __name__ == '__main__':
    fx_labels()
    fx_cca()
    fx_jointpca()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))
