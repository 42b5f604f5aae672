"""Seeded synthetic ECoG: a shared smooth latent per condition, a random
mixing matrix per patient, additive noise.  The reference ships no data and no generator; this
is the build's stated workload definition, used by bench.py and the tests."""
import numpy as np


def condition_sequences(n_cond, n_classes=9, seq_len=3, seed=999):
    rng = np.random.default_rng(seed)
    seen, out = set(), []
    while len(out) < n_cond:
        s = tuple(int(v) for v in rng.integers(1, n_classes + 1, seq_len))
        if s not in seen:
            seen.add(s)
            out.append(s)
    return np.array(out, dtype=np.int64)


def shared_latents(n_cond, T, k=16, seed=998):
    rng = np.random.default_rng(seed)
    z = np.cumsum(rng.standard_normal((n_cond, T, k)), axis=1)
    return ((z - z.mean(axis=1, keepdims=True)) / z.std(axis=1, keepdims=True)).astype(np.float32)


def make_patient(p, n_trials, T=200, C=128, n_cond=64, k=16, noise=0.5):
    """Returns X (n_trials, T, C) float32 and y_full (n_trials, 3) int64 in 1..9.
    Every condition occurs at least n_trials // (2 * n_cond) times."""
    rng = np.random.default_rng(1000 + p)
    seqs = condition_sequences(n_cond)
    Z = shared_latents(n_cond, T, k)
    base = np.repeat(np.arange(n_cond), max(1, n_trials // (2 * n_cond)))[:n_trials]
    cond = np.concatenate([base, rng.integers(0, n_cond, n_trials - len(base))])
    rng.shuffle(cond)
    A = (rng.standard_normal((k, C)) * 0.5).astype(np.float32)
    X = Z[cond] @ A + noise * rng.standard_normal((n_trials, T, C), dtype=np.float32)
    return X.astype(np.float32), seqs[cond]
