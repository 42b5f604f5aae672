"""Deterministic model weights from numpy's PCG64 (stable across numpy versions), so
golden fixtures need not store megabytes of parameters.  Used by the fixture
generator AND by the tests; it is data generation, not reference code."""
import numpy as np
import torch


def weights_from_seed(state_dict, seed):
    """Return a new state dict with the same keys/shapes/dtypes, values drawn from
    default_rng(seed) in key order."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in state_dict.items():
        shape = tuple(v.shape)
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
        elif k.endswith('running_var') or k.endswith('bn.weight'):
            out[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
        elif k.endswith('running_mean'):
            out[k] = torch.from_numpy(rng.uniform(-0.2, 0.2, shape).astype(np.float32))
        elif len(shape) >= 2:
            fan = int(np.prod(shape[1:]))
            s = 1.0 / np.sqrt(fan)
            if 'embedding' in k:
                s = 0.5
            out[k] = torch.from_numpy(rng.uniform(-s, s, shape).astype(np.float32))
        else:
            out[k] = torch.from_numpy(rng.uniform(-0.1, 0.1, shape).astype(np.float32))
    return out
