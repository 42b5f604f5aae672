"""Whole-batch augmentations — same names, arguments and random-draw sources (numpy global RNG / torch global RNG, one draw
per call, in the same order) as the reference's nn_models/data_utils/augmentations.py (:13,32,51,65,79), computed by the HIP
kernels of csrc/xps_augment.hip.  A tensor that already lives on the GPU stays there (augmentation per epoch on the device);
a host tensor (what DataModule.setup() holds, as in the reference) is uploaded, augmented and returned on the host.  There is
no CPU fallback: without the GPU these raise (deliberately: the product path never computes on the host).

Precision: the kernels compute in float32 -- the dtype of every tensor the DataModules hold (the reference's `torch.Tensor(...)`
casts, datamodules.py:574).  A float64 input is rounded to float32, augmented and cast back (the reference would keep float64
arithmetic there); the goldens in tests/golden/augmentations.npz are float32."""
import numpy as np
import torch

from ..._lib import call
from ..functional import _need_gpu, _ptr, _stream


def _on_device(data):
    if not torch.is_tensor(data):
        data = torch.as_tensor(data)
    if data.dim() != 3:
        raise ValueError('augmentations expect (n_trials, n_timepoints, n_features) tensors')
    host = not data.is_cuda
    if host and not torch.cuda.is_available():
        raise RuntimeError('cross_patient_speech_decoding_amd: the augmentation kernels need the MI355X (no CPU fallback)')
    x = data.to('cuda', dtype=torch.float32).contiguous()
    _need_gpu(x)
    return x, host, data.dtype


def _back(out, host, dtype):
    out = out if dtype == torch.float32 else out.to(dtype)
    return out.cpu() if host else out


def time_warping(data, factor_range=(0.8, 1.2)):
    """Linear warp of the time axis by a random factor (scipy.ndimage.zoom, order 1), then torchvision's Resize back to the
    original length (bilinear, antialias), in one kernel."""
    factor = np.random.uniform(*factor_range)
    x, host, dt = _on_device(data)
    N, T, C = x.shape
    T2 = int(round(T * factor))                       # scipy.ndimage.zoom's output length
    out = torch.empty_like(x)
    call('xps_aug_time_warp_f32', _ptr(x), _ptr(out), N, T, C, max(T2, 1), _stream())
    return _back(out, host, dt)


def time_masking(data, mask_ratio=0.1):
    n_time = data.size(1)
    mask_size = int(n_time * mask_ratio)
    mask_start = np.random.randint(0, n_time - mask_size)
    x, host, dt = _on_device(data)
    N, T, C = x.shape
    out = torch.empty_like(x)
    call('xps_aug_time_mask_f32', _ptr(x), _ptr(out), N, T, C, int(mask_start), int(mask_size), _stream())
    return _back(out, host, dt)


def time_shifting(data, shift_max=20):
    shift = np.random.randint(-shift_max, shift_max)
    x, host, dt = _on_device(data)
    N, T, C = x.shape
    out = torch.empty_like(x)
    call('xps_aug_time_shift_f32', _ptr(x), _ptr(out), N, T, C, int(shift), _stream())
    return _back(out, host, dt)


def noise_jitter(data, noise_level=0.01):
    # the N(0, 1) draw comes from the generator of the device the data lives on, like the reference's torch.randn_like
    noise = torch.randn_like(data if torch.is_tensor(data) else torch.as_tensor(data))
    x, host, dt = _on_device(data)
    nz = noise.to('cuda', dtype=torch.float32).contiguous()
    out = torch.empty_like(x)
    call('xps_aug_jitter_f32', _ptr(x), _ptr(nz), _ptr(out), x.numel(), float(noise_level), _stream())
    return _back(out, host, dt)


def scaling(data, scale_range=(0.9, 1.1)):
    scale = np.random.uniform(*scale_range)
    x, host, dt = _on_device(data)
    out = torch.empty_like(x)
    call('xps_aug_scale_f32', _ptr(x), _ptr(out), x.numel(), float(np.float32(scale)), _stream())
    return _back(out, host, dt)
