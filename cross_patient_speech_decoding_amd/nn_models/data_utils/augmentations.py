"""Whole-batch augmentations — same names, arguments and random-draw sources (numpy global RNG /
torch global RNG, one draw per call) as the reference's nn_models/data_utils/augmentations.py
(:13,32,51,65,79).  They run once per fold in DataModule.setup(), as in the reference."""
import numpy as np
import torch


def time_warping(data, factor_range=(0.8, 1.2)):
    """Linear temporal warp by a random factor, then linear resize back to the original length
    (the reference uses scipy.ndimage.zoom(order=1) + torchvision Resize; torchvision is not in the
    image, so both resamplings are torch linear interpolations along time)."""
    factor = np.random.uniform(*factor_range)
    x = data.permute(0, 2, 1)                                           # (N, C, T)
    T = x.shape[-1]
    warped = torch.nn.functional.interpolate(x, size=max(2, int(round(T * factor))), mode='linear',
                                             align_corners=True)
    back = torch.nn.functional.interpolate(warped, size=T, mode='linear', align_corners=False)
    return back.permute(0, 2, 1).contiguous()


def time_masking(data, mask_ratio=0.1):
    n_time = data.size(1)
    mask_size = int(n_time * mask_ratio)
    mask_start = np.random.randint(0, n_time - mask_size)
    out = data.clone()
    out[:, mask_start:mask_start + mask_size, :] = 0
    return out


def time_shifting(data, shift_max=20):
    shift = np.random.randint(-shift_max, shift_max)
    return torch.roll(data, shifts=shift, dims=1)


def noise_jitter(data, noise_level=0.01):
    return data + torch.randn_like(data) * noise_level


def scaling(data, scale_range=(0.9, 1.1)):
    return data * np.random.uniform(*scale_range)
