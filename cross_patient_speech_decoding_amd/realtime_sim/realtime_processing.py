"""Per-bin high-gamma feature extraction on the MI355X — counterpart of the reference's
realtime_sim/realtime_processing.py (process_HG :10, CAR :42, filter_HG_bin :60, FIR_filter_HG_bin :86,
IIR_filter_HG_bin :106, compute_bin_power :146): same names, arguments and return values (numpy float64).

Every function is one launch of the fused HIP kernel (xps_process_hg_f64: common average reference -> band-pass filters
with carried state -> RMS); ``process_HG`` runs all three stages in a single launch.  The IIR path reproduces
scipy.signal.lfilter's direct-form-II-transposed arithmetic bit for bit (no fused multiply-add), the CAR and RMS stages
numpy's summation order; the FIR path (scipy evaluates it with np.convolve / BLAS dot products, whose summation order
is not defined) agrees to rounding.  There is no CPU fallback."""

import numpy as np
import torch

from .._lib import call, lib

_F64 = torch.float64


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('realtime_processing needs the MI355X: the HIP path has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _up(x):
    return None if x is None else torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64)).to(_dev())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def lfilter_zi(b, a):
    """scipy.signal.lfilter_zi: steady-state filter state of a unit step (companion-matrix solve; float64)."""
    b, a = np.atleast_1d(np.asarray(b, dtype=np.float64)), np.atleast_1d(np.asarray(a, dtype=np.float64))
    while len(a) > 1 and a[0] == 0.0:
        a = a[1:]
    if a[0] != 1.0:
        b, a = b / a[0], a / a[0]
    n = max(len(a), len(b))
    a = np.r_[a, np.zeros(n - len(a))]
    b = np.r_[b, np.zeros(n - len(b))]
    comp = np.zeros((n - 1, n - 1))
    comp[0, :] = -a[1:] / a[0]
    comp[np.arange(1, n - 1), np.arange(0, n - 2)] = 1.0
    return np.linalg.solve(np.eye(n - 1) - comp.T, b[1:] - a[1:] * b[0])


def _run(data, b=None, a=None, zi=None, good=None, do_car=True, want='power'):
    """One launch.  ``want``: 'car' | 'filtered' | 'power'.  Returns (result ndarray, updated zi ndarray or None)."""
    data = np.asarray(data, dtype=np.float64)
    Cn, Tn = data.shape
    dev = _dev()
    d = _up(data)
    bands = 0 if b is None else b.shape[0]
    taps = 1 if b is None else b.shape[1]
    bd, ad = _up(b), _up(a)
    zd = _up(zi)
    gd = None if good is None else torch.as_tensor(np.ascontiguousarray(good, dtype=np.uint8)).to(dev)
    car = torch.empty(Cn, Tn, dtype=_F64, device=dev) if want == 'car' else None
    filt = torch.empty(Cn, Tn, max(bands, 1), dtype=_F64, device=dev) if want == 'filtered' else None
    power = torch.empty(Cn, dtype=_F64, device=dev) if want == 'power' else None
    nbytes = lib().xps_process_hg_f64_workspace(Cn, Tn, max(bands, 1))
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()
    call('xps_process_hg_f64', ptr(d), Cn, Tn, ptr(gd), ptr(bd), ptr(ad), bands, taps, ptr(zd), int(do_car), ptr(car),
         ptr(filt), ptr(power), ptr(ws), nbytes, _stream())
    out = car if want == 'car' else (filt if want == 'filtered' else power)
    return out.cpu().numpy(), (None if zd is None else zd.cpu().numpy())


def _good_mask(n_chan, bad_channels):
    good = np.ones(n_chan, dtype=np.uint8)
    for c in (bad_channels or []):
        good[c] = 0
    return good


def _split_coefs(bandpassCoefs, n_chan, band_ics):
    """(b, a, zi) arrays for the kernel from the reference's coefficient layouts."""
    coefs = np.asarray(bandpassCoefs, dtype=np.float64)
    if coefs.ndim == 3:                              # IIR: (bands, taps, [a, b])
        a, b = np.ascontiguousarray(coefs[:, :, 0]), np.ascontiguousarray(coefs[:, :, 1])
        if band_ics is None:                         # :129-136: lfilter_zi tiled over the channels
            zi = np.stack([np.tile(lfilter_zi(bb, aa), (n_chan, 1)) for bb, aa in zip(b, a)], axis=0)
        else:
            zi = np.asarray(band_ics, dtype=np.float64)
        return b, a, zi
    if coefs.ndim == 2:                              # FIR: (bands, taps); lfilter(coefs, 1.0, data): zero state, not returned
        return np.ascontiguousarray(coefs), None, None
    raise ValueError('bandpassCoefs must be either 2D or 3D array.')


def CAR(data, bad_channels=None):
    """Common average reference (reference :42-57)."""
    data = np.asarray(data, dtype=np.float64)
    out, _ = _run(data, good=_good_mask(data.shape[0], bad_channels), do_car=True, want='car')
    return out


def IIR_filter_HG_bin(data, bandpassCoefs, zi=None):
    """(channels, time) -> ((channels, time, bands), (bands, channels, order)) (reference :106-143)."""
    data = np.asarray(data, dtype=np.float64)
    b, a, z = _split_coefs(bandpassCoefs, data.shape[0], zi)
    return _run(data, b, a, z, do_car=False, want='filtered')


def FIR_filter_HG_bin(data, bandpassCoefs):
    """(channels, time) -> ((channels, time, bands), None) (reference :86-103)."""
    data = np.asarray(data, dtype=np.float64)
    b, _, _ = _split_coefs(bandpassCoefs, data.shape[0], None)
    out, _ = _run(data, b, None, None, do_car=False, want='filtered')
    return out, None


def filter_HG_bin(data, bandpassCoefs, band_ics=None):
    """Routes to the IIR (3-D coefficients) or FIR (2-D) filter (reference :60-83)."""
    bandpassCoefs = np.asarray(bandpassCoefs)
    if bandpassCoefs.ndim == 3:
        return IIR_filter_HG_bin(data, bandpassCoefs, band_ics)
    if bandpassCoefs.ndim == 2:
        return FIR_filter_HG_bin(data, bandpassCoefs)
    raise ValueError('bandpassCoefs must be either 2D or 3D array.')


def compute_bin_power(data):
    """RMS over (time, bands) per channel of a (channels, time, bands) array (reference :146-164)."""
    data = np.asarray(data, dtype=np.float64)
    Cn, Tn, nb = data.shape
    # the kernel's identity "filter" (one tap, b = 1) over the flattened (time*bands) samples as ONE band keeps the
    # contiguous summation order np.mean(axis=(1, 2)) uses
    out, _ = _run(data.reshape(Cn, Tn * nb), np.ones((1, 1)), None, None, do_car=False, want='power')
    return out


def process_HG(data, bandpassCoefs, bad_channels=None, filt_ics=None):
    """CAR -> band-pass -> RMS band power in ONE launch (reference :10-39).  Returns (power (channels,), filt_ics)."""
    data = np.asarray(data, dtype=np.float64)
    b, a, z = _split_coefs(bandpassCoefs, data.shape[0], filt_ics)
    return _run(data, b, a, z, good=_good_mask(data.shape[0], bad_channels), do_car=True, want='power')
