"""Oracle: numpy / scipy restatement of the reference's per-bin feature extraction
(realtime_sim/realtime_processing.py:10-164: CAR -> lfilter per band with carried state -> RMS).

TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/realtime_processing.npz (generated from the reference's own
module, tests/golden/make_processing_fixtures.py)."""
import numpy as np
import scipy.signal as signal


def car(data, bad_channels=None):                                   # :42-57
    good = [i for i in range(data.shape[0]) if i not in (bad_channels or [])]
    return data - np.mean(data[good, :], axis=0)


def iir_filter(data, coefs, zi=None):                               # :106-143
    out, ics = [], []
    for k, bc in enumerate(coefs):
        b, a = bc[:, 1], bc[:, 0]
        z = np.tile(signal.lfilter_zi(b, a), (data.shape[0], 1)) if zi is None else zi[k]
        y, zf = signal.lfilter(b, a, data, zi=z)
        out.append(y)
        ics.append(zf)
    return np.stack(out, axis=-1), np.stack(ics, axis=0)


def fir_filter(data, coefs):                                        # :86-103
    return np.stack([signal.lfilter(c, 1.0, data) for c in coefs], axis=-1), None


def bin_power(filtered):                                            # :146-164
    return np.sqrt(np.mean(np.square(filtered), axis=(1, 2)))


def process_hg(data, coefs, bad_channels=None, filt_ics=None):      # :10-39
    x = car(data, bad_channels)
    coefs = np.asarray(coefs)
    if coefs.ndim == 3:
        y, ics = iir_filter(x, coefs, filt_ics)
    elif coefs.ndim == 2:
        y, ics = fir_filter(x, coefs)
    else:
        raise ValueError('bandpassCoefs must be either 2D or 3D array.')
    return bin_power(y), ics
