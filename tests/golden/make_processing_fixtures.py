"""Generate tests/golden/realtime_processing.npz from the REFERENCE's realtime_sim/realtime_processing.py
(importable as is: numpy + scipy only).  Build container only."""
import os
import sys

import numpy as np
import scipy
import scipy.signal as signal

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference/aligned_decoding')
from realtime_sim import realtime_processing as rp     # noqa: E402

if __name__ == '__main__':
    rng = np.random.default_rng(11)
    fs, C, Tn = 2000, 20, 40
    iir = []
    for lo, hi in ((70, 90), (90, 110), (110, 130), (130, 150)):
        b, a = signal.butter(2, [lo, hi], btype='band', fs=fs)
        iir.append(np.stack([a, b], axis=1))
    iir = np.stack(iir)                                  # (bands, taps, [a, b])
    fir = np.stack([signal.firwin(15, [lo, hi], pass_zero=False, fs=fs) for lo, hi in ((70, 110), (110, 150))])
    bins = [rng.standard_normal((C, Tn)) * (1 + 0.3 * np.arange(C))[:, None] for _ in range(3)]
    bad = [3, 11]
    out = dict(iir=iir, fir=fir, bad=np.array(bad), bins=np.stack(bins), numpy_version=np.array(np.__version__),
               scipy_version=np.array(scipy.__version__))
    ics = None
    for i, d in enumerate(bins):                         # three consecutive bins with carried filter state
        p, ics = rp.process_HG(d, iir, bad_channels=bad, filt_ics=ics)
        out[f'iir_power{i}'], out[f'iir_ics{i}'] = p, ics
    out['car0'] = rp.CAR(bins[0], bad)
    y, z = rp.IIR_filter_HG_bin(out['car0'], iir)
    out['iir_filtered0'], out['iir_zf0'] = y, z
    out['power_of_filtered0'] = rp.compute_bin_power(y)
    pf, none = rp.process_HG(bins[1], fir)
    assert none is None
    out['fir_power1'] = pf
    out['fir_filtered1'] = rp.FIR_filter_HG_bin(rp.CAR(bins[1]), fir)[0]
    np.savez_compressed(os.path.join(HERE, 'realtime_processing.npz'), **out)
    print('realtime_processing.npz', {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.ndim})
