"""Pin oracle/realtime_processing_oracle.py to golden vectors of the reference's realtime_processing module.  CPU only."""
import os

import numpy as np

from oracle import realtime_processing_oracle as po


def test_process_hg_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'realtime_processing.npz'))
    bad = list(g['bad'])
    ics = None
    for i in range(3):                                   # consecutive bins, carried IIR state
        p, ics = po.process_hg(g['bins'][i], g['iir'], bad_channels=bad, filt_ics=ics)
        np.testing.assert_array_equal(p, g[f'iir_power{i}'])
        np.testing.assert_array_equal(ics, g[f'iir_ics{i}'])
    np.testing.assert_array_equal(po.car(g['bins'][0], bad), g['car0'])
    y, zf = po.iir_filter(g['car0'], g['iir'])
    np.testing.assert_array_equal(y, g['iir_filtered0'])
    np.testing.assert_array_equal(zf, g['iir_zf0'])
    np.testing.assert_array_equal(po.bin_power(y), g['power_of_filtered0'])
    p, none = po.process_hg(g['bins'][1], g['fir'])
    assert none is None
    np.testing.assert_array_equal(p, g['fir_power1'])
