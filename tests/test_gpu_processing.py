"""Per-bin high-gamma feature extraction (CAR -> band-pass with carried state -> RMS) on the GPU against the golden
vectors of the reference's realtime_sim/realtime_processing.py: the IIR chain is BIT-EXACT (scipy's direct-form-II-
transposed arithmetic, numpy's summation orders); the FIR path (scipy: np.convolve / BLAS dot) to rounding."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rp():
    from cross_patient_speech_decoding_amd.realtime_sim import realtime_processing as rp
    return rp


def test_iir_chain_bit_exact_with_carried_state(golden_dir):
    rp = _rp()
    g = np.load(os.path.join(golden_dir, 'realtime_processing.npz'))
    bad = [int(v) for v in g['bad']]
    ics = None
    for i in range(3):
        p, ics = rp.process_HG(g['bins'][i], g['iir'], bad_channels=bad, filt_ics=ics)
        assert p.dtype == np.float64 and ics.shape == g[f'iir_ics{i}'].shape
        np.testing.assert_array_equal(ics, g[f'iir_ics{i}'])
        np.testing.assert_array_equal(p, g[f'iir_power{i}'])


def test_stage_functions_match_reference(golden_dir):
    rp = _rp()
    g = np.load(os.path.join(golden_dir, 'realtime_processing.npz'))
    bad = [int(v) for v in g['bad']]
    car = rp.CAR(g['bins'][0], bad)
    np.testing.assert_array_equal(car, g['car0'])
    y, zf = rp.IIR_filter_HG_bin(car, g['iir'])
    np.testing.assert_array_equal(y, g['iir_filtered0'])
    np.testing.assert_array_equal(zf, g['iir_zf0'])
    np.testing.assert_array_equal(rp.compute_bin_power(y), g['power_of_filtered0'])
    y2, z2 = rp.filter_HG_bin(car, g['iir'])
    np.testing.assert_array_equal(y2, y)
    # FIR: scipy evaluates it with np.convolve (BLAS dot, undefined summation order): agreement to rounding
    yf, none = rp.FIR_filter_HG_bin(rp.CAR(g['bins'][1]), g['fir'])
    assert none is None
    np.testing.assert_allclose(yf, g['fir_filtered1'], rtol=0, atol=1e-14 * np.abs(g['fir_filtered1']).max())
    pf, none = rp.process_HG(g['bins'][1], g['fir'])
    assert none is None
    np.testing.assert_allclose(pf, g['fir_power1'], rtol=1e-13)
    with pytest.raises(ValueError):
        rp.filter_HG_bin(car, np.zeros(5))


def test_sizes_beyond_the_golden_vs_oracle():
    """128 channels x 40 samples x 8 bands (a realistic bin) and a long ragged one against the CPU oracle."""
    from oracle import realtime_processing_oracle as po
    import scipy.signal as signal
    rp = _rp()
    rng = np.random.default_rng(5)
    for C, Tn, nb, order in ((128, 40, 8, 3), (37, 301, 5, 4), (9, 7, 1, 1)):
        coefs = []
        for k in range(nb):
            b, a = signal.butter(order, [60 + 12 * k, 72 + 12 * k], btype='band', fs=2000)
            coefs.append(np.stack([a, b], axis=1))
        coefs = np.stack(coefs)
        ics_ref = ics = None
        for _ in range(2):
            d = rng.standard_normal((C, Tn))
            p_ref, ics_ref = po.process_hg(d, coefs, bad_channels=[1], filt_ics=ics_ref)
            p, ics = rp.process_HG(d, coefs, bad_channels=[1], filt_ics=ics)
            np.testing.assert_array_equal(ics, ics_ref)
            np.testing.assert_array_equal(p, p_ref)
