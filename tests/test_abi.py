"""The C-ABI library builds, loads and exports every symbol include/xps.h declares.
No compute calls (no GPU needed): argument validation returns before any HIP call."""
import ctypes as C

import pytest

from cross_patient_speech_decoding_amd import _build, _lib


@pytest.fixture(scope='module')
def lib():
    _build.build(verbose=False)
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = _lib.header_functions()
    assert len(declared) >= 30
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version_and_error_channel(lib):
    assert lib.xps_abi_version() == 4
    rm = _lib.rowmap(4)
    rc = lib.xps_gemm_nt_f32(None, C.byref(rm), None, C.byref(rm), None, C.byref(rm), None, 4, 4, 4, 0, None)
    assert rc == -1
    assert b'xps_gemm_nt_f32' in lib.xps_last_error()
    with pytest.raises(_lib.XpsError, match='null argument'):
        _lib.call('xps_gemm_nn_f32', None, C.byref(rm), None, C.byref(rm), None, C.byref(rm), 4, 4, 4, 0, None)
    rc = lib.xps_gru_seq_fwd_f32(None, None, None, None, None, None, 1, 1, 1, 1, None, 0, None)
    assert rc == -1


def test_workspace_queries(lib):
    assert lib.xps_gemm_tn_f32_workspace(384, 100, 40960) >= 384 * 100 * 4
    assert lib.xps_colsum_f32_workspace(1000, 64) >= 4 * 64 * 4
    assert lib.xps_xcov_f64_workspace(409600, 128, 128) >= 128 * 128 * 8
    assert lib.xps_sumsq_f32_workspace(10) >= 8
    assert lib.xps_jacobi_f64_workspace(64) >= 1


def test_product_path_refuses_cpu_tensors():
    import torch
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        XF.linear(torch.zeros(2, 3), torch.zeros(4, 3), torch.zeros(4))
