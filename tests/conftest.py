import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(params=['fp32', 'bf16x3'])
def gemm_precision(request):
    """Runs a GPU test in both product precisions of the matrix kernels (include/xps.h xps_set_gemm_precision);
    yields the mode name and restores the previous mode."""
    from cross_patient_speech_decoding_amd._lib import lib
    old = lib().xps_get_gemm_precision()
    assert lib().xps_set_gemm_precision({'fp32': 0, 'bf16x3': 1}[request.param]) == 0
    yield request.param
    lib().xps_set_gemm_precision(old)
