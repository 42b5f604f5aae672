"""world_size-2 (and 4) gloo tests of the data-parallel HOST logic on CPU: batch sharding, global
loss weighting, metric reduction, LR schedule.  The kernels themselves need the GPU and are covered
by tests/test_gpu_training.py::test_data_parallel_two_ranks_equal_single_process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('world,port', [(2, '29541'), (4, '29542')])
def test_gloo_host_logic(world, port):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), '--world', str(world),
                          '--device', 'cpu'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'DP_OK' in out.stdout


def test_overlapped_allreduce_has_no_rank_local_fallback():
    """A rank that caught an exception and switched to another collective sequence would pair mismatched collectives
    with its peers (RCCL hang / corruption).  The hook that issues the overlapped all-reduce must not swallow anything."""
    import ast
    import inspect
    import textwrap
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    tree = ast.parse(textwrap.dedent(inspect.getsource(FlatAdamW._reduce_tail_async)))
    assert not any(isinstance(n, (ast.Try, ast.ExceptHandler)) for n in ast.walk(tree))
    tree = ast.parse(textwrap.dedent(inspect.getsource(FlatAdamW.step)))
    assert not any(isinstance(n, (ast.Try, ast.ExceptHandler)) for n in ast.walk(tree))
