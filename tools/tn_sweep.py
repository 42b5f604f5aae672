"""TN (weight-gradient) GEMM: time vs number of blocks (XPS_TN_BLOCKS is read once per process)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.path.dirname(here))
    import torch
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    from bench_gemm import timeit  # noqa
    sys.exit(0)
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
dev = 'cuda'
out = []
for name, M, N, K in [('dWih L1', 384, 256, 40960), ('dWhh', 384, 128, 40960), ('conv', 100, 640, 40960)]:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    bias = torch.empty(M, device=dev)
    t = timeit(lambda: XF.gemm_tn_grouped([XF.tn_problem(A, B, C, M, N, K, colsum_out=bias)], dev))
    out.append('%%s %%6.1f us %%5.1f TF' %% (name, t * 1e6, 2 * M * N * K / t / 1e12))
print(os.environ.get('XPS_TN_BLOCKS'), ' | '.join(out))
''' % os.path.dirname(here)
for nb in (96, 192, 256, 384, 512, 768, 1024, 1536):
    env = dict(os.environ, XPS_TN_BLOCKS=str(nb))
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:])
