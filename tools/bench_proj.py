"""Input projections of the cfg-2 encoder layers: weight-stationary kernel (XPS_PROJ_WS=1) vs the tile kernels (0)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd._lib import call, rowmap
from cross_patient_speech_decoding_amd.nn_models import functional as F


def timeit(fn, iters=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M, N, K, nprob in [(40960, 384, 256, 2), (40960, 384, 100, 2), (40960, 192, 128, 2)]:
    A = torch.randn(M, K, device='cuda'); W = [torch.randn(N, K, device='cuda') * K ** -0.5 for _ in range(nprob)]
    bs = [torch.randn(N, device='cuda') for _ in range(nprob)]
    outs = [torch.empty(M, N, device='cuda') for _ in range(nprob)]
    ra, rb, rc = rowmap(K), rowmap(K), rowmap(N)
    fn = lambda: call('xps_gemm_nt_multi_f32', A.data_ptr(), C.byref(ra), F._ptr_array(W), C.byref(rb), F._ptr_array(outs), C.byref(rc),
                      F._ptr_array(bs), nprob, M, N, K, F._stream())
    res = []
    for v in ('0', '1', '0', '1'):
        os.environ['XPS_PROJ_WS'] = v
        res.append('%s: %6.1f us' % (v, timeit(fn)))
    mb = (M * K + nprob * M * N) * 4 / 1e6
    print(f'M={M} N={N} K={K} x{nprob}  ({mb:.0f} MB)  ' + '  '.join(res))
