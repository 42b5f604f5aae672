import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['XPS_LIB_OVERRIDE'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libxps_gstamp.so')
import numpy as np, torch
from cross_patient_speech_decoding_amd import _lib
from cross_patient_speech_decoding_amd.nn_models import functional as XF
l = _lib.lib(); l.xps_debug_read_gstamps.argtypes = [C.c_void_p, C.c_int]
for name, M, N, K in [('tn dWih L1', 384, 256, 40960), ('tn dWhh', 384, 128, 40960)]:
    A = torch.randn(K, M, device='cuda'); B = torch.randn(K, N, device='cuda'); Cc = torch.empty(M, N, device='cuda'); bias = torch.empty(M, device='cuda')
    for _ in range(3): XF.gemm_tn_grouped([XF.tn_problem(A, B, Cc, M, N, K, colsum_out=bias)], 'cuda')
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8192 * 4))(); l.xps_debug_read_gstamps(buf, 8192 * 4)
    a = np.array(buf[:], dtype=np.float64).reshape(8192, 4)
    a = a[a[:, 1] > 0]
    tot = a.sum(1)
    print(f'{name}: waves {len(a)}; share of k-loop cycles: load-issue {a[:,0].sum()/tot.sum():.2f}  mfma+lds {a[:,1].sum()/tot.sum():.2f}  lds-store {a[:,2].sum()/tot.sum():.2f}  barrier {a[:,3].sum()/tot.sum():.2f}; cycles per wave {np.median(tot):.0f}')
