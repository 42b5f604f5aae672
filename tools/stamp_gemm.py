import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['XPS_LIB_OVERRIDE'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libxps_gstamp.so')
os.environ['XPS_GEMM_SMALL_TILE_BLOCKS'] = '0'
import numpy as np, torch
from cross_patient_speech_decoding_amd import _lib
from cross_patient_speech_decoding_amd.nn_models import functional as XF
l = _lib.lib(); l.xps_debug_read_gstamps.argtypes = [C.c_void_p, C.c_int]
for name, M, N, K in [('nt proj L1', 40960, 384, 256), ('nt conv', 40960, 100, 640), ('nt sq', 4096, 4096, 1024)]:
    A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); Cc = torch.empty(M, N, device='cuda')
    for _ in range(3): XF.gemm_nt(A, B, Cc, M, N, K)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8192 * 4))(); l.xps_debug_read_gstamps(buf, 8192 * 4)
    nb = min(8192, ((M + 127) // 128) * ((N + 127) // 128) * 4)
    a = np.array(buf[:], dtype=np.float64).reshape(8192, 4)[:nb] / (K / 16)
    print(f'{name}: per k-tile cycles (median): load-issue {np.median(a[:,0]):.0f}  mfma+lds {np.median(a[:,1]):.0f}  lds-store {np.median(a[:,2]):.0f}  barrier {np.median(a[:,3]):.0f}   (pure MFMA = 2048)')
