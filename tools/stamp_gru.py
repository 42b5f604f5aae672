"""Diagnostic: run the stamped build of the resident GRU forward and print cycle shares."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libxps_stamp.so')
from cross_patient_speech_decoding_amd.nn_models import functional as XF
T, B, H = 20, 2048, 128
w_hh = [torch.randn(3 * H, H, device='cuda') * 0.08 for _ in range(2)]
b_hh = [torch.randn(3 * H, device='cuda') * 0.1 for _ in range(2)]
gi = torch.randn(2, T, B, 3 * H, device='cuda') * 0.5
for _ in range(3):
    XF._gru_forward(gi, w_hh, b_hh, None, T, B, H, 2, True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8192)()
l = _lib.lib()
l.xps_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
print('rc', l.xps_debug_read_stamps(buf, 8192))
raw = np.array(buf[:], dtype=np.float64).reshape(1024, 8)
print('prologue cycles median %.0f, loop %.0f; entry spread %.0f, exit spread %.0f, first-entry to last-exit %.0f' % (np.median(raw[:,4]), np.median(raw[:,5]), raw[:,6].max()-raw[:,6].min(), raw[:,7].max()-raw[:,7].min(), raw[:,7].max()-raw[:,6].min()))
a = raw[:, :4] / T
print('per-step cycles (median over waves): mfma %.0f  epilogue %.0f  stores %.0f  barrier %.0f' % tuple(np.median(a, 0)))
print('p10/p90 mfma', np.percentile(a[:,0],[10,90]), 'epi', np.percentile(a[:,1],[10,90]), 'store', np.percentile(a[:,2],[10,90]), 'bar', np.percentile(a[:,3],[10,90]))
