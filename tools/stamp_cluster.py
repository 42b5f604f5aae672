"""Diagnostic: stamped build (-DXPS_CL_STAMP) of the cluster GRU forward: per wave role, cycles per round spent working,
draining (vmcnt wait) and waiting at the barrier.  The role that waits least at the barrier paces the kernel.
build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DXPS_CL_STAMP -o tools/libxps_clstamp.so cross_patient_speech_decoding_amd/csrc/*.hip"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd import _lib
_lib.LIB_PATH = os.environ.get('XPS_STAMP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libxps_clstamp.so')
from cross_patient_speech_decoding_amd.nn_models import functional as xf
T, B, H, ndir = 20, 2048, 512, 2
torch.manual_seed(0)
gi = (torch.randn(ndir, T, B, 3 * H) * 0.5).cuda()
ws = [(torch.randn(3 * H, H) / H ** 0.5).cuda() for _ in range(ndir)]
bs = [(torch.randn(3 * H) * 0.1).cuda() for _ in range(ndir)]
l = _lib.lib()
l.xps_debug_read_cluster_stamps.argtypes = [C.c_void_p, C.c_int]
for save in (True, False):
    for _ in range(3):
        xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, save)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16384)()
    assert l.xps_debug_read_cluster_stamps(buf, 16384) == 0
    raw = np.array(buf[:], dtype=np.float64).reshape(2048, 8)
    rounds = T * 8
    print('forward, saving gates' if save else 'forward, eval', '- cycles per round (median over workgroups): work / drain / barrier wait')
    for role, ws_ in (('contraction', [0, 1, 2, 3]), ('gates', [4, 5, 6, 7])):
        sel = np.concatenate([raw[w::8] for w in ws_])
        print(f'   {role:12s} {np.median(sel[:, 0]) / rounds:8.0f} {np.median(sel[:, 1]) / rounds:8.0f} {np.median(sel[:, 2]) / rounds:8.0f}')

# ---- backward (BPTT): per wave role, cycles per ROUND (3 sub-iterations): work / drain (vmcnt wait) / barrier wait / poll
y_ext, saved = xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)
dy = (torch.randn(T, B, ndir * H) * 0.1).cuda()
for _ in range(3):
    xf._gru_backward(dy, None, y_ext, saved, ws, T, B, H, ndir, False)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16384)()
assert l.xps_debug_read_cluster_stamps(buf, 16384) == 0
raw = np.array(buf[:], dtype=np.float64).reshape(2048, 8)
rounds = (T - 1) * 8
print('backward - cycles per round (median over workgroups): work / drain / barrier wait / poll')
for role, ws_ in (('contraction', [0, 1, 2, 3]), ('gates 5-7', [5, 6, 7]), ('gate wave 4', [4])):
    sel = np.concatenate([raw[w::8] for w in ws_])
    print(f'   {role:12s} {np.median(sel[:, 0]) / rounds:8.0f} {np.median(sel[:, 1]) / rounds:8.0f} {np.median(sel[:, 2]) / rounds:8.0f} {np.median(sel[:, 3]) / rounds:8.0f}')
