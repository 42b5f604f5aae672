"""Diagnostic: stamped build (-DXPS_CL_STAMP) of the cluster GRU backward: cycle shares of the loop segments.
build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DXPS_CL_STAMP -o tools/libxps_clstamp.so cross_patient_speech_decoding_amd/csrc/*.hip"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libxps_clstamp.so')
from cross_patient_speech_decoding_amd.nn_models import functional as xf
T, B, H, ndir = 20, 2048, 512, 2
torch.manual_seed(0)
gi = (torch.randn(ndir, T, B, 3 * H) * 0.5).cuda()
ws = [(torch.randn(3 * H, H) / H ** 0.5).cuda() for _ in range(ndir)]
bs = [(torch.randn(3 * H) * 0.1).cuda() for _ in range(ndir)]
dy = (torch.randn(T, B, ndir * H) * 0.1).cuda()
l = _lib.lib()
l.xps_debug_read_cluster_stamps.argtypes = [C.c_void_p, C.c_int]
for mode in sys.argv[1:] or ['persistent']:
    xf.set_gru_cluster_mode(mode)
    y_ext, saved = xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)
    for _ in range(3):
        xf._gru_backward(dy, None, y_ext, saved, ws, T, B, H, ndir, False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16384)()
    assert l.xps_debug_read_cluster_stamps(buf, 16384) == 0
    raw = np.array(buf[:8192], dtype=np.float64).reshape(1024, 8)
    rt = np.array(buf[8192:], dtype=np.float64).reshape(1024, 8)[:, :6]
    t0 = rt[:, 0].min()
    print(mode, 'wall clock (us since the first wave entered; median / max over waves): entry %.1f/%.1f  weights done %.1f/%.1f  xcd check done %.1f/%.1f  pre-phase done %.1f/%.1f  loop begin %.1f/%.1f  loop end %.1f/%.1f' % tuple(
        v for i in range(6) for v in (np.median(rt[:, i] - t0) / 100, (rt[:, i] - t0).max() / 100)))
    nq = (T - 1) * 24 if mode == 'persistent' else 24
    names = ['(unused)', 'requests + contraction + DMA issue', 'poll check', 'drain: vmcnt(0)', 'barrier', 'publish + gate math of previous round']
    print(mode, 'per sub-iteration cycles (median over waves; wave 0 / others):')
    w0 = raw[0::4]; wo = np.concatenate([raw[1::4], raw[2::4], raw[3::4]])
    for i, nme in enumerate(names):
        print(f'   {nme:44s} {np.median(w0[:, i]) / nq:9.0f} {np.median(wo[:, i]) / nq:9.0f}')
    print('   loop total cycles median', np.median(raw[:, 7] - raw[:, 6]), ' per sub-iteration', np.median(raw[:, 7] - raw[:, 6]) / nq)
    dur = raw[:, 7] - raw[:, 6]
    print('   loop cycles: min %.0f p50 %.0f p90 %.0f max %.0f' % (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
    print('   begin spread %.0f  end spread %.0f  first begin -> last end %.0f' % (raw[:, 6].max() - raw[:, 6].min(), raw[:, 7].max() - raw[:, 7].min(), raw[:, 7].max() - raw[:, 6].min()))
    # per workgroup (4 waves each): blockIdx = wid // 4; cluster mapping for G = 256, CS = 16: xcd = bid & 7, slot = bid >> 3, cluster = xcd * 2 + slot // 16
    bid = np.arange(1024) // 4
    cl = (bid & 7) * 2 + (bid >> 3) // 16
    for c in range(16):
        m = cl == c
        print(f'   cluster {c:2d}: loop cycles p50 {np.median(dur[m]):10.0f} max {dur[m].max():10.0f}; waits (poll) {np.median(raw[m, 2]):9.0f} commit {np.median(raw[m, 3]):9.0f} barrier {np.median(raw[m, 4]):9.0f} epi {np.median(raw[m, 5]):9.0f} issue {np.median(raw[m, 0]):9.0f}')
