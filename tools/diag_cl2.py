"""Diagnostics of the cluster BPTT launch at the bench shape (one bidirectional H = 512 layer, 2048 trials, 20 steps): HIP-event
time, the status block of the last launch (failed look-ahead lookups and the time waited for them, workgroups in one-XCD
clusters) and -- with the stamped build (XPS_LIB_OVERRIDE=tools/libxps_clstamp.so) -- cycles per slot and wave role."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd import _lib
from cross_patient_speech_decoding_amd.nn_models import functional as xf
T, B, H, ndir = 20, int(os.environ.get('B', 2048)), int(os.environ.get('H', 512)), 2
torch.manual_seed(0)
gi = (torch.randn(ndir, T, B, 3 * H) * 0.5).cuda()
ws = [(torch.randn(3 * H, H) / H ** 0.5).cuda() for _ in range(ndir)]
bs = [(torch.randn(3 * H) * 0.1).cuda() for _ in range(ndir)]
dy = (torch.randn(T, B, ndir * H) * 0.1).cuda()
y_ext, saved = xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)
l = _lib.lib()
if os.environ.get('GRID'):
    l.xps_set_gru_bptt_grid(int(os.environ['GRID']))
print('BPTT grid:', '4 x 4' if l.xps_get_gru_bptt_grid() else '1-D')
ef0, ef1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)
ef0.record()
for _ in range(10):
    xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)
ef1.record(); torch.cuda.synchronize()
usf = ef0.elapsed_time(ef1) / 10 * 1e3
print(f'fwd launch (train, incl. init kernel) {usf:.1f} us ({4 * ndir * T * B * 8 * H / usf / 1e3:.0f} GB/s algorithmic)')
l.xps_debug_gru_bwd_status_offset.restype = C.c_longlong
off = l.xps_debug_gru_bwd_status_offset(B, H, ndir)
wt = [torch.empty(H, 3 * H, device='cuda') for _ in ws]
xf.call('xps_transpose_batched_f32', xf._ptr_array(ws), xf._ptr_array(wt), 2, 3 * H, H, xf._stream())
dgi = torch.empty(ndir, T, B, 3 * H, device='cuda'); dghn = torch.empty(ndir, T, B, H, device='cuda')
nb = l.xps_gru_seq_bwd_f32_workspace(T, B, H, ndir)
wsb = torch.empty(nb, dtype=torch.uint8, device='cuda')
def launch():
    xf.call('xps_gru_seq_bwd_split4_f32', dy.data_ptr(), None, y_ext.data_ptr(), saved.data_ptr(), xf._ptr_array(ws), xf._ptr_array(wt),
            dgi.data_ptr(), dghn.data_ptr(), None, T, B, H, ndir, 0.0, 0, wsb.data_ptr(), nb, xf._stream())
for _ in range(3):
    launch()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    launch()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
st = wsb[off:off + 64].view(torch.int32).tolist()
print(f'bwd launch {us:.1f} us ({4 * ndir * T * B * 10 * H / us / 1e3:.0f} GB/s algorithmic); status {st[0]}; image lookups missed {st[1]} (waited {st[2] / 100:.0f} us in all); '
      f'quarter lookups missed {st[6]} (waited {st[7] / 100:.0f} us); workgroups in one-XCD clusters {st[4]}, mixed {st[5]}')
if hasattr(l, 'xps_debug_read_cluster_stamps'):
    l.xps_debug_read_cluster_stamps.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 16384)()
    assert l.xps_debug_read_cluster_stamps(buf, 16384) == 0
    raw = np.array(buf[:], dtype=np.float64).reshape(2048, 8)
    slots = (T - 1) * 16 + 4
    print('cycles per slot (median over workgroups): work / drain / barrier wait / poll | contraction: dma issue, mfma loop; gates: flags+math+stores, flag wait+quarter requests')
    for role, wv in (('contraction', [0, 1, 2, 3]), ('gates 5-7', [5, 6, 7]), ('gate wave 4', [4])):
        sel = np.concatenate([raw[w::8] for w in wv])
        print(f'   {role:12s} {np.median(sel[:, 0]) / slots:8.0f} {np.median(sel[:, 1]) / slots:8.0f} {np.median(sel[:, 2]) / slots:8.0f} {np.median(sel[:, 3]) / slots:8.0f} | {np.median(sel[:, 4]) / slots:8.0f} {np.median(sel[:, 5]) / slots:8.0f}')

    if hasattr(l, 'xps_debug_read_cluster_stamps2') and l.xps_get_gru_bptt_grid():
        l.xps_debug_read_cluster_stamps2.argtypes = [C.c_void_p, C.c_int]
        assert l.xps_debug_read_cluster_stamps2(buf, 16384) == 0
        raw = np.array(buf[:], dtype=np.float64).reshape(2048, 8)
        pairs = slots / 2
        print('gate waves, cycles per (math slot + request slot): math slot: until inputs are there / epilogue issued / next inputs requested | request slot: exchange-row wait + flag / quarter flags / quarters requested')
        for role, wv in (('gates 5-7', [5, 6, 7]), ('gate wave 4', [4])):
            sel = np.concatenate([raw[w::8] for w in wv])
            print(f'   {role:12s} ' + ' '.join(f'{np.median(sel[:, k]) / pairs:8.0f}' for k in range(6)))
