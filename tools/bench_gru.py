"""Times the GRU recurrence entry points alone (HIP events): forward (training: saves gates / eval) and BPTT.
usage: bench_gru.py [H] [B] [T] [ndir]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as xf

H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ndir = int(sys.argv[4]) if len(sys.argv) > 4 else 2
modes = os.environ.get('MODES', 'persistent,steps,off').split(',')
precs = os.environ.get('PRECS', 'bf16x3,fp32').split(',')
torch.manual_seed(0)
gi = (torch.randn(ndir, T, B, 3 * H) * 0.5).cuda()
ws = [(torch.randn(3 * H, H) / H ** 0.5).cuda() for _ in range(ndir)]
bs = [(torch.randn(3 * H) * 0.1).cuda() for _ in range(ndir)]
dy = (torch.randn(T, B, ndir * H) * 0.1).cuda()


def ev(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for prec in precs:
    xf.set_gemm_precision(prec)
    for mode in modes:
        xf.set_gru_cluster_mode(mode)
        xf.check_gru_status()
        f_tr = ev(lambda: xf.gru_forward_training_form(gi, ws, bs, T, B, H, ndir))
        xf.check_gru_status()
        f_ev = ev(lambda: xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, False))
        y_ext, saved = xf._gru_forward(gi, ws, bs, None, T, B, H, ndir, True)
        xf.check_gru_status()
        sp4 = xf.split4_wanted(T, B, H, ndir)                     # (the output format the training step asks for)
        b_tr = ev(lambda: xf._gru_backward(dy, None, y_ext, saved, ws, T, B, H, ndir, False, split4=sp4))
        diag = ''
        xf.check_gru_status()
        by_f = 4 * ndir * T * B * (3 * H + H + 4 * H)
        by_b = 4 * ndir * T * B * (4 * H + H + H + 3 * H + H)
        print(f'H={H} B={B} T={T} ndir={ndir} {prec:7s} {mode:10s}: fwd(train) {f_tr:8.1f} us ({f_tr / T:6.1f}/step, {by_f / f_tr / 1e3:6.0f} GB/s)  '
              f'fwd(eval) {f_ev:8.1f} us  bwd {b_tr:8.1f} us ({b_tr / T:6.1f}/step, {by_b / b_tr / 1e3:6.0f} GB/s)' + diag, flush=True)
