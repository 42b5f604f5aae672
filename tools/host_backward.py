"""Host time of each autograd Function's forward / backward in a training step (small batch: the GPU never back-pressures)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cross_patient_speech_decoding_amd.nn_models import functional as XF
from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
acc = collections.defaultdict(float)
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t; return r
    return staticmethod(w)
for cls in [XF.GRULayerFn, XF.TemporalConvFn, XF.DecoderFn, XF.LinearFn, XF.CrossEntropyFn, XF.DropoutFn]:
    cls.forward = timed(cls.__name__ + '.fwd', cls.forward)
    cls.backward = timed(cls.__name__ + '.bwd', cls.backward)
c = bench.CFG
torch.manual_seed(1234)
model = bench.build_model(c).cuda()
opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_norm=0.5)
X, y = bench.make_data(0, c); X, y = X[:256].cuda(), y[:256].cuda()
model.train()
one = torch.ones((), device='cuda')
T = collections.defaultdict(float)
def step():
    t0 = time.perf_counter(); opt.zero_grad()
    t1 = time.perf_counter(); logits = model(X, y, teacher_forcing_ratio=0.5); loss = model.criterion(logits.view(-1, 9), y.view(-1))
    t2 = time.perf_counter(); loss.backward(one)
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter()
    T['zero'] += t1 - t0; T['fwd'] += t2 - t1; T['bwd'] += t3 - t2; T['opt'] += t4 - t3
for _ in range(20): step()
torch.cuda.synchronize(); acc.clear(); T.clear()
n = 100
for _ in range(n): step()
torch.cuda.synchronize()
print({k: round(v / n * 1e6, 1) for k, v in T.items()}, 'us per step')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f'{k:28s} {v / n * 1e6:8.1f} us')
