"""Host-side time of each phase of a training step (small batch: the GPU never back-pressures)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
c = bench.CFG
torch.manual_seed(1234)
model = bench.build_model(c).cuda()
opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_norm=0.5)
X, y = bench.make_data(0, c); X, y = X[:256].cuda(), y[:256].cuda()
model.train()
T = [0.0] * 5
def step(rec=False):
    t0 = time.perf_counter(); opt.zero_grad()
    t1 = time.perf_counter(); logits = model(X, y, teacher_forcing_ratio=0.5)
    t2 = time.perf_counter(); loss = model.criterion(logits.view(-1, 9), y.view(-1))
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[i] += d
for _ in range(10): step()
torch.cuda.synchronize()
n = 100
for _ in range(n): step(True)
torch.cuda.synchronize()
print('host us per step: zero_grad %.0f  forward %.0f  loss %.0f  backward %.0f  optimizer %.0f  total %.0f' % tuple([t / n * 1e6 for t in T] + [sum(T) / n * 1e6]))

# per-Function host time of backward (wrapped staticmethods)
from cross_patient_speech_decoding_amd.nn_models import functional as XF
acc = {}
def wrap(cls):
    orig = cls.backward
    def timed(ctx, *a):
        t0 = time.perf_counter(); r = orig(ctx, *a); acc[cls.__name__] = acc.get(cls.__name__, 0.0) + time.perf_counter() - t0; return r
    cls.backward = staticmethod(timed)
for name in dir(XF):
    obj = getattr(XF, name)
    if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
        wrap(obj)
for _ in range(n): step()
torch.cuda.synchronize()
print('backward host us per step by Function:', {k: round(v / n * 1e6) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}, 'sum', round(sum(acc.values()) / n * 1e6))
