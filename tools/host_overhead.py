"""How long does the HOST need to enqueue one training step (no sync) vs the GPU to execute it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
c = bench.CFG
torch.manual_seed(1234)
model = bench.build_model(c).cuda()
opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_norm=0.5)
X, y = bench.make_data(0, c); X, y = X.cuda(), y.cuda()
model.train()
def step():
    opt.zero_grad()
    logits = model(X, y, teacher_forcing_ratio=0.5)
    loss = model.criterion(logits.view(-1, 9), y.view(-1))
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'host enqueue per step {t_enq / n * 1e3:.3f} ms ; wall per step {t_all / n * 1e3:.3f} ms')
