"""cProfile of the host side of training steps (enqueue only) to find Python overhead."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
c = bench.CFG
torch.manual_seed(1234)
model = bench.build_model(c).cuda()
opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_norm=0.5)
X, y = bench.make_data(0, c); X, y = X[:256].cuda(), y[:256].cuda()      # small batch: the GPU never back-pressures the host
model.train()
def step():
    opt.zero_grad()
    logits = model(X, y, teacher_forcing_ratio=0.5)
    loss = model.criterion(logits.view(-1, 9), y.view(-1))
    loss.backward()
    opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'host enqueue per step (B=256, GPU not limiting) {t_enq / n * 1e3:.3f} ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(n): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats(os.environ.get('SORT', 'tottime')).print_stats(int(os.environ.get('TOP', '28')))
