"""Distribution of the 20-step timing window of bench.py (sync, 20 steps, sync) inside ONE process: how often does a window
come out slow, and does pinning the process to a few cores change it?  usage: jitter.py [ncores_to_pin]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and int(sys.argv[1]) > 0:
    cores = sorted(os.sched_getaffinity(0))[:int(sys.argv[1])]
    os.sched_setaffinity(0, cores)
import torch
import bench
from cross_patient_speech_decoding_amd.nn_models import functional as XF
from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
c = bench.CFG
torch.manual_seed(1234)
model = bench.build_model(c).cuda()
opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_norm=0.5)
X, y = bench.make_data(0, c); X, y = X.cuda(), y.cuda()
model.train()
one = XF.unit_gradient(X.device)
def step():
    opt.zero_grad()
    logits = model(X, y, teacher_forcing_ratio=0.5)
    loss = model.criterion(logits.view(-1, c['num_classes']), y.view(-1))
    loss.backward(one)
    opt.step()
for _ in range(400): step()
torch.cuda.synchronize()
gc.disable()
w = []
for _ in range(150):
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    w.append((time.perf_counter() - t0) / 20 * 1e3)
w.sort()
print(f'affinity {len(os.sched_getaffinity(0))} cores: 20-step windows ms/step  min {w[0]:.3f}  median {w[len(w)//2]:.3f}  p90 {w[int(len(w)*0.9)]:.3f}  '
      f'p97 {w[int(len(w)*0.97)]:.3f}  max {w[-1]:.3f}')
