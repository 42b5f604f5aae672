"""Upper bound of what a whole-step hipGraph would buy: capture one training step as it is (dropout seeds, the Adam step
count and the teacher-forcing pattern are frozen into the graph, so this is a TIMING probe only, not a valid trainer)
and compare replay time with eager stepping."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
c = bench.WORKLOADS[os.environ.get('WORKLOAD', 'configs1')]
torch.manual_seed(1234)
model = bench.build_model(c).cuda()
opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_norm=0.5)
X, y = bench.make_data(0, c); X, y = X.cuda(), y.cuda()
model.train()
one = torch.ones((), device='cuda')
coins = [True, False, True]
def step():
    opt.zero_grad()
    logits = model(X, y, coins=coins)
    loss = model.criterion(logits.view(-1, c['num_classes']), y.view(-1))
    loss.backward(one)
    opt.step()
    return loss
for _ in range(int(os.environ.get('PREWARM', '400'))): step()
torch.cuda.synchronize()
def timeit(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
N_T = int(os.environ.get('NT', '50'))
print(f'eager   {timeit(step, N_T):.3f} ms/step', flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
for _ in range(20): g.replay()
print(f'graphed {timeit(g.replay, N_T):.3f} ms/step   loss {float(loss):.4f}', flush=True)
print(f'eager   {timeit(step, N_T):.3f} ms/step', flush=True)
