"""Which torch (non-libxps) device kernels a training step still launches, with the Python frame that launched them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
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
    loss = model.criterion(logits.view(-1, c['num_classes']), y.view(-1))
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and ev.cpu_parent is not None \
            and not ev.cpu_parent.name.startswith('aten::') and any(k.duration > 0 for k in ev.kernels):
        frames = [s for s in (ev.stack or []) if 'cross_patient' in s or 'bench' in s or 'torch_kernels' in s][:2]
        print(f'{ev.name:28s} kernels={[k.name[:40] for k in ev.kernels]} parent={ev.cpu_parent.name[:30]} {frames}')
    elif ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and ev.cpu_parent is None \
            and any(k.duration > 0 for k in ev.kernels):
        frames = [s for s in (ev.stack or []) if 'cross_patient' in s or 'bench' in s or 'torch_kernels' in s][:2]
        print(f'{ev.name:28s} kernels={[k.name[:40] for k in ev.kernels]} top {frames}')
