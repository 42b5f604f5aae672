"""Which framework (at::native) kernels are still launched inside one training step, and from where: torch.profiler with stacks
on one step of a bench workload (WORKLOAD=configs1|configs3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

name = os.environ.get('WORKLOAD', 'configs1')
c = bench.WORKLOADS[name]
dev = torch.device('cuda', 0)
model, step = bench._make_step(c, dev, 0)
for _ in range(30):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
seen = []
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::'):
        ks = [k.name for k in e.kernels]
        if any(('at::native' in k) or ('rocclr' in k) or ('fillBuffer' in k) for k in ks):
            top = [f for f in (e.stack or []) if 'cross_patient' in f or 'bench.py' in f][:4]
            seen.append((e.name, [k[:70] for k in ks], top))
for s in seen:
    print(s[0], s[1]); [print('     ', f) for f in s[2]]
print(len(seen), 'framework launches in the step')
