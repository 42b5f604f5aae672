"""Stage timing of the replicated part of the 8-view MCCA fit (_gevp at D = 1024, k = 30) with synchronising timers."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
import importlib
M = importlib.import_module('cross_patient_speech_decoding_amd.alignment.AlignMCCA')
from cross_patient_speech_decoding_amd.alignment import _linalg as LA
pats = [make_patient(p, 512, T=200, C=128) for p in range(8)]
feats, labs = [torch.from_numpy(p[0]).cuda() for p in pats], [p[1] for p in pats]
cap = {}
orig = M._gevp
def spy(G, offs, k, regs):
    cap['a'] = (G.clone(), offs.copy(), k, regs)
    return orig(G, offs, k, regs)
M._gevp = spy
al = M.AlignMCCA(n_components=30, regs=0.5); al.fit(feats, labs)
M._gevp = orig
G, offs, k, regs = cap['a']
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orig(G, offs, k, regs); torch.cuda.synchronize()
    print(f'_gevp: {(time.perf_counter() - t0) * 1e3:.1f} ms')
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    al.fit(feats, labs); torch.cuda.synchronize()
    print(f'whole fit (resident inputs): {(time.perf_counter() - t0) * 1e3:.1f} ms')
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **kw):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **kw)
        torch.cuda.synchronize(); acc[name][0] += time.perf_counter() - t0; acc[name][1] += 1
        return r
    setattr(mod, name, g)
for n in ('dgemm', 'eigh_psd_batched', 'eigh_sym_top', 'to_device', 'xcov', 'col_mean', 'apply', 'eigh_psd'):
    wrap(LA, n)
wrap(M, '_gevp'); wrap(M, '_group_conditions_device')
torch.cuda.synchronize(); t0 = time.perf_counter()
al.fit(feats, labs); torch.cuda.synchronize()
print(f'instrumented fit: {(time.perf_counter() - t0) * 1e3:.1f} ms')
for n, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f'  {n:28s} {t * 1e3:8.2f} ms in {c:4d} calls')
