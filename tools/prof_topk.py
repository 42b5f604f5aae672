"""Where the time of eigh_sym_top (8-view MCCA matrix, k = 30) goes: synchronised timers around its building blocks."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
import importlib
M = importlib.import_module('cross_patient_speech_decoding_amd.alignment.AlignMCCA')
from cross_patient_speech_decoding_amd.alignment import _linalg as LA
pats = [make_patient(p, 512, T=200, C=128) for p in range(8)]
feats, labs = [p[0] for p in pats], [p[1] for p in pats]
cap = {}
orig = LA.eigh_sym_top
def spy(C, k, **kw):
    cap['C'] = C.clone(); cap['k'] = k
    return orig(C, k, **kw)
LA.eigh_sym_top = spy; M.LA.eigh_sym_top = spy
M.AlignMCCA(n_components=30, regs=0.5).fit(feats, labs)
LA.eigh_sym_top = orig; M.LA.eigh_sym_top = orig
C, k = cap['C'], cap['k']
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(name):
    f = getattr(LA, name)
    def g(*a, **kw):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **kw)
        torch.cuda.synchronize(); acc[name][0] += time.perf_counter() - t0; acc[name][1] += 1
        return r
    setattr(LA, name, g)
    return f
for rep in range(3):
    st = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w, V = LA.eigh_sym_top(C, k, stats=st)
    torch.cuda.synchronize()
    print(f'eigh_sym_top({C.shape[0]}, {k}): {(time.perf_counter() - t0) * 1e3:.1f} ms', st)
saved = {n: wrap(n) for n in ('dgemm', '_orthonormal_columns', 'eigh_psd', '_lanczos_bounds', '_jacobi')}
torch.cuda.synchronize(); t0 = time.perf_counter()
LA.eigh_sym_top(C, k)
torch.cuda.synchronize()
print(f'instrumented (synchronising): {(time.perf_counter() - t0) * 1e3:.1f} ms')
for n, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f'  {n:24s} {t * 1e3:8.2f} ms in {c:4d} calls ({t / max(c, 1) * 1e6:7.1f} us each)')
