"""Stage timing of the 8-view MCCA fit (north-star shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
import importlib
M = importlib.import_module('cross_patient_speech_decoding_amd.alignment.AlignMCCA')
from cross_patient_speech_decoding_amd.alignment import _linalg as LA
P, N = 8, 2048
pats = [make_patient(p, N, T=200, C=128) for p in range(P)]
feats, labs = [p[0] for p in pats], [p[1] for p in pats]
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f'{label:40s} {(time.perf_counter() - t0) * 1e3:8.1f} ms'); return r
for rep in range(2):
    print('--- pass', rep)
    avgs = T('group conditions (upload + cnd_avg)', lambda: M._group_conditions_device(feats, labs))
    avgs = [a.reshape(-1, a.shape[-1]) for a in avgs]
    Vd = [LA.to_device(v).reshape(-1, v.shape[-1]) for v in avgs]
    Z = torch.cat([v.to(LA.F64) for v in Vd], dim=1).contiguous()
    mean = LA.col_mean(Z)
    G = T('xcov D=1024 + D2H', lambda: LA.xcov(Z, None, mean).cpu().numpy())
    print('rows', Z.shape)
    offs = np.arange(0, 1025, 128)
    T('8 x eigh_psd(128)', lambda: [LA.eigh_psd(LA.to_device(0.5 * G[o:o + 128, o:o + 128] + 0.5 * np.eye(128))) for o in offs[:-1]])
    Gd = LA.to_device(G)
    T('2 x dgemm 1024^3', lambda: LA.dgemm(LA.dgemm(Gd, Gd), Gd))
    C = 0.5 * (Gd + Gd.t())
    T('eigh_sym_top(1024, 30)', lambda: LA.eigh_sym_top(C, 30))
    T('whole fit', lambda: M.AlignMCCA(n_components=30, regs=0.5).fit(feats, labs))
import cProfile, pstats
m = M.AlignMCCA(n_components=30, regs=0.5)
m.fit(feats, labs); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); m.fit(feats, labs); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
