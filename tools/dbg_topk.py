"""Spectrum of the 8-view MCCA matrix and the convergence history of the subspace iteration on it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
import importlib
M = importlib.import_module('cross_patient_speech_decoding_amd.alignment.AlignMCCA')
from cross_patient_speech_decoding_amd.alignment import _linalg as LA
P, N = int(os.environ.get('VIEWS', 8)), 2048
pats = [make_patient(p, N, T=200, C=128) for p in range(P)]
feats, labs = [p[0] for p in pats], [p[1] for p in pats]
captured = {}
orig = LA.eigh_sym_top
def spy(C, k, **kw):
    captured['C'] = C.clone(); captured['k'] = k
    return orig(C, k, **kw)
LA.eigh_sym_top = spy
M.LA.eigh_sym_top = spy
M.AlignMCCA(n_components=30, regs=0.5).fit(feats, labs)
C, k = captured['C'], captured['k']
w = np.linalg.eigvalsh(C.cpu().numpy())[::-1]
print('n', C.shape[0], 'k', k)
print('top 50:', np.array2string(w[:50], precision=4))
print('bottom 5:', w[-5:])
for kk in (10, 30):
    for tol in (2e-14, 1e-12):
        st = {}
        orig(C, kk, tol=tol)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ww, V = orig(C, kk, tol=tol, stats=st)
        torch.cuda.synchronize()
        print(f'k {kk} tol {tol}: {1e3 * (time.perf_counter() - t0):.1f} ms', st, float(np.abs(ww - w[:kk]).max()))
torch.cuda.synchronize(); t0 = time.perf_counter()
LA._eigh_sym_top_full(C, 30); torch.cuda.synchronize()
print(f'full Jacobi: {1e3 * (time.perf_counter() - t0):.1f} ms')
G = 0.5 * (captured['C'] + captured['C'].t())
import importlib
Z = None
# the raw Gram matrix of the views (PCA-like spectrum: 16 huge values over a flat bulk)
feat = torch.cat([LA.to_device(f).reshape(-1, f.shape[-1]).to(LA.F64) for f in feats[:4]], dim=1)[:20000].contiguous()
Gm = LA.xcov(feat, None, LA.col_mean(feat)); Gm = 0.5 * (Gm + Gm.t())
wg = np.linalg.eigvalsh(Gm.cpu().numpy())[::-1]
for kk in (10, 30):
    st = {}
    orig(Gm, kk)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ww, V = orig(Gm, kk, stats=st)
    torch.cuda.synchronize()
    print(f'gram n={Gm.shape[0]} k {kk}: {1e3 * (time.perf_counter() - t0):.1f} ms', st, float(np.abs(ww - wg[:kk]).max() / wg[0]))
