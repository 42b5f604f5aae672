"""Stage times of PCA(0.95).fit and AlignCCA.fit at bench.py's shapes (resident inputs).  python tools/prof_pca_cca.py"""
import os, sys, time, collections, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
from cross_patient_speech_decoding_amd import alignment as A
pats = [make_patient(p, 2048, T=200, C=128) for p in range(2)]
Xd = [torch.from_numpy(x).cuda() for x, _ in pats]; ys = [y for _, y in pats]
def timed(fn, n=5):
    fn(); ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[n // 2]
pca = A.PCA(0.95)
print('pca fit ms', timed(lambda: pca.fit(Xd[0].reshape(-1, 128))))
Z = [A.PCA(0.95).fit(x.reshape(-1, 128)).transform(x.reshape(-1, 128)).reshape(2048, 200, -1) for x in Xd]
al = A.AlignCCA()
print('cca fit ms', timed(lambda: al.fit(Z[0], Z[1], ys[0], ys[1])))
for name, fn in (('pca', lambda: pca.fit(Xd[0].reshape(-1, 128))), ('cca', lambda: al.fit(Z[0], Z[1], ys[0], ys[1]))):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): fn()
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(name); print('\n'.join(s.getvalue().splitlines()[4:34]))
