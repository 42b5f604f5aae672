"""Alignment timing at the north-star shape: PCA(0.95) + CCA fit + transform per patient pair, 8-view MCCA,
joint PCA; HIP path vs the CPU oracle (numpy/LAPACK/sklearn) on the same host."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
from cross_patient_speech_decoding_amd import alignment as A
from cross_patient_speech_decoding_amd.alignment import _linalg as LA
from oracle import align_oracle as ao, mcca_oracle as mo

P = int(os.environ.get('P', '8')); N = int(os.environ.get('N', '2048'))
pats = [make_patient(p, N, T=200, C=128) for p in range(P)]
def sync(): torch.cuda.synchronize()
def t(fn, n=3):
    fn(); sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    sync(); return (time.perf_counter() - t0) / n
res = {}
X0, y0 = pats[0]; X1, y1 = pats[1]
# k1: condition means
res['cnd_avg_ms'] = t(lambda: A.cnd_avg(X0, A.label2str(y0))) * 1e3
Xd = LA.to_device(X0); keys = A.label2str(y0); uniq, order, start = LA.condition_index(keys)
dt = t(lambda: LA.cnd_avg_device(Xd, order, start), 10)
res['cnd_avg_kernel_ms'] = dt * 1e3; res['cnd_avg_GBps'] = X0.nbytes / dt / 1e9
# PCA
pca = A.PCA(0.95)
res['pca_fit_ms (409600x128 f32, incl. upload)'] = t(lambda: pca.fit(X0.reshape(-1, 128))) * 1e3
X2 = Xd.reshape(-1, 128); mean = LA.col_mean(X2)
dt = t(lambda: LA.xcov(X2, None, mean), 10)
res['xcov_kernel_ms'] = dt * 1e3; res['xcov_TFLOPs_f64'] = 2 * X2.shape[0] * 128 * 128 / dt / 1e12
w = t(lambda: LA.eigh_psd(LA.xcov(X2, None, mean)))
res['xcov+jacobi_eigh128_ms'] = w * 1e3
Z0 = pca.transform(X0.reshape(-1, 128)).reshape(N, 200, -1); Z1 = A.PCA(0.95).fit_transform(X1.reshape(-1, 128)).reshape(N, 200, -1)
al = A.AlignCCA()
res['cca_fit_ms'] = t(lambda: al.fit(Z0, Z1, y0, y1)) * 1e3
res['cca_transform_ms'] = t(lambda: al.transform(Z1)) * 1e3
res['latent_dims'] = [int(Z0.shape[-1]), int(Z1.shape[-1])]
t0 = time.perf_counter(); ref = ao.AlignCCAOracle().fit(Z0, Z1, y0, y1); res['cpu_cca_fit_ms'] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); ao.pca_fit(X0.reshape(-1, 128), 0.95); res['cpu_pca_fit_ms'] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); ao.cnd_avg(X0, ao.label_keys(y0)); res['cpu_cnd_avg_ms'] = (time.perf_counter() - t0) * 1e3
# MCCA over P raw views (D = P*128)
feats, labs = [p[0] for p in pats], [p[1] for p in pats]
m = A.AlignMCCA(n_components=30, regs=0.5)
res[f'mcca_fit_ms (P={P}, D={P*128})'] = t(lambda: m.fit(feats, labs), 1) * 1e3
t0 = time.perf_counter(); mo.get_mcca_transforms(feats, labs, 30, 0.5, 1); res['cpu_mcca_fit_ms'] = (time.perf_counter() - t0) * 1e3
res['mcca_transform_ms (one view)'] = t(lambda: m.transform(feats[0], idx=0)) * 1e3
print(json.dumps(res, indent=1))
