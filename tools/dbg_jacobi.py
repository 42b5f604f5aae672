import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.alignment import _linalg as LA
from cross_patient_speech_decoding_amd._lib import call
rng = np.random.default_rng(0)
for n in (1024, 128):
    B = rng.standard_normal((n, 300)); C = torch.from_numpy(B @ B.T / 300 + 0.5 * np.eye(n)).cuda()
    mu = float(C.abs().sum(1).max()); W = (C + 2 * mu * torch.eye(n, dtype=torch.float64, device='cuda')).contiguous()
    off = torch.zeros(1, dtype=torch.float64, device='cuda')
    for s in range(22):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        call('xps_jacobi_sweeps_f64', W.data_ptr(), W.stride(0), None, n, n, n, 1, off.data_ptr(), None, 0, torch.cuda.current_stream().cuda_stream)
        o = off.item(); print(n, 'sweep', s + 1, 'off %.3e' % o, '%.1f ms' % ((time.perf_counter() - t0) * 1e3))
