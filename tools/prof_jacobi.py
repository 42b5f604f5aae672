"""Target for rocprofv3: one eigh_sym_top(1024, 30) (per-round Jacobi launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cross_patient_speech_decoding_amd.alignment import _linalg as LA
rng = np.random.default_rng(0)
B = rng.standard_normal((1024, 300)); C = torch.from_numpy(B @ B.T / 300 + 0.5 * np.eye(1024)).cuda()
for _ in range(2):
    w, V = LA.eigh_sym_top(C, 30)
torch.cuda.synchronize()
