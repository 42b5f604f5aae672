"""A few launches of each big-tile GEMM form (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF
dev = 'cuda'
R = 40960
A = torch.randn(R, 1024, device=dev); B = torch.randn(1536, 1024, device=dev); C = torch.empty(R, 1536, device=dev)
for _ in range(4): XF.gemm_nt(A, B, C, R, 1536, 1024)
A2 = torch.randn(R, 1536, device=dev); B2 = torch.randn(1536, 1024, device=dev); C2 = torch.empty(R, 1024, device=dev)
for _ in range(4): XF.gemm_nn(A2, B2, C2, R, 1024, 1536)
C3 = torch.empty(1536, 1024, device=dev); bias = torch.empty(1536, device=dev)
for _ in range(4): XF.gemm_tn_grouped([XF.tn_problem(A2, A, C3, 1536, 1024, R, colsum_out=bias)], dev)
torch.cuda.synchronize()
