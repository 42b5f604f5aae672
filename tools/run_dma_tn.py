"""A few launches of the weight-gradient GEMM dW_ih of configs[3]'s layer 1 (1536 x 1024 x 40960) on XPS_FMT_SPLIT4 operands:
the LDS-DMA k loop of gemm_big_tn_kernel (XPS_GEMM_DMA=0: the register-staged loop) -- for rocprofv3 --pmc / --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd._lib import rowmap
from cross_patient_speech_decoding_amd.nn_models import functional as XF
dev = 'cuda'
K, M, N = 40960, 1536, 1024
torch.manual_seed(0)
A = XF.split4(torch.randn(K, M, device=dev) * 0.1)
B = XF.split4(torch.randn(K, N, device=dev))
C = torch.empty(M, N, device=dev); cs = torch.empty(M, device=dev)
for _ in range(6):
    XF.gemm_tn_grouped([XF.tn_problem(A, B, C, M, N, K, ra=rowmap(M, fmt=1), rb=rowmap(N, fmt=1), colsum_out=cs)], dev)
torch.cuda.synchronize()
