import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF
M, N, K = 384, 128, 40960
A = torch.randn(K, M, device='cuda'); B = torch.randn(K, N, device='cuda'); C = torch.empty(M, N, device='cuda'); bias = torch.empty(M, device='cuda')
for _ in range(6):
    XF.gemm_tn_grouped([XF.tn_problem(A, B, C, M, N, K, colsum_out=bias)], 'cuda')
A2 = torch.randn(40960, 256, device='cuda'); W = torch.randn(384, 256, device='cuda'); Y = torch.empty(40960, 384, device='cuda')
for _ in range(6):
    XF.gemm_nt(A2, W, Y, 40960, 384, 256)
torch.cuda.synchronize(); print('done')
