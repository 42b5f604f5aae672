"""One GEMM shape launched a few times: target for rocprofv3 --pmc passes.  argv: form M N K [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF
form, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = 'cuda'
torch.manual_seed(0)
C = torch.empty(M, N, device=dev)
if form == 'nt':
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
    fn = lambda: XF.gemm_nt(A, B, C, M, N, K)
elif form == 'nn':
    A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev)
    fn = lambda: XF.gemm_nn(A, B, C, M, N, K)
else:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
    bias = torch.empty(M, device=dev)
    fn = lambda: XF.gemm_tn_grouped([XF.tn_problem(A, B, C, M, N, K, colsum_out=bias)], dev)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
