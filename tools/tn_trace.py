"""Target for `rocprofv3 --kernel-trace`: each TN shape 6 times (kernel durations without host effects)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF
dev = 'cuda'
for name, M, N, K in [('dWih L1', 384, 256, 40960), ('dWhh', 384, 128, 40960), ('dWih L0', 384, 100, 40960), ('conv', 100, 640, 40960),
                      ('dWih L1 half K', 384, 256, 20480), ('dWih L1 2K', 384, 256, 81920)]:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
    bias = torch.empty(M, device=dev)
    for _ in range(6):
        XF.gemm_tn_grouped([XF.tn_problem(A, B, C, M, N, K, colsum_out=bias)], dev)
    torch.cuda.synchronize()
