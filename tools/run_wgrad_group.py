"""One-shot driver for counter runs: the grouped weight-gradient launch of configs[1] encoder layer 1 (6 problems,
K = 40960 rows, H = 128) a few times on random operands -- the launch bench.py's configs1 roofline times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.nn_models import functional as XF
Tp, B, H = 20, 2048, 128
K, In = Tp * B, 2 * H
dev = 'cuda'
torch.manual_seed(0)
dgi = torch.randn(2, K, 3 * H, device=dev) * 0.1
dghn = torch.randn(2, K, H, device=dev) * 0.1
x = torch.randn(K, In, device=dev)
y_ext = torch.randn(Tp + 2, B, 2 * H, device=dev)
outs = [(torch.empty(3 * H, In, device=dev), torch.empty(3 * H, device=dev), torch.empty(3 * H, H, device=dev),
         torch.empty(3 * H, device=dev)) for _ in range(2)]
for _ in range(8):
    probs = []
    for d in range(2):
        dw_ih, db_ih, dw_hh, db_hh = outs[d]
        hprev = y_ext.view(-1)[(0 if d == 0 else 2) * B * 2 * H + d * H:]
        probs.append(XF.tn_problem(dgi[d], hprev, dw_hh, 2 * H, H, K, ra=XF.rowmap(3 * H), rb=XF.rowmap(2 * H), rc=XF.rowmap(H), colsum_out=db_hh))
        probs.append(XF.tn_problem(dghn[d], hprev, dw_hh[2 * H:], H, H, K, ra=XF.rowmap(H), rb=XF.rowmap(2 * H), rc=XF.rowmap(H), colsum_out=db_hh[2 * H:]))
        probs.append(XF.tn_problem(dgi[d], x, dw_ih, 3 * H, In, K, colsum_out=db_ih))
    XF.gemm_tn_grouped(probs, dev)
torch.cuda.synchronize()
print('done')
