"""Run-to-run determinism probe: N train steps from a fixed seed, print parameter/loss checksums."""
import hashlib, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from cross_patient_speech_decoding_amd.nn_models.models import Seq2SeqRNN
from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW, seed_everything
from cross_patient_speech_decoding_amd.nn_models import functional as XF

def run(H, B, C, steps=4, drop=0.3):
    seed_everything(7)
    XF._DROP_COUNTER[0] = 0
    m = Seq2SeqRNN(C, 60, H, 10, 2, 1, 10, 10, 0, drop, drop, 'gru', 1e-3, 1e-5).cuda()
    opt = FlatAdamW(m, lr=1e-3, max_norm=0.5)
    x = torch.randn(B, 200, C).cuda(); y = torch.randint(1, 10, (B, 3)).cuda()
    m.train()
    out = []
    for s in range(steps):
        opt.zero_grad()
        loss = m.training_step((x, y), s)
        loss.backward()
        g = hashlib.md5(opt.flat_g.cpu().numpy().tobytes()).hexdigest()[:8]
        opt.step()
        out.append((float(loss), g, hashlib.md5(opt.flat_p.cpu().numpy().tobytes()).hexdigest()[:8]))
    return out

for H, B, C in ((64, 150, 48), (128, 2048, 64)):
    a = run(H, B, C); b = run(H, B, C)
    print(H, B, 'same-process identical:', a == b)
    for s, (u, v) in enumerate(zip(a, b)):
        print('  step', s, u, v if u != v else '')
