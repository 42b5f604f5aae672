"""Print how far the bf16x3 training step is from the reference golden (tiny model, tf1)."""
import ast, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import numpy as np, torch
from cross_patient_speech_decoding_amd._lib import lib
from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
from test_gpu_seq2seq import build_hip, NOISE_KEY
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
for mode in (0, 1):
    lib().xps_set_gemm_precision(mode)
    for name in ('tiny', 'tiny_relu_dec2', 'cfg2'):
        for tag, coin in (('tf1', True), ('tf0', False)):
            g = np.load(os.path.join(gd, f'seq2seq_{name}.npz'))
            cfg = ast.literal_eval(str(g['cfg']))
            m = build_hip(cfg, int(g['seed'])).train()
            x, y = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['y']).cuda()
            opt = FlatAdamW(m, lr=1e-3, weight_decay=1e-5, max_norm=0.5)
            logits = m(x, y, coins=[coin] * 3)
            loss = m.criterion(logits.view(-1, 9), y.view(-1))
            opt.zero_grad(); loss.backward()
            le = np.abs(logits.detach().cpu().numpy() - g[f'{tag}_logits']).max()
            opt.step()
            worst = (0, '')
            if f'{tag}_grad/decoder.fc_out.weight' in g:
                for k, p in dict(m.named_parameters()).items():
                    ref = g[f'{tag}_grad/{k}']
                    d = np.abs(p.grad.cpu().numpy() - ref)
                    tol = 2e-3 * np.abs(ref) + 2e-5 * max(1.0, np.abs(ref).max())
                    r = (d / tol).max()
                    if r > worst[0]: worst = (r, k, float(d.max()), float(np.abs(ref).max()))
            print(f'mode {mode} {name} {tag}: loss rel {abs(loss.item() - float(g[tag + "_loss"])) / float(g[tag + "_loss"]):.2e} logits {le:.2e} '
                  f'gnorm rel {abs(float(opt.grad_norm()) - float(g[tag + "_gnorm"])) / float(g[tag + "_gnorm"]):.2e} worst grad/tol {worst}', flush=True)
