"""Config-5 family: CTC training step and full-sequence inference of RealtimeRNNModel at C=128, T=200, H=128, L=2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cross_patient_speech_decoding_amd.realtime_sim import RealtimeRNNModel
from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
B, T, C, H, L, ncls = int(os.environ.get('B', 2048)), 200, 128, 128, 2, 11
torch.manual_seed(0)
m = RealtimeRNNModel(14 * C, H, L, ncls, dropout=0.3).cuda().train()
opt = FlatAdamW(m, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
x = torch.randn(B, T, C).cuda(); tg = torch.randint(1, ncls, (B, 3)).cuda()
il = torch.full((B,), T); tl = torch.full((B,), 3)
def step():
    opt.zero_grad(); loss = m.training_step((x, tg, il, tl), 0); loss.backward(); opt.step(); return loss
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
nw = m.n_windows(T)
fl = 3 * (2 * nw * 3 * H * (14 * C + H) + 2 * nw * 3 * H * (H + H) + 2 * nw * H * ncls)
print(f'CTC train step B={B}: {dt * 1e3:.3f} ms  {B / dt:,.0f} trials/s  {B * fl / dt / 1e12:.1f} TFLOP/s  loss {float(loss):.3f}')
m.eval()
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f'inference (full sequences) B={B}: {dt * 1e3:.3f} ms  {B / dt:,.0f} trials/s')
