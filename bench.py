#!/usr/bin/env python
"""Headline benchmark: ECoG trials/s of seq2seq-GRU training (forward + backward + clip + AdamW)
on N MI355X GPUs of one node, one process per GPU over RCCL.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: single-patient seq2seq GRU, C=64 channels, T=200 samples,
Conv1d(k=s=10, F=100) -> T'=20, bidirectional 2-layer GRU encoder H=128, 1-layer GRU decoder,
3 x 9-way phoneme outputs; one "step" = one full-batch optimisation step over 2048 trials per GPU
(the reference trains full-batch: batch_size 5000 > dataset, scripts/train_seq2seq.py:100-113),
dropout 0.3/0.3 and teacher forcing 0.5 as in the reference script.  Weak scaling: every rank
holds its own 2048-trial shard; BatchNorm statistics and the flat gradient are all-reduced.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the CPU oracle = torch.nn restatement of the reference, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(in_channels=64, n_filters=100, hidden_size=128, num_classes=9, n_enc_layers=2, n_dec_layers=1,
           kernel_size=10, stride=10, T=200, trials_per_gpu=2048)
F32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2 / 16x16x4, 64 FLOP/clk/SIMD
BF16_MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16, 32 cycles)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def train_flops_per_trial(c):
    """BASELINE.md section 2 bookkeeping: 3 x forward FLOPs."""
    Tp = (c['T'] - c['kernel_size']) // c['stride'] + 1
    H, F = c['hidden_size'], c['n_filters']
    fwd = 2 * Tp * F * c['in_channels'] * c['kernel_size']
    n_in = F
    for _ in range(c['n_enc_layers']):
        fwd += 2 * (2 * 3 * H * (n_in + H)) * Tp
        n_in = 2 * H
    fwd += 3 * (c['n_dec_layers'] * 2 * 3 * H * 2 * H + 2 * H * c['num_classes'])
    return 3 * fwd


def make_data(rank, c):
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    X, y_full = make_patient(rank, c['trials_per_gpu'], T=c['T'], C=c['in_channels'])
    return torch.from_numpy(X), torch.from_numpy(y_full - 1)           # labels 0..8 (train_seq2seq.py:95)


def build_model(c, dropout=0.3):
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    return Seq2SeqRNN(c['in_channels'], c['n_filters'], c['hidden_size'], c['num_classes'], c['n_enc_layers'],
                      c['n_dec_layers'], c['kernel_size'], c['stride'], 0, dropout, dropout, 'gru', 1e-4, 1e-5,
                      activation=False, decay_iters=500)


def cpu_baseline(c):
    """The CPU oracle (plain torch.nn restatement of the reference, pinned to reference goldens) on the SAME workload:
    the full batch of 2048 trials per step (the reference trains full-batch), fp32, forward + backward + clip + AdamW.
    All cores of the GPU's host share (at most 16: oversubscribed small GEMMs run slower): 3 warm-up + 10 timed steps,
    median; and ONE thread: 1 warm-up + 3 timed steps, median (a bounded sample: a step takes seconds there)."""
    from oracle.seq2seq_oracle import Seq2SeqOracle, train_step
    torch.manual_seed(0)
    m = Seq2SeqOracle(c['in_channels'], c['n_filters'], c['hidden_size'], c['num_classes'], c['n_enc_layers'],
                      c['n_dec_layers'], c['kernel_size'], c['stride'], 0, 0.3, 0.3, learning_rate=1e-4,
                      l2_reg=1e-5, activation=False, decay_iters=500)
    opt, _ = m.make_optimizer()
    X, y = make_data(0, c)
    B = X.shape[0]

    def run(threads, warm, timed):
        torch.set_num_threads(threads)
        for _ in range(warm):
            train_step(m, opt, X, y, coins=[True, False, True])
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            train_step(m, opt, X, y, coins=[True, False, True])
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], sum(ts)
    threads = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    med_all, _ = run(threads, 3, 10)
    med_one, _ = run(1, 1, 3)
    el = time.perf_counter() - t0
    return {'value': round(B / med_all, 1), 'unit': 'trials/s', 'cores': threads, 'kind': 'port',
            'sample': f'full batch of {B} trials per step (fwd+bwd+clip+AdamW), same architecture/T/C, fp32 torch.nn CPU oracle: '
                      f'{threads} threads, 3 warm-up + 10 timed steps, median {med_all * 1e3:.0f} ms/step; whole baseline {el:.0f} s',
            'one_thread': {'value': round(B / med_one, 1), 'unit': 'trials/s', 'cores': 1,
                           'sample': f'1 warm-up + 3 timed steps, median {med_one * 1e3:.0f} ms/step'}}


def _event_time(fn, iters=20, warm=3):
    """Average launch duration by HIP events on torch's current stream (= the stream the C ABI launches on)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def capture_dominant_launch(step_fn):
    """Run ONE more training step with the grouped weight-gradient entry point wrapped: returns the problem list of
    its largest launch (by FLOPs) and the tensors that keep its operands alive -- the step's own operands."""
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    seen = []
    orig_group, orig_launch = XF.gemm_tn_grouped, XF._launch_weight_grads
    holder = {'tensors': ()}

    def launch(fn, device, tensors, direct):
        holder['tensors'] = tensors
        return orig_launch(fn, device, tensors, direct)

    def group(problems, device, stream=None):
        fl = sum(2.0 * q.M * q.N * q.K for q in problems)
        seen.append((fl, list(problems), holder['tensors']))
        return orig_group(problems, device, stream)
    XF.gemm_tn_grouped, XF._launch_weight_grads = group, launch
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        XF.gemm_tn_grouped, XF._launch_weight_grads = orig_group, orig_launch
    return max(seen, key=lambda t: t[0]) if seen else None


def time_dominant_kernel(model, c, captured=None):
    """Roofline of the kernel with the largest share of the step (profiles/round1/r1c_*): the fp32-MFMA
    GEMM tile kernel, measured on its largest single launch = the grouped weight-gradient GEMM of encoder
    layer 1 (6 problems, K = T'*B rows).  Algorithmic FLOPs = sum 2*M*N*K over the group.  The fused GRU
    recurrence (second largest) is reported beside it."""
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    Tp = (c['T'] - c['kernel_size']) // c['stride'] + 1
    B, H = c['trials_per_gpu'], c['hidden_size']
    K, In = Tp * B, 2 * H
    dev = 'cuda'
    dgi = torch.randn(2, K, 3 * H, device=dev) * 0.1
    dghn = torch.randn(2, K, H, device=dev) * 0.1
    x = torch.randn(K, In, device=dev)
    y_ext = torch.randn(Tp + 2, B, 2 * H, device=dev)
    outs = [(torch.empty(3 * H, In, device=dev), torch.empty(3 * H, device=dev), torch.empty(3 * H, H, device=dev),
             torch.empty(3 * H, device=dev)) for _ in range(2)]

    def launch():
        probs = []
        for d in range(2):
            dw_ih, db_ih, dw_hh, db_hh = outs[d]
            hprev = y_ext.view(-1)[(0 if d == 0 else 2) * B * 2 * H + d * H:]
            probs.append(XF.tn_problem(dgi[d], hprev, dw_hh, 2 * H, H, K, ra=XF.rowmap(3 * H), rb=XF.rowmap(2 * H),
                                       rc=XF.rowmap(H), colsum_out=db_hh))
            probs.append(XF.tn_problem(dghn[d], hprev, dw_hh[2 * H:], H, H, K, ra=XF.rowmap(H), rb=XF.rowmap(2 * H),
                                       rc=XF.rowmap(H), colsum_out=db_hh[2 * H:]))
            probs.append(XF.tn_problem(dgi[d], x, dw_ih, 3 * H, In, K, colsum_out=db_ih))
        XF.gemm_tn_grouped(probs, dev)
    dur_rand = _event_time(launch)
    flops = 2 * (2 * K * (3 * H * H + 3 * H * In))
    dur = dur_rand
    operands = 'random normal operands of the step\'s shapes'
    if captured is not None and abs(captured[0] - flops) < 1e-6 * flops:
        # the SAME launch on the operands of a real backward pass (gradients and dropout-masked activations): the
        # matrix pipe draws less power on them than on dense random data and holds higher clocks
        probs_real = captured[1]
        dur = _event_time(lambda: XF.gemm_tn_grouped(probs_real, dev))
        operands = 'operands captured from a training step'
    ach = flops / dur / 1e12
    precision = XF.get_gemm_precision()
    # second: the fused GRU recurrence of one encoder layer (both directions, one launch)
    rnn = model.encoder.rnn
    w_hh = [rnn.weight_hh_l1.detach().contiguous(), rnn.weight_hh_l1_reverse.detach().contiguous()]
    b_hh = [rnn.bias_hh_l1.detach().contiguous(), rnn.bias_hh_l1_reverse.detach().contiguous()]
    gi = torch.randn(2, Tp, B, 3 * H, device=dev) * 0.5
    dur_gru = _event_time(lambda: XF._gru_forward(gi, w_hh, b_hh, None, Tp, B, H, 2, True))
    fl_gru = 2 * Tp * B * 2 * 3 * H * H
    by_gru = 4 * 2 * Tp * B * (3 * H + H + 4 * H)          # gi in, y + saved gates (r, z, n, q) out
    # HBM traffic of this launch from the PMC counters (collected offline with rocprofv3 --pmc, separate passes,
    # gfx950 FETCH_SIZE correction applied): profiles/round1/pmc_traffic.json
    traffic = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'round1', 'pmc_traffic.json')) as f:
            key = 'gemm_tn_grouped_kernel' + ('' if precision == 'fp32' else '_' + precision)
            traffic = json.load(f)[key]['hbm_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        pass
    # algorithmic HBM bytes of the launch: every operand row read once (dgi, dghn, x, h_prev of both directions),
    # gradients written once
    bytes_alg = 4 * (K * (2 * 3 * H + 2 * H + In + 2 * H) + 2 * (3 * H * H + 3 * H * In + 6 * H))
    common = {'launch_us': round(dur * 1e6, 1), 'flops_per_launch': flops, 'bytes_per_launch': bytes_alg, 'operands': operands,
              'traffic': traffic, 'precision': precision,
              'random_operands': {'launch_us': round(dur_rand * 1e6, 1)}}
    if precision == 'fp32':
        out = {'bound': 'mfma', 'kernel': 'gemm_tn_grouped_kernel (fp32 MFMA 128x128x16 tile; encoder layer-1 weight '
                                          'gradients, 6 problems in one launch, incl. its reduce pass)',
               'achieved': round(ach, 3), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
               'frac': round(ach / F32_MFMA_PEAK_TFLOPS, 4)}
        out.update(common)
        out['also'] = {'kernel': 'gru_fwd_resident_kernel<128> (encoder layer, both directions)',
                       'achieved': round(fl_gru / dur_gru / 1e12, 3), 'frac': round(fl_gru / dur_gru / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                       'launch_us': round(dur_gru * 1e6, 1), 'flops_per_launch': fl_gru}
        return out
    # bf16 split products: three bf16 MFMAs per algorithmic multiply-add.  The matrix pipe is no longer what binds
    # (issued MFMA work = 3 * flops is reported beside it); the launch is priced against HBM: algorithmic bytes / time.
    gbs = bytes_alg / dur / 1e9
    out = {'bound': 'hbm', 'kernel': 'gemm_tn_grouped_kernel<bf16x3> (128x128x16 tile, operands split hi/lo while staged, 3 bf16 '
                                     'MFMAs per product; encoder layer-1 weight gradients, 6 problems in one launch, incl. its reduce pass)',
           'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4)}
    out.update(common)
    out['mfma_side'] = {'algorithmic_tflops': round(ach, 2), 'issued_bf16_tflops': round(3 * ach, 2), 'peak': BF16_MFMA_PEAK_TFLOPS,
                        'issued_frac': round(3 * ach / BF16_MFMA_PEAK_TFLOPS, 4)}
    gru_gbs = by_gru / dur_gru / 1e9
    out['also'] = {'kernel': 'gru_fwd_resident_kernel<128, bf16x3> (encoder layer, both directions)', 'bound': 'hbm',
                   'achieved': round(gru_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gru_gbs / HBM_PEAK_GBS, 4),
                   'launch_us': round(dur_gru * 1e6, 1), 'bytes_per_launch': by_gru, 'flops_per_launch': fl_gru}
    return out


def _timed_steps(step, steps, warm):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _make_step(c, dev, rank, dropout=0.3):
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    torch.manual_seed(1234)
    model = build_model(c, dropout).to(dev)
    opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_norm=0.5)
    X, y = make_data(rank, c)
    X, y = X.to(dev), y.to(dev)
    model.train()
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    one = XF.unit_gradient(dev)

    def step():
        opt.zero_grad()
        logits = model(X, y, teacher_forcing_ratio=0.5)
        loss = model.criterion(logits.view(-1, c['num_classes']), y.view(-1))
        loss.backward(one)
        opt.step()
        return loss
    return model, step


def north_star_shard(dev, steps=10, warm=10):
    """Second record: the per-GPU shard of configs[3] (8-patient MCCA + bidirectional 2-layer GRU, H = 512, on 8 GPUs): 2048
    trials per GPU and step of the aligned d = 30 latent input, F = 100, k = s = 10 (T' = 20), enc 2 x bi-GRU H = 512, dec
    1 x GRU.  Same step function as the headline (dropout 0.3, teacher forcing 0.5, clip 0.5, AdamW), single GPU, no
    collective; the recurrence runs the cluster-persistent kernels (csrc/xps_gru_cluster.hip), timed alone beside it."""
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    c = dict(CFG, in_channels=30, hidden_size=512)
    model, step = _make_step(c, dev, 0)
    dt = _timed_steps(step, steps, warm)
    XF.check_gru_status()
    fl = train_flops_per_trial(c)
    B, H, Tp = c['trials_per_gpu'], 512, 20
    rnn = model.encoder.rnn
    w_hh = [rnn.weight_hh_l1.detach().contiguous(), rnn.weight_hh_l1_reverse.detach().contiguous()]
    b_hh = [rnn.bias_hh_l1.detach().contiguous(), rnn.bias_hh_l1_reverse.detach().contiguous()]
    gi = torch.randn(2, Tp, B, 3 * H, device=dev) * 0.5
    dy = torch.randn(Tp, B, 2 * H, device=dev) * 0.1
    t_f = _event_time(lambda: XF._gru_forward(gi, w_hh, b_hh, None, Tp, B, H, 2, True), iters=10)
    y_ext, saved = XF._gru_forward(gi, w_hh, b_hh, None, Tp, B, H, 2, True)
    t_b = _event_time(lambda: XF._gru_backward(dy, None, y_ext, saved, w_hh, Tp, B, H, 2, False), iters=10)
    XF.check_gru_status()
    by_f = 4 * 2 * Tp * B * (3 * H + H + 4 * H)              # gi in, h + saved gates (r, z, n, q) out
    by_b = 4 * 2 * Tp * B * (4 * H + H + H + 3 * H + H)      # saved gates, dy, h_prev in; dgi, dghn out
    fl_rec = 2.0 * 2 * Tp * B * 3 * H * H
    traffic = {}
    try:                                   # HBM-side bytes per launch from the PMC passes kept under profiles/round2 (collected offline)
        with open(os.path.join(ROOT, 'profiles', 'round2', 'pmc_traffic.json')) as f:
            pj = json.load(f)
        sfx = '_bf16x3' if XF.get_gemm_precision() == 'bf16x3' else ''
        traffic = {'bwd': pj['gru_cluster_bwd_kernel' + sfx]['hbm_bytes_per_launch'], 'fwd': pj['gru_cluster_fwd_kernel' + sfx]['hbm_bytes_per_launch']}
    except (OSError, KeyError, ValueError):
        pass
    return {
        'workload': 'configs[3] per-GPU shard: 8-patient MCCA-aligned input (d = 30), enc 2x bi-GRU H=512, dec 1x GRU, T=200 '
                    "(T'=20), F=100, 2048 trials per GPU and step, dropout 0.3, teacher forcing 0.5, clip 0.5, AdamW; 1 GPU, no collective",
        'value': round(B / dt, 1), 'unit': 'trials/s', 'ms_per_step': round(dt * 1e3, 3), 'steps': steps, 'warmup': warm,
        'train_mflop_per_trial': round(fl / 1e6, 2), 'model_tflops': round(B / dt * fl / 1e12, 2),
        'dtype': 'bf16x3' if XF.get_gemm_precision() == 'bf16x3' else 'f32',
        'roofline': {'bound': 'hbm', 'kernel': 'gru_cluster_bwd_kernel (BPTT of one bidirectional H = 512 layer, 20 steps, ONE launch: '
                                               'W_hh resident in a 16-workgroup cluster, gate gradients exchanged in-kernel)',
                     'achieved': round(by_b / t_b / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(by_b / t_b / 1e9 / HBM_PEAK_GBS, 4),
                     'launch_us': round(t_b * 1e6, 1), 'bytes_per_launch': by_b, 'flops_per_launch': fl_rec, 'traffic': traffic.get('bwd'),
                     'also': {'kernel': 'gru_cluster_fwd_kernel (same layer, forward, gates saved)', 'bound': 'hbm',
                              'achieved': round(by_f / t_f / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': round(by_f / t_f / 1e9 / HBM_PEAK_GBS, 4), 'launch_us': round(t_f * 1e6, 1),
                              'bytes_per_launch': by_f, 'flops_per_launch': fl_rec, 'traffic': traffic.get('fwd')}}}


def alignment_record(dev):
    """Fourth record: latent alignment at the north-star shape (patients of 2048 trials x 200 samples x 128 channels, fp32,
    resident in HBM like the training inputs): fits / s of the per-patient PCA(0.95), of the pairwise CCA fit on the PCA latents
    (reference: AlignCCA inside process_aligner, datamodules.py:542-565) and of the 4-view MCCA fit (AlignMCCA.py:140-154);
    HIP events on the launch stream, median of 5."""
    import numpy as np
    from cross_patient_speech_decoding_amd import alignment as A
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    P = 4
    pats = [make_patient(p, 2048, T=200, C=128) for p in range(P)]
    Xd = [torch.from_numpy(x).to(dev) for x, _ in pats]
    ys = [y for _, y in pats]

    def med(fn, n=5):
        fn()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[n // 2]
    pca = A.PCA(0.95)
    t_pca = med(lambda: pca.fit(Xd[0].reshape(-1, 128)))
    Z = [A.PCA(0.95).fit(x.reshape(-1, 128)).transform(x.reshape(-1, 128)).reshape(2048, 200, -1) for x in Xd[:2]]
    al = A.AlignCCA()
    t_cca = med(lambda: al.fit(Z[0], Z[1], ys[0], ys[1]))
    t_tr = med(lambda: al.transform(Z[1]))
    m = A.AlignMCCA(n_components=30, regs=0.5)
    t_mcca = med(lambda: m.fit(Xd, ys), n=3)
    by = Xd[0].numel() * 4
    return {'workload': 'north-star patients (2048 trials x 200 x 128 ch fp32 = 210 MB each), inputs resident in HBM',
            'pca_fit': {'ms': round(t_pca * 1e3, 2), 'fits_per_s': round(1 / t_pca, 1), 'input_GB_per_s': round(by / t_pca / 1e9, 1)},
            'cca_fit': {'ms': round(t_cca * 1e3, 2), 'fits_per_s': round(1 / t_cca, 1), 'latent_dims': [int(Z[0].shape[-1]), int(Z[1].shape[-1])]},
            'cca_transform': {'ms': round(t_tr * 1e3, 2)},
            'mcca_fit_4_views': {'ms': round(t_mcca * 1e3, 2), 'fits_per_s': round(1 / t_mcca, 2), 'D': 4 * 128}}


def fp32_record(c, dev, steps=20, warm=30):
    """Third record: the headline workload with the matrix kernels in exact-fp32 MFMA mode (the reference's own arithmetic),
    priced against the 157.3 TFLOP/s fp32 matrix peak."""
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    old = XF.get_gemm_precision()
    XF.set_gemm_precision('fp32')
    try:
        _, step = _make_step(c, dev, 0)
        dt = _timed_steps(step, steps, warm)
    finally:
        XF.set_gemm_precision(old)
    fl = train_flops_per_trial(c)
    tf = c['trials_per_gpu'] / dt * fl / 1e12
    return {'dtype': 'f32', 'ms_per_step': round(dt * 1e3, 3), 'value': round(c['trials_per_gpu'] / dt, 1), 'unit': 'trials/s',
            'model_tflops': round(tf, 2), 'peak_tflops': F32_MFMA_PEAK_TFLOPS, 'frac': round(tf / F32_MFMA_PEAK_TFLOPS, 4),
            'steps': steps, 'warmup': warm}


def _set_affinity_all_threads(cores):
    """sched_setaffinity for every thread of this process (threads keep the mask they were created with)."""
    try:
        for tid in os.listdir('/proc/self/task'):
            try:
                os.sched_setaffinity(int(tid), cores)
            except OSError:
                pass
    except OSError:
        pass


def pin_host_threads(local_rank):
    """One trainer process per GPU, pinned to a few cores of its own (XPS_BENCH_PIN_CORES, default 4; 0 = leave the scheduler
    alone).  The enqueue path of a step needs ~1.0 ms of host time against ~1.05 ms of GPU time, so a main or autograd thread
    that migrates across a 256-core host shows up in a 20-step window: tools/jitter.py, 150 windows in one process:
    unpinned p90 / p97 / max = 1.124 / 1.347 / 1.638 ms per step (median 1.082), pinned to 2-4 cores 1.08 / 1.09 / 1.19-1.33.
    Returns the original mask (restored for the CPU-baseline leg, which wants all cores)."""
    full = os.sched_getaffinity(0)
    n = int(os.environ.get('XPS_BENCH_PIN_CORES', '4'))
    if n > 0:
        cores = sorted(full)
        sel = cores[local_rank * n:(local_rank + 1) * n]
        if len(sel) == n:
            _set_affinity_all_threads(set(sel))
    return full


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: 200 timed steps (~0.2 s) -- with 20 (a 22-ms window) one host-side hiccup of a few ms moved the headline by 10-25 %
    # (1.04-1.06 ms/step typical, single runs at 1.27-1.6 seen); the enqueue path leaves the host ~15 % of slack per step
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--headline-only', action='store_true', help='skip the configs[3] shard and fp32 sub-records')
    ap.add_argument('--precision', choices=['bf16x3', 'fp32'], default=None,
                    help='product precision of the matrix kernels (default: the library default, bf16x3)')
    ap.add_argument('--hidden', type=int, default=None, help='(exploration only) override the hidden size, e.g. 512 = cfg 4')
    ap.add_argument('--channels', type=int, default=None, help='(exploration only) override the input channels')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        raise SystemExit('launch multi-GPU runs with torch.distributed.run (one process per GPU)')
    full_affinity = pin_host_threads(local_rank)
    # rehearsal knobs (1-GPU box): XPS_BENCH_BACKEND=gloo XPS_BENCH_ONE_DEVICE=1 put every rank on cuda:0
    backend = os.environ.get('XPS_BENCH_BACKEND', 'nccl')
    if os.environ.get('XPS_BENCH_ONE_DEVICE'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # rehearsal knob (1-GPU box): XPS_BENCH_FORCE_DP=1 (+ XPS_DP_SINGLE_RANK_COLLECTIVES=1) runs the data-parallel code path --
    # SyncBN exchanges, flat-gradient all-reduces, hooks -- on a ONE-rank RCCL communicator: what the DP host path costs per step
    force_dp = world == 1 and os.environ.get('XPS_BENCH_FORCE_DP') == '1'
    if force_dp:
        os.environ.setdefault('MASTER_PORT', '29577')
    if world > 1 or force_dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    if args.precision:
        XF.set_gemm_precision(args.precision)
    precision = XF.get_gemm_precision()
    c = dict(CFG)
    if args.hidden:
        c['hidden_size'] = args.hidden
    if args.channels:
        c['in_channels'] = args.channels
    explore = bool(args.hidden or args.channels)
    torch.manual_seed(1234)                      # identical initial weights on every rank
    model = build_model(c).to(dev)
    if world > 1 or force_dp:
        model.temporal_conv.process_group = dist.group.WORLD
        model.temporal_conv.global_batch = c['trials_per_gpu'] * world
    opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_norm=0.5, group=dist.group.WORLD if (world > 1 or force_dp) else None)
    X, y = make_data(rank, c)
    X, y = X.to(dev), y.to(dev)                  # inputs resident in HBM before the timed region
    torch.manual_seed(99)                        # the same teacher-forcing coins on every rank
    model.train()

    one = XF.unit_gradient(dev)

    def step():
        opt.zero_grad()
        logits = model(X, y, teacher_forcing_ratio=0.5)
        loss = model.criterion(logits.view(-1, c['num_classes']), y.view(-1))
        loss.backward(one)                       # a resident 1.0 as the root gradient (autograd would launch a fill)
        opt.step()
        return loss

    # untimed pre-warm: the first second of a fresh process runs slower on this pool (clock ramp / cold code pages: 2.9 vs
    # 1.8 ms/step measured); it is spent here, BEFORE the W warm-up steps of the contract, never inside the timed region
    # (a fixed STEP count, so that every rank of a data-parallel run issues the same collectives)
    for _ in range(int(os.environ.get('XPS_BENCH_PREWARM_STEPS', '400'))):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    import gc
    if os.environ.get('XPS_BENCH_COLLECT', '0') == '1':
        gc.collect()
    if os.environ.get('XPS_BENCH_GC', '0') != '1':
        gc.disable()                             # no collector pauses inside the timed region (they gate every rank under DP)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    final_loss = float(loss.item())

    if rank == 0:
        trials = c['trials_per_gpu'] * world * args.steps
        value = trials / el
        fl = train_flops_per_trial(c)
        out = {
            'metric': 'ECoG trials/sec seq2seq-RNN training', 'value': round(value, 1), 'unit': 'trials/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(el / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            # bf16x3: every fp32 operand split hi + lo in bf16, 3 bf16 MFMAs per product, fp32 accumulation and fp32 everywhere
            # else (results within 2e-6 of the fp32 reference goldens); fp32: fp32 MFMA
            'dtype': 'bf16x3' if precision == 'bf16x3' else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'configs[1]: single-patient seq2seq GRU, H=128, T=200 (T\'=20), C=64, F=100, '
                                   'enc 2x bi-GRU, dec 1x GRU, full-batch step of 2048 trials per GPU, '
                                   'dropout 0.3, teacher forcing 0.5, clip 0.5, AdamW',
                       'trials_per_gpu': c['trials_per_gpu'], 'global_batch': c['trials_per_gpu'] * world,
                       'parallelism': f'dp{world}', 'train_mflop_per_trial': round(fl / 1e6, 2),
                       'precision': ('bf16x3 = fp32 operands split hi + lo in bf16, 3 bf16 MFMAs per product, fp32 accumulate and '
                                     'fp32 everywhere else' if precision == 'bf16x3' else 'fp32 MFMA')},
            'model_tflops': round(value * fl / 1e12, 3), 'final_loss': round(final_loss, 5),
        }
        out['config']['collective'] = (f'RCCL (torch.distributed backend {backend!r}) world {world}: SyncBN statistics + flat gradient '
                                       f'all-reduce per step' if world > 1 else 'none (single process)')
        out['rccl_world'] = world if (world > 1 and backend == 'nccl') else (0 if world == 1 else None)
        if not explore:
            # the capture runs one more training step: single process only (under DP it would issue collectives alone)
            out['roofline'] = time_dominant_kernel(model, c, capture_dominant_launch(step) if world == 1 else None)
            if world == 1 and not args.headline_only:
                # sub-records never cost the headline line: a failure is reported in place of the record
                for key, fn in (('north_star_shard', lambda: north_star_shard(dev)),
                                ('fp32', (lambda: fp32_record(c, dev)) if precision != 'fp32' else None),
                                ('alignment', lambda: alignment_record(dev))):
                    if fn is None:
                        continue
                    try:
                        out[key] = fn()
                    except Exception as e:                                   # noqa: BLE001
                        out[key] = {'error': f'{type(e).__name__}: {e}'[:300]}
        else:
            out['config']['workload'] = f"EXPLORATION (not the headline config): H={c['hidden_size']}, C={c['in_channels']}"
        if world == 1 and not args.no_cpu_baseline:
            try:
                _set_affinity_all_threads(full_affinity)        # the CPU oracle gets every core of the host share
                out['cpu_baseline'] = cpu_baseline(c)
            except Exception as e:                                           # noqa: BLE001
                out['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'[:300]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
