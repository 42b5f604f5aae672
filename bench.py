#!/usr/bin/env python
"""Headline benchmark: ECoG trials/s of seq2seq-GRU training (forward + backward + clip + AdamW)
on N MI355X GPUs of one node, one process per GPU over RCCL.

    python bench.py --gpus 1 --steps 100 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[3], the configuration the metric and the north-star target are quoted on: the
cross-patient seq2seq GRU on the MCCA-aligned latent of 8 patients (d = 30 components), T = 200 samples,
Conv1d(k = s = 10, F = 100) -> T' = 20, bidirectional 2-layer GRU encoder H = 512, 1-layer GRU decoder, 3 x 9-way phoneme
outputs (reference shape: scripts/train_seq2seq.py:125-138, H = 500 there).  One "step" = one full-batch optimisation step
(the reference trains full-batch: batch_size 5000 > dataset, scripts/train_seq2seq.py:100-113) over 2048 trials per GPU --
8 patients x 2048 trials sharded over 8 GPUs -- with dropout 0.3 / 0.3, teacher forcing 0.5, clip 0.5, AdamW as in the
reference script (:171-189).  Weak scaling: every rank holds its own 2048-trial shard; BatchNorm statistics and the flat
gradient are all-reduced.  `--workload configs1` measures BASELINE.json configs[1] (single patient, C = 64, H = 128) instead;
the default run carries it as a sub-record.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel of the step, HIP-event timed) and `cpu_baseline` (the CPU
oracle = torch.nn restatement of the reference on the SAME full batch, bounded sample) plus sub-records (single process
only): `configs1`, `fp32`, `alignment` (8 views), `realtime` (config 5) and `dp_rehearsal`.
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

BASE = dict(n_filters=100, num_classes=9, n_enc_layers=2, n_dec_layers=1, kernel_size=10, stride=10, T=200, trials_per_gpu=2048)
WORKLOADS = {
    'configs3': dict(BASE, in_channels=30, hidden_size=512),
    'configs1': dict(BASE, in_channels=64, hidden_size=128),
}
WORKLOAD_TEXT = {
    'configs3': "configs[3] per-GPU shard: 8-patient MCCA-aligned input (d = 30), enc 2x bi-GRU H=512, dec 1x GRU, T=200 (T'=20), "
                'F=100, full-batch step of 2048 trials per GPU (8 patients x 2048 trials over 8 GPUs), dropout 0.3, teacher '
                'forcing 0.5, clip 0.5, AdamW',
    'configs1': "configs[1]: single-patient seq2seq GRU, H=128, T=200 (T'=20), C=64, F=100, enc 2x bi-GRU, dec 1x GRU, "
                'full-batch step of 2048 trials per GPU, dropout 0.3, teacher forcing 0.5, clip 0.5, AdamW',
}
PREWARM = {'configs3': 150, 'configs1': 400}
F32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2 / 16x16x4, 64 FLOP/clk/SIMD
BF16_MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16, 32 cycles)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# HBM-side traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): collected OFFLINE by tools/round4_evidence.sh with the
# kernels of the commit named in the file, not measured inside this run (`roofline.traffic_source` says so in the line)
PMC_FILES = [os.path.join(ROOT, 'profiles', r, 'pmc_traffic.json') for r in ('round4', 'round3')]


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


def make_data(rank, c, trials=None):
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    X, y_full = make_patient(rank, trials or c['trials_per_gpu'], T=c['T'], C=c['in_channels'])
    return torch.from_numpy(X), torch.from_numpy(y_full - 1)           # labels 0..8 (train_seq2seq.py:95)


def build_model(c, dropout=0.3):
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    return Seq2SeqRNN(c['in_channels'], c['n_filters'], c['hidden_size'], c['num_classes'], c['n_enc_layers'],
                      c['n_dec_layers'], c['kernel_size'], c['stride'], 0, dropout, dropout, 'gru', 1e-4, 1e-5,
                      activation=False, decay_iters=500)


def _pmc(key):
    """HBM-side bytes per launch from the PMC passes kept under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    passes, gfx950 FETCH x2 correction; collected offline by tools/round4_evidence.sh): the newest file that has the key."""
    for path in PMC_FILES:
        try:
            with open(path) as f:
                return json.load(f)[key]['hbm_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            continue
    return None


def _pmc_source(key):
    for path in PMC_FILES:
        try:
            with open(path) as f:
                d = json.load(f)
            if key in d:
                return (f"committed PMC profile {os.path.relpath(path, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                        f"FETCH x2 gfx950 correction; kernels of commit {d.get('_commit', 'see profiles/ README')}; not measured in this run)")
        except (OSError, ValueError):
            continue
    return None


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(c):
    """The CPU oracle (plain torch.nn restatement of the reference, pinned to reference goldens) on the SAME workload:
    the full batch of 2048 trials per step (the reference trains full-batch), fp32, forward + backward + clip + AdamW.
    Thread counts {16, 32, 64} (those the host share offers) are swept, 1 warm-up + 3 timed steps each, the best median is
    `value`; one thread is timed on a 256-trial batch (a full batch takes ~30 s per step there)."""
    from oracle.seq2seq_oracle import Seq2SeqOracle, train_step
    torch.manual_seed(0)
    m = Seq2SeqOracle(c['in_channels'], c['n_filters'], c['hidden_size'], c['num_classes'], c['n_enc_layers'],
                      c['n_dec_layers'], c['kernel_size'], c['stride'], 0, 0.3, 0.3, learning_rate=1e-4,
                      l2_reg=1e-5, activation=False, decay_iters=500)
    opt, _ = m.make_optimizer()
    X, y = make_data(0, c)
    B = X.shape[0]

    def run(threads, warm, timed, xb, yb):
        torch.set_num_threads(threads)
        for _ in range(warm):
            train_step(m, opt, xb, yb, coins=[True, False, True])
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            train_step(m, opt, xb, yb, coins=[True, False, True])
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]
    avail = len(os.sched_getaffinity(0))
    counts = sorted({min(n, avail) for n in (16, 32, 64)})
    t0 = time.perf_counter()
    sweep = {n: run(n, 1, 3, X, y) for n in counts}
    best = min(sweep, key=sweep.get)
    sub = 256
    med_one = run(1, 1, 2, X[:sub], y[:sub])
    el = time.perf_counter() - t0
    return {'value': round(B / sweep[best], 1), 'unit': 'trials/s', 'cores': best, 'kind': 'port',
            'cpu_model': cpu_model_name(), 'host_cores_available': avail, 'host_cores_total': os.cpu_count(),
            'sample': f'full batch of {B} trials per step (fwd+bwd+clip+AdamW), same architecture / T / d as the GPU workload, fp32 torch.nn '
                      f'CPU oracle; thread sweep {counts}: 1 warm-up + 3 timed steps each, median; best = {best} threads, '
                      f'{sweep[best] * 1e3:.0f} ms/step; whole baseline {el:.0f} s',
            'thread_sweep': {str(n): {'ms_per_step': round(t * 1e3, 1), 'trials_per_s': round(B / t, 1)} for n, t in sweep.items()},
            'one_thread': {'value': round(sub / med_one, 1), 'unit': 'trials/s', 'cores': 1,
                           'sample': f'{sub}-trial batch, 1 warm-up + 2 timed steps, median {med_one * 1e3:.0f} ms/step'}}


def _event_time(fn, iters=20, warm=3, prepare=None):
    """Average launch duration by HIP events on torch's current stream (= the stream the C ABI launches on).
    prepare: enqueued in front of EVERY timed launch and outside its event pair (e.g. the producer pass that writes the launch's
    input, so that the launch finds its input where the training step leaves it -- partly in the Infinity Cache -- instead of
    re-reading a buffer the previous identical launch has long pushed out)."""
    if prepare is not None:
        for _ in range(warm):
            prepare(); fn()
        pairs = []
        for _ in range(iters):
            prepare()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in pairs) / iters * 1e-3
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


def roofline_wgrad(model, c, captured=None):
    """Roofline of the dominant launch of the configs[1] step: the grouped weight-gradient GEMM of encoder layer 1 (6 problems
    dW_hh / dW_ih / biases of both directions, K = T'*B rows, ONE split-K launch + its reduce pass), re-issued on the operands
    captured from a real backward pass.  bf16x3 mode: priced against HBM (algorithmic bytes = every operand row of dgi, dghn,
    x, h_prev read once + gradients written), MFMA side beside it; fp32 mode: against the fp32 matrix peak.  `also` = the
    resident GRU forward kernel of one encoder layer."""
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
    rnn = model.encoder.rnn
    w_hh = [rnn.weight_hh_l1.detach().contiguous(), rnn.weight_hh_l1_reverse.detach().contiguous()]
    b_hh = [rnn.bias_hh_l1.detach().contiguous(), rnn.bias_hh_l1_reverse.detach().contiguous()]
    gi = torch.randn(2, Tp, B, 3 * H, device=dev) * 0.5
    dur_gru = _event_time(lambda: XF._gru_forward(gi, w_hh, b_hh, None, Tp, B, H, 2, True))
    fl_gru = 2 * Tp * B * 2 * 3 * H * H
    by_gru = 4 * 2 * Tp * B * (3 * H + H + 4 * H)          # gi in, y + saved gates (r, z, n, q) out
    traffic = _pmc('gemm_tn_grouped_kernel' + ('' if precision == 'fp32' else '_' + precision))
    bytes_alg = 4 * (K * (2 * 3 * H + 2 * H + In + 2 * H) + 2 * (3 * H * H + 3 * H * In + 6 * H))
    common = {'launch_us': round(dur * 1e6, 1), 'flops_per_launch': flops, 'bytes_per_launch': bytes_alg, 'operands': operands,
              'traffic': traffic, 'traffic_source': _pmc_source('gemm_tn_grouped_kernel' + ('' if precision == 'fp32' else '_' + precision)),
              'precision': precision,
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


def roofline_cluster(model, c, dev):
    """Roofline of the dominant kernel of the configs[3] step: the cluster-persistent BPTT launch of one bidirectional H = 512
    layer (20 steps, 2048 trials, ONE launch; two such launches per step), HIP-event timed on the launch stream.  HBM-bound
    by design: W_hh stays in registers, what must move is the saved gates, dy and h_prev in, dgi and dghn out (algorithmic
    bytes below; the in-kernel exchange of gate gradients is on-chip traffic and not counted as algorithmic).  `traffic` = the
    PMC-measured HBM-side bytes of the same launch; `also` = the forward launch and the largest 256-tile GEMM (MFMA side)."""
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    B, H = c['trials_per_gpu'], c['hidden_size']
    Tp = (c['T'] - c['kernel_size']) // c['stride'] + 1
    rnn = model.encoder.rnn
    w_hh = [rnn.weight_hh_l1.detach().contiguous(), rnn.weight_hh_l1_reverse.detach().contiguous()]
    b_hh = [rnn.bias_hh_l1.detach().contiguous(), rnn.bias_hh_l1_reverse.detach().contiguous()]
    gi = torch.randn(2, Tp, B, 3 * H, device=dev) * 0.5
    dy = torch.randn(Tp, B, 2 * H, device=dev) * 0.1
    # forward: in the step gi is written by the layer's input-projection GEMM right in front of the launch, and the launch time depends
    # on what ran before it (tools/probe_fwd_context.py, one box: 625 us back to back with itself on a stale gi -- what round 3's
    # probe timed --, 516 us behind a copy into gi, 497 us behind the projection GEMM = the 496 us of the profiled step, 485 us behind
    # a 1-GiB fill).  The probe therefore issues THAT GEMM (same operands' shapes and formats as functional.GRULayerFmtFn.forward)
    # in front of every timed launch, outside the event pair.
    import ctypes as C
    from cross_patient_speech_decoding_amd._lib import call as _call, rowmap as _rowmap
    In1 = 2 * H
    w_ih = [rnn.weight_ih_l1.detach().contiguous(), rnn.weight_ih_l1_reverse.detach().contiguous()]
    b_ih = [rnn.bias_ih_l1.detach().contiguous(), rnn.bias_ih_l1_reverse.detach().contiguous()]
    x1 = torch.randn(Tp, B, In1, device=dev) * 0.5
    sfmt = 1 if XF.split4_wanted(Tp, B, H, 2) else 0
    if sfmt:
        x1, w_ih = XF.split4(x1), [XF.split4(w) for w in w_ih]

    def projection():
        ra, rb, rc = _rowmap(In1, fmt=sfmt), _rowmap(In1, fmt=sfmt), _rowmap(3 * H)
        _call('xps_gemm_nt_multi_f32', XF._ptr(x1), C.byref(ra), XF._ptr_array(w_ih), C.byref(rb), XF._ptr_array([gi[d] for d in range(2)]),
              C.byref(rc), XF._ptr_array(b_ih), 2, Tp * B, 3 * H, In1, XF._stream())
    # (the launch form the step uses: in bf16x3 mode with the split4 image of y_ext as its exchange buffer -- it also writes those 184 MB,
    #  which `traffic` contains and the algorithmic bytes below do not)
    t_f = _event_time(lambda: XF.gru_forward_training_form(gi, w_hh, b_hh, Tp, B, H, 2), iters=10, prepare=projection)
    y_ext, saved = XF.gru_forward_training_form(gi, w_hh, b_hh, Tp, B, H, 2)
    split = XF.split4_wanted(Tp, B, H, 2)
    t_b = _event_time(lambda: XF._gru_backward(dy, None, y_ext, saved, w_hh, Tp, B, H, 2, False, split4=split), iters=10)
    torch.cuda.synchronize()
    XF.check_gru_status()
    by_f = 4 * 2 * Tp * B * (3 * H + H + 4 * H)              # gi in, h + saved gates (r, z, n, q) out
    by_b = 4 * 2 * Tp * B * (4 * H + H + H + 3 * H + H)      # saved gates, dy, h_prev in; dgi, dghn out
    fl_rec = 2.0 * 2 * Tp * B * 3 * H * H
    sfx = '_bf16x3' if XF.get_gemm_precision() == 'bf16x3' else ''
    tr_b, tr_f = _pmc('gru_cluster_bwd_kernel' + sfx), _pmc('gru_cluster_fwd_kernel' + sfx)
    # the largest GEMM of the step beside it: dW_ih of encoder layer 1 (3H x 2H x T'B), the 256-tile TN kernel incl. its reduce
    K, In = Tp * B, 2 * H
    from cross_patient_speech_decoding_amd._lib import rowmap
    A = torch.randn(K, 3 * H, device=dev) * 0.1
    Bm = torch.randn(K, In, device=dev)
    ofmt = 1 if split else 0                  # bf16x3 mode: both operands reach this GEMM as XPS_FMT_SPLIT4 images in the step (dgi from
    if ofmt:                                  # the BPTT kernel, the layer input from the dropout pass): the LDS-DMA k loop
        A, Bm = XF.split4(A), XF.split4(Bm)
    Cw, cb = torch.empty(3 * H, In, device=dev), torch.empty(3 * H, device=dev)
    t_g = _event_time(lambda: XF.gemm_tn_grouped([XF.tn_problem(A, Bm, Cw, 3 * H, In, K, ra=rowmap(3 * H, fmt=ofmt), rb=rowmap(In, fmt=ofmt),
                                                                colsum_out=cb)], dev), iters=20)
    fl_g = 2.0 * 3 * H * In * K
    issued = 3 if sfx else 1
    peak = BF16_MFMA_PEAK_TFLOPS if sfx else F32_MFMA_PEAK_TFLOPS
    out = {'bound': 'hbm', 'kernel': 'gru_cluster_bwd_kernel (BPTT of one bidirectional H = 512 layer, 20 steps, 2048 trials, ONE launch: '
                                     'W_hh^T resident in the registers of 16-workgroup clusters, gate gradients exchanged in-kernel)',
           'achieved': round(by_b / t_b / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(by_b / t_b / 1e9 / HBM_PEAK_GBS, 4),
           'launch_us': round(t_b * 1e6, 1), 'bytes_per_launch': by_b, 'flops_per_launch': fl_rec, 'traffic': tr_b,
           'traffic_ratio': round(tr_b / by_b, 3) if tr_b else None,
           'traffic_source': _pmc_source('gru_cluster_bwd_kernel' + sfx),
           'precision': XF.get_gemm_precision(), 'operands': 'random normal operands of the step\'s shapes',
           'also': [{'kernel': 'gru_cluster_fwd_kernel (same layer, forward, gates saved' + ('; bf16x3: the split4 image of y_ext is its exchange buffer, + 184 MB written' if sfx else '') + ')', 'bound': 'hbm',
                     'achieved': round(by_f / t_f / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(by_f / t_f / 1e9 / HBM_PEAK_GBS, 4), 'launch_us': round(t_f * 1e6, 1),
                     'bytes_per_launch': by_f, 'flops_per_launch': fl_rec, 'traffic': tr_f,
                     'traffic_ratio': round(tr_f / by_f, 3) if tr_f else None},
                    {'kernel': 'gemm_big_tn_kernel + reduce (dW_ih of encoder layer 1: 1536 x 1024 x 40960, 256 x 256 tiles' +
                               (', XPS_FMT_SPLIT4 operands: LDS-DMA k loop)' if ofmt else ')'), 'bound': 'mfma',
                     'achieved': round(issued * fl_g / t_g / 1e12, 1), 'peak': peak, 'unit': 'TFLOP/s (issued MFMA work)',
                     'frac': round(issued * fl_g / t_g / 1e12 / peak, 4), 'launch_us': round(t_g * 1e6, 1), 'flops_per_launch': fl_g,
                     'algorithmic_tflops': round(fl_g / t_g / 1e12, 1),
                     # context, not the denominator: a pure MFMA loop on random bf16 operands reaches 0.70 of `peak` on this
                     # chip (clock 2.33 -> 1.67 GHz under toggling inputs): profiles/round3/mfma_clock.txt
                     'mfma_rate_on_random_operands_tflops': 1760.0}]}
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


def _make_step(c, dev, rank, dropout=0.3, group=None, world=1):
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    torch.manual_seed(1234)                      # identical initial weights on every rank
    model = build_model(c, dropout).to(dev)
    if group is not None:
        model.temporal_conv.process_group = group
        model.temporal_conv.global_batch = c['trials_per_gpu'] * world
    opt = FlatAdamW(model, lr=1e-4, weight_decay=1e-5, max_norm=0.5, group=group)
    X, y = make_data(rank, c)
    X, y = X.to(dev), y.to(dev)                  # inputs resident in HBM before any timed region
    model.train()
    one = XF.unit_gradient(dev)

    def step():
        opt.zero_grad()
        logits = model(X, y, teacher_forcing_ratio=0.5)
        loss = model.criterion(logits.view(-1, c['num_classes']), y.view(-1))
        loss.backward(one)                       # a resident 1.0 as the root gradient (autograd would launch a fill)
        opt.step()
        return loss
    return model, step


def workload_record(name, dev, steps, warm, with_roofline=True):
    """A sub-record: the named workload in a fresh model, single GPU, the same step function as the headline."""
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    c = WORKLOADS[name]
    model, step = _make_step(c, dev, 0)
    dt = _timed_steps(step, steps, warm + PREWARM[name] // 4)
    XF.check_gru_status()
    fl = train_flops_per_trial(c)
    B = c['trials_per_gpu']
    out = {'workload': WORKLOAD_TEXT[name], 'value': round(B / dt, 1), 'unit': 'trials/s', 'ms_per_step': round(dt * 1e3, 3),
           'steps': steps, 'warmup': warm + PREWARM[name] // 4, 'train_mflop_per_trial': round(fl / 1e6, 2),
           'model_tflops': round(B / dt * fl / 1e12, 2), 'dtype': 'bf16x3' if XF.get_gemm_precision() == 'bf16x3' else 'f32'}
    if with_roofline:
        out['roofline'] = (roofline_wgrad(model, c, capture_dominant_launch(step)) if name == 'configs1'
                           else roofline_cluster(model, c, dev))
    return out


def fp32_record(name, dev, steps, warm):
    """The headline workload with the matrix kernels in exact-fp32 MFMA mode (the reference's own arithmetic): the
    precision-MATCHED number.  Whole step priced against the 157.3 TFLOP/s fp32 matrix peak; `roofline` = the same dominant-kernel
    record as the headline's, measured in this mode (configs[3]: the cluster BPTT launch -- same algorithmic bytes, fp32
    recurrent products -- with the forward launch and the largest fp32-MFMA GEMM beside it)."""
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    c = WORKLOADS[name]
    old = XF.get_gemm_precision()
    XF.set_gemm_precision('fp32')
    roof = None
    try:
        model, step = _make_step(c, dev, 0)
        dt = _timed_steps(step, steps, warm)
        XF.check_gru_status()
        if name == 'configs3':
            roof = roofline_cluster(model, c, dev)
    finally:
        XF.set_gemm_precision(old)
    fl = train_flops_per_trial(c)
    tf = c['trials_per_gpu'] / dt * fl / 1e12
    out = {'workload': name, 'dtype': 'f32', 'ms_per_step': round(dt * 1e3, 3), 'value': round(c['trials_per_gpu'] / dt, 1),
           'unit': 'trials/s', 'model_tflops': round(tf, 2), 'peak_tflops': F32_MFMA_PEAK_TFLOPS,
           'frac': round(tf / F32_MFMA_PEAK_TFLOPS, 4), 'steps': steps, 'warmup': warm}
    if roof is not None:
        out['roofline'] = roof
    return out


def alignment_record(dev):
    """Latent alignment at the north-star shape: EIGHT patients of 2048 trials x 200 samples x 128 channels (fp32, 210 MB each,
    resident in HBM like the training inputs): fits / s of the per-patient PCA(0.95), of the pairwise CCA fit on the PCA
    latents (reference: AlignCCA inside process_aligner, datamodules.py:542-565) and of the 8-view MCCA fit on the raw
    channels (D = 8 x 128 = 1024, n_components 30, regs 0.5; AlignMCCA.py:140-154) -- single process, and the share ONE rank
    computes when the fit is sharded by patient over 8 ranks (its patient's condition means and block row of the
    cross-covariance + the replicated eigensolve; the exchanges -- 8 x 13 MB of views, 8 MB of block rows -- not included)."""
    import numpy as np
    from cross_patient_speech_decoding_amd import alignment as A
    from cross_patient_speech_decoding_amd.alignment.AlignMCCA import _gevp
    from cross_patient_speech_decoding_amd.alignment import _linalg as LA
    from cross_patient_speech_decoding_amd.alignment.alignment_utils import _group_conditions_device
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    P = 8
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

    # per-rank share of the patient-sharded fit (8 ranks): own condition means + own block row, then the replicated tail
    own = [i == 0 for i in range(P)]
    avgs_all = [a.reshape(-1, a.shape[-1]) for a in _group_conditions_device(Xd, ys)]
    Zc = torch.cat([a.to(LA.F64) for a in avgs_all], dim=1).contiguous()
    mean = torch.cat([LA.col_mean(a) for a in avgs_all])

    def rank_share():
        a0 = _group_conditions_device(Xd, ys, own=own)[0]
        a0 = a0.reshape(-1, a0.shape[-1])
        return LA.xcov(a0, Zc, mean[:128].contiguous(), mean)
    t_share = med(rank_share, n=3)
    G = torch.cat([LA.xcov(avgs_all[i], Zc, mean[128 * i:128 * (i + 1)].contiguous(), mean) for i in range(P)], dim=0).contiguous()
    offs = np.arange(P + 1) * 128
    t_tail = med(lambda: _gevp(G, offs, 30, 0.5), n=3)
    by = Xd[0].numel() * 4
    return {'workload': 'north-star patients: 8 x (2048 trials x 200 x 128 ch fp32 = 210 MB), inputs resident in HBM',
            'pca_fit': {'ms': round(t_pca * 1e3, 2), 'fits_per_s': round(1 / t_pca, 1), 'input_GB_per_s': round(by / t_pca / 1e9, 1)},
            'cca_fit': {'ms': round(t_cca * 1e3, 2), 'fits_per_s': round(1 / t_cca, 1), 'latent_dims': [int(Z[0].shape[-1]), int(Z[1].shape[-1])]},
            'cca_transform': {'ms': round(t_tr * 1e3, 2)},
            'mcca_fit_8_views': {'D': P * 128, 'n_components': 30, 'single_process_ms': round(t_mcca * 1e3, 2),
                                 'fits_per_s': round(1 / t_mcca, 2),
                                 'per_rank_of_8': {'own_condition_means_and_block_row_ms': round(t_share * 1e3, 2),
                                                   'replicated_eigensolve_ms': round(t_tail * 1e3, 2),
                                                   'compute_ms': round((t_share + t_tail) * 1e3, 2),
                                                   'exchanged_bytes': int(P * avgs_all[0].numel() * 8 + (P * 128) ** 2 * 8)}}}


def realtime_record(dev):
    """BASELINE.json configs[4] (config 5 of the survey): streaming decode, one 20 ms step = one 14-sample window of C = 128
    channels (1792 inputs) through a 2-layer GRU H = 128 + Linear(11), batch 1, replayed from a hipGraph (window upload
    included).  The reference's one published latency: 2.06 ms per prediction (figure_analyses/supp/supp_fig_24.ipynb cell 23)."""
    from cross_patient_speech_decoding_amd.realtime_sim import RealtimeRNNModel, StreamingDecoder
    torch.manual_seed(0)
    m = RealtimeRNNModel(14 * 128, 128, 2, 11, dropout=0.0).to(dev).eval()
    dec = StreamingDecoder(m, n_streams=1, use_graph=True)
    win = torch.randn(1, 14 * 128, device=dev)
    for _ in range(50):
        dec.step(win)
    torch.cuda.synchronize()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        dec.step(win)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    return {'workload': 'configs[4] streaming decode: C=128 (1792-wide window), GRU H=128 L=2 + Linear(11), batch 1, one hipGraph replay per 20 ms step',
            'us_per_step': round(us, 1), 'steps': n, 'reference_published_ms_per_prediction': 2.06,
            'vs_reference_published': round(2060.0 / us, 1)}


def dp_rehearsal_record(name, dev, steps, warm):
    """What the data-parallel host path costs per step, measured where no second GPU exists: the same workload on a ONE-rank
    RCCL communicator (init_process_group('nccl', device_id=), SyncBN exchanges, ReduceOp.AVG all-reduce of the flat gradient
    in two pieces, the tail issued asynchronously from the autograd thread) against the plain single-process step, both in
    this process, same weights.  The product's rule is "data-parallel paths for more than one rank"; this measurement
    replaces that rule from the outside (XF._dp_enabled) for its own duration."""
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    c = WORKLOADS[name]
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    _, plain = _make_step(c, dev, 0)
    dt_plain = _timed_steps(plain, steps, warm)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    real = XF._dp_enabled
    XF._dp_enabled = lambda group: group is not None
    try:
        _, dp = _make_step(c, dev, 0, group=dist.group.WORLD, world=1)
        dt_dp = _timed_steps(dp, steps, warm)
        XF.check_gru_status()
    finally:
        XF._dp_enabled = real
        XF.POST_SYNCBN_HOOKS[:] = []
        dist.destroy_process_group()
    return {'workload': name, 'plain_ms_per_step': round(dt_plain * 1e3, 3), 'one_rank_rccl_ms_per_step': round(dt_dp * 1e3, 3),
            'dp_overhead_pct': round((dt_dp / dt_plain - 1.0) * 100.0, 2), 'steps': steps, 'warmup': warm,
            'collectives_per_step': 'SyncBN fwd + bwd statistics (2 x 2F floats), flat-gradient all-reduce in two pieces (tail async)'}


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


def _gpu_local_cpus(local_rank):
    """CPUs of the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI device), or None."""
    try:
        out = os.popen('rocm-smi --showbus --json 2>/dev/null').read()
        bus = json.loads(out)[f'card{local_rank}']['PCI Bus'].lower()
        with open(f'/sys/bus/pci/devices/{bus}/local_cpulist') as f:
            txt = f.read().strip()
        cpus = set()
        for part in txt.split(','):
            a, _, b = part.partition('-')
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus or None
    except Exception:                                                        # noqa: BLE001
        return None


def pin_host_threads(local_rank):
    """One trainer process per GPU, pinned to a few cores of its own (XPS_BENCH_PIN_CORES, default 4; 0 = leave the scheduler
    alone), taken from the NUMA node of ITS GPU when sysfs says which that is (else from the affinity mask in order).  The
    enqueue path of a step needs ~1 ms of host time, so a main or autograd thread that migrates across a 256-core host shows
    up in short timing windows (tools/jitter.py).  Returns (original mask, description of the pinning for the JSON line)."""
    full = os.sched_getaffinity(0)
    n = int(os.environ.get('XPS_BENCH_PIN_CORES', '4'))
    if n <= 0:
        return full, 'none'
    local = _gpu_local_cpus(local_rank)
    pool = sorted(full & local) if local else []
    src = 'gpu-numa-node'
    if len(pool) < n:
        pool, src = sorted(full), 'affinity-order'
    # ranks whose GPUs share a node take consecutive slices of that node's cores
    k = local_rank % max(len(pool) // n, 1)
    sel = pool[k * n:(k + 1) * n]
    if len(sel) != n:
        return full, 'none'
    _set_affinity_all_threads(set(sel))
    return full, f'{n} cores ({src}): {sel}'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: 100 timed configs[3] steps (~0.7 s): short windows wobbled 10-25 % when the host hiccuped once
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='configs3')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--headline-only', action='store_true', help='skip the sub-records')
    ap.add_argument('--no-probes', action='store_true',
                    help='profiling runs only: skip the roofline probes too (their re-issued launches would be counted into a per-step '
                         'kernel profile); the line then has no roofline')
    ap.add_argument('--precision', choices=['bf16x3', 'fp32'], default=None,
                    help='product precision of the matrix kernels (default: the library default, bf16x3)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        raise SystemExit('launch multi-GPU runs with torch.distributed.run (one process per GPU)')
    full_affinity, pinning = pin_host_threads(local_rank)
    # rehearsal knobs (1-GPU box): XPS_BENCH_BACKEND=gloo XPS_BENCH_ONE_DEVICE=1 put every rank on cuda:0
    backend = os.environ.get('XPS_BENCH_BACKEND', 'nccl')
    if os.environ.get('XPS_BENCH_ONE_DEVICE'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    if args.precision:
        XF.set_gemm_precision(args.precision)
    precision = XF.get_gemm_precision()
    name = args.workload
    c = dict(WORKLOADS[name])
    model, step = _make_step(c, dev, rank, group=dist.group.WORLD if world > 1 else None, world=world)
    torch.manual_seed(99)                        # the same teacher-forcing coins on every rank

    # untimed pre-warm: the first second of a fresh process runs slower on this pool (clock ramp / cold code pages); it is
    # spent here, BEFORE the W warm-up steps of the contract, never inside the timed region, and stated in the JSON line
    # (a fixed STEP count, so that every rank of a data-parallel run issues the same collectives)
    prewarm = int(os.environ.get('XPS_BENCH_PREWARM_STEPS', str(PREWARM[name])))
    for _ in range(prewarm):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    import gc
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
    XF.check_gru_status()                        # (outside the timed region: one small device word)
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
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'prewarm_steps': prewarm,
            'ms_per_step': round(el / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            # bf16x3: every fp32 operand split hi + lo in bf16, 3 bf16 MFMAs per product, fp32 accumulation and fp32 everywhere
            # else (results within 2e-6 of the fp32 reference goldens); fp32: fp32 MFMA
            'dtype': 'bf16x3' if precision == 'bf16x3' else 'f32',
            'data': 'synthetic',
            'config': {'workload': WORKLOAD_TEXT[name],
                       'trials_per_gpu': c['trials_per_gpu'], 'global_batch': c['trials_per_gpu'] * world,
                       'parallelism': f'dp{world}', 'train_mflop_per_trial': round(fl / 1e6, 2),
                       'precision': ('bf16x3 = fp32 operands split hi + lo in bf16, 3 bf16 MFMAs per product, fp32 accumulate and '
                                     'fp32 everywhere else' if precision == 'bf16x3' else 'fp32 MFMA'),
                       'host_pinning': pinning, 'timed_window_ms': round(el * 1e3, 1)},
            'model_tflops': round(value * fl / 1e12, 3), 'final_loss': round(final_loss, 5),
        }
        out['config']['collective'] = (f'RCCL (torch.distributed backend {backend!r}) world {world}: SyncBN statistics + flat gradient '
                                       f'all-reduce per step' if world > 1 else 'none (single process)')
        out['rccl_world'] = world if (world > 1 and backend == 'nccl') else (0 if world == 1 else None)
        if world == 1 and args.no_probes:
            out['roofline'] = None
        elif world == 1:
            # (single process only: the roofline probes re-issue launches, a capture runs one more training step)
            try:
                out['roofline'] = (roofline_cluster(model, c, dev) if name == 'configs3'
                                   else roofline_wgrad(model, c, capture_dominant_launch(step)))
            except Exception as e:                                           # noqa: BLE001
                out['roofline'] = {'error': f'{type(e).__name__}: {e}'[:300]}
            if not args.headline_only:
                other = 'configs1' if name == 'configs3' else 'configs3'
                # sub-records never cost the headline line: a failure is reported in place of the record
                for key, fn in ((other, lambda: workload_record(other, dev, 200 if other == 'configs1' else 100, 20)),
                                ('fp32', (lambda: fp32_record(name, dev, 100, 10)) if precision != 'fp32' else None),
                                ('alignment', lambda: alignment_record(dev)),
                                ('realtime', lambda: realtime_record(dev)),
                                ('dp_rehearsal', lambda: dp_rehearsal_record(name, dev, 100, 30))):
                    if fn is None:
                        continue
                    try:
                        out[key] = fn()
                    except Exception as e:                                   # noqa: BLE001
                        out[key] = {'error': f'{type(e).__name__}: {e}'[:300]}
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
