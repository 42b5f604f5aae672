"""GPU tests of the training loop around the kernels: the PCA->CCA->pool pipeline, the Trainer
(Lightning-like surface), trajectory / PER parity with the CPU oracle, and data-parallel
equivalence (2 ranks, gloo rendezvous, both on the single GPU of the box)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pooled_views(P=3, N=96, T=40, C=(14, 12, 10), n_cond=12):
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    out = []
    for p in range(P):
        X, y = make_patient(p, N - 8 * p, T=T, C=C[p], n_cond=n_cond)
        out.append((X, y - 1))
    return out


def test_process_aligner_matches_oracle():
    from oracle import align_oracle as ao
    from cross_patient_speech_decoding_amd.alignment import AlignCCA
    from cross_patient_speech_decoding_amd.nn_models.data_utils.datamodules import process_aligner
    views = _pooled_views()
    (Xt, yt), pool = views[0], [(x, y, y) for x, y in views[1:]]
    ref_X, ref_y, ref_tar = ao.process_aligner(Xt, yt, yt, pool)
    Xp, yp, tar = process_aligner(torch.from_numpy(Xt), torch.from_numpy(yt), torch.from_numpy(yt),
                                  [(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(a)) for x, y, a in pool],
                                  AlignCCA)
    assert Xp.dtype == torch.float32 and yp.dtype == torch.int64
    assert tuple(Xp.shape) == ref_X.shape and tar.n_components_ == ref_tar.n_components_
    np.testing.assert_array_equal(yp.numpy(), ref_y)
    assert np.abs(Xp.numpy() - ref_X).max() <= 2e-4 * np.abs(ref_X).max()      # float32 inputs, float32 output
    z = tar.transform(Xt.reshape(-1, Xt.shape[-1])[:50])
    np.testing.assert_allclose(z, ref_tar.transform(Xt.reshape(-1, Xt.shape[-1])[:50]), atol=2e-4)


def test_process_aligner_multiview_mcca_matches_oracle_composition():
    from cross_patient_speech_decoding_amd.alignment import AlignMCCA
    from cross_patient_speech_decoding_amd.nn_models.data_utils.datamodules import process_aligner_multiview
    views = _pooled_views()
    (Xt, yt), pool = views[0], [(x, y, y) for x, y in views[1:]]
    Xp, yp, tmap = process_aligner_multiview(Xt, yt, yt, pool, lambda: AlignMCCA(n_components=4, regs=0.5))
    assert Xp.shape == (sum(v[0].shape[0] for v in views), 40, 4) and yp.shape == (Xp.shape[0], 3)
    z = tmap.transform(Xt.reshape(-1, Xt.shape[-1]))
    np.testing.assert_allclose(z.reshape(Xt.shape[0], 40, 4), Xp[:Xt.shape[0]].numpy(), atol=1e-4)
    # against the oracle composition: PCA(0.95) per patient (oracle/align_oracle.pca_fit) -> oracle MCCA on the condition means ->
    # every view through its own map, pooled; components up to sign (fixed by the first pooled sample with a clear value)
    from oracle import align_oracle as ao, mcca_oracle as mo
    dr = []
    for x, _ in views:
        _, zz = ao.pca_fit(x.reshape(-1, x.shape[-1]), 0.95)
        dr.append(zz.reshape(x.shape[0], -1, zz.shape[-1]))
    ref = mo.get_mcca_transforms(dr, [y for _, y in views], n_components=4, regs=0.5)
    gaps = np.abs(np.diff(ref.evals_)) / np.abs(ref.evals_).max()
    assert gaps.min() > 1e-6                                  # simple eigenvalues: the comparison below is meaningful
    ref_pool = np.vstack([mo.mcca_transform(ref, v, i) for i, v in enumerate(dr)])
    got = Xp.numpy().astype(np.float64)
    for c in range(4):
        sgn = np.sign(np.sum(got[..., c] * ref_pool[..., c]))
        assert np.abs(got[..., c] - sgn * ref_pool[..., c]).max() <= 5e-4 * np.abs(ref_pool[..., c]).max(), c


def test_trainer_fit_test_and_logged_metrics():
    from torch.utils.data import DataLoader, TensorDataset
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    from cross_patient_speech_decoding_amd.nn_models.trainer import ModelCheckpoint, Trainer, seed_everything
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    seed_everything(0)
    X, y = make_patient(0, 256, T=60, C=16, n_cond=8)
    X, y = torch.from_numpy(X), torch.from_numpy(y - 1)
    tr = DataLoader(TensorDataset(X[:192], y[:192]), batch_size=5000, shuffle=True)
    va = DataLoader(TensorDataset(X[192:], y[192:]), batch_size=5000)
    model = Seq2SeqRNN(16, 12, 24, 9, 2, 1, 6, 6, 0, 0.1, 0.1, 'gru', 5e-3, 1e-5, activation=False, decay_iters=40)
    ck = ModelCheckpoint(monitor='val_acc', mode='max')
    trainer = Trainer(max_epochs=40, gradient_clip_val=0.5, callbacks=[ck])
    trainer.fit(model, tr, va)
    m = trainer.logged_metrics
    for k in ('train_loss', 'train_acc', 'val_loss', 'val_acc'):
        assert k in m
    assert m['train_loss'] < 1.9 and m['train_acc'] > 0.2           # chance: ln 9 = 2.197, 0.111
    assert abs(m['lr'] - 5e-3 * (1 - 0.99 * 39 / 40)) < 1e-9        # LinearLR 1.0 -> 0.01, stepped per epoch
    res = trainer.test(model, va, ckpt_path='best')
    assert 'test_acc' in trainer.logged_metrics and res[0]['test_acc'] == pytest.approx(ck.best_score, abs=1e-6)


def test_training_trajectory_and_per_match_cpu_oracle():
    """40 full-batch steps from identical weights, dropout 0, identical teacher-forcing coins:
    loss curve within 2e-3, final PER within 0.5 % absolute (north-star tolerance)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from weights import weights_from_seed
    from oracle.seq2seq_oracle import Seq2SeqOracle, phoneme_error_rate, train_step
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    torch.set_num_threads(8)
    X, y = make_patient(3, 320, T=80, C=20, n_cond=10)
    X, y = torch.from_numpy(X), torch.from_numpy(y - 1)
    args = (20, 16, 32, 9, 2, 1, 8, 8, 0, 0.0, 0.0)
    orc = Seq2SeqOracle(*args, learning_rate=3e-3, l2_reg=1e-5, activation=False)
    sd = weights_from_seed(orc.state_dict(), 77)
    orc.load_state_dict(sd)
    hip = Seq2SeqRNN(*args, 'gru', 3e-3, 1e-5, activation=False)
    hip.load_state_dict(sd)
    hip = hip.cuda()
    opt_o, _ = orc.make_optimizer()
    opt_h = FlatAdamW(hip, lr=3e-3, weight_decay=1e-5, max_norm=0.5)
    rng = np.random.default_rng(5)
    Xg, yg = X.cuda(), y.cuda()
    worst = 0.0
    for step in range(40):
        coins = [bool(c) for c in rng.integers(0, 2, 3)]
        lo, _ = train_step(orc, opt_o, X, y, coins=coins, clip=0.5)
        hip.train()
        opt_h.zero_grad()
        logits = hip(Xg, yg, coins=coins)
        lh = hip.criterion(logits.view(-1, 9), yg.view(-1))
        lh.backward()
        opt_h.step()
        worst = max(worst, abs(lh.item() - lo.item()))
    assert worst <= 2e-3, worst
    orc.eval(); hip.eval()
    with torch.no_grad():
        po = orc(X, y, teacher_forcing_ratio=0).argmax(-1).numpy()
        ph = hip(Xg, yg, teacher_forcing_ratio=0).argmax(-1).cpu().numpy()
    per_o, per_h = phoneme_error_rate(po, y.numpy()), phoneme_error_rate(ph, y.numpy())
    assert abs(per_o - per_h) <= 0.5, (per_o, per_h)
    assert per_h < 80.0                                              # learned something (chance ~ 89 %)


def test_data_parallel_two_ranks_equal_single_process():
    """N-rank gradients (sharded batch, SyncBN statistics, flat all-reduce) == 1-rank gradients."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29517', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), '--world', '2', '--device', 'cuda'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'DP_OK' in out.stdout


@pytest.mark.parametrize('shape', ['h128', 'h512'])
def test_data_parallel_two_ranks_equal_single_process_at_baseline_shapes(shape):
    """SURVEY 8e's invariant -- N-rank gradients == 1-rank gradients -- ON THE KERNELS bench.py RUNS: configs[1] (C = 64,
    H = 128: resident GRU kernels, grouped weight-gradient launch, weight-stationary projections) and configs[3] (aligned
    d = 30, H = 512: cluster recurrence, wide decoder, 256-tile split-K groups), 512 trials over two ranks (gloo rendezvous,
    both ranks on the box's GPU), SyncBN statistics, the two-piece overlapped all-reduce of the flat gradient: flat gradient
    to 2e-4 of its largest element, clipped gradient norm, BatchNorm running statistics, AdamW-updated weights.  At H = 512
    the two PROCESSES share one GPU, so the recurrence runs one step per launch (XPS_GRU_CLUSTER=steps: the same kernels,
    bit-identical to the persistent form -- tests/test_gpu_gru_cluster.py -- without two persistent grids waiting for each
    other's CUs; one process per GPU, the production layout, is not affected)."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29561' if shape == 'h128' else '29563',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    if shape == 'h512':
        env['XPS_GRU_CLUSTER'] = 'steps'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), '--world', '2', '--device', 'cuda',
                          '--shape', shape], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'DP_OK' in out.stdout


def test_rccl_collectives_execute_on_a_one_rank_communicator():
    """A one-GPU box cannot hold two RCCL ranks, but it can hold ONE: the worker replaces functional._dp_enabled (test side: the
    product rule is "more than one rank") so that the data-parallel paths run at world size 1, and `init_process_group('nccl', device_id=)`, the SyncBN exchanges, `ReduceOp.AVG` on the flat
    gradient and the async tail all-reduce issued from the autograd thread all EXECUTE on RCCL (tests/dp_worker.py --device nccl1
    checks that they were issued and that the step equals the plain step bit for bit).  The 2-rank test below needs 2 GPUs."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29549')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), '--world', '1', '--device', 'nccl1'], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'DP_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]




def test_data_parallel_two_ranks_rccl_one_device_each():
    """The production configuration: backend 'nccl' (= RCCL), one process per GPU, device bound at init_process_group,
    ReduceOp.AVG, the tail of the flat gradient all-reduced asynchronously from the autograd thread.  Needs two GPUs: the
    1-GPU boxes of this pool skip it, the driver's 8-GPU node (and any 2-GPU box) runs it."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (one RCCL rank per device)')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), '--world', '2', '--device', 'nccl'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'DP_OK' in out.stdout


def test_patient_sharded_alignment_equals_single_process():
    """SURVEY 8e, alignment: patients sharded over ranks (PCA + CCA on the owner's GPU, aligned trials broadcast):
    the pooled training set of 3 ranks is identical to the single-process one."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29523', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), '--world', '3', '--device', 'align'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'DP_OK' in out.stdout


@pytest.mark.parametrize('world', [2, 4])
def test_patient_sharded_mcca_equals_single_process(world):
    """SURVEY 8e (2), MCCA: views sharded over ranks by patient -- condition means / signal ranks / PCA by the owner, every
    rank computes the block rows C_{p,.} of ITS views (xps_xcov_f64), block rows exchanged, eigensolve replicated.  Loadings,
    transforms of every view and the pooled training set of the PCA -> MCCA -> pool pipeline are identical, bit for bit,
    to the single-process fit (reference call site: alignment/AlignMCCA.py:140-154)."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29551 + world), HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), '--world', str(world), '--device', 'mcca'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert 'DP_OK' in out.stdout


def test_train_seq2seq_cli_end_to_end(tmp_path):
    """The counterpart of scripts/train_seq2seq.py: pooled + CCA-aligned k-fold training on synthetic patients."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'train_seq2seq.py'), '-pt', 'SYN', '-p', 'True',
                          '--synthetic', '3', '--iters', '1', '--folds', '2', '--epochs', '60', '--hidden', '64', '--seed', '3',
                          '--out', str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    accs = np.load(os.path.join(str(tmp_path), 'accs', 'SYN', 'SYN_pooled_accs.npy'))
    assert accs.shape == (1, 2) and accs.mean() > 0.2             # chance 1/9


def test_config3_mcca_aligned_cross_patient_training():
    """BASELINE config 3 in miniature: 3 patients -> per-patient PCA -> ONE multiview MCCA (device) -> pooled
    trials -> seq2seq GRU trained by the HIP trainer; k-fold DataModule in multiview mode."""
    from cross_patient_speech_decoding_amd.alignment import AlignMCCA
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    from cross_patient_speech_decoding_amd.nn_models.data_utils.datamodules import AlignedMicroValDataModule
    from cross_patient_speech_decoding_amd.nn_models.trainer import Trainer, seed_everything
    seed_everything(1)
    views = _pooled_views(P=3, N=120, T=60, C=(20, 16, 18), n_cond=10)
    (Xt, yt), pool = views[0], [(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(y)) for x, y in views[1:]]
    dm = AlignedMicroValDataModule(torch.from_numpy(Xt), torch.from_numpy(yt), torch.from_numpy(yt), pool,
                                   lambda: AlignMCCA(n_components=6, regs=0.5), batch_size=5000, folds=3, val_size=0.2,
                                   multiview=True)
    dm.setup()
    dm.set_fold(0)
    shape = dm.get_data_shape()
    assert shape[1:] == (60, 6) and shape[0] > 150                   # pooled over the three patients
    model = Seq2SeqRNN(6, 16, 32, 9, 2, 1, 6, 6, 0, 0.1, 0.1, 'gru', 5e-3, 1e-5, activation=False, decay_iters=30)
    trainer = Trainer(max_epochs=30, gradient_clip_val=0.5)
    trainer.fit(model, dm.train_dataloader(), dm.val_dataloader())
    trainer.test(model, dm.test_dataloader())
    m = trainer.logged_metrics
    assert m['train_loss'] < 2.0 and m['test_acc'] > 0.15           # chance: ln 9 = 2.197 / 0.111


@pytest.mark.parametrize('H,B,C', [(64, 150, 48), (128, 512, 64), (160, 96, 40)])
def test_training_steps_are_bitwise_reproducible(H, B, C):
    """No atomics anywhere on the path (split-K slabs, two-stage reductions, counter-based dropout): the same
    seed gives the same bits, step after step -- what makes N-GPU runs comparable with 1-GPU runs."""
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW, seed_everything

    def run():
        seed_everything(7)
        XF._DROP_COUNTER[0] = 0
        m = Seq2SeqRNN(C, 60, H, 10, 2, 1, 10, 10, 0, 0.3, 0.3, 'gru', 1e-3, 1e-5).cuda()
        opt = FlatAdamW(m, lr=1e-3, max_norm=0.5)
        x = torch.randn(B, 200, C).cuda()
        y = torch.randint(1, 10, (B, 3)).cuda()
        m.train()
        trail = []
        for s in range(3):
            opt.zero_grad()
            loss = m.training_step((x, y), s)
            loss.backward()
            g = opt.flat_g.clone()
            opt.step()
            trail.append((loss.detach().clone(), g, opt.flat_p.clone()))
        return trail

    a, b = run(), run()
    for (la, ga, pa), (lb, gb, pb) in zip(a, b):
        assert torch.equal(la, lb) and torch.equal(ga, gb) and torch.equal(pa, pb)
