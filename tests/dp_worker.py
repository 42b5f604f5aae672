"""Spawned by the data-parallel tests: world ranks (gloo rendezvous on 127.0.0.1) each run one
sharded training step and the result is compared with the single-process full-batch step.

--device cuda : the HIP model on the box's single GPU (all ranks share cuda:0; gloo moves the
                CUDA tensors) -> checks the product path: SyncBN statistics + flat all-reduce.
--device cpu  : the host-side logic only (sharding, loss weighting, metric reduction, LinearLR),
                with the CPU oracle as compute stand-in -> runs in the no-GPU CI."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

# --shape: the model / batch the data-parallel step is checked on.  'tiny' is the host-logic case (unequal shards); the
# other two are BASELINE shapes (SURVEY 8e's invariant "N-rank gradients == 1-rank gradients" on the kernels bench.py runs):
#   h128 = configs[1] (C = 64, F = 100, k = s = 10, H = 128: resident GRU kernels, grouped weight-gradient launch),
#   h512 = configs[3] (aligned d = 30, H = 512: cluster recurrence, wide decoder, 256-tile split-K groups), 512 trials
SHAPES = {'tiny': ((10, 12, 16, 9, 2, 1, 5, 5, 0, 0.0, 0.0), (46, 40, 10), True),
          'h128': ((64, 100, 128, 9, 2, 1, 10, 10, 0, 0.0, 0.0), (512, 200, 64), False),
          'h512': ((30, 100, 512, 9, 2, 1, 10, 10, 0, 0.0, 0.0), (512, 200, 30), False)}
SHAPE = ['tiny']


def data():
    rng = np.random.default_rng(3)
    shp = SHAPES[SHAPE[0]][1]
    X = torch.from_numpy(rng.standard_normal(shp).astype(np.float32))      # tiny: 46 rows = unequal shards at world 4
    y = torch.from_numpy(rng.integers(0, 9, (shp[0], 3)))
    return X, y


def hip_step(rank, world, X, y, group):
    from weights import weights_from_seed
    from cross_patient_speech_decoding_amd.nn_models import Seq2SeqRNN
    from cross_patient_speech_decoding_amd.nn_models.trainer import FlatAdamW
    args, _, act = SHAPES[SHAPE[0]]
    m = Seq2SeqRNN(*args, 'gru', 1e-3, 1e-5, activation=act)
    m.load_state_dict(weights_from_seed(m.state_dict(), 5))
    m = m.cuda().train()
    m.temporal_conv.process_group = group
    m.temporal_conv.global_batch = X.shape[0] if group is not None else None
    opt = FlatAdamW(m, lr=1e-3, weight_decay=1e-5, max_norm=0.5, group=group)
    xs, ys = X[rank::world].cuda(), y[rank::world].cuda()
    opt.zero_grad()
    logits = m(xs, ys, coins=[True, False, True])
    loss = m.criterion(logits.view(-1, 9), ys.view(-1)) * (xs.shape[0] * world / X.shape[0])
    loss.backward()
    opt.step()
    from cross_patient_speech_decoding_amd.nn_models import functional as XF
    XF.check_gru_status()                     # (cluster recurrence: no hand-off gave up)
    gnorm = opt.grad_norm()
    # numpy (pickled by value): torch tensors would travel as file descriptors of a process that may be gone
    return (opt.flat_g.cpu().numpy(), opt.flat_p.cpu().numpy(), float(gnorm), m.temporal_conv.bn.running_var.cpu().numpy())


def _patients():
    from cross_patient_speech_decoding_amd.utils.synthetic import make_patient
    return [make_patient(p, 90 - 6 * p, T=40, C=(14, 12, 10, 16)[p], n_cond=10) for p in range(4)]


def sharded_alignment(group):
    """process_aligner_sharded over `group` (None: the single-process process_aligner)."""
    from cross_patient_speech_decoding_amd.alignment import AlignCCA
    from cross_patient_speech_decoding_amd.nn_models.data_utils.datamodules import process_aligner_sharded
    pats = _patients()
    (Xt, yt), pool = pats[0], [(torch.from_numpy(x), torch.from_numpy(y - 1), torch.from_numpy(y - 1)) for x, y in pats[1:]]
    Xp, yp, tar = process_aligner_sharded(torch.from_numpy(Xt), torch.from_numpy(yt - 1), torch.from_numpy(yt - 1), pool, AlignCCA,
                                          group=group)
    return Xp.numpy(), yp.numpy(), int(tar.n_components_)


def sharded_mcca(group):
    """(a) AlignMCCA.fit(..., group=) on raw 4-view data with pca_var < 1 (signal ranks from the owners) and (b) the whole
    PCA -> MCCA -> pool pipeline, patients sharded over `group` (None: single process)."""
    from cross_patient_speech_decoding_amd.alignment import AlignMCCA
    from cross_patient_speech_decoding_amd.nn_models.data_utils.datamodules import process_aligner_multiview_sharded
    pats = _patients()
    al = AlignMCCA(n_components=5, regs=0.5, pca_var=0.9)
    al.fit([x for x, _ in pats], [y for _, y in pats], group=group)
    loads = [np.asarray(l) for l in al.mcca.loadings_]
    tr = [np.asarray(al.transform(x, idx=i)) for i, (x, _) in enumerate(pats)]
    (Xt, yt), pool = pats[0], [(torch.from_numpy(x), torch.from_numpy(y - 1), torch.from_numpy(y - 1)) for x, y in pats[1:]]
    Xp, yp, tmap = process_aligner_multiview_sharded(torch.from_numpy(Xt), torch.from_numpy(yt - 1), torch.from_numpy(yt - 1), pool,
                                                     lambda: AlignMCCA(n_components=4, regs=0.5), group=group)
    zt = np.asarray(tmap.transform(Xt.reshape(-1, Xt.shape[-1])[:64]))
    return loads, tr, list(al.mcca.block_rows_computed_), Xp.numpy(), yp.numpy(), zt


def worker(rank, world, device, q, shape='tiny'):
    SHAPE[0] = shape
    if device == 'nccl':
        # the production configuration: one process per GPU, RCCL (backend 'nccl'), device bound at init
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
        X, y = data()
        res = hip_step(rank, world, X, y, dist.group.WORLD)
        if rank == 0:
            q.put(res)
        dist.barrier()
        dist.destroy_process_group()
        return
    dist.init_process_group('gloo', rank=rank, world_size=world)
    X, y = data()
    if device == 'align':
        torch.cuda.set_device(0)
        res = sharded_alignment(dist.group.WORLD)
        if rank == 1:                       # a rank that does NOT own the target
            q.put(res)
    elif device == 'mcca':
        torch.cuda.set_device(0)
        res = sharded_mcca(dist.group.WORLD)
        if rank == world - 1:
            q.put((rank,) + res)
    elif device == 'cuda':
        torch.cuda.set_device(0)
        res = hip_step(rank, world, X, y, dist.group.WORLD)
        if rank == 0:
            q.put(res)
    else:
        from cross_patient_speech_decoding_amd.nn_models.trainer import LinearLR, Trainer, _shard
        # shards partition the batch; the weighted local means add up to the global mean
        xs = _shard(X, rank, world)
        t = torch.tensor([float(xs.shape[0])]); dist.all_reduce(t)
        assert int(t.item()) == X.shape[0]
        local = xs.mean() * (xs.shape[0] * world / X.shape[0])
        dist.all_reduce(local); local /= world
        assert abs(local.item() - X.mean().item()) < 1e-6
        tr = Trainer.__new__(Trainer); tr.group = None
        tr._device = lambda: torch.device('cpu')
        m = tr._reduce_metrics({'acc': float(rank + 1) * xs.shape[0]}, xs.shape[0])
        exp = sum((r + 1) * len(X[r::world]) for r in range(world)) / X.shape[0]
        assert abs(m['acc'] - exp) < 1e-9

        class O:
            base_lr = lr = 1.0
        s = LinearLR(O, 1.0, 0.01, 10)
        for _ in range(12):
            s.step()
        assert abs(O.lr - 0.01) < 1e-12
        if rank == 0:
            q.put('ok')
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=2)
    ap.add_argument('--device', default='cpu')
    ap.add_argument('--shape', default='tiny', choices=sorted(SHAPES))
    a = ap.parse_args()
    SHAPE[0] = a.shape
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    if a.device == 'nccl1':
        # ONE rank on a real RCCL communicator.  The product runs its data-parallel paths for more than one rank only
        # (functional._dp_enabled); THIS test replaces that rule from the outside so that ReduceOp.AVG, device_id= init, the
        # SyncBN exchanges and the async tail all-reduce issued from the autograd thread all execute on RCCL on a one-GPU box;
        # averaging over one rank must give the plain step bit for bit
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        X, y = data()
        from cross_patient_speech_decoding_amd.nn_models import functional as XF
        XF._dp_enabled = lambda group: group is not None
        calls = []
        real = dist.all_reduce
        def spy(t, *args, **kw):
            calls.append((tuple(t.shape), kw.get('op', args[0] if args else None), bool(kw.get('async_op', False))))
            return real(t, *args, **kw)
        dist.all_reduce = spy
        r_dp = hip_step(0, 1, X, y, dist.group.WORLD)
        dist.all_reduce = real
        r_1 = hip_step(0, 1, X, y, None)
        for u, v in zip(r_dp, r_1):
            assert (np.asarray(u) == np.asarray(v)).all(), 'one-rank RCCL step differs from the plain step'
        ops = [c[1] for c in calls]
        assert any(c[2] for c in calls), f'no async all-reduce was issued: {calls}'
        assert dist.ReduceOp.AVG in ops and dist.ReduceOp.SUM in ops, ops
        dist.barrier()
        dist.destroy_process_group()
        print('DP_OK', len(calls), 'collectives on RCCL')
        return
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, a.world, a.device, q, a.shape)) for r in range(a.world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, p.exitcode
    if a.device == 'align':
        Xs, ys, ks = res
        X1, y1, k1 = sharded_alignment(None)
        assert ks == k1 and Xs.shape == X1.shape and (ys == y1).all()
        assert (Xs == X1).all(), float(abs(Xs - X1).max())          # deterministic decompositions: identical pooled set
    if a.device == 'mcca':
        rank, loads, tr, rows, Xp, yp, zt = res
        loads1, tr1, rows1, Xp1, yp1, zt1 = sharded_mcca(None)
        assert rows == [i for i in range(4) if i % a.world == rank] and rows1 == [0, 1, 2, 3], (rows, rows1)   # each rank: ITS block rows only
        for u, v in zip(loads + tr + [Xp, yp, zt], loads1 + tr1 + [Xp1, yp1, zt1]):
            assert u.shape == v.shape and (u == v).all(), float(np.abs(u - v).max())       # bit for bit
    if a.device in ('cuda', 'nccl'):
        X, y = data()
        g1, p1, n1, rv1 = [torch.as_tensor(v) if not isinstance(v, float) else v for v in hip_step(0, 1, X, y, None)]
        g2, p2, n2, rv2 = [torch.as_tensor(v) if not isinstance(v, float) else v for v in res]
        eg = (g1 - g2).abs().max().item() / max(g1.abs().max().item(), 1e-12)
        assert eg < 2e-4, f'gradient mismatch {eg}'
        assert abs(n1 - n2) < 2e-4 * n1, (n1, n2)
        assert (rv1 - rv2).abs().max().item() < 1e-5
        # updated weights: Adam's first step moves a weight by lr * g / (|g| + eps) -- for |g| >> eps that is lr * sign(g), blind
        # to everything but the sign, and for |g| ~ eps (1e-8) it amplifies rounding noise to +-lr.  So (1) elements whose gradient
        # is above the noise floor (1e-3 of the flat buffer's largest; the two runs differ by < 2e-4 of it) must have the same
        # sign and hence the same update to 1e-5 * lr ... and (2) every element must have moved by at most lr (+ weight decay)
        lr = 1e-3
        d = (p1 - p2).abs()
        big = g1.abs() > 1e-3 * g1.abs().max()
        assert big.float().mean().item() > 0.05, 'too few elements above the noise floor: the check would be empty'
        assert d[big].max().item() <= 1e-5, f'updated weights differ by {d[big].max().item()} on elements above the noise floor'
        assert d.max().item() <= 2.0 * lr * 1.001, d.max().item()
    print('DP_OK')


if __name__ == '__main__':
    main()
