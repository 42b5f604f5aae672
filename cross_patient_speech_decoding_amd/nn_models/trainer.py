"""Data-parallel trainer for the seq2seq GRU path on MI355X.

Plays the role ``lightning.Trainer`` plays for the reference (scripts/train_seq2seq.py:171-189):
``Trainer(max_epochs=..., gradient_clip_val=0.5, callbacks=[ModelCheckpoint(monitor='val_acc',
mode='max')]).fit(model, train_loader, val_loader)``, ``.test(model, loader, ckpt_path='best')``,
``.logged_metrics``.  What it adds is the MI355X-first part the reference does not have:

* one process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI; ``gloo`` on CPU
  for tests).  Every batch is sharded by trial across ranks (rank r takes rows r::world);
* all parameters, gradients and AdamW moments live in ONE flat fp32 buffer each, so a step does a
  single RCCL all-reduce (2.6 MB cfg-2, 33 MB cfg-4) followed by ONE fused
  clip-by-global-norm + AdamW kernel over the flat buffers;
* BatchNorm statistics are all-reduced (SyncBN) so N-GPU training is the same computation as the
  reference's single-process full-batch step; the global-norm clip is applied to the REDUCED
  gradient, so every rank clips identically.
"""
import os

import torch
import torch.distributed as dist

from . import functional as XF


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


class FlatAdamW:
    """torch.optim.AdamW semantics over flat buffers + gradient_clip_val (clip-by-norm).

    ``module`` parameters are re-pointed at views of ``flat_p``; their ``.grad`` are views of
    ``flat_g`` (autograd accumulates in place), so there is nothing to gather before the
    all-reduce and nothing to scatter after the update."""

    ALIGN = 64       # floats: every tensor starts on a 256-byte boundary (vector loads in the GEMMs)

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=None,
                 group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdamW drives the HIP optimiser kernel: move the model to the GPU first')
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = self.flat_g[o:o + n].view(p.shape)
        self._grad_views = [p.grad for p in self.params]
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.base_lr = self.lr = lr
        self.betas, self.eps, self.weight_decay, self.max_norm = betas, eps, weight_decay, max_norm
        self.group = group
        self.step_count = 0
        # data-parallel overlap: the gradients of everything downstream of `module.temporal_conv` are complete
        # before the convolution's own backward starts; their slice of the flat buffer is all-reduced
        # asynchronously while that backward (~8 % of the step) runs.  The trigger sits right after the SyncBN
        # statistics exchange of the backward pass (functional.POST_SYNCBN_HOOKS): one communicator, and the
        # small latency-critical exchange is never queued behind the large one.  XPS_DP_OVERLAP=0 disables.
        self._split = None
        self._early = None
        self._comm_stream = None
        self._dp = XF._dp_enabled(group)        # more than one rank (the one-rank RCCL rehearsals patch XF._dp_enabled)
        # RCCL averages in the collective (no extra pass over the buffer); gloo (CPU tests) sums, then one scale
        self._avg = self._dp and dist.get_backend(group) == 'nccl'
        self._op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if (self._dp and os.environ.get('XPS_DP_OVERLAP', '1') != '0' and XF.DIRECT_GRAD
                and hasattr(module, 'temporal_conv')):
            first = {id(p) for p in module.temporal_conv.parameters()}
            idx = [i for i, p in enumerate(self.params) if id(p) not in first]
            if idx and idx == list(range(idx[0], len(self.params))) and idx[0] > 0:
                self._split = offs[idx[0]]
                XF.POST_SYNCBN_HOOKS[:] = [self._reduce_tail_async]       # one optimiser drives the model

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()

    def _reduce_tail_async(self):
        """Hook (autograd thread, during backward): all-reduce flat_g[split:] without blocking the main stream.
        Issued from an ordinary torch stream that first waits for the main stream and for the side stream carrying
        the weight-gradient GEMMs; ``step()`` waits for the returned work handle.

        Every rank must issue the same collectives in the same order, so nothing here is allowed to change the
        sequence on ONE rank: there is no local fallback.  A failure (of the stream set-up or of the collective call)
        propagates out of ``loss.backward()`` and ends this rank; the other ranks end on the communicator's timeout /
        abort instead of pairing a different collective with this one."""
        if self._early is not None:
            return
        # the tail is reduced in place: every gradient of it must still live in the flat buffer.  A rank-local
        # deviation from that (someone replaced .grad) would desynchronise the collective sequence: fail hard.
        for p, view in zip(self.params, self._grad_views):
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                raise RuntimeError('FlatAdamW: a parameter gradient left the flat buffer while the overlapped gradient '
                                   'all-reduce is enabled (use FlatAdamW.zero_grad(), not zero_grad(set_to_none=True)); '
                                   'XPS_DP_OVERLAP=0 disables the overlap on EVERY rank')
        dev = self.flat_g.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=idx)       # an ordinary torch stream for the collective
        cs = self._comm_stream
        cs.wait_stream(torch.cuda.current_stream(idx))              # gradients written on the main stream ...
        side = XF._side_streams.get(idx)
        if side is not None:
            cs.wait_stream(side)                                    # ... and by the weight-gradient GEMMs on the side stream
        with torch.cuda.stream(cs):
            self._early = dist.all_reduce(self.flat_g[self._split:], op=self._op, group=self.group, async_op=True)

    @torch.no_grad()
    def step(self):
        """all-reduce (mean) -> global grad norm -> fused clip + AdamW.  The pre-clip norm of the step
        (what clip_grad_norm_ returns) is available from ``grad_norm()``."""
        world, _ = _world(self.group)
        # robustness: if someone reset .grad (model.zero_grad(set_to_none=True)) autograd allocated fresh
        # gradient tensors; fold them back into the flat buffer and restore the views
        for p, view in zip(self.params, self._grad_views):
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
        if self._dp:
            if self._early is not None:                    # the tail went out during backward: only the head is left
                self._early.wait()                         # (current stream waits for the collective)
                self._early = None
                dist.all_reduce(self.flat_g[:self._split], op=self._op, group=self.group)
            else:
                dist.all_reduce(self.flat_g, op=self._op, group=self.group)
            if not self._avg:
                self.flat_g.mul_(1.0 / world)
        self.step_count += 1
        XF.clip_adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.sumsq, self.max_norm or 0.0,
                           self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count)

    def grad_norm(self):
        """Global gradient norm (before clipping) of the last step: device scalar."""
        return self.sumsq.sqrt()[0]

    def state_dict(self):
        return {'m': self.flat_m.clone(), 'v': self.flat_v.clone(), 'step': self.step_count, 'lr': self.lr}

    def load_state_dict(self, sd):
        self.flat_m.copy_(sd['m']); self.flat_v.copy_(sd['v'])
        self.step_count, self.lr = sd['step'], sd['lr']


class LinearLR:
    """torch.optim.lr_scheduler.LinearLR closed form (reference models.py:381-384), stepped per epoch."""

    def __init__(self, opt, start_factor=1.0, end_factor=0.01, total_iters=20):
        self.opt, self.s, self.e, self.n = opt, start_factor, end_factor, total_iters
        self.epoch = 0
        self._apply()

    def _apply(self):
        f = self.s + (self.e - self.s) * min(self.n, self.epoch) / self.n
        self.opt.lr = self.opt.base_lr * f

    def step(self):
        self.epoch += 1
        self._apply()

    def get_last_lr(self):
        return [self.opt.lr]


class ModelCheckpoint:
    """Keeps the best ``state_dict`` by a monitored metric (in memory; ``dirpath`` optional file)."""

    def __init__(self, monitor='val_acc', mode='max', dirpath=None, filename='best', **_):
        self.monitor, self.mode, self.dirpath, self.filename = monitor, mode, dirpath, filename
        self.best_score, self.best_state, self.best_model_path = None, None, None

    def update(self, metrics, model):
        if self.monitor not in metrics:
            return
        v = float(metrics[self.monitor])
        better = self.best_score is None or (v > self.best_score if self.mode == 'max' else v < self.best_score)
        if better:
            self.best_score = v
            self.best_state = {k: t.detach().clone() for k, t in model.state_dict().items()}
            if self.dirpath is not None and _world()[1] == 0:
                os.makedirs(self.dirpath, exist_ok=True)
                self.best_model_path = os.path.join(self.dirpath, self.filename + '.ckpt')
                torch.save({'state_dict': self.best_state}, self.best_model_path)


class LearningRateMonitor:
    def __init__(self, logging_interval='epoch', **_):
        self.history = []


def _shard(t, rank, world):
    return t if world == 1 else t[rank::world]


class Trainer:
    """Minimal Lightning-style loop around the HIP model; see module docstring."""

    def __init__(self, max_epochs=500, gradient_clip_val=None, accelerator='auto', devices='auto', callbacks=None,
                 logger=True, enable_progress_bar=False, process_group=None, **_):
        self.max_epochs = max_epochs
        self.gradient_clip_val = gradient_clip_val
        self.callbacks = callbacks or []
        self.group = process_group
        self.logged_metrics = {}
        self.current_epoch = 0
        self.optimizer = None
        self.scheduler = None

    # ---- helpers ------------------------------------------------------------------------------
    def _device(self):
        if not torch.cuda.is_available():
            raise RuntimeError('the HIP path needs a GPU; there is no CPU fallback')
        return torch.device('cuda', torch.cuda.current_device())

    def _setup_optimizer(self, model):
        cfg = model.configure_optimizers()
        sched = None
        if isinstance(cfg, dict):
            opt = cfg['optimizer']
            if 'lr_scheduler' in cfg:
                sched = cfg['lr_scheduler']['scheduler']
        elif isinstance(cfg, (tuple, list)):             # Lightning's ([optimizers], [schedulers]) form
            opts, scheds = (cfg[0], cfg[1]) if len(cfg) == 2 and isinstance(cfg[0], (tuple, list)) else (cfg, [])
            if len(opts) != 1 or len(scheds) > 1:
                raise NotImplementedError('the HIP trainer drives one optimizer (and at most one scheduler)')
            opt, sched = opts[0], (scheds[0] if scheds else None)
        else:
            opt = cfg
        g = opt.param_groups[0]
        group = self.group
        if group is None and _world()[0] > 1:
            group = dist.group.WORLD                      # one process per GPU: the default group is the data-parallel group
        self.optimizer = FlatAdamW(model, lr=g['lr'], betas=g['betas'], eps=g['eps'], weight_decay=g['weight_decay'],
                                   max_norm=self.gradient_clip_val, group=group)
        self.scheduler = None
        if sched is not None:
            sch = sched
            if isinstance(sch, torch.optim.lr_scheduler.LinearLR):
                self.scheduler = LinearLR(self.optimizer, sch.start_factor, sch.end_factor, sch.total_iters)
            else:
                raise NotImplementedError(f'scheduler {type(sch).__name__} is not supported by the HIP trainer')

    def _reduce_metrics(self, sums, count):
        world, _ = _world(self.group)
        if world > 1:
            t = torch.tensor([count] + [float(v) for v in sums.values()], dtype=torch.float64, device=self._device())
            dist.all_reduce(t, group=self.group)
            count = t[0].item()
            sums = {k: t[i + 1].item() for i, k in enumerate(sums)}
        return {k: v / max(count, 1) for k, v in sums.items()}

    def _run_eval(self, model, loader, stage):
        world, rank = _world(self.group)
        dev = self._device()
        model.eval()
        sums, count = {}, 0
        with torch.no_grad():
            for bi, batch in enumerate(loader):
                batch = tuple(_shard(t, rank, world).to(dev) for t in batch)
                x = batch[0]
                if x.shape[0] == 0:
                    continue
                model._xps_logged = {}
                getattr(model, f'{stage}_step' if stage != 'val' else 'validation_step')(batch, bi)
                n = x.shape[0]
                for k, v in model._xps_logged.items():
                    sums[k] = sums.get(k, 0.0) + float(v) * n
                count += n
        out = self._reduce_metrics(sums, count)
        XF.check_gru_status()
        return out

    # ---- public API ---------------------------------------------------------------------------
    def fit(self, model, train_dataloaders=None, val_dataloaders=None):
        dev = self._device()
        world, rank = _world(self.group)
        model.to(dev)
        model.trainer = self
        if hasattr(model, 'temporal_conv'):
            model.temporal_conv.process_group = self.group if world > 1 else None
            if world > 1 and self.group is None:
                model.temporal_conv.process_group = dist.group.WORLD
        self._setup_optimizer(model)
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            model.train()
            sums, count = {}, 0
            for bi, batch in enumerate(train_dataloaders):
                n_global = batch[0].shape[0]
                batch = tuple(_shard(t, rank, world).to(dev) for t in batch)      # (x, y) or (x, y, lengths...)
                x = batch[0]
                if hasattr(model, 'temporal_conv'):
                    model.temporal_conv.global_batch = n_global
                model._xps_logged = {}
                self.optimizer.zero_grad()
                loss = model.training_step(batch, bi)
                if world > 1:                       # global mean over unequal shards
                    loss = loss * (x.shape[0] * world / n_global)
                loss.backward(XF.unit_gradient(loss.device) if loss.dtype == torch.float32 else torch.ones_like(loss))
                # (resident root gradient: no fill launch per step, and the fused cross-entropy returns its gradient unscaled)
                self.optimizer.step()
                n = x.shape[0]
                for k, v in model._xps_logged.items():
                    sums[k] = sums.get(k, 0.0) + float(v) * n
                count += n
            metrics = self._reduce_metrics(sums, count)
            XF.check_gru_status()                     # (the loss was just read: the device is synchronised anyway)
            if val_dataloaders is not None:
                metrics.update(self._run_eval(model, val_dataloaders, 'val'))
            metrics['lr'] = self.optimizer.lr
            self.logged_metrics.update(metrics)
            for cb in self.callbacks:
                if isinstance(cb, ModelCheckpoint):
                    cb.update(metrics, model)
            if self.scheduler is not None:
                self.scheduler.step()
        torch.cuda.synchronize(dev)
        XF.check_gru_status()
        return self

    def test(self, model, dataloaders=None, ckpt_path=None):
        model.to(self._device())
        if ckpt_path == 'best':
            for cb in self.callbacks:
                if isinstance(cb, ModelCheckpoint) and cb.best_state is not None:
                    model.load_state_dict(cb.best_state)
        elif ckpt_path:
            model.load_state_dict(torch.load(ckpt_path, map_location='cpu')['state_dict'])
        metrics = self._run_eval(model, dataloaders, 'test')
        self.logged_metrics.update(metrics)
        return [metrics]

    def predict(self, model, dataloaders=None):
        """Lightning's predict loop: list of ``predict_step`` outputs (this rank's shard of every batch)."""
        world, rank = _world(self.group)
        dev = self._device()
        model.to(dev).eval()
        outs = []
        with torch.no_grad():
            for bi, batch in enumerate(dataloaders):
                batch = tuple(_shard(t, rank, world).to(dev) for t in batch)
                outs.append(model.predict_step(batch, bi))
        torch.cuda.synchronize(dev)
        XF.check_gru_status()
        return outs

    def validate(self, model, dataloaders=None):
        metrics = self._run_eval(model, dataloaders, 'val')
        self.logged_metrics.update(metrics)
        return [metrics]


def seed_everything(seed):
    import random

    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed
