"""Realtime (streaming) CTC-RNN inference on MI355X — counterpart of the reference's
realtime_sim/realtime_nn_model.py (StackedRNN :22, DenseClassifier :66, RealtimeRNNModel :93,
forward :153, reformat_time_windows :172).  BASELINE config 5: per-step GRU inference captured in a
hipGraph.

Scope: forward with the reference's parameter names (``rnn.rnn.*``, ``h0``,
``classifier.fc.*``) so reference checkpoints load, and CTC training (training/validation/test steps,
fused CTC loss kernel, PER).

* ``forward(x)``: the right-aligned sliding windows are never materialised — window w of trial b is the
  contiguous ``win*C`` floats at ``x[b, w*stride]``, read by the input-projection GEMM through a row map;
  the recurrence runs in the fused GRU kernel with the trainable ``h0``.
* ``StreamingDecoder``: one 20 ms step (window -> L GRU cells -> Linear) recorded ONCE into a hipGraph on
  static device buffers and replayed per step; the hidden state never leaves the device.
"""
import torch
import torch.nn as nn

from ..nn_models import functional as XF
from ..nn_models._lightning import LightningModule
from .._lib import rowmap  # noqa: F401
from .ctc_decoder import greedy_decode_batch


class StackedRNN(nn.Module):
    def __init__(self, input_size, hidden_size, n_layers, dropout=0.3, bidirectional=False):
        super().__init__()
        self.rnn = nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=n_layers,
                          dropout=dropout if n_layers > 1 else 0, bidirectional=bidirectional, batch_first=True)


class DenseClassifier(nn.Module):
    def __init__(self, input_size, n_classes):
        super().__init__()
        self.fc = nn.Linear(input_size, n_classes)


def _layer_params(rnn, layer, ndir):
    out = []
    for d in range(ndir):
        sfx = f'_l{layer}' + ('_reverse' if d else '')
        out.append(tuple(getattr(rnn, n + sfx) for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')))
    return out


class _HipCTCLoss(nn.CTCLoss):
    """nn.CTCLoss whose default configuration dispatches to the fused HIP kernel.  Takes (T, B, C) log-probs as
    nn.CTCLoss does (the kernel re-normalises: a no-op on log-probabilities)."""

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        if self.reduction == 'mean' and log_probs.is_cuda and log_probs.dim() == 3 and targets.dim() == 2:
            return XF.ctc_loss(log_probs, targets, input_lengths, target_lengths, blank=self.blank,
                               zero_infinity=self.zero_infinity)
        return super().forward(log_probs, targets, input_lengths, target_lengths)


def edit_distance(a, b):
    """Levenshtein distance of two label sequences (torchaudio.functional.edit_distance)."""
    a, b = [int(v) for v in a], [int(v) for v in b]
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def calc_PER(decoded, targets, target_lengths):
    """Phoneme error rate in percent (reference :303-324)."""
    targets, target_lengths = targets.cpu(), target_lengths.cpu()
    dist = sum(edit_distance(p.cpu().tolist(), t[:int(l)].tolist()) for p, t, l in zip(decoded, targets, target_lengths))
    return dist / float(target_lengths.sum()) * 100


class RealtimeRNNModel(LightningModule):
    def __init__(self, input_size, hidden_size, n_layers, n_classes, dropout=0.3, win_size=14, stride=4,
                 bidirectional=False, learning_rate=1e-3, decay_steps=100, weight_decay=1e-5, blank=0):
        super().__init__()
        self.hparams_ = dict(input_size=input_size, hidden_size=hidden_size, n_layers=n_layers, n_classes=n_classes,
                             dropout=dropout, win_size=win_size, stride=stride, bidirectional=bidirectional,
                             learning_rate=learning_rate, decay_steps=decay_steps, weight_decay=weight_decay,
                             blank=blank)
        self.rnn = StackedRNN(input_size, hidden_size, n_layers, dropout, bidirectional)
        for name, param in self.rnn.named_parameters():
            if 'weight_hh' in name:
                nn.init.orthogonal_(param)
            if 'weight_ih' in name:
                nn.init.xavier_uniform_(param)
        ndir = 2 if bidirectional else 1
        self.h0 = nn.Parameter(torch.zeros(n_layers * ndir, 1, hidden_size))
        nn.init.xavier_uniform_(self.h0)
        self.classifier = DenseClassifier(hidden_size * ndir, n_classes)
        with torch.no_grad():
            self.classifier.fc.bias[:] = -2.0
            self.classifier.fc.bias[blank] = 2.0
        self.win_size, self.stride, self.blank = win_size, stride, blank
        self.criterion = _HipCTCLoss(blank=blank, zero_infinity=True)

    def n_windows(self, T):
        return (T - self.win_size) // self.stride + 1

    def forward_tm(self, x):
        """x (B, T, C) -> TIME-major logits (n_windows, B, n_classes); differentiable (weights and h0)."""
        rnn = self.rnn.rnn
        ndir = 2 if rnn.bidirectional else 1
        H, L = rnn.hidden_size, rnn.num_layers
        x = x.contiguous()
        B, T, Cc = x.shape
        nw = self.n_windows(T)
        if self.win_size * Cc != rnn.input_size:
            raise ValueError(f'input_size {rnn.input_size} != win_size * channels = {self.win_size * Cc}')
        inp = None
        for l in range(L):
            params = _layer_params(rnn, l, ndir)
            if l == 0:          # windows addressed through a row map, never materialised
                gis = [XF.WindowLinearFn.apply(x, w_ih, b_ih, self.win_size, self.stride) for w_ih, _, b_ih, _ in params]
            else:
                gis = [XF.linear(inp, w_ih, b_ih) for w_ih, _, b_ih, _ in params]
            gi = gis[0].unsqueeze(0) if ndir == 1 else torch.stack(gis, dim=0)
            h0 = self.h0[l * ndir:(l + 1) * ndir].expand(-1, B, -1).contiguous()
            y_ext = XF.GRURecurFn.apply(gi, h0, ndir, *[p[1] for p in params], *[p[3] for p in params])
            inp = y_ext[1:nw + 1]
            if l < L - 1:       # nn.GRU: dropout on the outputs of every layer but the last
                inp = XF.dropout(inp, rnn.dropout, self.training)
        return XF.linear(inp.contiguous(), self.classifier.fc.weight, self.classifier.fc.bias)     # (nw, B, C)

    def forward(self, x):
        """x (B, T, C) -> logits (B, n_windows, n_classes)  (reference :153-170)."""
        return self.forward_tm(x).permute(1, 0, 2).contiguous()

    # ---- training (reference :201-300): CTC loss on the window outputs, PER on validation -------------------
    def _adjusted_lengths(self, input_lengths):
        return ((input_lengths - self.win_size) // self.stride) + 1

    def _ctc(self, batch, adjust=True):
        inputs, targets, input_lengths, target_lengths = batch
        il = self._adjusted_lengths(input_lengths) if adjust else input_lengths
        logits_tm = self.forward_tm(inputs)
        # the fused kernel takes raw time-major scores (the log-softmax is inside): same value as
        # criterion(log_softmax(logits).permute(1, 0, 2), ...) without the (B, T, C) log-prob tensor
        loss = XF.ctc_loss(logits_tm, targets, il, target_lengths, blank=self.blank, zero_infinity=True)
        return loss, logits_tm

    def training_step(self, batch, batch_idx):
        loss, _ = self._ctc(batch)
        self.log('train_loss', loss, on_step=False, on_epoch=True, prog_bar=True)
        return loss

    def validation_step(self, batch, batch_idx):
        loss, logits_tm = self._ctc(batch)
        self.log('val_loss', loss, on_step=False, on_epoch=True, prog_bar=True)
        with torch.no_grad():
            decoded = greedy_decode_batch(logits_tm.permute(1, 0, 2), blank=self.blank)   # argmax: softmax-invariant
            per = calc_PER(decoded, batch[1], batch[3])
            self.log('val_PER', per, on_step=False, on_epoch=True, prog_bar=True)
        return loss

    def test_step(self, batch, batch_idx):
        loss, _ = self._ctc(batch, adjust=False)           # the reference passes the raw lengths here (:283-286)
        self.log('test_loss', loss)
        return loss

    def configure_optimizers(self):
        hp = self.hparams_
        optimizer = torch.optim.AdamW(self.parameters(), lr=hp['learning_rate'], weight_decay=hp['weight_decay'])
        scheduler = torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=1.0, end_factor=0.0,
                                                      total_iters=hp['decay_steps'])
        return [optimizer], [scheduler]

    def reformat_time_windows(self, x):
        """(B, T, C) -> (B, n_windows, win*C) right-aligned windows (materialised; for inspection only)."""
        B, T, Cc = x.shape
        nw = self.n_windows(T)
        idx = (torch.arange(nw, device=x.device) * self.stride)[:, None] + torch.arange(self.win_size, device=x.device)
        return x[:, idx, :].reshape(B, nw, self.win_size * Cc)


class StreamingDecoder:
    """Per-step streaming inference replayed from a hipGraph: ``step(window)`` runs one window (win*C
    floats per stream, 1..8 streams) through all GRU layers and the classifier with the weight-streaming
    GEMV kernels (xps_gru_cell_gemv_f32: one launch per layer, xps_gemv_f32 for the classifier); the
    hidden state lives in two static device buffers that alternate between steps (two captured graphs)."""

    def __init__(self, model, n_streams=1, use_graph=True):
        from .._lib import call
        rnn = model.rnn.rnn
        if rnn.bidirectional:
            raise ValueError('streaming decode needs a unidirectional model')
        if not 1 <= n_streams <= 8:
            raise ValueError('1..8 streams per decoder')
        self.model, self.B = model, n_streams
        self.H, self.L, self.K = rnn.hidden_size, rnn.num_layers, rnn.input_size
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('StreamingDecoder needs the model on the GPU (no CPU fallback)')
        self.dev, self._call = dev, call
        S = 1 << (n_streams - 1).bit_length()                 # rows allocated: power of two >= B
        self.n_classes = model.classifier.fc.out_features
        self.window = torch.zeros(S, self.K, dtype=torch.float32, device=dev)
        self.hbuf = torch.zeros(2, self.L, S, self.H, dtype=torch.float32, device=dev)    # ping-pong state
        self._logits = torch.zeros(S, self.n_classes, dtype=torch.float32, device=dev)
        self._token = torch.zeros(S, dtype=torch.int64, device=dev)
        self._params = [tuple(p.detach().contiguous() for p in _layer_params(rnn, l, 1)[0]) for l in range(self.L)]
        self._fc = (model.classifier.fc.weight.detach().contiguous(), model.classifier.fc.bias.detach().contiguous())
        self.parity = 0
        self.graphs = None
        self.reset()
        if use_graph:
            for par in (0, 1):                                # warm-up: module load, allocator
                self._body(par)
            torch.cuda.synchronize()
            self.graphs = []
            for par in (0, 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body(par)
                self.graphs.append(g)
            self.reset()

    @property
    def logits(self):
        return self._logits[:self.B]

    @property
    def token(self):
        return self._token[:self.B]

    @property
    def h(self):
        return self.hbuf[self.parity, :, :self.B]

    def reset(self):
        with torch.no_grad():
            self.hbuf.zero_()
            self.hbuf[:, :, :self.B] = self.model.h0.detach().expand(-1, self.B, -1)
        self.parity = 0

    @torch.no_grad()
    def _body(self, par):
        st = torch.cuda.current_stream().cuda_stream
        src, dst = self.hbuf[par], self.hbuf[par ^ 1]
        inp, k = self.window, self.K
        for l, (w_ih, w_hh, b_ih, b_hh) in enumerate(self._params):
            self._call('xps_gru_cell_gemv_f32', inp.data_ptr(), k, w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(),
                       b_hh.data_ptr(), src[l].data_ptr(), dst[l].data_ptr(), self.H, self.B, st)
            inp, k = dst[l], self.H
        self._call('xps_gemv_f32', inp.data_ptr(), self._fc[0].data_ptr(), self._fc[1].data_ptr(), self._logits.data_ptr(),
                   self.n_classes, self.H, self.B, st)
        self._call('xps_next_token', self._logits.data_ptr(), self.n_classes, None, 0, None, self._token.data_ptr(),
                   self.B, st)

    @torch.no_grad()
    def step(self, window=None):
        """window: (n_streams, win*C) device or host tensor (None: reuse the static buffer).  Returns the
        logits (n_streams, n_classes) in a static buffer; ``self.token`` holds the argmax class."""
        if window is not None:
            self.window[:self.B].copy_(window.reshape(self.B, self.K), non_blocking=True)
        if self.graphs is not None:
            self.graphs[self.parity].replay()
        else:
            self._body(self.parity)
        self.parity ^= 1
        return self.logits
