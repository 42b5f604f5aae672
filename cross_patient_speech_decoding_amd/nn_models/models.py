"""Seq2seq GRU phoneme decoder on MI355X — drop-in surface of the reference's
``nn_models/models.py`` (BaseLightningModel :15-108, Seq2SeqRNN :208-390, TemporalConv
:599-636, EncoderRNN :639-716, DecoderRNN :719-761, cmat_acc :875-889).

Same constructor argument order, attribute names, ``forward`` signature, Lightning hooks,
logged metric names and ``state_dict`` keys as the reference, so scripts/train_seq2seq.py
and reference checkpoints work unchanged.  The arithmetic runs in hand-written HIP kernels
(libxps.so) through ``functional``; the ``torch.nn`` sub-modules below are parameter
containers only (identical names, shapes and default initialisation order) and their own
``forward`` is never called.
"""
import os

import torch
import torch.nn as nn

from . import functional as XF
from ._lightning import LightningModule


def cmat_acc(y_hat, y, num_classes):
    """Accuracy = trace / sum of the confusion matrix of argmax(y_hat) (reference :875-889).
    The confusion matrix is a device bincount; no torchmetrics dependency."""
    y_pred = torch.argmax(y_hat, dim=1)
    cm = torch.bincount(y * num_classes + y_pred, minlength=num_classes * num_classes)
    cm = cm.view(num_classes, num_classes)
    return cm.diag().sum() / cm.sum()


class _HipCrossEntropyLoss(nn.CrossEntropyLoss):
    """nn.CrossEntropyLoss() whose default configuration dispatches to the HIP kernel."""

    def forward(self, input, target):
        plain = (self.weight is None and self.reduction == 'mean' and self.ignore_index == -100
                 and self.label_smoothing == 0.0 and input.dim() == 2 and input.is_cuda)
        if plain:
            return XF.cross_entropy(input, target)
        return super().forward(input, target)


class BaseLightningModel(LightningModule):
    """Shared step/optimiser logic (reference :15-108)."""

    def __init__(self, criterion=None, learning_rate=1e-3, l2_reg=1e-5):
        super().__init__()
        self.criterion = criterion if criterion is not None else _HipCrossEntropyLoss()
        self.learning_rate = learning_rate
        self.l2_reg = l2_reg

    def _shared_step(self, batch, stage):
        x, y = batch
        y_hat = self(x)
        loss = self.criterion(y_hat, y)
        acc = cmat_acc(y_hat, y, self.num_classes)
        self.log_dict({f'{stage}_loss': loss, f'{stage}_acc': acc}, prog_bar=True)
        return loss

    def training_step(self, batch, batch_idx):
        return self._shared_step(batch, 'train')

    def validation_step(self, batch, batch_idx):
        return self._shared_step(batch, 'val')

    def test_step(self, batch, batch_idx):
        return self._shared_step(batch, 'test')

    def predict_step(self, batch, batch_idx):
        x, _ = batch
        return self(x)

    def configure_optimizers(self):
        return torch.optim.AdamW(self.parameters(), lr=self.learning_rate, weight_decay=self.l2_reg)


class TemporalConv(nn.Module):
    """Conv1d -> BatchNorm1d -> [ReLU] -> Dropout (reference :599-636), fused HIP path.
    ``forward`` keeps the reference layout (B, C, T) -> (B, F, T'); ``forward_tm`` takes
    (B, T, C) and returns the time-major (T', B, F) tensor the encoder consumes."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dropout=0.2,
                 activation=True):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding)
        self.bn = nn.BatchNorm1d(out_channels)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        self.activation = activation
        self.process_group = None        # set by the data-parallel trainer (SyncBN statistics)
        self.global_batch = None         # trials of the current batch summed over ranks (set by the trainer)

    def forward_tm(self, x):
        k, s, pad = self.conv.kernel_size[0], self.conv.stride[0], self.conv.padding[0]
        if pad > 0:
            x = torch.nn.functional.pad(x, (0, 0, pad, pad))
        B, T, _ = x.shape
        Tp = (T - k) // s + 1
        F = self.conv.out_channels
        training = self.training
        mask, scale, p = None, 1.0, self.dropout.p
        if training and p > 0:
            mask = XF.dropout_mask((Tp, B, F), p, x.device)
            scale = 1.0 / (1.0 - p)
        momentum = 0.1 if self.bn.momentum is None else self.bn.momentum
        # num_batches_tracked is bumped by the statistics kernel (no separate launch for one integer)
        nbt = self.bn.num_batches_tracked if training and self.bn.track_running_stats else None
        return XF.TemporalConvFn.apply(x, self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias,
                                       self.bn.running_mean, self.bn.running_var, s, training,
                                       bool(self.activation), mask, scale, momentum, self.bn.eps,
                                       self.process_group, self.global_batch if self.process_group is not None else None,
                                       nbt)

    def forward(self, x):
        return self.forward_tm(x.permute(0, 2, 1)).permute(1, 2, 0)


def _gru_layer_weights(rnn, layer, ndir):
    out = []
    for d in range(ndir):
        sfx = f'_l{layer}' + ('_reverse' if d == 1 else '')
        out += [getattr(rnn, 'weight_ih' + sfx), getattr(rnn, 'weight_hh' + sfx),
                getattr(rnn, 'bias_ih' + sfx), getattr(rnn, 'bias_hh' + sfx)]
    return out


class EncoderRNN(nn.Module):
    """Bidirectional multi-layer GRU encoder (reference :639-716); last hidden =
    h_n[-1, fwd] + h_n[-1, bwd] as (1, B, H)."""

    def __init__(self, input_size, hidden_size, num_layers, dropout=0.3, model_type='gru'):
        super().__init__()
        self.model_type = model_type
        if model_type == 'gru':
            self.rnn = nn.GRU(input_size, hidden_size, num_layers, batch_first=True, dropout=dropout,
                              bidirectional=True)
        elif model_type == 'lstm':
            raise NotImplementedError("model_type='lstm' is not on the accelerated path (the reference's LSTM "
                                      'branch is itself broken: models.py:280 repeats a tuple)')
        else:
            raise ValueError('model_type must be one of "gru" or "lstm"')

    def forward_tm(self, x):
        """x: (T, B, In) time-major -> (y (T, B, 2H), last_hidden (1, B, H))."""
        y, last = self.forward_tm_last(x)
        return y, last.unsqueeze(0)

    def forward_tm_last(self, x):
        """As forward_tm with the summed final state as a plain (B, H) tensor."""
        rnn = self.rnn
        L = rnn.num_layers
        y = x
        fmt = 0
        for l in range(L):
            # inter-layer dropout (all layers but the last, training only) travels with the layer: its recurrence kernels
            # write the dropped output and re-make the decisions in the backward pass (no separate passes over y / dy)
            p_drop = float(rnn.dropout) if (l < L - 1 and self.training) else 0.0
            # a dropped output that only the next layer's GEMMs read is written as XPS_FMT_SPLIT4 groups (include/xps.h) where
            # the layer runs a dropout pass of its own (H = 512 / 500): those GEMMs then stage it without conversion arithmetic
            T_, B_ = y.shape[0], y.shape[1]
            out_split = XF.layer_output_split4_ok(T_, B_, rnn.hidden_size, 2, p_drop)
            y, last = XF.GRULayerFmtFn.apply(y, 2, XF.HN_SUM if l == L - 1 else XF.HN_NONE, p_drop,
                                             fmt | (XF.FMT_Y_SPLIT4 if out_split else 0), *_gru_layer_weights(rnn, l, 2))
            fmt = XF.FMT_X_SPLIT4 if out_split else 0
        return y, last                                    # last = h_fwd(T-1) + h_bwd(0)

    def forward(self, x):
        y, h = self.forward_tm(x.permute(1, 0, 2))
        return y.permute(1, 0, 2), h


class DecoderRNN(nn.Module):
    """Embedding -> GRU (one step) -> Linear (reference :719-761)."""

    def __init__(self, hidden_size, output_size, num_layers, dropout=0.3, model_type='gru'):
        super().__init__()
        self.embedding = nn.Embedding(output_size + 1, hidden_size)
        if model_type != 'gru':
            raise NotImplementedError("model_type='lstm' is not on the accelerated path")
        self.rnn = nn.GRU(hidden_size, hidden_size, num_layers, batch_first=True, dropout=dropout)
        self.fc_out = nn.Linear(hidden_size, output_size)

    def token_projection(self):
        """(n_tokens, 3H) table  E W_ih^T + b_ih : the layer-0 input projection of every
        possible token, computed once per forward instead of once per step and trial."""
        return XF.linear(self.embedding.weight, self.rnn.weight_ih_l0, self.rnn.bias_ih_l0)

    def step(self, tok, hidden, table):
        """tok (B,), hidden (L, B, H) -> logits (B, n_out), new hidden (L, B, H)."""
        rnn = self.rnn
        B = tok.shape[0]
        H = rnn.hidden_size
        new_h = []
        gi = XF.gather_rows(table, tok).view(1, 1, B, 3 * H)
        y_ext = XF.GRURecurFn.apply(gi, hidden[0:1], 1, rnn.weight_hh_l0, rnn.bias_hh_l0)
        h = y_ext[1]
        new_h.append(h)
        for l in range(1, rnn.num_layers):
            inp = XF.dropout(h, rnn.dropout, self.training)
            gi = XF.linear(inp, getattr(rnn, f'weight_ih_l{l}'), getattr(rnn, f'bias_ih_l{l}')).view(1, 1, B, 3 * H)
            y_ext = XF.GRURecurFn.apply(gi, hidden[l:l + 1], 1, getattr(rnn, f'weight_hh_l{l}'),
                                        getattr(rnn, f'bias_hh_l{l}'))
            h = y_ext[1]
            new_h.append(h)
        logits = XF.linear(h, self.fc_out.weight, self.fc_out.bias)
        return logits, torch.stack(new_h, dim=0)

    def forward(self, x, hidden):
        return self.step(x, hidden, self.token_projection())


class Seq2SeqRNN(BaseLightningModel):
    """TemporalConv -> bidirectional GRU encoder -> autoregressive GRU decoder
    (reference :208-390).  Positional constructor order is the reference's (:235-239)."""

    def __init__(self, in_channels, n_filters, hidden_size, num_classes, n_enc_layers, n_dec_layers,
                 kernel_size, stride=1, padding=0, cnn_dropout=0.3, rnn_dropout=0.3, model_type='gru',
                 learning_rate=1e-3, l2_reg=1e-5, criterion=None, activation=True, seq_length=3,
                 decay_iters=20):
        super().__init__(learning_rate=learning_rate, l2_reg=l2_reg, criterion=criterion)
        self.num_classes = num_classes
        self.seq_length = seq_length
        self.temporal_conv = TemporalConv(in_channels, n_filters, kernel_size, stride, padding, cnn_dropout,
                                          activation=activation)
        self.encoder = EncoderRNN(n_filters, hidden_size, n_enc_layers, dropout=rnn_dropout,
                                  model_type=model_type)
        self.decoder = DecoderRNN(hidden_size, num_classes, n_dec_layers, dropout=rnn_dropout,
                                  model_type=model_type)
        self.decay_iters = decay_iters

    def draw_teacher_coins(self, y, teacher_forcing_ratio):
        """The reference flips one host coin per decode step for the whole batch
        (``torch.rand(1).item() < ratio``, :295, evaluated only when y is given).  The same
        draws from the same global CPU generator are made here, up front, so a fixed seed
        gives the reference's coin sequence and the decode loop itself never syncs."""
        coins = []
        for _ in range(self.seq_length):
            coins.append(bool(y is not None and torch.rand(1).item() < teacher_forcing_ratio))
        return coins

    def forward(self, x, y=None, teacher_forcing_ratio=0.5, coins=None):
        """x (B, T, C), y (B, seq_length) or None -> logits (B, seq_length, num_classes)."""
        if coins is None:
            coins = self.draw_teacher_coins(y, teacher_forcing_ratio)
        return self.forward_device(x, y, self._device_flags(tuple(bool(c) and y is not None for c in coins), x.device))

    def _device_flags(self, bits, device):
        """Device int32 vector of teacher-forcing decisions.  One small tensor per coin pattern is uploaded once and kept
        (2^seq_length patterns at most): the per-step host-to-device copy cost 5 us of GPU timeline."""
        cache = self.__dict__.setdefault('_flag_cache', {})
        key = (bits, device)
        t = cache.get(key)
        if t is None:
            t = torch.tensor([int(b) for b in bits], dtype=torch.int32).to(device)
            if len(cache) < 4096:
                cache[key] = t
        return t

    def forward_device(self, x, y, flags):
        """Sync-free forward: ``flags`` is a DEVICE int32 vector (seq_length,) of teacher-forcing
        decisions, so the whole step can be captured in a hipGraph."""
        z = self.temporal_conv.forward_tm(x)                       # (T', B, F)
        _, enc_last = self.encoder.forward_tm_last(z)               # (B, H)
        rnn = self.decoder.rnn
        if (rnn.num_layers == 1 and self.num_classes + 1 <= 16
                and XF.decoder_supported(rnn.hidden_size, self.num_classes, self.seq_length)):
            # fused path: every decode step in one launch, tokens chosen on the device
            table = self.decoder.token_projection()
            logits, _ = XF.DecoderFn.apply(table, enc_last, rnn.weight_hh_l0, rnn.bias_hh_l0,
                                           self.decoder.fc_out.weight, self.decoder.fc_out.bias,
                                           y if y is not None else None, flags if y is not None else None,
                                           self.num_classes, self.seq_length)
            return logits
        if rnn.num_layers == 1 and rnn.hidden_size % 4 == 0 and os.environ.get('XPS_DECODER_WIDE', '1') != '0':
            # other hidden sizes (H = 500 / 512 of the north-star shape): the same schedule on the general GRU entry points,
            # ONE backward launch per kind for all decode steps (XF.DecoderWideFn)
            table = self.decoder.token_projection()
            logits, _ = XF.DecoderWideFn.apply(table, enc_last, rnn.weight_hh_l0, rnn.bias_hh_l0,
                                               self.decoder.fc_out.weight, self.decoder.fc_out.bias,
                                               y if y is not None else None, flags if y is not None else None,
                                               self.num_classes, self.seq_length)
            return logits
        dec_hidden = enc_last.unsqueeze(0).repeat(self.decoder.rnn.num_layers, 1, 1)
        B = x.size(0)
        tok = torch.full((B,), self.num_classes, dtype=torch.long, device=x.device)
        table = self.decoder.token_projection()
        outputs = []
        for i in range(self.seq_length):
            logits, dec_hidden = self.decoder.step(tok, dec_hidden, table)
            outputs.append(logits)
            if i + 1 < self.seq_length:
                teacher = y[:, i] if y is not None else None
                tok = XF.next_token(logits.detach(), teacher, flags[i:i + 1] if y is not None else None)
        return torch.stack(outputs, dim=1)

    def _seq_step(self, batch, stage, ratio):
        x, y = batch
        y_hat = self(x, y, teacher_forcing_ratio=ratio).view(-1, self.num_classes)
        y = y.reshape(-1)
        loss = self.criterion(y_hat, y)
        acc = cmat_acc(y_hat, y, self.num_classes)
        self.log_dict({f'{stage}_loss': loss, f'{stage}_acc': acc}, prog_bar=True)
        return loss

    def training_step(self, batch, batch_idx):
        return self._seq_step(batch, 'train', 0.5)

    def validation_step(self, batch, batch_idx):
        return self._seq_step(batch, 'val', 0)

    def test_step(self, batch, batch_idx):
        return self._seq_step(batch, 'test', 0)

    def configure_optimizers(self):
        """AdamW + LinearLR(1.0 -> 0.01 over decay_iters epochs), reference :367-390."""
        optim = torch.optim.AdamW(self.parameters(), lr=self.learning_rate, weight_decay=self.l2_reg)
        lr_sch = torch.optim.lr_scheduler.LinearLR(optim, start_factor=1.0, end_factor=0.01,
                                                   total_iters=self.decay_iters)
        return {'optimizer': optim, 'lr_scheduler': {'scheduler': lr_sch, 'interval': 'epoch', 'frequency': 1}}
