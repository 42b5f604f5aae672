"""Oracle: plain torch (CPU) restatement of the reference's realtime CTC-RNN forward
(realtime_sim/realtime_nn_model.py:153-199: right-aligned unfold windows -> nn.GRU with the trainable
h0 expanded over the batch -> Linear) and of greedy_decode_batch (ctc_decoder.py:172-189).

TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/realtime_*.npz (generated from the reference's own
RealtimeRNNModel with glue modules for lightning / torchaudio / torchmetrics, see
tests/golden/make_realtime_fixtures.py)."""
import torch
import torch.nn as nn


class RealtimeOracle(nn.Module):
    def __init__(self, input_size, hidden_size, n_layers, n_classes, win_size=14, stride=4):
        super().__init__()
        self.gru = nn.GRU(input_size, hidden_size, n_layers, batch_first=True)
        self.h0 = nn.Parameter(torch.zeros(n_layers, 1, hidden_size))
        self.fc = nn.Linear(hidden_size, n_classes)
        self.win, self.stride = win_size, stride

    def load_reference_state(self, sd):
        own = {k.replace('rnn.rnn.', 'gru.').replace('classifier.fc.', 'fc.'): v for k, v in sd.items()}
        self.load_state_dict(own)

    def windows(self, x):                       # :172-199
        B, T, C = x.shape
        u = x.permute(0, 2, 1).unsqueeze(2).unfold(3, self.win, self.stride).squeeze(2)     # (B, C, nw, win)
        return u.permute(0, 2, 3, 1).reshape(B, u.shape[2], self.win * C)

    def forward(self, x):                       # :153-170
        w = self.windows(x)
        out, _ = self.gru(w, self.h0.expand(-1, x.shape[0], -1).contiguous())
        return self.fc(out)


def greedy_decode_batch(log_probs, blank=0):    # ctc_decoder.py:172-189
    best = log_probs.argmax(dim=2)
    out = []
    for b in range(best.size(0)):
        p = best[b]
        keep = torch.ones_like(p, dtype=torch.bool)
        keep[1:] = p[1:] != p[:-1]
        out.append(p[keep & (p != blank)])
    return out


def ctc_step_loss(model, batch, adjust=True, blank=0):
    """training_step / validation_step loss of the reference (realtime_nn_model.py:201-232): the input lengths are
    converted to window counts, log_softmax over classes, nn.CTCLoss(blank, zero_infinity=True) on (T, B, C).
    ``adjust=False`` is test_step (:283-286), which hands the lengths over as they are."""
    inputs, targets, input_lengths, target_lengths = batch
    il = ((input_lengths - model.win) // model.stride) + 1 if adjust else input_lengths
    log_probs = model(inputs).log_softmax(2).permute(1, 0, 2)
    return nn.CTCLoss(blank=blank, zero_infinity=True)(log_probs, targets, il, target_lengths)


def edit_distance(a, b):
    a, b = [int(v) for v in a], [int(v) for v in b]
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def calc_per(decoded, targets, target_lengths):                 # :303-324
    dist = sum(edit_distance(p.tolist(), t[:int(l)].tolist()) for p, t, l in zip(decoded, targets, target_lengths))
    return dist / float(target_lengths.sum()) * 100
