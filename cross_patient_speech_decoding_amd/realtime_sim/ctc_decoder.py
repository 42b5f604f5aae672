"""Greedy CTC decoding — counterpart of the reference's realtime_sim/ctc_decoder.py:172-189 (the only
decoder the scripts import, scripts/train_ctc_rnn.py:26).  Collapse repeats, drop blanks."""
import torch


def greedy_decode_batch(log_probs, blank=0):
    """log_probs (B, T, C) -> list of 1-D LongTensors (on the input's device)."""
    best = log_probs.argmax(dim=2)
    keep = torch.ones_like(best, dtype=torch.bool)
    keep[:, 1:] = best[:, 1:] != best[:, :-1]
    keep &= best != blank
    return [best[b][keep[b]] for b in range(best.size(0))]
