"""CPU restatement of the reference's permutation-invariant label assignment
-- TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

    batch_pit_n_speaker_loss   FS-EEND/train/utils/loss.py:257-327  (LS-EEND train/utils/loss.py:276-348, same body)
    pit_loss_multispk          LS-EEND/train/utils/loss.py:350-379

Pinned: tests/golden/pit_*.npz hold the outputs of the reference's own function bodies (evaluated out of the
reference files by oracle/gen_golden_pit.py: the modules import torchmetrics, absent here) on seeded inputs.
Bar: the chosen permutations / permuted labels are exact; the loss value within 1e-5 relative (the reference sums
fp32 element losses, a restatement may sum in another order).

Both functions reduce to one (C x C) cost matrix per utterance,
    cost[i][j] = sum_t BCEwithLogits(y[t, i], label[t, j])   over the -1-padded batch length,
(batch_pit: losses[b, i, s] = cost[i][(i + s) % C]; multispk: exactly `cost_mxs`) followed by an assignment on
its leading n_speakers x n_speakers block, the remaining slots keeping their place.
"""
from itertools import permutations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def cost_matrices(ys, ts):
    """(B, C, C) float64: cost[b, i, j] = sum over the padded length of BCE(y[b, t, i], t[b, t, j]), pad value -1."""
    y = nn.utils.rnn.pad_sequence(ys, padding_value=-1, batch_first=True).double()
    t = nn.utils.rnn.pad_sequence(ts, padding_value=-1, batch_first=True).double()
    sp = F.softplus(y).sum(dim=1)                                  # BCE(y, t) = softplus(y) - y t
    return sp[:, :, None] - torch.einsum("bti,btj->bij", y, t)


def batch_pit_n_speaker_loss(ys, ts, n_speakers_list):
    C = max(n_speakers_list)
    cost = cost_matrices(ys, ts)
    perms = list(permutations(range(C)))
    total, labels = 0.0, []
    for b, (t, n) in enumerate(zip(ts, n_speakers_list)):
        best, best_p = None, None
        for p in perms:                                            # lexicographic order, first minimum wins (torch.argmin)
            if p[n:] != tuple(range(n, C)):                        # only extensions of a permutation of the first n speakers
                continue
            v = sum(float(cost[b, i, p[i]]) for i in range(C)) / C
            if best is None or v < best:
                best, best_p = v, p
        total += best
        labels.append(t[:, list(best_p)][:, :n])
    n_frames = sum(t.shape[0] for t in ts)
    return torch.tensor(total / n_frames, dtype=torch.float32), labels


def pit_loss_multispk(logits, target, n_speakers):
    """target: list of (T_i, C) label tensors (what the reference pads with pad_sequence); returns the permuted,
    truncated targets."""
    from scipy.optimize import linear_sum_assignment
    C = max(n_speakers)
    cost = cost_matrices(logits, target).numpy()
    out = []
    for b, (tg, n) in enumerate(zip(target, n_speakers)):
        cm = cost[b].copy()
        if C > n:
            mv = np.abs(cm).sum()
            cm[-(C - n):] = mv
            cm[:, -(C - n):] = mv
        rows, cols = linear_sum_assignment(cm)
        assert np.all(rows == np.arange(C))
        out.append(tg[:, torch.as_tensor(cols)][: logits[b].shape[0], :n])
    return out
