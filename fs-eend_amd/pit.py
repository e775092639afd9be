"""Permutation-invariant label assignment on the device (csrc/pit.hip; no CPU fallback):

    batch_pit_n_speaker_loss   FS-EEND/train/utils/loss.py:257-327 (LS-EEND train/utils/loss.py:276-348)
    pit_loss_multispk          LS-EEND/train/utils/loss.py:350-379
    pad_labels / pad_preds     LS-EEND/train/utils/loss.py:47-71

Same names, arguments and return structure as the reference.  Values only: PIT chooses labels, the loss that is
back-propagated is standard_loss on the permuted labels (train/oln_tfm_enc_dec_spk_pit.py:78-87).  Neither function leaves the GPU for the
assignment (the reference's pit_loss_multispk copies every cost matrix to the host for scipy).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib as _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _assign(ys, ts, n_speakers_list):
    L = _lib.load()
    if not ys:
        raise _lib.EendHipError("pit: empty batch")
    from .postproc import _to_device
    ys, ts = [_to_device(t, "ys") for t in ys], [_to_device(t, "ts") for t in ts]      # the callers may hold CPU tensors
    y = nn.utils.rnn.pad_sequence([t.to(torch.float32) for t in ys], padding_value=-1, batch_first=True).contiguous()
    lab = nn.utils.rnn.pad_sequence([t.to(torch.float32) for t in ts], padding_value=-1, batch_first=True).contiguous()
    B, T, C = y.shape
    if lab.shape != y.shape:
        raise _lib.EendHipError("pit: logits and labels must have the same (padded) shape")
    dev = y.device
    cost = torch.empty(B, C, C, dtype=torch.float64, device=dev)
    perm = torch.empty(B, C, dtype=torch.int32, device=dev)
    loss = torch.empty(B, dtype=torch.float64, device=dev)
    nspk = torch.tensor([int(n) for n in n_speakers_list], dtype=torch.int32, device=dev)
    _lib.check(L.eend_pit_cost_f64(y.data_ptr(), lab.data_ptr(), B, T, C, cost.data_ptr(), _stream()), "eend_pit_cost_f64")
    _lib.check(L.eend_pit_assign_i32(cost.data_ptr(), nspk.data_ptr(), B, C, perm.data_ptr(), loss.data_ptr(), _stream()),
               "eend_pit_assign_i32")
    return perm.long(), loss


def batch_pit_n_speaker_loss(ys, ts, n_speakers_list):
    """-> (mean BCE of the best permutation per utterance over the batch, permuted labels)."""
    perm, loss = _assign(ys, ts, n_speakers_list)
    n_frames = sum(t.shape[0] for t in ts)
    min_loss = (loss.sum() / n_frames).to(torch.float32)
    labels_perm = [t[:, perm[b].to(t.device)][:, :n] for b, (t, n) in enumerate(zip(ts, n_speakers_list))]
    return min_loss, labels_perm


def pit_loss_multispk(logits, target, n_speakers, detach_attractor_loss=False):
    """target: (B, T, C) padded tensor (as the reference's trainer passes it) or a list of (T_i, C) tensors."""
    if isinstance(target, torch.Tensor):
        clip = [l.shape[0] for l in logits]
        tl = [target[i, :clip[i]] for i in range(target.shape[0])]
    else:
        tl = list(target)
    if detach_attractor_loss:
        tl = [t.clone() for t in tl]
        for t, n in zip(tl, n_speakers):
            t[:, int(n):] = -1
    perm, _ = _assign(logits, tl, [int(n) for n in n_speakers])
    return [t[:, perm[b].to(t.device)][: logits[b].shape[0], : int(n)] for b, (t, n) in enumerate(zip(tl, n_speakers))]


def _widen(t, width):
    extra = width - t.shape[1]
    if extra < 0:
        raise ValueError(f"{t.shape[1]} speaker columns do not fit into {width}")
    return t if extra == 0 else torch.cat([t, t.new_zeros(t.shape[0], extra)], dim=1)


def pad_labels(ts, out_size):
    """Zero-extend every (T_i, n_i) label tensor to `out_size` speaker columns.  Like the reference helper
    (train/utils/loss.py:47-57) the list is updated in place and returned; too many columns is a ValueError."""
    ts[:] = [_widen(t, out_size) for t in ts]
    return ts


def pad_preds(ys, out_size):
    """Zero-extend every (T_i, n_i) logit tensor to `out_size` columns into a new list (train/utils/loss.py:59-72)."""
    return [_widen(y, out_size) for y in ys]
