"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of one FS-EEND training step
(BASELINE config 4): label preparation, train-mode forward, BCE + embedding-consistency loss, gradients
(torch autograd over the explicit tensor algebra of fs_eend_ref.py), gradient clipping, Adam, Noam schedule and
the BatchNorm running-statistics update.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Pinned by tests/golden/fs_train_*.npz, produced by oracle/gen_golden_train.py from the reference's own
`training_step`, `standard_loss`, model, NoamScheduler and torch.optim.Adam (tests/test_oracle_train.py).
Citations are file:line under /root/reference/FS-EEND/.
"""
import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

from oracle import fs_eend_ref as R

Tensor = torch.Tensor

# state-dict tensors of the FS model that never receive a gradient: the decoder's unused input
# projection and the fusion layers' unused norm12 -- the reference's Adam skips them (grad is None)
def never_graded(name: str) -> bool:
    return name.startswith("dec.encoder.") or name.startswith("dec.encoder_norm.") or ".norm12." in name


def prepare_labels(labels: Sequence[Tensor], clip_lengths: Sequence[int]) -> List[Tensor]:
    """train/oln_tfm_enc_dec.py:53-75: pad the speaker columns, order the speakers of every utterance by first
    activity (never-active speakers last, stable), prepend the silence column 1 - max_c label, append the all-zero
    "none speaker" column, cut every utterance to (ilen, nspk_i + 2)."""
    n_spks = [l.shape[1] for l in labels]
    max_spk = max(n_spks)
    lab = [F.pad(l, (0, max_spk - l.shape[1])) for l in labels]
    lab = torch.nn.utils.rnn.pad_sequence(lab, padding_value=0.0, batch_first=True)        # (B, T, S)
    B, T, _ = lab.shape
    frame_index = torch.arange(1, T + 1, dtype=lab.dtype)[None, :, None]
    first = frame_index * lab
    first = first.masked_fill(first == 0, float("inf")).min(dim=1)[0]                       # (B, S) first active frame
    order = torch.argsort(first, dim=1)                                                     # :66
    lab = torch.gather(lab, 2, order[:, None, :].expand(B, T, max_spk))                     # :67
    silence = 1.0 - lab.max(dim=-1)[0]
    lab = torch.cat([silence[..., None], lab, torch.zeros(B, T, 1, dtype=lab.dtype)], dim=-1)   # :69-73
    return [lab[b, :l, :n + 2] for b, (l, n) in enumerate(zip(clip_lengths, n_spks))]


def standard_loss(ys: Sequence[Tensor], ts: Sequence[Tensor]) -> Tensor:
    """train/utils/loss.py:119-125, label_delay = 0: per-utterance mean BCE-with-logits (over frames x columns)
    times the utterance's frame count, summed, divided by the total frame count."""
    losses = [F.binary_cross_entropy_with_logits(y, t) * len(y) for y, t in zip(ys, ts)]
    return torch.stack(losses).sum() / sum(t.shape[0] for t in ts)


def noam_lr(opt_step: int, d_model: int, warmup: int, scale: float = 1.0, base_lr: float = 1.0) -> float:
    """Learning rate of optimiser step `opt_step` (1-based) under utlis/scheduler.py:3-28 as Lightning drives it
    (interval "step", oln_tfm_enc_dec.py:274): the scheduler is stepped once at construction and once after every
    optimiser step, so step k runs with last_epoch = k - 1 and get_lr clamps that to >= 1."""
    e = max(1, opt_step - 1)
    return base_lr * scale * d_model ** (-0.5) * min(e ** (-0.5), e * warmup ** (-1.5))


def clip_coef(total_norm: float, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ (Lightning gradient_clip_val, train_dia.py:153)."""
    return min(1.0, max_norm / (total_norm + 1e-6))


def train_loss(sd: Dict[str, Tensor], feats, labels_raw, cfg: dict, pit_labels=None, bn_stats=None, dtype=torch.float32,
               drop=None):
    """One reference training_step forward: returns (total, bce, emb_loss, logits, prepared labels)."""
    ilens = [int(f.shape[0]) for f in feats]
    labels = prepare_labels([l.to(dtype) for l in labels_raw], ilens)
    logits, emb_loss, _, _ = R.fs_forward(feats, labels, ilens, sd, n_heads=cfg["n_heads"],
                                          enc_n_layers=cfg["enc_n_layers"], dec_n_layers=cfg["dec_n_layers"],
                                          has_mask=cfg["has_mask"], mask_delay=cfg["mask_delay"], dtype=dtype,
                                          bn_batch_stats=bn_stats if bn_stats is not None else {}, drop=drop)
    use = labels
    if pit_labels is not None:
        use = pit_labels(logits, labels)
    bce = standard_loss(logits, use)
    return bce + emb_loss, bce, emb_loss, logits, labels


def pit_permute(logits, labels):
    """train/oln_tfm_enc_dec_spk_pit.py:78-87: the speaker columns (1..-2) of the prepared labels are re-ordered by
    batch_pit_n_speaker_loss on the same columns of the logits; silence / none-speaker columns keep their place."""
    from oracle import pit_ref as P
    n_spks = [l.shape[1] - 2 for l in labels]
    C = max(n_spks)
    ys = [F.pad(y.detach()[:, 1:-1], (0, C - n)) for y, n in zip(logits, n_spks)]
    ts = [F.pad(l[:, 1:-1], (0, C - n)) for l, n in zip(labels, n_spks)]
    _, perm = P.batch_pit_n_speaker_loss(ys, ts, n_spks)
    return [torch.cat([l[:, :1], p, l[:, -1:]], dim=-1) for l, p in zip(labels, perm)]


class TrainRef:
    """Reference-equivalent trainer state over a flat state_dict: Adam(betas (0.9, 0.98), eps 1e-9, lr 1) x Noam."""

    def __init__(self, sd: Dict[str, Tensor], cfg: dict, warmup: int, clip: float, pit: bool = False,
                 dtype=torch.float32):
        self.cfg, self.warmup, self.clip, self.pit, self.dtype = cfg, warmup, clip, pit, dtype
        self.sd = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        self.pnames = [k for k, v in self.sd.items() if v.is_floating_point() and v.dim() >= 1
                       and not k.endswith(("running_mean", "running_var", "pos_enc.pe"))]
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.pnames}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.pnames}
        self.opt_step = 0
        self.b1, self.b2, self.eps = 0.9, 0.98, 1e-9

    def step(self, feats, labels_raw):
        """-> dict(loss, bce, emb, lr, gradnorm, grads {name: unclipped grad or None})."""
        leaves = {k: self.sd[k].clone().requires_grad_(True) for k in self.pnames}
        sd = dict(self.sd)
        sd.update(leaves)
        bn = {}
        tot, bce, emb, _, _ = train_loss(sd, feats, labels_raw, self.cfg, pit_permute if self.pit else None, bn,
                                         self.dtype)
        grads = torch.autograd.grad(tot, [leaves[k] for k in self.pnames], allow_unused=True)
        grads = dict(zip(self.pnames, grads))
        gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None))
        coef = clip_coef(gn, self.clip)
        self.opt_step += 1
        lr = noam_lr(self.opt_step, self.cfg["n_units"], self.warmup)
        t = self.opt_step
        for k in self.pnames:                                   # torch.optim.Adam, no weight decay / amsgrad
            g = grads[k]
            if g is None:
                continue
            g = g * coef
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(1 - self.b2 ** t)).add_(self.eps)
            self.sd[k] = self.sd[k] - (lr / (1 - self.b1 ** t)) * (self.m[k] / denom)
        # BatchNorm1d running statistics, momentum 0.1, unbiased variance (torch.nn.BatchNorm1d in train mode)
        n = bn["count"]
        self.sd["enc.bn.running_mean"] = 0.9 * self.sd["enc.bn.running_mean"] + 0.1 * bn["mean"]
        self.sd["enc.bn.running_var"] = 0.9 * self.sd["enc.bn.running_var"] + 0.1 * bn["var_biased"] * (n / (n - 1))
        if "enc.bn.num_batches_tracked" in self.sd:
            self.sd["enc.bn.num_batches_tracked"] = self.sd["enc.bn.num_batches_tracked"] + 1
        return dict(loss=float(tot.detach()), bce=float(bce.detach()), emb=float(emb.detach()), lr=lr, gradnorm=gn, grads=grads)
