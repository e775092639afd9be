"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of one LS-EEND training step (the LS half of
BASELINE config 4): label preparation, train-mode forward (the Conformer conv modules' BatchNorm1d on batch
statistics), BCE + length-masked embedding-consistency loss, gradients (torch autograd over the explicit tensor
algebra of ls_eend_ref.py -- whose retention keeps the reference's `.detach()`ed scales), gradient clipping, Adam,
Noam schedule and the BatchNorm running-statistics update.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Pinned by tests/golden/ls_train_*.npz, produced by oracle/gen_golden_train_ls.py from the reference's own
`training_step` (LS-EEND/train/oln_tfm_enc_dec_on_the_fly.py:52-92, the module train_dia_simu.py:35 trains with, and
train/oln_tfm_enc_dec_spk_pit_on_the_fly.py:52-105 for the PIT variant), `standard_loss`, model, NoamScheduler and
torch.optim.Adam (tests/test_oracle_train_ls.py).  Citations are file:line under /root/reference/LS-EEND/.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

from oracle import ls_eend_ref as R
from oracle.train_ref import clip_coef, noam_lr, prepare_labels, standard_loss

Tensor = torch.Tensor


def never_graded(name: str) -> bool:
    """Tensors of the LS model that are not on the forward path: the decoder's unused input projection and the
    fusion layers' unused norm12 (nnet/modules/merge_retnet_layer.py:93) -- the reference's Adam skips them."""
    return name.startswith("dec.encoder.") or name.startswith("dec.encoder_norm.") or ".norm12." in name


def pit_permute(logits, labels):
    """train/oln_tfm_enc_dec_spk_pit_on_the_fly.py:82-96: the speaker columns (1..n_spk) of the prepared labels are
    re-ordered by `pit_loss_multispk` (Hungarian assignment on the BCE cost matrix, train/utils/loss.py:350-379) of the
    same columns of the logits; silence / none-speaker columns keep their place."""
    from oracle import pit_ref as P
    n_spks = [l.shape[1] - 2 for l in labels]
    C = max(n_spks)
    ys = [F.pad(y.detach()[:, 1:n + 1], (0, C - n)) for y, n in zip(logits, n_spks)]
    ts = [F.pad(l[:, 1:-1], (0, C - n)) for l, n in zip(labels, n_spks)]
    perm = P.pit_loss_multispk(ys, ts, n_spks)
    return [torch.cat([l[:, :1], p, l[:, -1:]], dim=-1) for l, p in zip(labels, perm)]


def train_loss(sd: Dict[str, Tensor], feats, labels_raw, cfg: dict, pit: bool = False, bn_train=None, dtype=torch.float32,
               drop=None):
    """One reference training_step forward: (total, bce, emb_loss, logits, prepared labels)."""
    ilens = [int(f.shape[0]) for f in feats]
    labels = prepare_labels([l.to(dtype) for l in labels_raw], ilens)               # oln_tfm_enc_dec_on_the_fly.py:53-75
    logits, emb_loss, _, _ = R.ls_forward(feats, labels, ilens, sd, n_heads=cfg["n_heads"], enc_n_layers=cfg["enc_n_layers"],
                                          dec_n_layers=cfg["dec_n_layers"], chunk=cfg["recurrent_chunk_size"],
                                          conv_delay=cfg.get("conv_delay", 9), dtype=dtype,
                                          bn_train=bn_train if bn_train is not None else {}, drop=drop)
    use = pit_permute(logits, labels) if pit else labels
    bce = standard_loss(logits, use)                                                # loss.py:136-142, label_delay 0
    return bce + emb_loss, bce, emb_loss, logits, use


class LsTrainRef:
    """Reference-equivalent trainer state over a flat state_dict: Adam(betas (0.9, 0.98), eps 1e-9, lr 1) x Noam
    (train_dia_simu.py:97-117), gradient_clip_val (:171), BatchNorm1d running statistics (momentum 0.1)."""

    def __init__(self, sd: Dict[str, Tensor], cfg: dict, warmup: int, clip: float, pit: bool = False, dtype=torch.float32):
        self.cfg, self.warmup, self.clip, self.pit, self.dtype = cfg, warmup, clip, pit, dtype
        self.sd = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        self.pnames = [k for k, v in self.sd.items() if v.is_floating_point() and v.dim() >= 1
                       and not k.endswith(("running_mean", "running_var", "pos_enc.pe", ".angle", ".decay"))]
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.pnames}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.pnames}
        self.opt_step = 0
        self.b1, self.b2, self.eps = 0.9, 0.98, 1e-9

    def grads(self, feats, labels_raw):
        leaves = {k: self.sd[k].clone().requires_grad_(True) for k in self.pnames}
        sd = dict(self.sd)
        sd.update(leaves)
        bn = {}
        tot, bce, emb, logits, labels = train_loss(sd, feats, labels_raw, self.cfg, self.pit, bn, self.dtype)
        g = torch.autograd.grad(tot, [leaves[k] for k in self.pnames], allow_unused=True)
        return tot, bce, emb, dict(zip(self.pnames, g)), bn, logits, labels

    def step(self, feats, labels_raw):
        """-> dict(loss, bce, emb, lr, gradnorm, grads {name: unclipped grad or None})."""
        tot, bce, emb, grads, bn, _, _ = self.grads(feats, labels_raw)
        gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None))
        coef = clip_coef(gn, self.clip)
        self.opt_step += 1
        lr = noam_lr(self.opt_step, self.cfg["n_units"], self.warmup)
        t = self.opt_step
        for k in self.pnames:
            g = grads[k]
            if g is None:
                continue
            g = g * coef
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(1 - self.b2 ** t)).add_(self.eps)
            self.sd[k] = self.sd[k] - (lr / (1 - self.b1 ** t)) * (self.m[k] / denom)
        for pfx, st in bn.items():                      # conv-module BatchNorm1d (convolution.py:143), momentum 0.1
            b = pfx + "sequential.5."
            n = st["count"]
            self.sd[b + "running_mean"] = 0.9 * self.sd[b + "running_mean"] + 0.1 * st["mean"]
            self.sd[b + "running_var"] = 0.9 * self.sd[b + "running_var"] + 0.1 * st["var_biased"] * (n / (n - 1))
            if b + "num_batches_tracked" in self.sd:
                self.sd[b + "num_batches_tracked"] = self.sd[b + "num_batches_tracked"] + 1
        return dict(loss=float(tot.detach()), bce=float(bce.detach()), emb=float(emb.detach()), lr=lr, gradnorm=gn, grads=grads)
