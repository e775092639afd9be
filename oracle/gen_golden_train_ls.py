"""Golden vectors for the LS-EEND TRAINING step (the LS half of BASELINE config 4) -- runs ONLY in the build
container, where /root/reference exists.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_train_ls.py

What runs is the reference itself:
  * the model: LS-EEND/nnet/model/onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_
    emb_loss_mask.py (imported), train() mode (the Conformer conv modules' BatchNorm1d on batch statistics), dropout 0;
  * the step: `SpeakerDiarization.training_step` / `.detect` of LS-EEND/train/oln_tfm_enc_dec_on_the_fly.py:48-92 (the
    module train_dia_simu.py:35 trains with) and, for the PIT case, of train/oln_tfm_enc_dec_spk_pit_on_the_fly.py:52-105.
    Those modules import pytorch_lightning / torchaudio (absent here), so only the method definitions are evaluated out
    of the reference file and bound to a plain object carrying the attributes they read;
  * the losses: `standard_loss`, `pit_loss_multispk`, `pad_labels`, `pad_preds` of train/utils/loss.py (same way);
  * the optimiser set-up of LS-EEND/train_dia_simu.py:97-117: torch.optim.Adam(lr, betas=(0.9, 0.98), eps=1e-9) +
    utlis/scheduler.py NoamScheduler (imported), stepped per optimiser step, and Lightning's gradient_clip_val (:171)
    = torch.nn.utils.clip_grad_norm_ before optimizer.step().

Only data is written (tests/golden/ls_train_*.npz): seeds, per-step losses / learning rates / gradient norms,
per-parameter gradient norms and a few gradient entries, parameter entries and the conv modules' BatchNorm running
statistics after each step.  No reference source or bytecode enters the repository.
"""
import ast
import copy
import functools
import os
import math
import sys
import types
from itertools import permutations
from typing import List

sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import fixtures as FX

REF = "/root/reference/LS-EEND"
NSLICE = 24            # gradient entries stored per parameter (first 12 + 12 strided)

LS_FULL = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.0, max_seqlen=1000,
               recurrent_chunk_size=500, feed_forward_expansion_factor=4, dec_dim_feedforward=2048,
               conv_expansion_factor=2, conv_kernel_size=16, half_step_residual=True, conv_delay=9)


def cfg(**kw):
    c = dict(LS_FULL)
    c.update(kw)
    return c


SMALL = cfg(enc_n_layers=2, dec_n_layers=1, dec_dim_feedforward=512, recurrent_chunk_size=100)
CASES = [
    # three chunks of 100 (carried retention state, T % chunk != 0), ragged lengths and speaker counts, 3 optimiser steps
    dict(name="ls_train_small", cfg=SMALL, lengths=[300, 250, 170], nspk=[2, 3, 1], seed=51, pseed=61, xseed=811,
         lseed=812, steps=3, warm=25, clip=5.0, pit=False),
    dict(name="ls_train_clip", cfg=cfg(enc_n_layers=1, dec_n_layers=2, dec_dim_feedforward=512, recurrent_chunk_size=64),
         lengths=[96, 130], nspk=[2, 2], seed=52, pseed=62, xseed=813, lseed=814, steps=2, warm=10, clip=0.02, pit=False),
    dict(name="ls_train_pit", cfg=SMALL, lengths=[150, 150, 149], nspk=[3, 2, 3], seed=53, pseed=63, xseed=815,
         lseed=816, steps=1, warm=25, clip=5.0, pit=True),
    # BASELINE config 4 shapes: the shipped yaml, 4-speaker mixtures, T = 1000 chunks (two retention chunks of 500)
    dict(name="ls_train_full", cfg=cfg(), lengths=[1000, 1000, 930], nspk=[4, 4, 3], seed=54, pseed=64, xseed=817,
         lseed=818, steps=1, warm=100, clip=5.0, pit=False),
    # ... at the batch size bench.py --mode train --flavour ls times (round 5): 64 utterances, a few of them shorter
    dict(name="ls_train_b64", cfg=cfg(), lengths=[1000] * 60 + [930, 777, 501, 1000], nspk=[4, 4, 3, 4] * 16, seed=55, pseed=65, xseed=819,
         lseed=820, steps=1, warm=100, clip=5.0, pit=False, ckpt64=True),
]


def reference_defs(path, names, ns, cls=None):
    """Evaluate the named function definitions (module level, or methods of class `cls`) out of a reference file."""
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def grad_slice_index(numel):
    a = np.arange(min(12, numel))
    b = (np.arange(12) * 7919 + 13) % numel
    return np.concatenate([a, b]).astype(np.int64)[:NSLICE]


def main():
    sys.path.insert(0, REF)
    from nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_pos_enc_l2norm_emb_loss_mask \
        import OnlineConformerRetentionDADiarization
    from utlis.scheduler import NoamScheduler
    from scipy.optimize import linear_sum_assignment
    from torch.nn.functional import logsigmoid

    ns = {"torch": torch, "nn": nn, "F": F, "np": np, "permutations": permutations, "List": List,
          "linear_sum_assignment": linear_sum_assignment, "logsigmoid": logsigmoid}
    standard_loss, pit_multispk, pad_labels, pad_preds = reference_defs(
        f"{REF}/train/utils/loss.py", ["standard_loss", "pit_loss_multispk", "pad_labels", "pad_preds"], ns)
    ns_plain = dict(ns)
    step_plain, detect = reference_defs(f"{REF}/train/oln_tfm_enc_dec_on_the_fly.py", ["training_step", "detect"], ns_plain,
                                        cls="SpeakerDiarization")
    ns_pit = dict(ns)
    ns_pit.update(pad_labels=pad_labels, pad_preds=pad_preds)
    (step_pit,) = reference_defs(f"{REF}/train/oln_tfm_enc_dec_spk_pit_on_the_fly.py", ["training_step"], ns_pit,
                                 cls="SpeakerDiarization")

    only = set(sys.argv[1:])                                  # optional: the case names to (re)generate
    for case in CASES:
        if only and case["name"] not in only:
            continue
        torch.manual_seed(case["seed"])
        model = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **case["cfg"])
        FX.perturb_(model, case["pseed"])
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1.0, betas=(0.9, 0.98), eps=1e-9)     # train_dia_simu.py:97-113, lr: 1
        sched = NoamScheduler(opt, case["cfg"]["n_units"], case["warm"], scale=1.0)            # train_dia_simu.py:117

        logged = {}
        me = types.SimpleNamespace(model=model, loss_func1=pit_multispk, loss_func2=standard_loss, label_delay=0, opt=opt,
                                   log=lambda k, v, **kw: logged.__setitem__(k, float(v)))
        me.detect = types.MethodType(detect, me)
        step_fn = types.MethodType(step_pit if case["pit"] else step_plain, me)

        feats = FX.make_src(case["lengths"], 345, case["xseed"])
        labels = FX.make_labels(case["lengths"], case["nspk"], case["lseed"])
        arrays = {}
        names = [n for n, _ in model.named_parameters()]
        bn_keys = [k for k in model.state_dict() if k.endswith(("running_mean", "running_var"))]
        for s in range(case["steps"]):
            batch = [tuple(f.clone() for f in feats), tuple(l.clone() for l in labels), tuple(range(len(feats)))]
            opt.zero_grad()
            loss = step_fn(batch, s)
            loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), case["clip"])      # Lightning gradient_clip_val
            arrays[f"s{s}_loss"] = np.array([float(loss), logged["train/pit_loss"], logged["train/emb_loss"]], dtype=np.float64)
            arrays[f"s{s}_lr"] = np.array([opt.param_groups[-1]["lr"]], dtype=np.float64)
            arrays[f"s{s}_gradnorm"] = np.array([float(gn)], dtype=np.float64)
            if s == 0:
                coef = min(1.0, case["clip"] / (float(gn) + 1e-6))
                norms, slices, nograd = [], [], []
                for n, p in model.named_parameters():
                    if p.grad is None:
                        nograd.append(n)
                        norms.append(-1.0)
                        slices.append(np.zeros(NSLICE, dtype=np.float32))
                        continue
                    g = p.grad.detach().flatten() / coef                                 # un-clipped gradient
                    norms.append(float(g.double().norm()))
                    idx = grad_slice_index(g.numel())
                    sl = np.zeros(NSLICE, dtype=np.float32)
                    sl[:len(idx)] = g[torch.as_tensor(idx)].numpy()
                    slices.append(sl)
                arrays["grad_norms"] = np.array(norms, dtype=np.float64)
                arrays["grad_slices"] = np.stack(slices)
                # Conditioning of every gradient tensor, measured with the reference itself: the same step in float64 (a double copy
                # of the model, default dtype float64 while it runs so that the tensors the reference creates on the fly follow).
                # gap_norm / gap_l2 = |fp32 - fp64| relative to max(||g64||, 1e-3 ||g_total||): what the reference's OWN working
                # precision does to this tensor.  tests/test_train_step_ls.py scales its per-tensor bar with it (COND_K).
                m64 = copy.deepcopy(model).double()
                for p_ in m64.parameters():
                    p_.grad = None
                if case.get("ckpt64"):
                    # the 64-utterance case does not fit this container's memory in float64: every encoder block / decoder layer of
                    # the double copy is recomputed in the backward (torch.utils.checkpoint around the reference module's own
                    # forward; same arithmetic, same gradients)
                    from torch.utils.checkpoint import checkpoint
                    for layer in list(m64.enc.encoder.layers) + list(m64.dec.layers):
                        layer.forward = functools.partial(checkpoint, layer.forward, use_reentrant=False)
                me64 = types.SimpleNamespace(model=m64, loss_func1=pit_multispk, loss_func2=standard_loss, label_delay=0, opt=opt,
                                             log=lambda k, v, **kw: None)
                me64.detect = types.MethodType(detect, me64)
                step64 = types.MethodType(step_pit if case["pit"] else step_plain, me64)
                torch.set_default_dtype(torch.float64)
                try:
                    b64 = [tuple(f.double() for f in feats), tuple(l.double() for l in labels), tuple(range(len(feats)))]
                    loss64 = step64(b64, 0)
                    loss64.backward()
                finally:
                    torch.set_default_dtype(torch.float32)
                g64 = {n: (None if p_.grad is None else p_.grad.detach().flatten()) for n, p_ in m64.named_parameters()}
                tot64 = math.sqrt(sum(float((g ** 2).sum()) for g in g64.values() if g is not None))
                n64, gap_n, gap_l2 = [], [], []
                for n, p in model.named_parameters():
                    if p.grad is None:
                        n64.append(-1.0); gap_n.append(0.0); gap_l2.append(0.0)
                        continue
                    a, b = p.grad.detach().flatten().double() / coef, g64[n]
                    den = max(float(b.norm()), 1e-3 * tot64)
                    n64.append(float(b.norm()))
                    gap_n.append(abs(float(a.norm()) - float(b.norm())) / den)
                    gap_l2.append(float((a - b).norm()) / den)
                arrays["grad_norms_f64"] = np.array(n64, dtype=np.float64)
                arrays["grad_gap_norm"] = np.array(gap_n, dtype=np.float64)
                arrays["grad_gap_l2"] = np.array(gap_l2, dtype=np.float64)
                arrays["s0_loss_f64"] = np.array([float(loss64)], dtype=np.float64)
                worst = sorted(zip(gap_n, names), reverse=True)[:4]
                print(f"  fp64 rerun: loss {float(loss64):.9f}; median gap {np.median(gap_n):.2e}; worst " +
                      ", ".join(f"{k} {g:.2e}" for g, k in worst))
            opt.step()
            sched.step()
            sd = model.state_dict()
            arrays[f"s{s}_bn"] = np.stack([sd[k].numpy().copy() for k in bn_keys])
            arrays[f"s{s}_param_slices"] = np.stack([
                (lambda f, idx: np.pad(f[torch.as_tensor(idx)].numpy(), (0, NSLICE - len(idx))))(p.detach().flatten(), grad_slice_index(p.numel()))
                for _, p in model.named_parameters()])
            print(f"{case['name']} step {s}: loss {float(loss):.6f} (pit {logged['train/pit_loss']:.6f} emb "
                  f"{logged['train/emb_loss']:.6f}) |g| {float(gn):.4e} lr {arrays[f's{s}_lr'][0]:.3e}")
        torch.manual_seed(case["seed"])
        m0 = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **case["cfg"])
        FX.perturb_(m0, case["pseed"])
        meta = dict(kind="ls_train", cfg=case["cfg"], lengths=case["lengths"], nspk=case["nspk"], seed=case["seed"],
                    pseed=case["pseed"], xseed=case["xseed"], lseed=case["lseed"], steps=case["steps"], warm=case["warm"],
                    clip=case["clip"], pit=case["pit"], in_size=345, param_names=names, nograd=nograd, bn_keys=bn_keys,
                    checksum_keys=sorted(FX.param_checksums(m0.state_dict())),
                    checksums=FX.param_checksums(m0.state_dict()), torch=torch.__version__)
        p = FX.save_case(case["name"], meta, arrays)
        print(f"  -> {os.path.relpath(p)} ({os.path.getsize(p) / 1024:.0f} KiB); never-graded tensors: {nograd}")


if __name__ == "__main__":
    main()
