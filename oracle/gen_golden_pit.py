"""Golden vectors for the PIT label-assignment row (runs ONLY in the build container, where /root/reference exists).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_pit.py

train/utils/loss.py imports torchmetrics (absent here); as in gen_golden_post.py the generator evaluates only the
definitions of `batch_pit_n_speaker_loss` (FS-EEND) and `pit_loss_multispk` (LS-EEND) out of the reference files,
with the names they use bound to the installed packages.  Only seeds and outputs are written (tests/golden/pit_*.npz).
"""
import os
import sys
from itertools import permutations

sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

import ast

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def reference_functions(path, names):
    from typing import List, Tuple
    tree = ast.parse(open(path).read())
    ns = {"torch": torch, "nn": nn, "F": F, "np": np, "permutations": permutations, "linear_sum_assignment": linear_sum_assignment,
          "logsigmoid": F.logsigmoid, "List": List, "Tuple": Tuple}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


CASES = [dict(name="pit_a", seed=21, lens=[120, 97, 64], nspk=[3, 2, 4], C=4),
         dict(name="pit_b", seed=22, lens=[500, 500], nspk=[4, 4], C=4),
         dict(name="pit_c6", seed=23, lens=[200, 180, 150, 33], nspk=[6, 1, 3, 5], C=6),
         dict(name="pit_single", seed=24, lens=[50], nspk=[2], C=2),
         dict(name="pit_silent", seed=25, lens=[80, 60], nspk=[3, 3], C=3, silent=True)]


def pit_inputs(c):
    """Seeded logits / labels: labels are a hidden permutation of thresholded smooth tracks, logits a noisy copy, so
    the assignment is non-trivial; columns >= nspk[b] of the labels are zero (pad_labels)."""
    g = torch.Generator().manual_seed(c["seed"])
    ys, ts = [], []
    for T, n in zip(c["lens"], c["nspk"]):
        act = (torch.rand(T, c["C"], generator=g) < 0.4).float()
        act[:, n:] = 0
        if c.get("silent"):
            act[:, 1:] = 0                      # identical (all-zero) label columns: exercises ties
        perm = torch.randperm(n, generator=g)
        lab = act.clone()
        lab[:, :n] = act[:, perm]
        y = (act * 2 - 1) * 2.0 + torch.randn(T, c["C"], generator=g) * 1.5
        ys.append(y)
        ts.append(lab)
    return ys, ts


def main():
    (bpit,) = reference_functions(f"{REF}/FS-EEND/train/utils/loss.py", ["batch_pit_n_speaker_loss"])
    (multi,) = reference_functions(f"{REF}/LS-EEND/train/utils/loss.py", ["pit_loss_multispk"])
    for c in CASES:
        ys, ts = pit_inputs(c)
        loss, labels = bpit([y.clone() for y in ys], [t.clone() for t in ts], list(c["nspk"]))
        d = {"meta": np.array(repr(c)), "loss": np.array([float(loss)], dtype=np.float64)}
        for i, l in enumerate(labels):
            d[f"bpit_label{i}"] = l.numpy().astype(np.int8)
        tgt = nn.utils.rnn.pad_sequence([t.clone() for t in ts], padding_value=-1, batch_first=True)   # what the LS trainer passes
        perm = multi([y.clone() for y in ys], tgt, np.array(c["nspk"]))
        for i, l in enumerate(perm):
            d[f"multi_label{i}"] = l.numpy().astype(np.int8)
        np.savez_compressed(os.path.join(OUT, c["name"] + ".npz"), **d)
        print(c["name"], float(loss), [tuple(l.shape) for l in labels], [tuple(l.shape) for l in perm])


if __name__ == "__main__":
    main()
