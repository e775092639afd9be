"""Golden vectors for the post-processing row (runs ONLY in the build container, where /root/reference exists).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_post.py

train/utils/make_rttm.py imports h5py and train/utils/loss.py imports torchmetrics, neither of which this
image has, although the functions of interest do not use them.  So the generator parses the reference files,
takes the definitions of `make_rttm`, `calc_diarization_error` and `report_diarization_error` as they stand
and evaluates only those, with the names they really use (torch, F, medfilt, defaultdict) bound to the
installed packages.  Inputs are seeded; only inputs-seed and outputs are written (tests/golden/post_*.npz).
"""
import ast
import os
import sys
from collections import defaultdict

sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import medfilt

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def reference_functions(path, names):
    tree = ast.parse(open(path).read())
    ns = {"torch": torch, "F": F, "medfilt": medfilt, "defaultdict": defaultdict, "np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def post_inputs(seed, T, S, kind):
    g = torch.Generator().manual_seed(seed)
    if kind == "smooth":          # slowly varying tracks: realistic segment structure
        x = torch.randn(T + 40, S, generator=g)
        x = torch.nn.functional.avg_pool1d(x.t().unsqueeze(0), 41, 1).squeeze(0).t() * 6
        return torch.sigmoid(x + 0.3 * torch.randn(T, S, generator=g))
    if kind == "noise":
        return torch.rand(T, S, generator=g)
    if kind == "ones":
        return torch.ones(T, S)
    if kind == "zeros":
        return torch.zeros(T, S)
    raise ValueError(kind)


RTTM_CASES = [dict(name="post_rttm_smooth500", seed=1, T=500, S=5, kind="smooth", median=11, threshold=0.5),
              dict(name="post_rttm_noise137", seed=2, T=137, S=3, kind="noise", median=11, threshold=0.5),
              dict(name="post_rttm_short7", seed=3, T=7, S=2, kind="noise", median=11, threshold=0.4),
              dict(name="post_rttm_ones", seed=4, T=40, S=2, kind="ones", median=11, threshold=0.5),
              dict(name="post_rttm_zeros", seed=5, T=40, S=2, kind="zeros", median=11, threshold=0.5),
              dict(name="post_rttm_med1", seed=6, T=90, S=4, kind="noise", median=1, threshold=0.7),
              dict(name="post_rttm_med5", seed=7, T=333, S=9, kind="smooth", median=5, threshold=0.5),
              dict(name="post_rttm_T1", seed=8, T=1, S=3, kind="ones", median=11, threshold=0.5)]
DER_CASES = [dict(name="post_der_a", seed=11, T=500, C=4, delay=0), dict(name="post_der_b", seed=12, T=777, C=6, delay=0),
             dict(name="post_der_delay5", seed=13, T=200, C=3, delay=5), dict(name="post_der_T1", seed=14, T=1, C=2, delay=0)]


def der_inputs(seed, T, C):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(T, C, generator=g) * 2
    label = (torch.rand(T, C, generator=g) < 0.35).float()
    return pred, label


def main():
    (make_rttm,) = reference_functions(f"{REF}/FS-EEND/train/utils/make_rttm.py", ["make_rttm"])
    calc, report = reference_functions(f"{REF}/FS-EEND/train/utils/loss.py", ["calc_diarization_error", "report_diarization_error"])
    for c in RTTM_CASES:
        pred = post_inputs(c["seed"], c["T"], c["S"], c["kind"])
        rttm = make_rttm(rec="rec0", pred=pred, threshold=c["threshold"], median=c["median"])
        lines = [f"{k}\t{l}" for k in sorted(rttm, key=int) for l in rttm[k]]
        np.savez_compressed(os.path.join(OUT, c["name"] + ".npz"), meta=np.array(repr(c)), lines=np.array(lines, dtype=object).astype(str))
        print(c["name"], len(lines), "segments")
    from oracle import postproc_ref as P
    for c in DER_CASES:
        pred, label = der_inputs(c["seed"], c["T"], c["C"])
        res = calc(pred, label, c["delay"])
        vals = np.array([float(res[k]) for k in P.DER_KEYS], dtype=np.float64)
        rep = report([pred, pred[: max(1, c["T"] // 2)]], [label, label[: max(1, c["T"] // 2)]], c["delay"] if c["T"] > 12 else 0)
        repv = np.array([rep[k] for k in P.DER_KEYS], dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, c["name"] + ".npz"), meta=np.array(repr(c)), values=vals, report=repv)
        print(c["name"], dict(zip(P.DER_KEYS, vals)))


if __name__ == "__main__":
    main()
