"""GPU: HIP post-processing (threshold + median + segments -> RTTM lines, DER counters) against the golden
outputs of the reference's own functions and against the oracle at sizes beyond the fixtures.  Bit exact."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import gen_golden_post as G
from oracle import postproc_ref as P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return ast.literal_eval(str(z["meta"])), z


def lines(rttm):
    return [f"{k}\t{l}" for k in sorted(rttm, key=int) for l in rttm[k]]


@pytest.mark.parametrize("case", [c["name"] for c in G.RTTM_CASES])
def test_make_rttm_vs_reference_golden(hip_lib, dev, case):
    from fs_eend_amd import postproc
    meta, z = load(case)
    pred = G.post_inputs(meta["seed"], meta["T"], meta["S"], meta["kind"]).to(dev)
    got = lines(postproc.make_rttm("rec0", pred, threshold=meta["threshold"], median=meta["median"]))
    assert got == [str(l) for l in z["lines"]]


@pytest.mark.parametrize("case", [c["name"] for c in G.DER_CASES])
def test_der_vs_reference_golden(hip_lib, dev, case):
    from fs_eend_amd import postproc
    meta, z = load(case)
    pred, label = G.der_inputs(meta["seed"], meta["T"], meta["C"])
    res = postproc.calc_diarization_error(pred.to(dev), label.to(dev), meta["delay"])
    assert [float(res[k]) for k in P.DER_KEYS] == list(z["values"])
    h = max(1, meta["T"] // 2)
    rep = postproc.report_diarization_error([pred.to(dev), pred[:h].to(dev)], [label.to(dev), label[:h].to(dev)],
                                            meta["delay"] if meta["T"] > 12 else 0)
    assert np.array_equal(np.array([rep[k] for k in P.DER_KEYS]), z["report"])


@pytest.mark.parametrize("T,S,median,thr", [(36000, 8, 11, 0.5), (5000, 10, 11, 0.5), (12345, 3, 7, 0.3), (64, 1, 11, 0.5), (65, 12, 3, 0.9)])
def test_activity_and_segments_vs_oracle(hip_lib, dev, T, S, median, thr):
    """1-hour stream sized inputs; activity map and change points must equal the oracle's exactly."""
    from fs_eend_amd import postproc
    pred = G.post_inputs(T + S, T, S, "smooth" if T > 100 else "noise")
    act = postproc.activity(pred.to(dev), thr, median)
    want = P.activity(pred, thr, median)
    assert np.array_equal(act.cpu().numpy().astype(np.int64), want)
    assert postproc.segments(act) == P.segments(want)
    assert lines(postproc.make_rttm("r", pred.to(dev), threshold=thr, median=median)) == lines(P.make_rttm("r", pred, threshold=thr, median=median))


def test_threshold_edges_and_strided_input(hip_lib, dev):
    from fs_eend_amd import postproc
    # exactly-at-threshold values are inactive (strict >); logits of exactly 0 are inactive (sigmoid = 0.5)
    pred = torch.tensor([[0.5, 0.50000006], [0.5, 0.7]] * 10, dtype=torch.float32)
    assert np.array_equal(postproc.activity(pred.to(dev), 0.5, 1).cpu().numpy(), P.activity(pred, 0.5, 1))
    logit = torch.tensor([[0.0, 1e-3, -1e-3, 5.0]] * 7)
    label = torch.tensor([[0.0, 1.0, 1.0, 0.0]] * 7)
    assert postproc.calc_diarization_error(logit.to(dev), label.to(dev)) == P.calc_diarization_error(logit, label)
    # a column slice of the model output (preds[:, 1:], dia_pred.py:56) is a strided view
    big = torch.rand(300, 6, generator=torch.Generator().manual_seed(5))
    view = big.to(dev)[:, 1:]
    assert lines(postproc.make_rttm("x", view)) == lines(P.make_rttm("x", big[:, 1:]))


def test_model_to_rttm_end_to_end(hip_lib, dev):
    """model.test -> sigmoid -> make_rttm on the device, as dia_pred.predict does (FS-EEND/dia_pred.py:55-62)."""
    from oracle import fixtures as FX
    from fs_eend_amd import postproc
    from tests.helpers import build_fs_mirror
    meta, arr = FX.load_case("fs_full_T500_c6")
    m = build_fs_mirror(meta).to(dev)
    src = [s.to(dev) for s in FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])]
    preds, _, _ = m.test(src, meta["lengths"], max_nspks=meta["C"])
    prob = torch.sigmoid(preds[0][:, 1:])
    got = postproc.make_rttm("utt", prob)
    want = P.make_rttm("utt", prob.cpu())
    assert lines(got) == lines(want)
