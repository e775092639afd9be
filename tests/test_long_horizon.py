"""GPU: long-horizon parity against the REFERENCE's own outputs (oracle/gen_golden_long.py; BASELINE config 5).

  * one hour (T = 36 000, max_nspks = 10) of LS-EEND: `test_chunked` / `test` vs the reference's batch `model.test`
    (tests/golden/ls_hour_c10.npz) -- 72 retention chunks with the state carried;
  * the same hour frame by frame through `LsStreamSession` vs the reference's frame-by-frame driver
    (ls_hour_stream_c10.npz) -- the recurrent form `kv_t = kv_{t-1} sqrt((t-1)/t) + k (x) v / sqrt(t)` over 36 000 steps;
  * FS-EEND K/V-cache decode to t = 5000 through `FsStreamSession` vs the reference's streaming model and vs its batch
    test (fs_stream_T5000.npz).
Bar: per-frame activity logits within 1e-3 of the reference (north_star), measured on the stored rows."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as FX
from tests.helpers import build_fs_mirror, build_ls_mirror

pytestmark = pytest.mark.gpu


def _have(name):
    return os.path.exists(os.path.join(FX.GOLDEN_DIR, name + ".npz"))


def test_ls_one_hour_batch_vs_reference(hip_lib, dev):
    assert _have("ls_hour_c10")
    meta, arr = FX.load_case("ls_hour_c10")
    m = build_ls_mirror(meta).to(dev)
    T, C = meta["lengths"][0], meta["C"]
    src = [s.to(dev) for s in FX.make_src([T], meta["in_size"], meta["xseed"])]
    rows = torch.as_tensor(arr["rows"], device=dev)
    want = torch.as_tensor(arr["logits"], device=dev)
    lg, em, _ = m.test_chunked(src, [T], C, return_attractors=False)
    torch.cuda.synchronize()
    d = (lg[0][rows] - want).abs()
    print(f"LS one hour, test_chunked vs reference batch: max |d logit| {float(d.max()):.2e} (first 600 frames {float(d[:600].max()):.2e}, "
          f"last 600 {float(d[-600:].max()):.2e})")
    assert float(d.max()) < 1e-3
    e_want = torch.as_tensor(arr["emb"], device=dev)
    assert float((em[0][rows[::8]] - e_want).abs().max()) < 3e-3
    lg2, _, _ = m.test(src, [T], C)
    assert torch.equal(lg2[0], lg[0])


def test_ls_one_hour_streaming_vs_reference_streaming(hip_lib, dev):
    assert _have("ls_hour_stream_c10")
    from fs_eend_amd.ls_stream import LsStreamSession
    meta, arr = FX.load_case("ls_hour_stream_c10")
    m = build_ls_mirror(meta).to(dev)
    T, C = meta["lengths"][0], meta["C"]
    src = FX.make_src([T], meta["in_size"], meta["xseed"])[0].to(dev)
    sess = LsStreamSession(m, C)
    keep = {int(r): i for i, r in enumerate(arr["rows"])}
    got = torch.zeros(len(keep), C, device=dev)
    n = 0
    for t in range(T):
        y = sess.push(src[t])
        if y is not None:
            if n in keep:
                got[keep[n]] = y[0, 0]
            n += 1
    for y in sess.flush():
        if n in keep:
            got[keep[n]] = y[0, 0]
        n += 1
    torch.cuda.synchronize()
    assert n == meta["frames_out"] == T
    want = torch.as_tensor(arr["stream_logits"], device=dev)
    d = (got - want).abs()
    print(f"LS one hour, LsStreamSession vs reference streaming: max |d logit| {float(d.max()):.2e} (first 600 {float(d[:600].max()):.2e}, "
          f"last 600 {float(d[-600:].max()):.2e})")
    # Streaming is compared with streaming: the reference's OWN two fp32 forms of the recurrence (frame-by-frame vs chunked
    # batch, same weights, same input) differ by ~1e-2 in the logits after an hour (SURVEY 4 notes the looseness of its
    # streaming == batch self-checks already at T = 30).  The arbiter between the reference's fp32 streaming and this build
    # is the float64 evaluation of the same recurrence (ls_hour_stream64_c10, oracle pinned at T = 120):
    _, barr = FX.load_case("ls_hour_c10")
    assert np.array_equal(barr["rows"], arr["rows"])
    ref_gap = float(np.abs(barr["logits"] - arr["stream_logits"]).max())
    our_gap = float((got - torch.as_tensor(barr["logits"], device=dev)).abs().max())
    print(f"   streaming vs batch over the hour: reference {ref_gap:.2e}, this build {our_gap:.2e}")
    # the 1e-3 bar of the batch forms holds on every stored frame of the hour (measured 1.6e-4, which is the reference's own fp32
    # error: see below; 8e-4 with only the decoder step in f32, 1.2e-3 with only the retention projections, 2.0e-3 all-f16),
    # and the gap to the batch form stays the reference's own
    assert float(d.max()) < 1e-3 and our_gap < ref_gap + 1e-3
    if _have("ls_hour_stream64_c10"):
        _, a64 = FX.load_case("ls_hour_stream64_c10")
        assert np.array_equal(a64["rows"], arr["rows"])
        truth = torch.as_tensor(a64["stream_logits64"], device=dev, dtype=torch.float64)
        e_ref = float((want.double() - truth).abs().max())
        eo = (got.double() - truth).abs().flatten()
        e_our, p999 = float(eo.max()), float(torch.quantile(eo, 0.999))
        print(f"   against the float64 recurrence: reference fp32 streaming {e_ref:.2e}, this build max {e_our:.2e}, "
              f"mean {float(eo.mean()):.2e}, 99.9th percentile {p999:.2e}")
        # Measured (profiles/r03_ls_hour_stream_profile.txt): max 5.7e-5, mean 5.9e-6 in every 3000-frame window -- closer to
        # the float64 recurrence than the reference's own fp32 streaming (1.75e-4), and no growth with the stream position.
        assert e_our < 3e-4 and float(eo.mean()) < 3e-5 and p999 < 2e-4
        assert float(eo[-600 * C:].mean()) < 2.0 * float(eo[:600 * C].mean()) + 1e-5      # last minute vs first minute


@pytest.mark.parametrize("S", [20, 64])
def test_ls_one_hour_20_streams_vs_reference_streaming(hip_lib, dev, S):
    """(S = 64, round 5: 640 rows per decoder frame step on the f32 MFMA kernel of gemm_f32.hip -- VERDICT r04 item 7.)
    VERDICT r03 item 3a: a multi-stream session (20 streams x 10 slots = 200 rows per frame, i.e. far more than one 16-row
    group) meets the same 1e-3 bar over the hour as the single-stream session: stream 0 carries the golden input, the others
    perturbed copies; every frame step runs the all-f32 path in row groups of 16 (round 3 ran > 16 rows on the f16 MFMA step,
    which its own measurement put at 2.0e-3 after an hour)."""
    assert _have("ls_hour_stream_c10")
    from fs_eend_amd.ls_stream import LsStreamSession
    meta, arr = FX.load_case("ls_hour_stream_c10")
    m = build_ls_mirror(meta).to(dev)
    T, C = meta["lengths"][0], meta["C"]
    src = FX.make_src([T], meta["in_size"], meta["xseed"])[0].to(dev)
    g = torch.Generator(device="cpu").manual_seed(4321)
    noise = (0.3 * torch.randn(S - 1, 1, src.shape[1], generator=g)).to(dev)         # per-stream offset of the features
    sess = LsStreamSession(m, C, batch=S)
    keep = {int(r): i for i, r in enumerate(arr["rows"])}
    got = torch.zeros(len(keep), C, device=dev)
    last = torch.zeros(len(keep), C, device=dev)
    n = 0
    x = torch.empty(S, src.shape[1], device=dev)

    def take(y):
        nonlocal n
        if n in keep:
            got[keep[n]] = y[0, 0]
            last[keep[n]] = y[S - 1, 0]
        n += 1

    for t in range(T):
        x[0] = src[t]
        x[1:] = src[t] + noise[:, 0] * (1.0 + 0.1 * ((t * 7) % 13))
        y = sess.push(x)
        if y is not None:
            take(y)
    for y in sess.flush():
        take(y)
    torch.cuda.synchronize()
    assert n == meta["frames_out"] == T
    want = torch.as_tensor(arr["stream_logits"], device=dev)
    d = (got - want).abs()
    print(f"LS one hour, {S}-stream LsStreamSession, stream 0 vs reference streaming: max |d logit| {float(d.max()):.2e} "
          f"(first 600 {float(d[:600].max()):.2e}, last 600 {float(d[-600:].max()):.2e})")
    assert float(d.max()) < 1e-3
    assert torch.isfinite(last).all() and float((last - got).abs().max()) > 1e-3          # the other streams are other streams
    if _have("ls_hour_stream64_c10"):
        _, a64 = FX.load_case("ls_hour_stream64_c10")
        truth = torch.as_tensor(a64["stream_logits64"], device=dev, dtype=torch.float64)
        eo = (got.double() - truth).abs().flatten()
        print(f"   against the float64 recurrence: max {float(eo.max()):.2e}, mean {float(eo.mean()):.2e}")
        assert float(eo.max()) < 3e-4 and float(eo.mean()) < 3e-5
        assert float(eo[-600 * C:].mean()) < 2.0 * float(eo[:600 * C].mean()) + 1e-5      # no growth with the stream position


def test_fs_streaming_to_5000_frames_vs_reference(hip_lib, dev):
    assert _have("fs_stream_T5000")
    from fs_eend_amd.fs_stream import FsStreamSession, StreamingTransformerEDADiarization, copy_params_from_masked_to_streaming
    meta, arr = FX.load_case("fs_stream_T5000")
    m = build_fs_mirror(meta).to(dev)
    sm = StreamingTransformerEDADiarization(in_size=meta["in_size"], **meta["cfg"]).eval().to(dev)
    copy_params_from_masked_to_streaming(m, sm)
    T, C = meta["T"], meta["C"]
    src = FX.make_src([T], meta["in_size"], meta["xseed"])[0].to(dev)
    sess = FsStreamSession(sm, C, cap=1024)                   # three cache growths on the way to 5000
    keep = {int(r): i for i, r in enumerate(arr["rows"])}
    got = torch.zeros(len(keep), C, device=dev)
    n = 0
    for t in range(T):
        y = sess.push(src[t])
        if y is not None:
            if n in keep:
                got[keep[n]] = y[0, 0]
            n += 1
    for y in sess.flush():
        if n in keep:
            got[keep[n]] = y[0, 0]
        n += 1
    torch.cuda.synchronize()
    assert n == T
    d = (got - torch.as_tensor(arr["stream_logits"], device=dev)).abs()
    print(f"FS streaming to t={T}: vs reference streaming max |d logit| {float(d.max()):.2e} (last 600 frames {float(d[-600:].max()):.2e})")
    assert float(d.max()) < 1e-3
    # batch path on the same 5000 frames (tiled attention kernel, Tp > 512) vs the reference's batch output
    lg, _, _ = m.test([src], [T], C)
    rows = torch.as_tensor(arr["rows"], device=dev)
    db = (lg[0][rows] - torch.as_tensor(arr["batch_logits"], device=dev)).abs()
    print(f"FS batch T={T} vs reference batch: max |d logit| {float(db.max()):.2e}")
    assert float(db.max()) < 1e-3
