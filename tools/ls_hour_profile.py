"""Per-window error profile of LsStreamSession against the fp64 recurrence over the one-hour fixture."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from oracle import fixtures as FX
from tests.helpers import build_ls_mirror
from fs_eend_amd.ls_stream import LsStreamSession
dev = torch.device("cuda:0")
meta, arr = FX.load_case("ls_hour_stream_c10")
_, a64 = FX.load_case("ls_hour_stream64_c10")
m = build_ls_mirror(meta).to(dev)
T, C = meta["lengths"][0], meta["C"]
src = FX.make_src([T], meta["in_size"], meta["xseed"])[0].to(dev)
sess = LsStreamSession(m, C)
rows = arr["rows"]
keep = {int(r): i for i, r in enumerate(rows)}
got = torch.zeros(len(keep), C, device=dev)
n = 0
for t in range(T):
    y = sess.push(src[t])
    if y is not None:
        if n in keep: got[keep[n]] = y[0, 0]
        n += 1
for y in sess.flush():
    if n in keep: got[keep[n]] = y[0, 0]
    n += 1
torch.cuda.synchronize()
truth = a64["stream_logits64"]
e = np.abs(got.double().cpu().numpy() - truth)
er = np.abs(arr["stream_logits"].astype(np.float64) - truth)
print("rows stored", len(rows), "first", rows[:5], "last", rows[-5:])
for lo in range(0, T, 3000):
    sel = (rows >= lo) & (rows < lo + 3000)
    if sel.any():
        i = np.unravel_index(np.argmax(e[sel]), e[sel].shape)
        print(f"[{lo:5d},{lo+3000:5d}) n={sel.sum():4d} ours max {e[sel].max():.2e} mean {e[sel].mean():.2e} | ref32 max {er[sel].max():.2e} mean {er[sel].mean():.2e} | worst row {rows[sel][i[0]]} slot {i[1]} |logit| {abs(truth[sel][i]):.3f}")
dref = np.abs(got.cpu().numpy().astype(np.float64) - arr["stream_logits"].astype(np.float64))
print(f"vs the reference's fp32 streaming: max {dref.max():.2e}, first 600 frames {dref[:600].max():.2e}; vs float64: max {e.max():.2e}, mean {e.mean():.2e}, "
      f"99.9th percentile {np.quantile(e, 0.999):.2e}, entries > 1e-3: {(e > 1e-3).sum()} of {e.size}")
big = np.argwhere(e > 8e-4)
print("entries > 8e-4:", [(int(rows[i]), int(c), float(e[i, c]), float(truth[i, c])) for i, c in big][:20])
