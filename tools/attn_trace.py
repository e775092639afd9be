"""Perf study: phase timeline of attn_causal_full_kernel from s_memtime stamps (library built with -DEEND_ATT_TRACE:
tools/ab_variants.sh build trace=-DEEND_ATT_TRACE; run with EEND_HIP_LIB=.../libeend_hip_trace.so)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
from fs_eend_amd.train import _call
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
T, Tp, H = 500, 512, 4
NAMES = {0: "start", 1: "Q+g0 issued", 2: "g0 ready", 3: "g1-3 issued", 4: "arrive g1", 5: "g1 ready", 6: "arrive g2", 7: "g2 ready",
         8: "arrive g3", 9: "g3 ready", 10: "late tiles done", 11: "late stored", 12: "early tiles done", 13: "end"}
for nseq in (64, 384):
    q = (torch.randn(nseq, H, Tp, 64, generator=g) * ops.QSCALE_LOG2).to(dev).to(torch.bfloat16)
    k = torch.randn(nseq, H, Tp, 64, generator=g).to(dev).to(torch.bfloat16)
    vt = torch.randn(nseq, H, 64, Tp, generator=g).to(dev).to(torch.bfloat16)
    o = torch.empty(nseq * Tp, 256, dtype=torch.float16, device=dev)
    tr = torch.zeros(nseq * H * 8 * 16, dtype=torch.int64, device=dev)
    for _ in range(3):
        _call("eend_attn_causal_lse_bf16", q, k, vt, o, tr, nseq, H, Tp, 256, 0, T, ops.LN2, None)
    torch.cuda.synchronize()
    t = tr.view(nseq * H, 8, 16).cpu().double()
    t0 = t[:, :, 0].min()
    tick_us = 1.0 / 100.0            # s_memtime: 100 MHz constant clock
    print(f"nseq={nseq}: blocks start {((t[:, :, 0].amin(1) - t0) * tick_us).quantile(torch.tensor([0., .5, 1.], dtype=torch.float64)).tolist()} us; "
          f"all end {((t[:, :, 13].amax() - t0) * tick_us):.2f} us")
    for blk in (0, nseq * H // 2, nseq * H - 1):
        b0 = t[blk, :, 0].min()
        print(f"  block {blk} (start +{(b0 - t0) * tick_us:.2f} us): per-stamp min/max over waves [us since block start]")
        for kk in range(14):
            v = (t[blk, :, kk] - b0) * tick_us
            print(f"    {kk:2d} {NAMES[kk]:18s} {v.min():7.2f} {v.max():7.2f}   waves: " + " ".join(f"{x:6.2f}" for x in v.tolist()))
