"""A/B study: causal attention kernel at the encoder (64 seq) and decoder (384 seq) launch sizes, T = 500 (Tp = 512)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def timeit(name, fn, flop, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    print(f"{name:26s} {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s  {flop / us / 1e6 / 2500:.3f} of MFMA peak", flush=True)
T, Tp, H = 500, 512, 4
for nseq in (64, 384):
    q = (torch.randn(nseq, H, Tp, 64, generator=g) * ops.QSCALE_LOG2).to(dev).to(torch.bfloat16)
    k = torch.randn(nseq, H, Tp, 64, generator=g).to(dev).to(torch.bfloat16)
    vt = torch.randn(nseq, H, 64, Tp, generator=g).to(dev).to(torch.bfloat16)
    o = torch.empty(nseq * Tp, 256, dtype=torch.float16, device=dev)
    flop = nseq * 2.0 * 256 * T * (T + 1)
    timeit(f"attn_causal nseq={nseq}", lambda: ops.attn_causal(q, k, vt, o, nseq, H, Tp, 0, T, scale=ops.LN2), flop)
    ref = o.clone()
    ops.attn_causal(q, k, vt, o, nseq, H, Tp, 0, T, scale=ops.LN2)
    chk = float(o.float().abs().sum())
    print(f"   checksum {chk:.6e}")
# fused in-projection + attention vs the two kernels it replaces
for nseq in (64, 384):
    x = torch.randn(nseq * Tp, 256, generator=g).to(dev).to(torch.float16)
    w = (torch.randn(768, 256, generator=g) / 16).to(dev).to(torch.float16)
    b = (torch.randn(768, generator=g) * 0.1).to(dev)
    q, k, vt = (torch.empty(nseq * Tp * 256, dtype=torch.bfloat16, device=dev) for _ in range(3))
    o = torch.empty(nseq * Tp, 256, dtype=torch.float16, device=dev)
    flop = nseq * 2.0 * 256 * T * (T + 1) + 2.0 * nseq * Tp * 768 * 256
    def two():
        ops.inproj_heads(x, w, b, q, k, vt, nseq, Tp, H)
        ops.attn_causal(q, k, vt, o, nseq, H, Tp, 0, T, scale=ops.LN2)
    timeit(f"inproj + attn nseq={nseq}", two, flop)
    timeit(f"fused        nseq={nseq}", lambda: ops.inproj_attn_causal(x, w, b, q, o, nseq, H, Tp, 0, T), flop)
