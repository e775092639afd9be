"""Perf study: fused FFN time vs hidden width F.  t(F) = fixed + F/64 * per_chunk separates the per-tile
prologue/epilogue cost of the kernel from the marginal rate of its chunk loop."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
dev = torch.device("cuda")
M = 196608
x = torch.randn(M, 256, device=dev).half()
res = torch.randn(M, 256, device=dev); g = torch.ones(256, device=dev); be = torch.zeros(256, device=dev)
o32 = torch.empty_like(res); o16 = torch.empty_like(x)
for F in (64, 128, 512, 1024, 2048, 4096):
    w1 = (torch.randn(F, 256, device=dev) * 0.08).half(); b1 = torch.randn(F, device=dev) * 0.3
    w2 = (torch.randn(256, F, device=dev) * 0.04).half(); b2 = torch.randn(256, device=dev) * 0.3
    for _ in range(3): ops.ffn_fused(x, w1, b1, w2, b2, res, g, be, o32, o16)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.ffn_fused(x, w1, b1, w2, b2, res, g, be, o32, o16)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print("F=%5d  %.4f ms  %.1f TFLOP/s  (%.2f us per 64-chunk-round)" % (F, ms, 4.0 * M * F * 256 / ms / 1e9, ms * 1e3 / (F / 64) / 6), flush=True)
