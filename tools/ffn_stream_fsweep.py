"""Perf study: F sweep of the packed-stream layer kernel at fixed M -- the slope is the cost of one stream item (32 hidden
units x 192 rows per workgroup), the intercept the per-tile fixed cost (out-projection items, LayerNorms, epilogue)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def rn(*s, scale=1.0, dt=torch.float32):
    return (torch.randn(*s, generator=g) * scale).to(dev).to(dt)
M = int(os.environ.get("AB_M", 196608))
a16 = rn(M, 256, dt=torch.float16); res16 = rn(M, 256, dt=torch.float16); res32 = rn(M, 256)
wo, bo = rn(256, 256, scale=0.06, dt=torch.float16), rn(256, scale=0.2)
one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
o16 = torch.empty(M, 256, dtype=torch.float16, device=dev); o32 = torch.empty(M, 256, device=dev)
def timeit(fn, n=10, rounds=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return statistics.median(ts)
res = {}
for Fh in (64, 256, 512, 1024, 2048):
    w1, b1 = rn(Fh, 256, scale=0.08, dt=torch.float16), rn(Fh, scale=0.3)
    w2, b2 = rn(256, Fh, scale=0.04, dt=torch.float16), rn(256, scale=0.3)
    wsp, wsn = ops.ffn_stream_pack(wo, w1, w2), ops.ffn_stream_pack(None, w1, w2)
    t_pre = timeit(lambda: ops.attnout_ffn_stream(a16, wsp, bo, None, res16, one, zero, 1e-5, b1, b2, one, zero, 1e-5, None, o16))
    t_pre32 = timeit(lambda: ops.attnout_ffn_stream(a16, wsp, bo, res32, None, one, zero, 1e-5, b1, b2, one, zero, 1e-5, o32, o16))
    t_plain = timeit(lambda: ops.ffn_stream(a16, wsn, b1, b2, res32, one, zero, o32, o16))
    res[Fh] = (t_pre, t_pre32, t_plain)
    print(f"F={Fh:5d}: attnout_ffn_stream res16 {t_pre:8.1f} us   f32res+out32 {t_pre32:8.1f} us   ffn_stream {t_plain:8.1f} us", flush=True)
tiles = (M + 191) // 192
rounds = (tiles + 255) // 256
for k, name in enumerate(("attnout res16", "attnout f32res", "ffn_stream")):
    slope = (res[2048][k] - res[256][k]) / ((2048 - 256) / 32) / rounds
    fixed = res[256][k] / rounds - slope * (256 / 32)
    print(f"{name:16s}: {slope:6.3f} us per item per tile (MFMA floor 1632 cycles = 0.78 us at 2.1 GHz), fixed {fixed:6.2f} us per tile ({rounds} rounds)")
