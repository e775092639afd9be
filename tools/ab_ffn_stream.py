"""Same-box A/B of the layer-tail kernels: ffn.hip (un-packed, 8 waves x 128-row tiles) vs ffn_stream.hip (packed weight
stream, 4 waves x 48 rows).  Interleaved rounds in one process, HIP events, random operands; prints median / min per variant
and the MFMA fraction on algorithmic rows (T = 500 of Tp = 512)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def rn(*s, scale=1.0, dt=torch.float32):
    return (torch.randn(*s, generator=g) * scale).to(dev).to(dt)
B, C, Tp, T = 64, 6, 512, 500
M = int(os.environ.get("AB_M", B * C * Tp))
Fh = int(os.environ.get("AB_F", 2048))
a16 = rn(M, 256, dt=torch.float16)
wo, bo = rn(256, 256, scale=0.06, dt=torch.float16), rn(256, scale=0.2)
w1, b1 = rn(Fh, 256, scale=0.08, dt=torch.float16), rn(Fh, scale=0.3)
w2, b2 = rn(256, Fh, scale=0.04, dt=torch.float16), rn(256, scale=0.3)
res16 = rn(M, 256, dt=torch.float16); res32 = rn(M, 256)
one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
o32 = torch.empty(M, 256, device=dev); o16 = torch.empty(M, 256, dtype=torch.float16, device=dev)
ws_pre = ops.ffn_stream_pack(wo, w1, w2)
ws_plain = ops.ffn_stream_pack(None, w1, w2)
variants = {
    "attnout_ffn_fused_res16 (ffn.hip)": lambda: ops.attnout_ffn_fused_res16(a16, wo, bo, res16, one, zero, 1e-5, w1, b1, w2, b2, one, zero, 1e-5, None, o16),
    "attnout_ffn_stream res16": lambda: ops.attnout_ffn_stream(a16, ws_pre, bo, None, res16, one, zero, 1e-5, b1, b2, one, zero, 1e-5, None, o16),
    "attnout_ffn_fused f32res (ffn.hip)": lambda: ops.attnout_ffn_fused(a16, wo, bo, res32, one, zero, 1e-5, w1, b1, w2, b2, one, zero, 1e-5, o32, o16),
    "attnout_ffn_stream f32res": lambda: ops.attnout_ffn_stream(a16, ws_pre, bo, res32, None, one, zero, 1e-5, b1, b2, one, zero, 1e-5, o32, o16),
    "ffn_fused (ffn.hip)": lambda: ops.ffn_fused(a16, w1, b1, w2, b2, res32, one, zero, o32, o16),
    "ffn_stream": lambda: ops.ffn_stream(a16, ws_plain, b1, b2, res32, one, zero, o32, o16),
}
times = {k: [] for k in variants}
for k, fn in variants.items():
    for _ in range(3): fn()
torch.cuda.synchronize()
for rnd in range(int(os.environ.get("AB_ROUNDS", 7))):
    for k, fn in variants.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 10)
rows = M * T // Tp
for k, v in times.items():
    pre = "attnout" in k
    flop = rows * (4 * Fh * 256 + (2 * 256 * 256 if pre else 0))
    med = statistics.median(v)
    print(f"{k:38s} median {med*1e3:8.1f} us  min {min(v)*1e3:8.1f} us   {flop/med/1e9:7.1f} TFLOP/s = {flop/med/1e9/2500:.3f} of peak", flush=True)
