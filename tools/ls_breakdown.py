"""Per-kernel time breakdown of the LS-EEND batch forward (HIP events around every C-ABI call)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fs_eend_amd import ops
from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization

NAMES = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith("_") and n not in ("frames_pad",)]
B, T, C = int(os.environ.get("B", 16)), int(os.environ.get("T", 2000)), int(os.environ.get("C", 10))
dev = torch.device("cuda")
cfg = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, max_seqlen=1000, recurrent_chunk_size=500,
           feed_forward_expansion_factor=4, dec_dim_feedforward=2048, conv_expansion_factor=2, conv_kernel_size=16,
           half_step_residual=True, conv_delay=9)
torch.manual_seed(0)
m = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **cfg).eval().to(dev)
g = torch.Generator().manual_seed(1)
src = [(torch.randn(T, 345, generator=g) * 2 - 3).to(dev) for _ in range(B)]
for _ in range(2):
    m.test(src, [T] * B, C)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m.test(src, [T] * B, C)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"LS-EEND B={B} T={T} C={C}: {dt*1e3:.2f} ms/step, {B*T/dt/1e6:.2f} M frames/s")
rec, orig = [], {}
def wrap(name, fn):
    def w(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = fn(*a, **k); e.record()
        shp = tuple(a[0].shape) if hasattr(a[0], "shape") else ((len(a[0]),) if hasattr(a[0], "__len__") else ())
        rec.append((name, shp, s, e)); return r
    return w
for n in NAMES:
    f = getattr(ops, n)
    if getattr(f, "__module__", "") == ops.__name__:
        orig[n] = f; setattr(ops, n, wrap(n, f))
for _ in range(3):
    m.test(src, [T] * B, C)
torch.cuda.synchronize()
agg = {}
for name, shp, s, e in rec:
    d = agg.setdefault((name, shp), [0.0, 0]); d[0] += s.elapsed_time(e); d[1] += 1
tot = sum(v[0] for v in agg.values())
for (name, shp), (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:24s} {str(shp):18s} x{n//3:3d}  avg {ms/n:.4f} ms  share {ms/tot:.3f}")
