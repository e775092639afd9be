"""Same-box timing of conv1d_l2norm_stream (conv_stream.hip) against conv1d_l2norm at the FS model.test shape (64 x 512, k = 19)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("fs-eend_amd.ops")
g = torch.Generator().manual_seed(0)
dev = "cuda"
nseq, Tp, kt, pad = 64, 512, 19, 9
x = torch.randn(nseq * Tp, 256, generator=g).to(dev).half()
wr = (torch.randn(256, kt * 256, generator=g) / 40).to(dev).half()
bias = (torch.randn(256, generator=g) * 0.1).to(dev)
il = torch.full((nseq,), 500, dtype=torch.int32, device=dev)
ws = ops.conv_stream_pack(wr, kt)
o32 = torch.empty(nseq * Tp, 256, device=dev); o16 = torch.empty(nseq * Tp, 256, dtype=torch.float16, device=dev)
r32 = torch.empty_like(o32); r16 = torch.empty_like(o16)
fns = {"conv_stream": lambda: ops.conv1d_l2norm_stream(x, ws, bias, il, o32, o16, nseq, Tp, kt, pad),
       "implicit GEMM": lambda: ops.conv1d_l2norm(x, wr, bias, il, r32, r16, nseq, Tp, 256, kt, pad)}
res = {k: [] for k in fns}
for _ in range(5):
    for k, fn in fns.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 10 * 1e3)
print("max |diff|", (o32 - r32).abs().max().item())
fl = 2.0 * nseq * Tp * 256 * kt * 256
for k, v in res.items():
    print(f"{k}: min {min(v):.1f} us  median {sorted(v)[2]:.1f} us  = {fl / min(v) / 1e6:.0f} TFLOP/s")
