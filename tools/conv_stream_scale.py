import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("fs-eend_amd.ops")
g = torch.Generator().manual_seed(0)
dev = "cuda"
Tp, kt, pad = 512, 19, 9
wr = (torch.randn(256, kt * 256, generator=g) / 40).to(dev).half()
bias = (torch.randn(256, generator=g) * 0.1).to(dev)
ws = ops.conv_stream_pack(wr, kt)
for nseq in (16, 64, 128, 256):
    x = torch.randn(nseq * Tp, 256, generator=g).to(dev).half()
    il = torch.full((nseq,), 500, dtype=torch.int32, device=dev)
    o32 = torch.empty(nseq * Tp, 256, device=dev); o16 = torch.empty(nseq * Tp, 256, dtype=torch.float16, device=dev)
    fn = lambda: ops.conv1d_l2norm_stream(x, ws, bias, il, o32, o16, nseq, Tp, kt, pad)
    ts = []
    for _ in range(5):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(nseq, "sequences:", round(min(ts), 1), "us")
