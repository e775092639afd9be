"""Perf study: duration of ONE launch of the layer-tail kernels right behind different predecessor kernels (cache / clock state)."""
import os, sys, statistics, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def rn(*s, scale=1.0, dt=torch.float32):
    return (torch.randn(*s, generator=g) * scale).to(dev).to(dt)
M, Fh = 196608, 2048
a16 = rn(M, 256, dt=torch.float16); res16 = rn(M, 256, dt=torch.float16)
wo, bo = rn(256, 256, scale=0.06, dt=torch.float16), rn(256, scale=0.2)
w1, b1 = rn(Fh, 256, scale=0.08, dt=torch.float16), rn(Fh, scale=0.3)
w2, b2 = rn(256, Fh, scale=0.04, dt=torch.float16), rn(256, scale=0.3)
one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
o16 = torch.empty(M, 256, dtype=torch.float16, device=dev); scratch = torch.empty(M, 256, dtype=torch.float16, device=dev)
big = torch.empty(200 * 1024 * 1024, dtype=torch.uint8, device=dev)
wsp = ops.ffn_stream_pack(wo, w1, w2)
new = lambda: ops.attnout_ffn_stream(a16, wsp, bo, None, res16, one, zero, 1e-5, b1, b2, one, zero, 1e-5, None, o16)
old = lambda: ops.attnout_ffn_fused_res16(a16, wo, bo, res16, one, zero, 1e-5, w1, b1, w2, b2, one, zero, 1e-5, None, o16)
fillers = {
    "none (same kernel before)": None,
    "host sleep 2 ms": lambda: (torch.cuda.synchronize(), time.sleep(0.002)),
    "linear_res16_ln on the same tensors": lambda: ops.linear_res16_ln(a16, wo, bo, res16, one, zero, None, scratch, 1e-5),
    "the other layer-tail kernel": "other",
    "200 MB memset": lambda: big.zero_(),
    "inproj_attn_causal (decoder size)": "attn",
}
q = torch.empty(M * 256, dtype=torch.bfloat16, device=dev)
w_in, b_in = rn(768, 256, scale=0.06, dt=torch.float16), rn(768, scale=0.2)
attn = lambda: ops.inproj_attn_causal(a16, w_in, b_in, q, scratch, 384, 4, 512, 0, 500)
for name, f in fillers.items():
    res = {}
    for kn, fn, other in (("stream", new, old), ("ffn.hip", old, new)):
        ff = other if f == "other" else attn if f == "attn" else f
        for _ in range(3): fn()
        ts = []
        for _ in range(7):
            if ff is None: fn()
            else: ff()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        res[kn] = statistics.median(ts)
    print(f"behind {name:38s}: stream {res['stream']:7.1f} us   ffn.hip {res['ffn.hip']:7.1f} us", flush=True)
