"""Same-box A/B of ffn_stream.hip variant builds (tools/fs_variants.sh): each variant library is loaded in its own process,
rounds are interleaved across processes by the caller (run with: for r in 1 2 3; do for v in ...; do EEND_HIP_LIB=... python
tools/ab_fs_variants.py; done; done).  Prints one line: median time of the three layer-tail forms at the headline shape."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def rn(*s, scale=1.0, dt=torch.float32):
    return (torch.randn(*s, generator=g) * scale).to(dev).to(dt)
REPS = int(os.environ.get('AB_REPS', 4))
M, Fh = 196608, int(os.environ.get("AB_F", 2048))
a16 = rn(M, 256, dt=torch.float16); res16 = rn(M, 256, dt=torch.float16); res32 = rn(M, 256)
wo, bo = rn(256, 256, scale=0.06, dt=torch.float16), rn(256, scale=0.2)
w1, b1 = rn(Fh, 256, scale=0.08, dt=torch.float16), rn(Fh, scale=0.3)
w2, b2 = rn(256, Fh, scale=0.04, dt=torch.float16), rn(256, scale=0.3)
one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
o32 = torch.empty(M, 256, device=dev); o16 = torch.empty(M, 256, dtype=torch.float16, device=dev)
wsp, wsn = ops.ffn_stream_pack(wo, w1, w2), ops.ffn_stream_pack(None, w1, w2)
fns = [lambda: ops.attnout_ffn_stream(a16, wsp, bo, None, res16, one, zero, 1e-5, b1, b2, one, zero, 1e-5, None, o16),
       lambda: ops.attnout_ffn_stream(a16, wsp, bo, res32, None, one, zero, 1e-5, b1, b2, one, zero, 1e-5, o32, o16),
       lambda: ops.ffn_stream(a16, wsn, b1, b2, res32, one, zero, o32, o16),
       lambda: ops.attnout_ffn_fused_res16(a16, wo, bo, res16, one, zero, 1e-5, w1, b1, w2, b2, one, zero, 1e-5, None, o16)]
filler = lambda: ops.linear_res16_ln(a16, wo, bo, res16, one, zero, None, o16, 1e-5)     # something HBM-bound in between, as in the model
out = []
for fn in fns:
    for _ in range(3): fn()
    ts = []
    for _ in range(5):
        filler()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / REPS * 1e3)
    out.append(statistics.median(ts))
print(f"{os.path.basename(os.environ.get('EEND_HIP_LIB', 'default')):36s} attnout res16 {out[0]:7.1f} us   attnout f32res+out32 {out[1]:7.1f} us   ffn_stream {out[2]:7.1f} us   [ffn.hip attnout res16 {out[3]:7.1f} us]", flush=True)
