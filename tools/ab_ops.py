"""Perf study helper: times a few C-ABI ops in isolation (HIP events, 20 launches each) -- used with tools/ab_variants.sh
to compare library builds on the same GPU box (box-to-box spread is +-7 %, far above most kernel-level effects)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def rn(*s, scale=1.0, dt=torch.float32):
    return (torch.randn(*s, generator=g) * scale).to(dev).to(dt)
def timeit(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:28s} {a.elapsed_time(b) / n * 1e3:9.1f} us", flush=True)
B, C, Tp, H = 64, 6, 512, 4
M = B * C * Tp
x16 = rn(M, 256, dt=torch.float16)
w_in, b_in = rn(768, 256, scale=0.06, dt=torch.float16), rn(768, scale=0.2)
q = torch.empty(M * 256, dtype=torch.bfloat16, device=dev); k = torch.empty_like(q); vt = torch.empty_like(q)
timeit("inproj_heads M=196608", lambda: ops.inproj_heads(x16, w_in, b_in, q, k, vt, B * C, Tp, H))
o16 = torch.empty(M, 256, dtype=torch.float16, device=dev)
timeit("attn_causal nseq=384", lambda: ops.attn_causal(q, k, vt, o16, B * C, H, Tp, 0, Tp, scale=ops.LN2))
timeit("spk_qkv_attn M=196608", lambda: ops.spk_qkv_attn(x16, w_in, b_in, o16, B, C, Tp, H))
wo, bo = rn(256, 256, scale=0.06, dt=torch.float16), rn(256, scale=0.2)
w1, b1 = rn(2048, 256, scale=0.08, dt=torch.float16), rn(2048, scale=0.3)
w2, b2 = rn(256, 2048, scale=0.04, dt=torch.float16), rn(256, scale=0.3)
res = rn(M, 256); one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
o32 = torch.empty_like(res); out16 = torch.empty_like(x16)
timeit("attnout_ffn_fused M=196608", lambda: ops.attnout_ffn_fused(x16, wo, bo, res, one, zero, 1e-5, w1, b1, w2, b2, one, zero, 1e-5, o32, out16))
timeit("linear_res_ln M=196608", lambda: ops.linear_res_ln(x16, wo, bo, res, one, zero, o32, out16, 1e-5))
src = [rn(500, 345) for _ in range(64)]
bn = tuple(rn(345).abs() + 0.5 if i in (0, 3) else rn(345) for i in range(4))
xin = torch.zeros(64 * 512, 384, dtype=torch.float16, device=dev)
timeit("gather_bn_cast_pad B=64", lambda: ops.gather_bn_cast_pad(src, bn, xin, 500, 512, -1.0, True, 1e-5))
