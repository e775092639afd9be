"""Perf study (needs a -DEEND_FFN_ABLATE build, tools/ab_variants.sh build "ablate=-DEEND_FFN_ABLATE"): the fused FFN kernel
with and without its in-loop weight stream -- is the chunk loop bound by the L2 -> LDS delivery of the weight slices?"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch; sys.path.insert(0, %r)
from fs_eend_amd import ops
dev = torch.device("cuda")
M, F = 196608, 2048
x = torch.randn(M, 256, device=dev).half()
w1 = (torch.randn(F, 256, device=dev) * 0.08).half(); b1 = torch.randn(F, device=dev) * 0.3
w2 = (torch.randn(256, F, device=dev) * 0.04).half(); b2 = torch.randn(256, device=dev) * 0.3
res = torch.randn(M, 256, device=dev); g = torch.ones(256, device=dev); be = torch.zeros(256, device=dev)
o32 = torch.empty_like(res); o16 = torch.empty_like(x)
for _ in range(3): ops.ffn_fused(x, w1, b1, w2, b2, res, g, be, o32, o16)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.ffn_fused(x, w1, b1, w2, b2, res, g, be, o32, o16)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print("%%-40s %%.4f ms  %%.1f TFLOP/s" %% (%r, ms, 4.0 * M * F * 256 / ms / 1e9), flush=True)
'''
lib = os.path.join(ROOT, "fs-eend_amd", "csrc", "variants", "libeend_hip_ablate.so")
for flag, label in [(0, "normal"), (4, "no in-loop weight DMA"), (6, "no weight DMA, no residual read"), (7, "... and no output stores")]:
    env = dict(os.environ, EEND_FFN_DBG=str(flag), EEND_HIP_LIB=lib)
    subprocess.run([sys.executable, "-c", CODE % (ROOT, label)], env=env, check=False)
