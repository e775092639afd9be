"""Perf study: time one GEMM shape under the runtime ablation flags (EEND_GEMM_DBG: 1 skip stores, 2 reload k-tile 0,
4 skip MFMA).  The flags are only compiled in when gemm.hip / api.hip are built with -DEEND_GEMM_ABLATE (the shipped
library has none of these branches); without that build every row below times the normal kernel."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch; sys.path.insert(0, %r)
from fs_eend_amd import ops
dev = torch.device("cuda")
M, N, K = %d, %d, %d
a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.1).half(); b = torch.randn(N, device=dev)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
for _ in range(3): ops.linear(a, w, b, out, relu=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): ops.linear(a, w, b, out, relu=True)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("%%-28s %%.4f ms  %%.1f TFLOP/s" %% (%r, ms, 2.0 * M * N * K / ms / 1e9))
'''
def run(flag, label, shape):
    env = dict(os.environ, EEND_GEMM_DBG=str(flag))
    subprocess.run([sys.executable, "-c", CODE % (ROOT, *shape, label)], env=env, check=False)
for shape in [(196608, 2048, 256), (196608, 768, 256)]:
    print("shape", shape)
    for flag, label in [(0, "normal"), (1, "no stores"), (2, "reload k-tile 0"), (4, "no MFMA"), (3, "no stores + tile0"),
                        (5, "no stores + no MFMA"), (7, "only tile-0 loads + LDS")]:
        run(flag, label, shape)
