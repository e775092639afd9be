"""Profiling driver: LS-EEND model.test at BASELINE config 3's size (16 x T=2000, C=10), eager launches, for
`rocprofv3 --kernel-trace --stats` (tools/gpu_ls_prof.sh) -- per-kernel shares of the LS batch path."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd.ls_model import OnlineConformerRetentionDADiarization  # noqa: E402

LS_CFG = dict(n_units=256, n_heads=4, enc_n_layers=4, dec_n_layers=2, dropout=0.1, max_seqlen=1000,
              recurrent_chunk_size=500, feed_forward_expansion_factor=4, dec_dim_feedforward=2048,
              conv_expansion_factor=2, conv_kernel_size=16, half_step_residual=True, conv_delay=9)
dev = torch.device("cuda:0")
torch.manual_seed(0)
ls = OnlineConformerRetentionDADiarization(n_speakers=None, in_size=345, **LS_CFG).eval().to(dev)
g = torch.Generator().manual_seed(1)
B, T, C = 16, 2000, 10
src = [(torch.randn(T, 345, generator=g) * 2 - 3).to(dev) for _ in range(B)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(2):
    ls.test(src, [T] * B, C)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ls.test(src, [T] * B, C)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"LS-EEND test {B} x {T}, C={C}: {dt * 1e3:.3f} ms/step eager = {B * T / dt / 1e6:.2f} M frames/s")
