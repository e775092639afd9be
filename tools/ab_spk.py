"""A/B study: speaker-axis fused kernel at the FS decoder size (use with tools/ab_variants.sh / EEND_HIP_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
B, C, Tp, T, H = 64, 6, 512, 500, 4
M = B * C * Tp
x = torch.randn(M, 256, generator=g).to(dev).half()
w = (torch.randn(768, 256, generator=g) * 0.06).to(dev).half()
bias = (torch.randn(768, generator=g) * 0.2).to(dev)
o = torch.empty(M, 256, dtype=torch.float16, device=dev)
for _ in range(5): ops.spk_qkv_attn(x, w, bias, o, B, C, Tp, H, t_valid=T)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): ops.spk_qkv_attn(x, w, bias, o, B, C, Tp, H, t_valid=T)
b.record(); torch.cuda.synchronize()
print(f"spk_qkv_attn M={M}: {a.elapsed_time(b) / 50 * 1e3:.1f} us   checksum {float(o.float().abs().sum()):.6e}")
