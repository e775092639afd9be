"""Same-box timing of the decoder input fan-out: store-shaped kernel (convert_rows.hip) vs the GEMM epilogue (EEND_CONVERT_GEMM=1
in a second process) at the FS model.test shape (64 x 512 rows, 6 slots)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("fs-eend_amd.ops")
g = torch.Generator().manual_seed(0)
B, Tp, C = 64, 512, 6
e = torch.randn(B * Tp, 256, generator=g).cuda().half()
w1 = (torch.randn(256, 256, generator=g) / 16).cuda().half()
pc = torch.randn(C, 256, generator=g).cuda()
o16 = torch.empty(B * C * Tp, 256, dtype=torch.float16, device="cuda")
fn = lambda: ops.convert_fanout(e, w1, pc, None, o16, B, Tp, C)
ts = []
for _ in range(5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(("GEMM epilogue" if os.environ.get("EEND_CONVERT_GEMM") == "1" else "store-shaped kernel"), f"min {min(ts):.1f} us  median {sorted(ts)[2]:.1f} us  checksum {o16.float().abs().sum().item():.6e}")
