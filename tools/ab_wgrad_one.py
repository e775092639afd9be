"""One weight-gradient shape, a few launches (PMC target): python tools/ab_wgrad_one.py M N K"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd.train import _call, WS_FLOATS
M, N, K = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
ws = torch.empty(WS_FLOATS, dtype=torch.float32, device=dev)
dy = (torch.randn(M, N, generator=g) * 1e-3).to(dev).to(torch.bfloat16)
x = torch.randn(M, K, generator=g).to(dev).to(torch.float16)
out = torch.empty(N, K, dtype=torch.float32, device=dev)
for _ in range(4):
    _call("eend_wgrad_bf16", dy, N, x, K, 1, M, N, K, ws, WS_FLOATS, out, K, K, 1.0, 0)
torch.cuda.synchronize()
