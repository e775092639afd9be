"""One weight-gradient shape, a few launches (PMC / kernel-trace target): python tools/ab_wgrad_one.py M N K [lda [ldb]]
(lda / ldb > N / K: the operands are column blocks of wider row-major tensors)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd.train import _call, WS_FLOATS
M, N, K = (int(a) for a in sys.argv[1:4])
lda = int(sys.argv[4]) if len(sys.argv) > 4 else N
ldb = int(sys.argv[5]) if len(sys.argv) > 5 else K
xf16 = int(os.environ.get("X_F16", "1"))
dev = torch.device("cuda")
ws = torch.empty(WS_FLOATS, dtype=torch.float32, device=dev)
dy = (torch.randn(M, lda, device=dev) * 1e-3).to(torch.bfloat16)
x = torch.randn(M, ldb, device=dev).to(torch.float16 if xf16 else torch.bfloat16)
out = torch.empty(N, K, dtype=torch.float32, device=dev)
for _ in range(4):
    _call("eend_wgrad_bf16", dy, lda, x, ldb, xf16, M, N, K, ws, WS_FLOATS, out, K, K, 1.0, 0)
torch.cuda.synchronize()
