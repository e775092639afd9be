"""Perf study: retention backward (eend_retention_bwd_bf16) at the LS decoder (384 seq) and encoder (64 seq) sizes, T = 1000, L = 500."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd.train import _call
dev = torch.device("cuda")
H, D, Tv, Tp, L = 4, 256, 1000, 1024, 500
n_iter = int(os.environ.get("N_ITER", "20"))
for nseq in (64, 384):
    g = torch.Generator(device="cpu").manual_seed(nseq)
    M = nseq * Tp
    def heads(scale):
        x = (torch.randn(nseq, H, Tp, 64, generator=g) * scale).to(torch.bfloat16)
        return x.to(dev).contiguous(), x.transpose(-1, -2).contiguous().to(dev)
    q, qt = heads(0.5); k, kt = heads(0.5); v, vt = heads(1.0)
    dctx = (torch.randn(M, D, generator=g) * 1e-3).to(dev)
    gate = torch.randn(M, D, generator=g).to(torch.float16).to(dev)
    rhat = torch.randn(M, D, generator=g).to(torch.float16).to(dev)
    rc = (0.5 + torch.rand(M, H, generator=g)).to(dev)
    nc = Tv // L
    ot = torch.empty(M * D, dtype=torch.bfloat16, device=dev); ott = torch.empty(M * D, dtype=torch.bfloat16, device=dev)
    kv_ws = torch.empty(nseq * H * nc * 4096, dtype=torch.float32, device=dev); g_ws = torch.empty_like(kv_ws)
    st = torch.empty(nseq * H * nc * 6 * 4096, dtype=torch.bfloat16, device=dev)
    dq = torch.empty(M, 4 * D, dtype=torch.bfloat16, device=dev)
    fn = lambda: _call("eend_retention_bwd_bf16", q, qt, k, kt, v, vt, dctx, gate, D, rhat, rc, ot, ott, kv_ws, g_ws, st, dq, 4 * D, nseq, H, Tp, L, Tv, 0.125)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n_iter): fn()
    b.record(); torch.cuda.synchronize()
    print(f"retention bwd nseq={nseq}: {a.elapsed_time(b) / n_iter * 1e3:8.1f} us   chk {float(dq.float().abs().sum()):.5e}", flush=True)
