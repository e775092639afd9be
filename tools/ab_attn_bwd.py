"""Perf study: attention backward (dq + dkv kernels) at the decoder (384 seq) and encoder (64 seq) sizes, T = 500."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import ops
from fs_eend_amd.train import _call
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
T, Tp, H = 500, 512, 4
for nseq in (64, 384):
    mk = lambda *s_: (torch.randn(*s_, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    q, k, v = mk(nseq, H, Tp, 64), mk(nseq, H, Tp, 64), mk(nseq, H, Tp, 64)
    qt, kt = q.transpose(-1, -2).contiguous(), k.transpose(-1, -2).contiguous()
    dO = (torch.randn(nseq * Tp, 256, generator=g) * 1e-3).to(dev).to(torch.bfloat16)
    O = torch.randn(nseq * Tp, 256, generator=g).to(dev).to(torch.float16)
    lse = torch.randn(nseq * H * Tp, generator=g).to(dev) + 6
    dot_ws = torch.empty(nseq * Tp * 256, dtype=torch.bfloat16, device=dev)
    dh_ws = torch.empty(nseq * H * Tp, device=dev)
    dqkv = torch.empty(nseq * Tp, 768, dtype=torch.bfloat16, device=dev)
    import ctypes
    from fs_eend_amd import lib as L
    for pd in (0.0, 0.1):
        spec = None if pd == 0 else L.Dropout(12345, int(round(pd * (1 << 24))), 1.0 / (1.0 - pd))
        dref = None if spec is None else ctypes.byref(spec)
        fn = lambda: _call("eend_attn_causal_bwd_bf16", q, qt, k, kt, v, dO, 256, O, 256, lse, dot_ws, dh_ws, dqkv, 768, nseq, H, Tp, 0, T, T,
                           1.0, 0.125, ops.LN2, dref)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        print(f"attn bwd nseq={nseq} p_drop={pd}: {a.elapsed_time(b) / 20 * 1e3:8.1f} us   chk {float(dqkv.float().abs().sum()):.5e}", flush=True)
