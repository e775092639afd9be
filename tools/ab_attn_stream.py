"""Same-box timing of the packed in-projection + attention kernel (attn_stream.hip) against attn_fused.hip at the FS
model.test shapes (decoder: 384 sequences, encoder: 64; Tp = 512, kv_len = 500)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("fs-eend_amd.ops")
dev = "cuda"
g = torch.Generator().manual_seed(0)
Tp = 512
w = torch.randn(768, 256, generator=g) / 16
b = torch.randn(768, generator=g) * 0.1
w[:256] *= ops.QSCALE_LOG2; b[:256] *= ops.QSCALE_LOG2
w = w.to(dev).half(); b = b.to(dev)
wp = ops.inproj_attn_pack(w)
for nseq in (384, 64):
    x = torch.randn(nseq * Tp, 256, generator=g).to(dev).half()
    qs = torch.empty(nseq * Tp * 256, dtype=torch.bfloat16, device=dev)
    o0 = torch.empty_like(x); o1 = torch.empty_like(x)
    kq, kk, kv = (torch.empty(nseq * Tp * 256, dtype=torch.bfloat16, device=dev) for _ in range(3))
    def two_kernels():
        ops.inproj_heads(x, w, b, kq, kk, kv, nseq, Tp, 4)
        ops.attn_causal(kq, kk, kv, o0, nseq, 4, Tp, 0, 500, scale=ops.LN2)
    fns = {"inproj_heads + attn_causal": two_kernels,
           "attn_stream": lambda: ops.inproj_attn_causal_packed(x, wp, b, o1, nseq, 4, Tp, 0, 500)}
    res = {k: [] for k in fns}
    for _ in range(5):
        for k, fn in fns.items():
            for _ in range(2): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"nseq {nseq}: max |diff| {(o0.float() - o1.float()).abs().max().item():.3e}  " +
          "  ".join(f"{k} min {min(v):.1f} med {sorted(v)[2]:.1f} us" for k, v in res.items()))
