"""Same-box A/B of the training forward of the time-axis attention: eend_inproj_heads_train_bf16 + eend_attn_causal_lse_bf16 against
eend_inproj_attn_train_bf16 (attn_stream.hip TRAIN), with and without dropout of the probabilities."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fs_eend_amd  # noqa
from fs_eend_amd import ops, train as T, lib as L

dev = torch.device("cuda:0")
F16, BF16 = torch.float16, torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[n // 2] * 1e3


for nseq, Tp in ((384, 512), (64, 512)):
    M, n = nseq * Tp, nseq * Tp * 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, 256, generator=g).to(dev).to(F16)
    w = (torch.randn(768, 256, generator=g) / 16).to(dev).to(F16)
    b = (torch.randn(768, generator=g) * 0.1).to(dev)
    wp = ops.inproj_attn_pack(w)
    q, k, v, vt = (torch.empty(n, dtype=BF16, device=dev) for _ in range(4))
    ctx = torch.empty(M, 256, dtype=F16, device=dev)
    lse = torch.empty(nseq * 4 * Tp, device=dev)
    for pd in (0.0, 0.1):
        spec = L.Dropout(12345, int(round(pd * (1 << 24))), 1.0 / (1.0 - pd)) if pd else None
        dr = ctypes.byref(spec) if spec is not None else None
        t_p = timeit(lambda: T._call("eend_inproj_heads_train_bf16", x, 256, w, b, q, None, k, None, v, vt, nseq, Tp, 4))
        t_a = timeit(lambda: T._call("eend_attn_causal_lse_bf16", q, k, vt, ctx, lse, nseq, 4, Tp, 256, 0, 500, ops.LN2, dr))
        t_f = timeit(lambda: T._call("eend_inproj_attn_train_bf16", x, 256, wp, b, ctx, 256, q, k, v, lse, nseq, 4, Tp, 0, 500, dr))
        o = torch.empty(M, 256, dtype=F16, device=dev)
        t_i = timeit(lambda: ops.inproj_attn_causal_packed(x, wp, b, o, nseq, 4, Tp, 0, 500))
        print(f"nseq {nseq} p {pd}: in-projection {t_p:.1f} + attention {t_a:.1f} = {t_p + t_a:.1f} us -> one launch {t_f:.1f} us (inference form {t_i:.1f} us)")
