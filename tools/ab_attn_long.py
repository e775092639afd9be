"""A/B on one box: the grouped long-window form of the packed in-projection + attention kernel against the two-kernel path
(in-projection to HBM + tiled attention) and against the 512-frame kernel on the same number of frames."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fs_eend_amd  # noqa
from fs_eend_amd import ops

dev = torch.device("cuda:0")
F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32


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


for frames, Tp in ((163840, 512), (163840, 1024), (163840, 2048), (32768, 512), (32768, 1024), (32768, 2048), (163840, 4096)):
    nseq = frames // Tp
    g = torch.Generator().manual_seed(1)
    x = torch.randn(nseq * Tp, 256, generator=g).to(dev).to(F16)
    w = (torch.randn(768, 256, generator=g) / 16).to(dev).to(F16)
    b = (torch.randn(768, generator=g) * 0.1).to(dev)
    wp = ops.inproj_attn_pack(w)
    o = torch.empty(nseq * Tp, 256, dtype=F16, device=dev)
    kv = Tp - 12
    if Tp <= 512:
        t = timeit(lambda: ops.inproj_attn_causal_packed(x, wp, b, o, nseq, 4, Tp, 0, kv))
        print(f"frames {frames} Tp {Tp} nseq {nseq}: packed {t:.1f} us")
        continue
    need = ops.inproj_attn_long_scratch(nseq, Tp, 0, kv)
    part = torch.empty(need[0], dtype=F16, device=dev); lse = torch.empty(need[1], dtype=F32, device=dev)
    t_long = timeit(lambda: ops.inproj_attn_causal_long(x, wp, b, o, part, lse, nseq, 4, Tp, 0, kv))
    q, k, vt = (torch.empty(nseq * Tp * 256, dtype=BF16, device=dev) for _ in range(3))

    def two():
        ops.inproj_heads(x, w, b, q, k, vt, nseq, Tp, 4)
        ops.attn_causal(q, k, vt, o, nseq, 4, Tp, 0, kv, scale=ops.LN2)
    t_two = timeit(two)
    print(f"frames {frames} Tp {Tp} nseq {nseq}: grouped {t_long:.1f} us, two kernels {t_two:.1f} us")
