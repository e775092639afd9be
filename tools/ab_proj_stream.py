"""Same-box A/B: the packed-stream projection (proj_stream.hip) against the launches it replaces in the training forward."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fs_eend_amd  # noqa
from fs_eend_amd import ops, train as T

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


def case(M, N):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, 256, generator=g).to(dev).to(F16)
    w = (torch.randn(N, 256, generator=g) / 16).to(dev).to(F16)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    return x, w, b


for nseq, Tp in ((384, 512), (64, 512), (768, 512), (128, 512)):
    M, n = nseq * Tp, nseq * Tp * 256
    # FS: bf16 Q / K / V head rows
    x, w, b = case(M, 768)
    qb, kb, vb = (torch.empty(n, dtype=BF16, device=dev) for _ in range(3))
    t_old = timeit(lambda: T._call("eend_inproj_heads_train_bf16", x, x.stride(0), w, b, qb, None, kb, None, vb, None, nseq, Tp, 4))
    ws = ops.proj_stream_pack(w)
    gr = [dict(rows=qb, kind=2), dict(rows=kb, kind=2), dict(rows=vb, kind=2)]
    t_new = timeit(lambda: ops.proj_stream(x, ws, b, M, 768, Tp, 4, gr))
    # ... as the FS step calls it: with V^T for the forward attention kernel
    vtb = torch.empty(n, dtype=BF16, device=dev)
    t_old_vt = timeit(lambda: T._call("eend_inproj_heads_train_bf16", x, x.stride(0), w, b, qb, None, kb, None, vb, vtb, nseq, Tp, 4))
    gr_vt = [dict(rows=qb, kind=2), dict(rows=kb, kind=2), dict(rows=vb, kind=2, heads_t=vtb)]
    t_new_vt = timeit(lambda: ops.proj_stream(x, ws, b, M, 768, Tp, 4, gr_vt))
    print(f"M {M}: FS in-proj heads + V^T {t_old_vt:.1f} -> {t_new_vt:.1f} us")
    # row-major [M][768] f16 (speaker-axis in-projection)
    o = torch.empty(M, 768, dtype=F16, device=dev)
    t_lin = timeit(lambda: ops.linear(x, w, b, o))
    gr2 = [dict(rows=o.view(-1)[i * 256:], kind=1, ld=768) for i in range(3)]
    t_lin_new = timeit(lambda: ops.proj_stream(x, ws, b, M, 768, 0, 4, gr2))
    # LS: f16 q, k, k^T, v^T, g + bf16 q, k, v
    x, w, b = case(M, 1024)
    q, k, kt, vt = (torch.empty(n, dtype=F16, device=dev) for _ in range(4))
    g = torch.empty(M, 256, dtype=F16, device=dev)

    def two():
        ops.retention_proj(x, w, b, q, k, kt, vt, g, nseq, Tp, 4)
        T._call("eend_inproj_heads_train_bf16", x, x.stride(0), w, b, qb, None, kb, None, vb, None, nseq, Tp, 4)
    t_ls_old = timeit(two)
    ws4 = ops.proj_stream_pack(w)
    gr4 = [dict(rows=q, kind=2, rows2=qb), dict(rows=k, kind=2, rows2=kb, heads_t=kt), dict(rows2=vb, heads_t=vt), dict(rows=g, kind=1, ld=256)]
    t_ls_new = timeit(lambda: ops.proj_stream(x, ws4, b, M, 1024, Tp, 4, gr4))
    print(f"M {M}: FS in-proj heads {t_old:.1f} -> {t_new:.1f} us; row-major [M][768] {t_lin:.1f} -> {t_lin_new:.1f} us; "
          f"LS retention_proj + inproj_heads_train {t_ls_old:.1f} -> {t_ls_new:.1f} us")
