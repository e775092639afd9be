"""GPU: the packed-stream projection kernel (csrc/proj_stream.hip) against fp32 torch on the same f16 operands, in every destination
layout (row-major, head rows f16 / bf16, second bf16 copy, transposed head rows) and against the two launches it replaces in the
training forward (eend_inproj_heads_train_bf16, ops.retention_proj)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32


def _case(dev, M, N, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, 256, generator=g).to(dev).to(F16)
    w = (torch.randn(N, 256, generator=g) / 16).to(dev).to(F16)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    return x, w, b


def _heads(y, nseq, Tp):          # [M][256] -> [seq][H][Tp][64]
    return y.view(nseq, Tp, 4, 64).permute(0, 2, 1, 3)


@pytest.mark.parametrize("nseq,Tp,N", [(1, 64, 256), (3, 128, 768), (5, 192, 1024), (2, 512, 768), (40, 512, 1024), (7, 320, 512)])
def test_proj_stream_layouts(hip_lib, dev, nseq, Tp, N):
    from fs_eend_amd import ops
    M = nseq * Tp
    x, w, b = _case(dev, M, N, nseq * 3 + N)
    ws = ops.proj_stream_pack(w)
    assert ws is not None
    want = x.float() @ w.float().t() + b
    ng = N // 256
    nan16 = lambda *s, dt=F16: torch.full(s, float("nan"), dtype=dt, device=dev)
    rows_rm = nan16(M, N)
    heads16 = [nan16(nseq, 4, Tp, 64) for _ in range(ng)]
    heads_b = [nan16(nseq, 4, Tp, 64, dt=BF16) for _ in range(ng)]
    heads_t = [nan16(nseq, 4, 64, Tp, dt=BF16 if i & 1 else F16) for i in range(ng)]
    # pass 1: row-major group 0, head rows elsewhere, everything with the second copy and the transposed copy
    groups = []
    for i in range(ng):
        gd = dict(rows2=heads_b[i], heads_t=heads_t[i])
        if i == 0:
            gd.update(rows=rows_rm, kind=1, ld=N)
        else:
            gd.update(rows=heads16[i], kind=2)
        groups.append(gd)
    groups[0]["rows"] = rows_rm                    # (the pointer of the group's first column; ld = N)
    assert ops.proj_stream_ok(256, M, N, Tp, 4, groups)
    ops.proj_stream(x, ws, b, M, N, Tp, 4, groups)
    torch.cuda.synchronize()
    assert (rows_rm[:, :256].float() - want[:, :256]).abs().max().item() < 4e-3
    assert torch.isnan(rows_rm[:, 256:]).all() if N > 256 else True
    for i in range(ng):
        wi = want[:, i * 256:(i + 1) * 256]
        if i > 0:
            assert (heads16[i].float() - _heads(wi, nseq, Tp)).abs().max().item() < 4e-3
        assert (heads_b[i].float() - _heads(wi, nseq, Tp)).abs().max().item() < 3e-2
        assert (heads_t[i].float() - _heads(wi, nseq, Tp).transpose(2, 3)).abs().max().item() < (3e-2 if i & 1 else 4e-3)
        # the second copy is the bf16 rounding of the SAME accumulators
        if i > 0:
            assert (heads_b[i].float() - heads16[i].float()).abs().max().item() < 3e-2
    # pass 2: transposed only / bf16 head rows only
    groups = [dict(heads_t=heads_t[i]) if i & 1 else dict(rows=heads_b[i], kind=2) for i in range(ng)]
    for t in heads_t + heads_b:
        t.fill_(float("nan"))
    ops.proj_stream(x, ws, b, M, N, Tp, 4, groups)
    for i in range(ng):
        wi = want[:, i * 256:(i + 1) * 256]
        if i & 1:
            assert (heads_t[i].float() - _heads(wi, nseq, Tp).transpose(2, 3)).abs().max().item() < 3e-2
            assert torch.isnan(heads_b[i]).all()
        else:
            assert (heads_b[i].float() - _heads(wi, nseq, Tp)).abs().max().item() < 3e-2


@pytest.mark.parametrize("M,N", [(100, 768), (1000, 256), (333, 1024)])
def test_proj_stream_rowmajor_ragged(hip_lib, dev, M, N):
    """row-major destinations take any M (rows beyond M are dropped by the buffer bounds)"""
    from fs_eend_amd import ops
    x, w, b = _case(dev, M, N, M + N)
    ws = ops.proj_stream_pack(w)
    out = torch.full((M + 64, N), float("nan"), dtype=F16, device=dev)
    groups = [dict(rows=out.view(-1)[i * 256:], kind=1, ld=N) for i in range(N // 256)]      # each group's pointer: its first column
    ops.proj_stream(x, ws, b, M, N, 0, 4, groups)
    want = x.float() @ w.float().t() + b
    assert (out[:M].float() - want).abs().max().item() < 4e-3
    assert torch.isnan(out[M:]).all()


def test_proj_stream_matches_train_entries(hip_lib, dev):
    """one launch in place of ops.retention_proj + eend_inproj_heads_train_bf16 (LS training forward): same layouts, values within the
    rounding of two different accumulation orders"""
    from fs_eend_amd import ops, train as T
    nseq, Tp = 6, 512
    M, n = nseq * Tp, nseq * Tp * 256
    x, w, b = _case(dev, M, 1024, 5)
    q, k, kt, vt = (torch.empty(n, dtype=F16, device=dev) for _ in range(4))
    g = torch.empty(M, 256, dtype=F16, device=dev)
    ops.retention_proj(x, w, b, q, k, kt, vt, g, nseq, Tp, 4)
    qb, kb, vb = (torch.empty(n, dtype=BF16, device=dev) for _ in range(3))
    T._call("eend_inproj_heads_train_bf16", x, x.stride(0), w, b, qb, None, kb, None, vb, None, nseq, Tp, 4)
    q2, k2, kt2, vt2 = (torch.full((n,), float("nan"), dtype=F16, device=dev) for _ in range(4))
    g2 = torch.full((M, 256), float("nan"), dtype=F16, device=dev)
    qb2, kb2, vb2 = (torch.full((n,), float("nan"), dtype=BF16, device=dev) for _ in range(3))
    ws = ops.proj_stream_pack(w)
    ops.proj_stream(x, ws, b, M, 1024, Tp, 4, [dict(rows=q2, kind=2, rows2=qb2), dict(rows=k2, kind=2, rows2=kb2, heads_t=kt2),
                                               dict(rows2=vb2, heads_t=vt2), dict(rows=g2, kind=1, ld=256)])
    for a, c, tol in ((q, q2, 4e-3), (k, k2, 4e-3), (kt, kt2, 4e-3), (vt, vt2, 4e-3), (g, g2, 4e-3), (qb, qb2, 3e-2), (kb, kb2, 3e-2), (vb, vb2, 3e-2)):
        assert torch.isfinite(c.float()).all()
        assert (a.float() - c.float()).abs().max().item() < tol


def test_proj_stream_rejects(hip_lib, dev):
    from fs_eend_amd import ops
    x, w, b = _case(dev, 128, 768, 1)
    assert ops.proj_stream_pack(torch.empty(300, 256, dtype=F16, device=dev)) is None          # N not a multiple of 256
    assert ops.proj_stream_pack(torch.empty(1280, 256, dtype=F16, device=dev)) is None         # N > 1024
    hb = torch.empty(128 * 256, dtype=BF16, device=dev)
    assert not ops.proj_stream_ok(256, 128, 768, 96, 4, [dict(rows=hb, kind=2)] * 3)           # Tp not a multiple of 64
    assert not ops.proj_stream_ok(256, 128, 768, 64, 4, [dict(rows=hb, kind=2), dict(), dict(rows=hb, kind=2)])   # a group without destination
    assert ops.proj_stream_ok(256, 128, 768, 64, 4, [dict(rows=hb, kind=2)] * 3)
