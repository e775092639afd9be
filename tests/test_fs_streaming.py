"""GPU: FS-EEND frame-by-frame streaming (K/V-cache decode attention) driven as the reference's
FS-EEND/streaming_infer_dia.py:70-86 does, against the reference's own streaming logits."""
import pytest
import torch

from oracle import fixtures as FX
from tests.helpers import build_fs_mirror, max_abs

pytestmark = pytest.mark.gpu
F16, F32 = torch.float16, torch.float32


def test_attn_decode_kernel(hip_lib, dev):
    from fs_eend_amd import ops
    N, H, D, cap = 3, 4, 256, 256
    g = torch.Generator().manual_seed(5)
    kc = torch.zeros(N, H, cap, 64, dtype=F16, device=dev)
    vc = torch.zeros(N, H, cap, 64, dtype=F16, device=dev)
    hist = []
    for t in range(150):
        qkv = torch.randn(N, 3 * D, generator=g).to(dev).to(F16)
        out = torch.empty(N, D, dtype=F16, device=dev)
        ops.attn_decode(qkv, kc, vc, out, N, H, cap, t)
        hist.append(qkv.float())
        if t in (0, 1, 63, 64, 65, 149):
            x = torch.stack(hist, dim=1)                                    # (N,t+1,3D)
            q = x[:, -1, :D].view(N, H, 1, 64)
            k = x[:, :, D:2 * D].view(N, t + 1, H, 64).transpose(1, 2)
            v = x[:, :, 2 * D:].view(N, t + 1, H, 64).transpose(1, 2)
            want = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).reshape(N, D)
            assert max_abs(out, want.cpu()) < 4e-3, f"t={t}"


def test_fs_streaming_vs_reference_streaming(hip_lib, dev):
    from fs_eend_amd.fs_stream import StreamingTransformerEDADiarization, copy_params_from_masked_to_streaming
    meta, arr = FX.load_case("fs_stream_T60")
    m = build_fs_mirror(meta).to(dev)
    sm = StreamingTransformerEDADiarization(in_size=meta["in_size"], **meta["cfg"]).eval().to(dev)
    copy_params_from_masked_to_streaming(m, sm)
    src = FX.make_src([meta["T"]], meta["in_size"], meta["xseed"])[0].to(dev)
    ys = []
    for t in range(meta["T"]):
        y = sm.test(src[t].view(1, 1, -1), meta["C"])
        if t < 9:
            assert y is None                                   # look-ahead not filled yet
        if y is not None:
            ys.append(y)
    for _ in range(m.delay):
        y = sm.test(src[0].view(1, 1, -1), meta["C"], dummy_conv_input=True)
        if y is not None:
            ys.append(y)
    ys = torch.cat(ys, dim=1)[0]
    assert ys.shape == arr["stream_logits"].shape
    err = max_abs(ys, arr["stream_logits"])
    print(f"FS streaming logits vs reference streaming: {err:.2e}")
    assert err < 1e-3
    batch = m.test([src], [meta["T"]], meta["C"])[0][0]
    assert max_abs(ys, batch.cpu()) < 1e-3                     # the reference asserts streaming == batch (atol 1e-4 in fp32)
    # a second stream after reset reproduces the first
    sm.reset_streaming_state()
    y0 = [sm.test(src[t].view(1, 1, -1), meta["C"]) for t in range(12)]
    assert torch.equal(y0[9], ys[0].view(1, 1, -1).to(dev))


def test_cache_growth(hip_lib, dev):
    """K/V cache doubles transparently (cap 512 -> 1024) without changing results."""
    from fs_eend_amd.fs_stream import _KvCache
    from fs_eend_amd import ops
    kv = _KvCache(1, 4, dev, cap=8)
    g = torch.Generator().manual_seed(9)
    outs = []
    for t in range(20):
        kv.ensure_room()
        qkv = torch.randn(1, 768, generator=g).to(dev).to(F16)
        out = torch.empty(1, 256, dtype=F16, device=dev)
        ops.attn_decode(qkv, kv.k, kv.v, out, 1, 4, kv.cap, kv.t)
        kv.t += 1
        outs.append(out.clone())
    assert kv.cap >= 32 and all(torch.isfinite(o).all() for o in outs)


@pytest.mark.parametrize("use_graph", [True, False])
def test_fs_stream_session_matches_eager(hip_lib, dev, use_graph):
    """FsStreamSession (state in fixed HBM buffers, device-side history counters, three hipGraphs per cache-capacity
    bucket) reproduces the eager frame-by-frame API bit for bit -- across two cache growths (cap 16 -> 32 -> 64), the
    flush frames, and a second stream after reset."""
    from fs_eend_amd.fs_stream import (FsStreamSession, StreamingTransformerEDADiarization,
                                       copy_params_from_masked_to_streaming)
    meta, arr = FX.load_case("fs_stream_T60")
    m = build_fs_mirror(meta).to(dev)
    sm = StreamingTransformerEDADiarization(in_size=meta["in_size"], **meta["cfg"]).eval().to(dev)
    copy_params_from_masked_to_streaming(m, sm)
    src = FX.make_src([meta["T"]], meta["in_size"], meta["xseed"])[0].to(dev)
    want = []
    for t in range(meta["T"]):
        y = sm.test(src[t].view(1, 1, -1), meta["C"])
        if y is not None:
            want.append(y)
    for _ in range(m.delay):
        y = sm.test(src[0].view(1, 1, -1), meta["C"], dummy_conv_input=True)
        if y is not None:
            want.append(y)
    want = torch.cat(want, dim=1)
    ses = FsStreamSession(sm, meta["C"], cap=16, use_graph=use_graph)
    for rep in range(2):
        got = []
        for t in range(meta["T"]):
            y = ses.push(src[t])
            assert (y is None) == (t < 9)
            if y is not None:
                got.append(y)
        got += ses.flush()
        got = torch.cat(got, dim=1)
        assert got.shape == want.shape
        assert torch.equal(got, want), f"stream {rep}: max diff {float((got - want).abs().max()):.3e}"
        assert ses.cap == 64 and int(ses.t_enc) == meta["T"] and int(ses.t_dec) == meta["T"]
        ses.reset()
    assert max_abs(want[0], arr["stream_logits"]) < 1e-3


@pytest.mark.parametrize("N", [1, 6])
def test_attn_decode_split_matches_single_wave(hip_lib, dev, N):
    """The key-split decode (long histories) against the single-wave kernel at history lengths around the 512-key split
    boundaries and far beyond them; both append the new token's k / v at row t."""
    from fs_eend_amd import ops
    H, cap = 4, 8192
    g = torch.Generator().manual_seed(N)
    kc = (torch.randn(N, H, cap, 64, generator=g) * 0.7).to(F16).to(dev)
    vc = torch.randn(N, H, cap, 64, generator=g).to(F16).to(dev)
    ws = torch.empty(ops.attn_decode_split_ws(N, H, cap), dtype=torch.float32, device=dev)
    for t in (0, 1, 63, 511, 512, 513, 1024, 5000, cap - 1):
        qkv = torch.randn(N, 768, generator=g).to(F16).to(dev)
        k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        o1 = torch.empty(N, 256, dtype=F16, device=dev)
        o2 = torch.full((N, 256), float("nan"), dtype=F16, device=dev)
        ops.attn_decode(qkv, k1, v1, o1, N, H, cap, t)
        td = torch.tensor([t], dtype=torch.int32, device=dev)
        ops.attn_decode_split(qkv, k2, v2, o2, ws, N, H, cap, td)
        torch.cuda.synchronize()
        assert torch.equal(k1, k2) and torch.equal(v1, v2)                   # same append
        assert float((o1.float() - o2.float()).abs().max()) < 2e-3, t
