"""GPU: every C-ABI kernel against a plain PyTorch fp32 reference of the same op, on the
same (already f16/bf16-quantised) inputs.  Inputs are asymmetric random data so that a
transposed fragment or swapped operand cannot pass."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32


def rnd(shape, dev, seed, dtype=F32, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


def close(got, want, atol, rtol, what):
    got, want = got.float(), want.float()
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    bad = (err > bound)
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements off, max err {err.max().item():.3e} "
                           f"(first bad index {bad.nonzero()[0].tolist()})")


@pytest.mark.parametrize("M,N,K,relu", [(1, 128, 64, False), (100, 768, 256, False), (333, 2048, 256, True),
                                        (128, 128, 2048, True), (4100, 256, 384, False)])
def test_linear(hip_lib, dev, M, N, K, relu):
    from fs_eend_amd import ops
    a, w, b = rnd((M, K), dev, 1, F16), rnd((N, K), dev, 2, F16, 0.1), rnd((N,), dev, 3)
    out = torch.full((M, N), float("nan"), dtype=F16, device=dev)
    ops.linear(a, w, b, out, relu=relu)
    want = a.float() @ w.float().t() + b
    if relu:
        want = want.relu()
    close(out, want, 2e-3, 2e-3, f"linear {M}x{N}x{K}")


def test_linear_strided_output(hip_lib, dev):
    from fs_eend_amd import ops
    a, w = rnd((70, 256), dev, 4, F16), rnd((128, 256), dev, 5, F16, 0.1)
    big = torch.zeros(70, 384, dtype=F16, device=dev)
    out = big[:, 128:256]
    L = __import__("fs_eend_amd.lib", fromlist=["x"]).load()
    rc = L.eend_linear_f16(a.data_ptr(), 256, w.data_ptr(), 256, None, out.data_ptr(), 384, 70, 128, 256, 0,
                           torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    close(big[:, 128:256], a.float() @ w.float().t(), 2e-3, 2e-3, "strided linear")
    assert (big[:, :128] == 0).all() and (big[:, 256:] == 0).all()


def test_invalid_arguments_are_rejected(hip_lib, dev):
    L = hip_lib
    a = torch.zeros(64, 64, dtype=F16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert L.eend_linear_f16(a.data_ptr(), 64, a.data_ptr(), 64, None, a.data_ptr(), 64, 64, 100, 64, 0, s) == -1  # N%128
    assert L.eend_linear_f16(a.data_ptr(), 64, a.data_ptr(), 64, None, a.data_ptr(), 64, 64, 128, 60, 0, s) == -1  # K%64
    assert L.eend_linear_f16(None, 64, a.data_ptr(), 64, None, a.data_ptr(), 64, 64, 128, 64, 0, s) == -1
    assert L.eend_attn_causal_bf16(a.data_ptr(), a.data_ptr(), a.data_ptr(), a.data_ptr(), 1, 4, 100, 256, 0, 100,
                                   0.125, s) == -1                                                    # Tp%64
    assert L.eend_spk_attn_f16(a.data_ptr(), a.data_ptr(), 1, 13, 64, 4, 0.125, s) == -1               # C>12


@pytest.mark.parametrize("nseq,Tp", [(1, 64), (3, 192), (2, 512)])
def test_inproj_heads(hip_lib, dev, nseq, Tp):
    from fs_eend_amd import ops
    H, D = 4, 256
    M = nseq * Tp
    a, w, b = rnd((M, D), dev, 6, F16), rnd((3 * D, D), dev, 7, F16, 0.1), rnd((3 * D,), dev, 8)
    q = torch.full((M * D,), float("nan"), dtype=BF16, device=dev)
    k, vt = q.clone(), q.clone()
    ops.inproj_heads(a, w, b, q, k, vt, nseq, Tp, H)
    y = (a.float() @ w.float().t() + b).view(nseq, Tp, 3, H, 64)
    wq = y[:, :, 0].permute(0, 2, 1, 3)               # (nseq,H,Tp,64)
    wk = y[:, :, 1].permute(0, 2, 1, 3)
    wv = y[:, :, 2].permute(0, 2, 3, 1)               # (nseq,H,64,Tp)
    close(q.view(nseq, H, Tp, 64), wq, 1e-2, 1e-2, "Q heads")
    close(k.view(nseq, H, Tp, 64), wk, 1e-2, 1e-2, "K heads")
    close(vt.view(nseq, H, 64, Tp), wv, 1e-2, 1e-2, "V^T heads")


@pytest.mark.parametrize("M,K,res", [(64, 256, True), (200, 2048, True), (130, 384, False), (1, 256, True)])
def test_linear_res_ln(hip_lib, dev, M, K, res):
    from fs_eend_amd import ops
    a, w, b = rnd((M, K), dev, 9, F16), rnd((256, K), dev, 10, F16, 0.1), rnd((256,), dev, 11)
    r = rnd((M, 256), dev, 12) + 0.7 if res else None
    g, be = rnd((256,), dev, 13) * 0.2 + 1, rnd((256,), dev, 14) * 0.1
    o32 = torch.full((M, 256), float("nan"), dtype=F32, device=dev)
    o16 = torch.full((M, 256), float("nan"), dtype=F16, device=dev)
    ops.linear_res_ln(a, w, b, r, g, be, o32, o16, 1e-5)
    y = a.float() @ w.float().t() + b + (r if res else 0)
    want = torch.nn.functional.layer_norm(y, (256,), g, be, 1e-5)
    close(o32, want, 2e-4, 2e-4, "res+LN f32")
    close(o16, want, 3e-3, 2e-3, "res+LN f16")


def test_linear_res_ln_inplace_residual(hip_lib, dev):
    from fs_eend_amd import ops
    M, K = 300, 256
    a, w, b = rnd((M, K), dev, 15, F16), rnd((256, K), dev, 16, F16, 0.1), rnd((256,), dev, 17)
    r = rnd((M, 256), dev, 18)
    g, be = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    want = torch.nn.functional.layer_norm(a.float() @ w.float().t() + b + r, (256,), g, be, 1e-5)
    o16 = torch.empty((M, 256), dtype=F16, device=dev)
    ops.linear_res_ln(a, w, b, r, g, be, r, o16, 1e-5)       # out32 aliases res (how the model calls it)
    close(r, want, 2e-4, 2e-4, "in-place res+LN")


def test_linear_res_scale(hip_lib, dev):
    from fs_eend_amd import ops
    M, K = 150, 1024
    a, w, b = rnd((M, K), dev, 19, F16), rnd((256, K), dev, 20, F16, 0.05), rnd((256,), dev, 21)
    r = rnd((M, 256), dev, 22)
    o32 = torch.empty((M, 256), dtype=F32, device=dev)
    o16 = torch.empty((M, 256), dtype=F16, device=dev)
    ops.linear_res_scale(a, w, b, r, 0.5, o32, o16)
    want = (a.float() @ w.float().t() + b) * 0.5 + r
    close(o32, want, 2e-4, 2e-4, "res+scale f32")
    close(o16, want, 3e-3, 2e-3, "res+scale f16")


@pytest.mark.parametrize("nseq,Tp,ilens", [(2, 64, [64, 30]), (3, 128, [100, 128, 1]), (1, 512, [500])])
def test_conv1d_l2norm(hip_lib, dev, nseq, Tp, ilens):
    from fs_eend_amd import ops
    D, k, pad = 256, 19, 9
    x = rnd((nseq, Tp, D), dev, 23, F16)
    w = rnd((D, D, k), dev, 24, F32, 0.05)
    b = rnd((D,), dev, 25)
    wr = w.permute(0, 2, 1).reshape(D, k * D).to(F16).contiguous()
    il = torch.tensor(ilens, dtype=torch.int32, device=dev)
    o32 = torch.empty((nseq * Tp, D), dtype=F32, device=dev)
    o16 = torch.empty((nseq * Tp, D), dtype=F16, device=dev)
    ops.conv1d_l2norm(x.view(-1, D), wr, b, il, o32, o16, nseq, Tp, D, k, pad)
    xm = x.float().clone()
    for i, l in enumerate(ilens):
        xm[i, l:] = 0                                   # truncate to ilen, zero re-pad
    y = torch.nn.functional.conv1d(xm.transpose(1, 2), wr.view(D, k, D).permute(0, 2, 1).float(), b, padding=pad)
    y = y.transpose(1, 2)
    want = y / torch.linalg.vector_norm(y, dim=-1, keepdim=True)
    close(o32.view(nseq, Tp, D), want, 2e-4, 1e-3, "conv+l2 f32")
    close(o16.view(nseq, Tp, D), want, 1e-3, 2e-3, "conv+l2 f16")


@pytest.mark.parametrize("B,Tp,C", [(2, 64, 4), (1, 128, 10), (3, 64, 1)])
def test_convert_fanout(hip_lib, dev, B, Tp, C):
    from fs_eend_amd import ops
    e, w1 = rnd((B * Tp, 256), dev, 26, F16), rnd((256, 256), dev, 27, F16, 0.1)
    pc = rnd((C, 256), dev, 28)
    o32 = torch.full((B * C * Tp, 256), float("nan"), dtype=F32, device=dev)
    o16 = torch.full((B * C * Tp, 256), float("nan"), dtype=F16, device=dev)
    ops.convert_fanout(e, w1, pc, o32, o16, B, Tp, C)
    y = (e.float() @ w1.float().t()).view(B, 1, Tp, 256) + pc.view(1, C, 1, 256)
    close(o32.view(B, C, Tp, 256), y, 2e-4, 2e-4, "convert fan-out f32")
    close(o16.view(B, C, Tp, 256), y, 3e-3, 2e-3, "convert fan-out f16")


@pytest.mark.parametrize("B,Tp,C", [(2, 64, 4), (1, 128, 12), (3, 64, 1)])
def test_convert_fanout_f32_with_remainder(hip_lib, dev, B, Tp, C):
    """f32-operand form (exact-f32 MFMA) with the f16 remainder rows of round 5: out16 + out16lo == out32 to ~2^-22."""
    from fs_eend_amd import ops
    e, w1 = rnd((B * Tp, 256), dev, 36), rnd((256, 256), dev, 37, F32, 0.1)
    pc = rnd((C, 256), dev, 38)
    o32 = torch.full((B * C * Tp, 256), float("nan"), dtype=F32, device=dev)
    o16 = torch.full((B * C * Tp, 256), float("nan"), dtype=F16, device=dev)
    l16 = torch.full((B * C * Tp, 256), float("nan"), dtype=F16, device=dev)
    ops.convert_fanout_f32(e, w1, pc, o32, o16, B, Tp, C, out16lo=l16)
    y = ((e.double() @ w1.double().t()).view(B, 1, Tp, 256) + pc.double().view(1, C, 1, 256)).float()
    close(o32.view(B, C, Tp, 256), y, 2e-5, 2e-5, "convert fan-out f32 operands")
    assert torch.equal(o16, o32.half())
    assert torch.isfinite(l16).all()
    assert ((o16.float() + l16.float()) - o32).abs().max().item() < 2e-6 * max(1.0, o32.abs().max().item())
    o32b, o16b = torch.empty_like(o32), torch.empty_like(o16)
    ops.convert_fanout_f32(e, w1, pc, o32b, o16b, B, Tp, C)                       # without the remainder: same main outputs
    assert torch.equal(o32b, o32) and torch.equal(o16b, o16)


def attn_ref(q, k, v, delay, kv_len):
    """q,k,v (nseq,H,Tp,64) fp32 -> (nseq,Tp,H*64)."""
    Tp = q.shape[2]
    s = q @ k.transpose(-1, -2) / 8.0
    i = torch.arange(Tp, device=q.device)[:, None]
    j = torch.arange(Tp, device=q.device)[None, :]
    allowed = ((j - i) <= delay) & (j < kv_len)
    s = s.masked_fill(~allowed, float("-inf"))
    o = torch.softmax(s, -1) @ v
    return o.permute(0, 2, 1, 3).reshape(q.shape[0], Tp, -1)


@pytest.mark.parametrize("nseq,Tp,delay,kv_len,scale", [(1, 64, 0, 64, 1.0), (2, 128, 0, 128, 1.0), (1, 192, 0, 192, 3.0),
                                                        (2, 512, 0, 512, 1.0), (1, 256, 2, 256, 1.0),
                                                        (1, 128, 128, 100, 1.0), (1, 512, 0, 512, 6.0),
                                                        (1, 1024, 0, 1024, 1.0), (2, 640, 3, 600, 1.0), (3, 384, 0, 384, 1.0)])
def test_attn_causal(hip_lib, dev, nseq, Tp, delay, kv_len, scale):
    from fs_eend_amd import ops
    H = 4
    q = rnd((nseq, H, Tp, 64), dev, 29, BF16, scale)
    k = rnd((nseq, H, Tp, 64), dev, 30, BF16, scale)
    v = rnd((nseq, H, Tp, 64), dev, 31, BF16)
    vt = v.transpose(-1, -2).contiguous()
    o = torch.full((nseq * Tp, 256), float("nan"), dtype=F16, device=dev)
    ops.attn_causal(q.view(-1), k.view(-1), vt.view(-1), o, nseq, H, Tp, delay, kv_len)
    want = attn_ref(q.float(), k.float(), v.float(), delay, kv_len)
    valid = min(Tp, kv_len) if delay >= Tp else Tp       # rows that see >= 1 key are all rows here
    # P is rounded to bf16 before PV: tolerance ~ 2^-8 relative on O(1) values
    close(o.view(nseq, Tp, 256)[:, :valid], want[:, :valid], 1.5e-2, 1e-2, f"attention Tp={Tp} delay={delay}")
    # bit-exact mask indexing: row 0 with delay 0 attends only key 0 -> O[0] == V[0] exactly (p = 1)
    if delay == 0:
        o0 = o.view(nseq, Tp, H, 64)[:, 0]
        assert torch.equal(o0.float(), v[:, :, 0].float().to(F16).float()), "row 0 must equal V[0] exactly"


def test_attn_causality_property(hip_lib, dev):
    """Changing keys/values at frames > t must not change output rows <= t (bit exact)."""
    from fs_eend_amd import ops
    nseq, H, Tp = 1, 4, 256
    q, k, v = (rnd((nseq, H, Tp, 64), dev, s, BF16) for s in (32, 33, 34))
    o1 = torch.empty((Tp, 256), dtype=F16, device=dev)
    o2 = torch.empty((Tp, 256), dtype=F16, device=dev)
    ops.attn_causal(q.view(-1), k.view(-1), v.transpose(-1, -2).contiguous().view(-1), o1, nseq, H, Tp, 0, Tp)
    k2, v2 = k.clone(), v.clone()
    k2[:, :, 150:] = 7.0
    v2[:, :, 150:] = -3.0
    ops.attn_causal(q.view(-1), k2.view(-1), v2.transpose(-1, -2).contiguous().view(-1), o2, nseq, H, Tp, 0, Tp)
    assert torch.equal(o1[:150], o2[:150])
    assert not torch.equal(o1[150:], o2[150:])


@pytest.mark.parametrize("C", [1, 2, 3, 4, 6, 10, 12])
def test_spk_attn(hip_lib, dev, C):
    from fs_eend_amd import ops
    B, Tp, H, D = 2, 64, 4, 256
    M = B * C * Tp
    qkv = rnd((M, 3 * D), dev, 35 + C, F16)
    o = torch.full((M, D), float("nan"), dtype=F16, device=dev)
    ops.spk_attn(qkv, o, B, C, Tp, H)
    x = qkv.float().view(B, C, Tp, 3, H, 64).permute(3, 0, 2, 4, 1, 5)      # (3,B,Tp,H,C,64)
    s = x[0] @ x[1].transpose(-1, -2) / 8.0
    want = (torch.softmax(s, -1) @ x[2]).permute(0, 3, 1, 2, 4).reshape(B, C, Tp, D)   # (B,C,Tp,H*64)
    close(o.view(B, C, Tp, D), want, 3e-3, 3e-3, f"speaker attention C={C}")


def test_bn_cast_pad(hip_lib, dev):
    from fs_eend_amd import ops
    B, T, Tp, Fin, Fpad = 3, 70, 128, 345, 384
    x = rnd((B, T, Fin), dev, 50) * 2 - 3
    w, b = rnd((Fin,), dev, 51) * 0.2 + 1, rnd((Fin,), dev, 52) * 0.1
    mean, var = rnd((Fin,), dev, 53) - 3, rnd((Fin,), dev, 54).abs() + 0.5
    out = torch.full((B * Tp, Fpad), float("nan"), dtype=F16, device=dev)
    ops.bn_cast_pad(x, (w, b, mean, var), out, T, Tp, True, 1e-5)
    want = torch.zeros(B, Tp, Fpad, device=dev)
    want[:, :T, :Fin] = (x - mean) / torch.sqrt(var + 1e-5) * w + b
    close(out.view(B, Tp, Fpad), want, 1e-3, 1e-3, "bn+cast+pad")
    out2 = torch.full((B * Tp, Fpad), float("nan"), dtype=F16, device=dev)
    ops.bn_cast_pad(x, None, out2, T, Tp, False)
    want2 = torch.zeros(B, Tp, Fpad, device=dev)
    want2[:, :T, :Fin] = x
    close(out2.view(B, Tp, Fpad), want2, 1e-3, 1e-3, "cast+pad")


@pytest.mark.parametrize("B,T,Tp,C,D,a16", [(2, 50, 64, 6, 256, False), (1, 7, 64, 3, 256, False), (3, 61, 64, 5, 256, True), (1, 1, 64, 1, 256, True),
                                             (2, 130, 192, 10, 256, False), (2, 33, 64, 4, 128, False), (2, 33, 64, 4, 512, True)])
def test_head_l2dot(hip_lib, dev, B, T, Tp, C, D, a16):
    """rows-in-flight form (d_model 256, four rows per wave, ragged last wave) and the general one-row-per-wave form"""
    from fs_eend_amd import ops
    emb = rnd((B, Tp, D), dev, 55)
    emb = emb / emb.norm(dim=-1, keepdim=True)
    attr = rnd((B * C, Tp, D), dev, 56) * 3
    if a16:
        attr = attr.to(F16)
    ao = torch.full((B, T, C, D), float("nan"), dtype=F32, device=dev)
    lg = torch.full((B, T, C), float("nan"), dtype=F32, device=dev)
    ops.head_l2dot(emb.view(-1, D), attr.view(-1, D), ao, lg, B, T, Tp, C, D)
    attr = attr.float()
    a = attr.view(B, C, Tp, D)[:, :, :T].permute(0, 2, 1, 3)
    a = a / a.norm(dim=-1, keepdim=True)
    close(ao, a, 1e-6, 1e-5, "attractor l2norm")
    close(lg, (emb[:, :T, None, :] * a).sum(-1), 2e-6, 1e-5, "logits")


# ------------------------------------------------------------------------------------ LS-EEND kernels
def test_linear_swish(hip_lib, dev):
    from fs_eend_amd import ops
    a, w, b = rnd((300, 256), dev, 60, F16), rnd((1024, 256), dev, 61, F16, 0.1), rnd((1024,), dev, 62)
    out = torch.empty((300, 1024), dtype=F16, device=dev)
    ops.linear(a, w, b, out, act=ops.ACT_SWISH)
    y = a.float() @ w.float().t() + b
    close(out, y * torch.sigmoid(y), 2e-3, 2e-3, "linear+swish")


def test_linear_glu(hip_lib, dev):
    from fs_eend_amd import ops
    M, D = 200, 256
    a, w, b = rnd((M, D), dev, 63, F16), rnd((2 * D, D), dev, 64, F16, 0.1), rnd((2 * D,), dev, 65)
    wi = torch.stack([w[:D], w[D:]], dim=1).reshape(2 * D, D).contiguous()
    bi = torch.stack([b[:D], b[D:]], dim=1).reshape(2 * D).contiguous()
    out = torch.full((M, D), float("nan"), dtype=F16, device=dev)
    ops.linear_glu(a, wi, bi, out)
    y = a.float() @ w.float().t() + b
    close(out, y[:, :D] * torch.sigmoid(y[:, D:]), 2e-3, 2e-3, "pointwise conv + GLU")


def test_linear_res_scale_ln16(hip_lib, dev):
    from fs_eend_amd import ops
    M, K = 170, 1024
    a, w, b = rnd((M, K), dev, 66, F16), rnd((256, K), dev, 67, F16, 0.05), rnd((256,), dev, 68)
    r = rnd((M, 256), dev, 69)
    g, be = rnd((256,), dev, 70) * 0.2 + 1, rnd((256,), dev, 71) * 0.1
    o32 = torch.empty((M, 256), dtype=F32, device=dev)
    o16 = torch.empty((M, 256), dtype=F16, device=dev)
    ops.linear_res_scale_ln16(a, w, b, r, 0.5, g, be, o32, o16, 1e-5)
    y = (a.float() @ w.float().t() + b) * 0.5 + r
    close(o32, y, 2e-4, 2e-4, "residual stream")
    close(o16, torch.nn.functional.layer_norm(y, (256,), g, be, 1e-5), 3e-3, 2e-3, "LN for the next module")
    # alpha on the LN'd variant (block-final half-step FFN)
    o32b = torch.empty_like(o32)
    ops.linear_res_ln(a, w, b, r, g, be, o32b, o16, 1e-5, alpha=0.5)
    close(o32b, torch.nn.functional.layer_norm(y, (256,), g, be, 1e-5), 2e-4, 2e-4, "alpha + LN")


def test_layernorm_f16(hip_lib, dev):
    from fs_eend_amd import ops
    x = rnd((77, 256), dev, 72) * 3 + 1
    g, be = rnd((256,), dev, 73) * 0.2 + 1, rnd((256,), dev, 74) * 0.1
    out = torch.empty((77, 256), dtype=F16, device=dev)
    ops.layernorm_f16(x, g, be, out, 1e-5)
    close(out, torch.nn.functional.layer_norm(x, (256,), g, be, 1e-5), 3e-3, 2e-3, "layernorm")


@pytest.mark.parametrize("k", [16, 7, 31, 5])
def test_dwconv_bn_swish(hip_lib, dev, k):
    from fs_eend_amd import ops
    nseq, Tp, D = 2, 128, 256
    x = rnd((nseq * Tp, D), dev, 75, F16)
    w = rnd((D, k), dev, 76) * 0.3
    bn = (rnd((D,), dev, 77) * 0.2 + 1, rnd((D,), dev, 78) * 0.1, rnd((D,), dev, 79) * 0.1, rnd((D,), dev, 80).abs() + 0.5)
    out = torch.empty((nseq * Tp, D), dtype=F16, device=dev)
    ops.dwconv_bn_swish(x, w, bn, out, nseq, Tp, 1e-5)
    xi = x.float().view(nseq, Tp, D).transpose(1, 2)
    y = torch.nn.functional.conv1d(torch.nn.functional.pad(xi, (k - 1, 0)), w[:, None, :], groups=D)
    y = (y - bn[2][:, None]) / torch.sqrt(bn[3][:, None] + 1e-5) * bn[0][:, None] + bn[1][:, None]
    y = (y * torch.sigmoid(y)).transpose(1, 2).reshape(nseq * Tp, D)
    close(out, y, 3e-3, 2e-3, f"depthwise conv k={k}")


def _ret_inputs(dev, nseq, Tp, seed, qs=1.0):
    H = 4
    q = rnd((nseq, H, Tp, 64), dev, seed, F16, qs)
    k = rnd((nseq, H, Tp, 64), dev, seed + 1, F16, qs / 8)
    v = rnd((nseq, H, Tp, 64), dev, seed + 2, F16)
    g = rnd((nseq * Tp, 256), dev, seed + 3, F16)
    return q, k, v, g


def _ret_reference(q, k, v, g, L, T):
    """oracle retention (validated against the reference) on the first T frames, + LN + gate."""
    from oracle import ls_eend_ref as R
    nseq, H = q.shape[0], q.shape[1]
    qc, kc = q[:, :, :T].float().cpu(), k[:, :, :T].float().cpu()
    vc = v[:, :, :T].float().cpu().permute(0, 2, 1, 3).reshape(nseq, T, H * 64)
    o = R.retention_chunk(qc, kc, vc, L)                          # (nseq,T,H,64)
    o = R.layer_norm(o, None, None, 1e-6).reshape(nseq, T, 256)
    gg = g.float().cpu().view(nseq, -1, 256)[:, :T]
    return R.swish(gg) * o


@pytest.mark.parametrize("nseq,Tp,L,T,qs", [(2, 128, 64, 128, 1.0), (1, 512, 500, 500, 1.0), (2, 64, 10, 60, 1.0),
                                            (1, 256, 100, 200, 1.0), (1, 192, 32, 192, 3.0), (1, 1024, 500, 1000, 0.3),
                                            (2, 2048, 500, 2000, 1.0), (1, 128, 6, 126, 1.0), (1, 1088, 544, 1088, 1.0)])
def test_retention_chunk(hip_lib, dev, nseq, Tp, L, T, qs):
    from fs_eend_amd import ops
    H = 4
    q, k, v, g = _ret_inputs(dev, nseq, Tp, 81, qs)
    kt = k.transpose(-1, -2).contiguous()
    vt = v.transpose(-1, -2).contiguous()
    nc = (Tp + L - 1) // L
    st = torch.empty(nseq * H * nc * 2 * 4096, dtype=F16, device=dev)
    cs = torch.empty(nseq * H * nc, dtype=F32, device=dev)
    se = torch.empty(nseq * H * nc, dtype=F32, device=dev)
    o = torch.full((nseq * Tp, 256), float("nan"), dtype=F16, device=dev)
    ops.retention_chunk(q.view(-1), k.view(-1), kt.view(-1), vt.view(-1), g, o, st, cs, se, nseq, H, Tp, L)
    want = _ret_reference(q, k, v, g, L, T)
    got = o.view(nseq, Tp, 256)[:, :T].float().cpu()
    assert torch.isfinite(o).all()
    # per-head LayerNorm output is O(1); S is rounded to f16 before S.V
    close(got, want, 2e-2, 1e-2, f"retention L={L} Tp={Tp}")
    # cross_scale of the state before chunk 1 == max(1, max_d sum_k |S|/sqrt(L)) (retention.py:180)
    if nc > 1 and T >= L:
        kv = (k[:, :, :L].float().transpose(-1, -2) @ v[:, :, :L].float()) / (L ** 0.5)     # (nseq,H,64,64)
        ref = kv.abs().sum(dim=-2).max(dim=-1).values.clamp(min=1)
        close(cs.view(nseq, H, nc)[:, :, 1], ref, 1e-3, 1e-3, "cross_scale")


def test_retention_causality(hip_lib, dev):
    """Frames <= t must be bit-identical when K/V/Q change only after t (chunk-local and cross-chunk)."""
    from fs_eend_amd import ops
    nseq, H, Tp, L = 1, 4, 256, 100
    q, k, v, g = _ret_inputs(dev, nseq, Tp, 90)
    nc = (Tp + L - 1) // L
    ws = lambda: (torch.empty(nseq * H * nc * 2 * 4096, dtype=F16, device=dev), torch.empty(nseq * H * nc, dtype=F32, device=dev),
                  torch.empty(nseq * H * nc, dtype=F32, device=dev))
    o1 = torch.empty((Tp, 256), dtype=F16, device=dev)
    o2 = torch.empty((Tp, 256), dtype=F16, device=dev)
    ops.retention_chunk(q.view(-1), k.view(-1), k.transpose(-1, -2).contiguous().view(-1),
                        v.transpose(-1, -2).contiguous().view(-1), g, o1, *ws(), nseq, H, Tp, L)
    k2, v2 = k.clone(), v.clone()
    k2[:, :, 150:] = 0.7
    v2[:, :, 150:] = -2.0
    ops.retention_chunk(q.view(-1), k2.view(-1), k2.transpose(-1, -2).contiguous().view(-1),
                        v2.transpose(-1, -2).contiguous().view(-1), g, o2, *ws(), nseq, H, Tp, L)
    assert torch.equal(o1[:150], o2[:150])
    assert not torch.equal(o1[150:], o2[150:])


def test_retention_proj(hip_lib, dev):
    from fs_eend_amd import ops
    nseq, Tp, H, D = 2, 128, 4, 256
    M = nseq * Tp
    a, w, b = rnd((M, D), dev, 95, F16), rnd((4 * D, D), dev, 96, F16, 0.1), rnd((4 * D,), dev, 97)
    bufs = [torch.full((M * D,), float("nan"), dtype=F16, device=dev) for _ in range(4)]
    g = torch.full((M, D), float("nan"), dtype=F16, device=dev)
    ops.retention_proj(a, w, b, *bufs, g, nseq, Tp, H)
    y = (a.float() @ w.float().t() + b).view(nseq, Tp, 4, H, 64)
    close(bufs[0].view(nseq, H, Tp, 64), y[:, :, 0].permute(0, 2, 1, 3), 3e-3, 2e-3, "Q")
    close(bufs[1].view(nseq, H, Tp, 64), y[:, :, 1].permute(0, 2, 1, 3), 3e-3, 2e-3, "K")
    close(bufs[2].view(nseq, H, 64, Tp), y[:, :, 1].permute(0, 2, 3, 1), 3e-3, 2e-3, "K^T")
    close(bufs[3].view(nseq, H, 64, Tp), y[:, :, 2].permute(0, 2, 3, 1), 3e-3, 2e-3, "V^T")
    close(g, y[:, :, 3].reshape(M, D), 3e-3, 2e-3, "G")


@pytest.mark.parametrize("B,T,Tp,C,masked", [(2, 130, 192, 4, False), (3, 64, 64, 3, True), (1, 500, 512, 6, False), (2, 77, 128, 10, True)])
def test_emb_consistency_loss(hip_lib, dev, B, T, Tp, C, masked):
    """HIP embedding-consistency loss vs the reference formula in torch fp32 (FS model :46-57; LS :92-113)."""
    from fs_eend_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + T)
    emb = torch.randn(B, Tp, 256, generator=g).to(dev)
    emb = emb / emb.norm(dim=-1, keepdim=True)
    lab = (torch.rand(B, T, C, generator=g) > 0.6).float().to(dev)
    lens = [T - 7 * b for b in range(B)]
    e = emb[:, :T].clone()
    if masked:
        for b, l in enumerate(lens):
            e[b, l:] = 0
            lab[b, l:] = 0
    a = e @ e.transpose(-1, -2)
    n = e.norm(dim=-1, keepdim=True)
    a = a / (n @ n.transpose(-1, -2) + 1e-6)
    lm = lab @ lab.transpose(-1, -2)
    tn = lab.norm(dim=-1, keepdim=True)
    lm = lm / (tn @ tn.transpose(-1, -2) + 1e-6)
    if masked:
        want = torch.nn.functional.mse_loss(a, lm, reduction="sum") / sum(l * l for l in lens)
        got = ops.emb_consistency(emb, lab.contiguous(), T, lens=torch.tensor(lens, dtype=torch.int32, device=dev),
                                  inv_count=1.0 / sum(l * l for l in lens))
    else:
        want = torch.nn.functional.mse_loss(a, lm)
        got = ops.emb_consistency(emb, lab.contiguous(), T)
    assert abs(got.item() - want.item()) < 1e-6 + 1e-5 * abs(want.item()), (got.item(), want.item())


@pytest.mark.parametrize("M", [128, 1000, 4133])
def test_f16_residual_variants_equal_the_f32_residual_forms(hip_lib, dev, M):
    """eend_linear_res16_ln_f16 / eend_attnout_ffn_fused_res16_f16 read the residual from the f16 stream: given a residual that is
    exactly representable in f16 they must reproduce the f32-residual entry points bit for bit, with and without the f32 output."""
    from fs_eend_amd import ops
    g = torch.Generator().manual_seed(M)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    a16 = rn(M, 256).half()
    res16 = rn(M, 256).half()
    res32 = res16.float()
    wo, bo = rn(256, 256, sc=0.06).half(), rn(256, sc=0.1)
    g1, be1 = rn(256, sc=0.2) + 1.0, rn(256, sc=0.1)
    o32a, o16a = torch.empty(M, 256, device=dev), torch.empty(M, 256, dtype=torch.float16, device=dev)
    o32b, o16b, o16c = torch.empty_like(o32a), torch.empty_like(o16a), torch.empty_like(o16a)
    ops.linear_res_ln(a16, wo, bo, res32, g1, be1, o32a, o16a, 1e-5)
    ops.linear_res16_ln(a16, wo, bo, res16, g1, be1, o32b, o16b, 1e-5)
    ops.linear_res16_ln(a16, wo, bo, res16, g1, be1, None, o16c, 1e-5)
    assert torch.equal(o32a, o32b) and torch.equal(o16a, o16b) and torch.equal(o16a, o16c)
    w1, b1 = rn(2048, 256, sc=0.08).half(), rn(2048, sc=0.3)
    w2, b2 = rn(256, 2048, sc=0.03).half(), rn(256, sc=0.2)
    g2, be2 = rn(256, sc=0.2) + 1.0, rn(256, sc=0.1)
    ops.attnout_ffn_fused(a16, wo, bo, res32, g1, be1, 1e-5, w1, b1, w2, b2, g2, be2, 1e-5, o32a, o16a)
    ops.attnout_ffn_fused_res16(a16, wo, bo, res16, g1, be1, 1e-5, w1, b1, w2, b2, g2, be2, 1e-5, o32b, o16b)
    ops.attnout_ffn_fused_res16(a16, wo, bo, res16, g1, be1, 1e-5, w1, b1, w2, b2, g2, be2, 1e-5, None, o16c)
    assert torch.equal(o32a, o32b) and torch.equal(o16a, o16b) and torch.equal(o16a, o16c)
    inplace = res16.clone()                      # the model's use: residual and f16 output are the same buffer
    ops.attnout_ffn_fused_res16(a16, wo, bo, inplace, g1, be1, 1e-5, w1, b1, w2, b2, g2, be2, 1e-5, None, inplace)
    assert torch.equal(inplace, o16a)
