"""GPU: fused FFN kernel (linear1 + act + linear2 + residual + LayerNorm, hidden stays on chip)
against plain torch fp32 on the same f16-quantised operands, and against the two-launch path."""
import pytest
import torch

pytestmark = pytest.mark.gpu
F16, F32 = torch.float16, torch.float32


def rnd(shape, dev, seed, dtype=F32, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


@pytest.mark.parametrize("M,Fh,act,alpha,unnorm", [(128, 2048, 1, 1.0, False), (300, 1024, 2, 0.5, True),
                                                   (1, 256, 1, 1.0, False), (1000, 2048, 1, 1.0, False),
                                                   (77, 1024, 2, 0.5, False)])
def test_ffn_fused(hip_lib, dev, M, Fh, act, alpha, unnorm):
    from fs_eend_amd import ops
    x = rnd((M, 256), dev, 1, F16)
    w1, b1 = rnd((Fh, 256), dev, 2, F16, 0.08), rnd((Fh,), dev, 3) * 0.3
    w2, b2 = rnd((256, Fh), dev, 4, F16, 0.04), rnd((256,), dev, 5) * 0.3
    res = rnd((M, 256), dev, 6)
    g, be = rnd((256,), dev, 7) * 0.2 + 1, rnd((256,), dev, 8) * 0.1
    o32 = torch.full((M, 256), float("nan"), dtype=F32, device=dev)
    o16 = torch.full((M, 256), float("nan"), dtype=F16, device=dev)
    ops.ffn_fused(x, w1, b1, w2, b2, res, g, be, o32, o16, act, alpha, 1e-5, residual_unnormalised=unnorm)
    h = x.float() @ w1.float().t() + b1
    h = h.relu() if act == 1 else h * torch.sigmoid(h)
    h = h.to(F16).float()                                  # the hidden activations are f16 MFMA operands
    y = (h @ w2.float().t() + b2) * alpha + res
    ln = torch.nn.functional.layer_norm(y, (256,), g, be, 1e-5)
    assert torch.isfinite(o32).all() and torch.isfinite(o16).all()
    e32 = (o32 - (y if unnorm else ln)).abs().max().item()
    e16 = (o16.float() - ln).abs().max().item()
    assert e32 < 2e-3 and e16 < 5e-3, (e32, e16)
    # same arithmetic as the two-launch path up to the position of the bias in the fp32 sum (the fused
    # kernel seeds the accumulator with b1, the GEMM epilogue adds it last): a hidden unit can round to
    # the neighbouring f16, which moves an output by <= ulp_f16(h) * |w2| ~ 1e-4
    ff = torch.empty((M, Fh), dtype=F16, device=dev)
    p32, p16 = torch.empty_like(o32), torch.empty_like(o16)
    ops.linear(x, w1, b1, ff, act=act)
    if unnorm:
        ops.linear_res_scale_ln16(ff, w2, b2, res, alpha, g, be, p32, p16, 1e-5)
    else:
        ops.linear_res_ln(ff, w2, b2, res, g, be, p32, p16, 1e-5, alpha=alpha)
    assert (o32 - p32).abs().max().item() < 5e-4
    assert (o16.float() - p16.float()).abs().max().item() < 2e-3


def test_ffn_fused_inplace_residual(hip_lib, dev):
    from fs_eend_amd import ops
    M, Fh = 260, 2048
    x = rnd((M, 256), dev, 11, F16)
    w1, b1 = rnd((Fh, 256), dev, 12, F16, 0.08), rnd((Fh,), dev, 13) * 0.3
    w2, b2 = rnd((256, Fh), dev, 14, F16, 0.04), rnd((256,), dev, 15) * 0.3
    res = rnd((M, 256), dev, 16)
    g, be = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    want32, want16 = torch.empty_like(res), torch.empty((M, 256), dtype=F16, device=dev)
    ops.ffn_fused(x, w1, b1, w2, b2, res, g, be, want32, want16)
    x2 = x.clone()
    ops.ffn_fused(x2, w1, b1, w2, b2, res, g, be, res, x2)      # out32 aliases res, out16 aliases x (as the model calls it)
    assert torch.equal(res, want32) and torch.equal(x2, want16)


@pytest.mark.parametrize("M,unnorm,alpha", [(128, False, 1.0), (300, True, 1.0), (1, False, 1.0), (70000, False, 1.0)])
def test_proj256_ln_matches_generic_gemm(hip_lib, dev, M, unnorm, alpha):
    """K = 256 out-proj + residual + LN goes to the X-resident 128-row kernel (ffn.hip PROJ mode):
    check against torch fp32."""
    from fs_eend_amd import ops
    a, w, b = rnd((M, 256), dev, 21, F16), rnd((256, 256), dev, 22, F16, 0.1), rnd((256,), dev, 23)
    r = rnd((M, 256), dev, 24)
    g, be = rnd((256,), dev, 25) * 0.2 + 1, rnd((256,), dev, 26) * 0.1
    o32 = torch.full((M, 256), float("nan"), dtype=F32, device=dev)
    o16 = torch.full((M, 256), float("nan"), dtype=F16, device=dev)
    if unnorm:
        ops.linear_res_scale_ln16(a, w, b, r, alpha, g, be, o32, o16, 1e-5)
    else:
        ops.linear_res_ln(a, w, b, r, g, be, o32, o16, 1e-5, alpha=alpha)
    y = (a.float() @ w.float().t() + b) * alpha + r
    ln = torch.nn.functional.layer_norm(y, (256,), g, be, 1e-5)
    assert (o32 - (y if unnorm else ln)).abs().max().item() < 3e-4
    assert (o16.float() - ln).abs().max().item() < 4e-3


@pytest.mark.parametrize("M,Fh", [(128, 2048), (1000, 2048), (77, 1024), (40000, 2048)])
def test_attnout_ffn_fused(hip_lib, dev, M, Fh):
    """out-proj + residual + norm1 + FFN + residual + norm2 in one launch == linear_res_ln followed by
    ffn_fused (same MFMA k order and the same f16 rounding of x), and both match torch fp32."""
    from fs_eend_amd import ops
    a = rnd((M, 256), dev, 21, F16)
    wo, bo = rnd((256, 256), dev, 22, F16, 0.06), rnd((256,), dev, 23) * 0.2
    w1, b1 = rnd((Fh, 256), dev, 24, F16, 0.08), rnd((Fh,), dev, 25) * 0.3
    w2, b2 = rnd((256, Fh), dev, 26, F16, 0.04), rnd((256,), dev, 27) * 0.3
    res = rnd((M, 256), dev, 28)
    g1, be1 = rnd((256,), dev, 29) * 0.2 + 1, rnd((256,), dev, 30) * 0.1
    g2, be2 = rnd((256,), dev, 31) * 0.2 + 1, rnd((256,), dev, 32) * 0.1
    o32 = torch.full((M, 256), float("nan"), dtype=F32, device=dev)
    o16 = torch.full((M, 256), float("nan"), dtype=F16, device=dev)
    ops.attnout_ffn_fused(a, wo, bo, res, g1, be1, 1e-5, w1, b1, w2, b2, g2, be2, 1e-5, o32, o16)
    # torch fp32 on the same f16-quantised operands
    x = torch.nn.functional.layer_norm(a.float() @ wo.float().t() + bo + res, (256,), g1, be1, 1e-5)
    h = (x.to(F16).float() @ w1.float().t() + b1).relu().to(F16).float()
    want = torch.nn.functional.layer_norm(h @ w2.float().t() + b2 + x, (256,), g2, be2, 1e-5)
    assert torch.isfinite(o32).all() and torch.isfinite(o16).all()
    assert (o32 - want).abs().max().item() < 3e-3
    assert (o16.float() - want).abs().max().item() < 6e-3
    # two-launch path
    x32, x16 = torch.empty_like(o32), torch.empty_like(o16)
    p32, p16 = torch.empty_like(o32), torch.empty_like(o16)
    ops.linear_res_ln(a, wo, bo, res, g1, be1, x32, x16, 1e-5)
    ops.ffn_fused(x16, w1, b1, w2, b2, x32, g2, be2, p32, p16)
    assert (o32 - p32).abs().max().item() < 3e-3          # f16 roundings of x / h may flip: fp32 sum order differs
    # in place, as the model calls it: out32 aliases res, out16 aliases a
    res2, a2 = res.clone(), a.clone()
    ops.attnout_ffn_fused(a2, wo, bo, res2, g1, be1, 1e-5, w1, b1, w2, b2, g2, be2, 1e-5, res2, a2)
    assert torch.equal(res2, o32) and torch.equal(a2, o16)


@pytest.mark.parametrize("M,Fh", [(128, 2048), (1500, 2048), (77, 1024), (20011, 2048)])
def test_attnout_ffn_stream_lo_matches_unpacked(hip_lib, dev, M, Fh):
    """Round 6: the LO form of the packed-stream layer tail (ffn_stream.hip: Wo hi / lo pair in the stream, f32 + f16 hi / lo rows out) ==
    eend_attnout_ffn_fused_f16 with wo_lo / out16lo on un-packed weights (the launch it replaces in the LS-EEND decoder), up to f32
    summation order; remainder rows exact; in place on the residual stream; optional lo output."""
    from fs_eend_amd import ops
    a = rnd((M, 256), dev, 61, F16)
    wo32 = rnd((256, 256), dev, 62, F32, 0.06)
    wo, bo = wo32.half(), rnd((256,), dev, 63) * 0.2
    wlo = (wo32 - wo.float()).half()
    w1, b1 = rnd((Fh, 256), dev, 64, F16, 0.08), rnd((Fh,), dev, 65) * 0.3
    w2, b2 = rnd((256, Fh), dev, 66, F16, 0.04), rnd((256,), dev, 67) * 0.3
    res = rnd((M, 256), dev, 68)
    g1, be1 = rnd((256,), dev, 69) * 0.2 + 1, rnd((256,), dev, 70) * 0.1
    g2, be2 = rnd((256,), dev, 71) * 0.2 + 1, rnd((256,), dev, 72) * 0.1
    nan = lambda dt: torch.full((M, 256), float("nan"), dtype=dt, device=dev)
    p32, p16, pl = nan(F32), nan(F16), nan(F16)
    ops.attnout_ffn_fused(a, wo, bo, res, g1, be1, 1e-5, w1, b1, w2, b2, g2, be2, 1e-5, p32, p16, out16lo=pl, wo_lo=wlo)
    ws = ops.ffn_stream_pack_lo(wo, wlo, w1, w2)
    o32, o16, ol = nan(F32), nan(F16), nan(F16)
    ops.attnout_ffn_stream_lo(a, ws, bo, res, g1, be1, 1e-5, b1, b2, g2, be2, 1e-5, o32, o16, ol)
    for t in (o32, o16, ol):
        assert torch.isfinite(t).all()
    assert (o32 - p32).abs().max().item() < 3e-3                 # f16 roundings of x / h may flip: fp32 sum order differs
    assert ((o16.float() + ol.float()) - o32).abs().max().item() < 2e-6 * max(1.0, o32.abs().max().item())
    # against float64 on the f32 out-projection weight
    x = torch.nn.functional.layer_norm(a.double() @ wo32.double().t() + bo.double() + res.double(), (256,), g1.double(), be1.double(), 1e-5).float()
    h = (x.to(F16).float() @ w1.float().t() + b1).relu().to(F16).float()
    full = torch.nn.functional.layer_norm(h @ w2.float().t() + b2 + x, (256,), g2, be2, 1e-5)
    assert (o32 - full).abs().max().item() < 3e-3
    # in place on the f32 stream (out32 = res, out16 = a), without the lo output: the same rows
    res2, a2 = res.clone(), a.clone()
    ops.attnout_ffn_stream_lo(a2, ws, bo, res2, g1, be1, 1e-5, b1, b2, g2, be2, 1e-5, res2, a2, None)
    assert torch.equal(res2, o32) and torch.equal(a2, o16)
    # the split weight matters: the single-product stream differs from the f32-weight result by the f16 weight rounding
    z2, zb2 = torch.zeros_like(w2), torch.zeros_like(b2)
    one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    s32, s16 = nan(F32), nan(F16)
    ops.attnout_ffn_stream_lo(a, ops.ffn_stream_pack_lo(wo, wlo, w1, z2), bo, res, g1, be1, 1e-5, b1, zb2, one, zero, 1e-5, s32, s16, None)
    want = torch.nn.functional.layer_norm(x.double(), (256,), None, None, 1e-5).float()
    assert (s32 - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("M,Fh", [(128, 2048), (1500, 2048), (77, 1024)])
def test_attnout_ffn_fused_split_weight_and_remainder(hip_lib, dev, M, Fh):
    """Round 5 (LS-EEND decoder): wo_lo = the f16 remainder of the f32 out-projection weight (second MFMA product in the
    same launch) brings x = LN1(a Wo^T + ...) to the f32-weight result; out16lo = the f16 remainder of the f32 output rows, so
    out16 + out16lo carries the row to ~2^-21 for the next layer's query path."""
    from fs_eend_amd import ops
    a = rnd((M, 256), dev, 41, F16)
    wo32 = rnd((256, 256), dev, 42, F32, 0.06)
    wo, bo = wo32.half(), rnd((256,), dev, 43) * 0.2
    wlo = (wo32 - wo.float()).half()
    w1, b1 = rnd((Fh, 256), dev, 44, F16, 0.08), rnd((Fh,), dev, 45) * 0.3
    w2, b2 = rnd((256, Fh), dev, 46, F16, 0.04), rnd((256,), dev, 47) * 0.3
    res = rnd((M, 256), dev, 48)
    g1, be1 = rnd((256,), dev, 49) * 0.2 + 1, rnd((256,), dev, 50) * 0.1
    g2, be2 = rnd((256,), dev, 51) * 0.2 + 1, rnd((256,), dev, 52) * 0.1
    # the FFN half is switched off (zero W2 / b2, unit norm2 skipped by comparing through it) to expose x: out = LN2(x)
    z2, zb2 = torch.zeros_like(w2), torch.zeros_like(b2)
    one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    outs = {}
    for tag, lo in (("hi", None), ("split", wlo)):
        o32 = torch.full((M, 256), float("nan"), dtype=F32, device=dev)
        o16 = torch.full((M, 256), float("nan"), dtype=F16, device=dev)
        l16 = torch.full((M, 256), float("nan"), dtype=F16, device=dev)
        ops.attnout_ffn_fused(a, wo, bo, res, g1, be1, 1e-5, w1, b1, z2, zb2, one, zero, 1e-5, o32, o16, out16lo=l16, wo_lo=lo)
        assert torch.isfinite(o32).all() and torch.isfinite(o16).all() and torch.isfinite(l16).all()
        # remainder rows: out16 + out16lo == out32 to f16-of-remainder precision
        assert ((o16.float() + l16.float()) - o32).abs().max().item() < 2e-6 * max(1.0, o32.abs().max().item())
        outs[tag] = o32
    x64 = torch.nn.functional.layer_norm(a.double() @ wo32.double().t() + bo.double() + res.double(), (256,), g1.double(), be1.double(), 1e-5)
    want = torch.nn.functional.layer_norm(x64, (256,), None, None, 1e-5).float()
    e_hi = (outs["hi"] - want).abs().max().item()
    e_split = (outs["split"] - want).abs().max().item()
    assert e_split < 2e-5, e_split                       # f32 accumulation noise only
    assert e_hi > 4 * e_split, (e_hi, e_split)           # the f16 weight rounding was the error of the single product
    # full layer with the split weight against float64 on the f32 weight (h rounds to f16 as an MFMA operand: same bar as above)
    o32 = torch.empty((M, 256), dtype=F32, device=dev)
    o16 = torch.empty((M, 256), dtype=F16, device=dev)
    ops.attnout_ffn_fused(a, wo, bo, res, g1, be1, 1e-5, w1, b1, w2, b2, g2, be2, 1e-5, o32, o16, wo_lo=wlo)
    x = x64.float()
    h = (x.to(F16).float() @ w1.float().t() + b1).relu().to(F16).float()
    full = torch.nn.functional.layer_norm(h @ w2.float().t() + b2 + x, (256,), g2, be2, 1e-5)
    assert (o32 - full).abs().max().item() < 3e-3
