"""GPU: the row / channel kernels of the LS-EEND training step (csrc/ls_train.hip, the train outputs of
csrc/retention_full.hip) one by one, each against a plain fp64 torch restatement of the same reference op
(torch autograd where a backward is tested).  Whole-step parity with the reference is tests/test_train_step_ls.py."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
D = 256


def _call(name, *a):
    from fs_eend_amd.train import _call as c
    return c(name, *a)


def _ws(dev):
    return torch.empty(8 * 1024 * 1024, dtype=torch.float32, device=dev)


def _spec(p, seed=77):
    from fs_eend_amd import lib as L
    return L.Dropout(seed, int(round(p * (1 << 24))), 1.0 / (1.0 - p))


def test_swish_dropout_forward_backward(hip_lib, dev):
    M, Fh = 777, 1024
    g = torch.Generator().manual_seed(1)
    z = torch.randn(M, Fh, generator=g).to(torch.float16).to(dev)
    a = torch.empty_like(z)
    _call("eend_swish_dropout_f16", z, a, M, Fh, None)
    zz = z.double()
    assert float((a.double() - zz * torch.sigmoid(zz)).abs().max()) < 2e-3
    spec = _spec(0.25)
    ad = torch.empty_like(z)
    _call("eend_swish_dropout_f16", z, ad, M, Fh, ctypes.byref(spec))
    keep = ad != 0
    assert abs(float(keep.float().mean()) - 0.75) < 5e-3
    assert float((ad.double()[keep] - (a.double()[keep] / 0.75)).abs().max()) < 4e-3
    da = (torch.randn(M, Fh, generator=g) * 1e-4).to(torch.bfloat16).to(dev)
    dz = da.clone()
    _call("eend_swish_bwd_bf16", dz, z, M, Fh, ctypes.byref(spec))
    s = torch.sigmoid(zz)
    want = da.double() * keep.double() / 0.75 * (s * (1 + zz * (1 - s)))
    assert float((dz.double() - want).abs().max()) <= 8e-3 * float(want.abs().max())


def test_layernorm_train_and_general_backward(hip_lib, dev):
    M = 1029
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(M, D, generator=g) * 3 + 1).to(dev)
    gam, bet = (1 + 0.2 * torch.randn(D, generator=g)).to(dev), (0.1 * torch.randn(D, generator=g)).to(dev)
    y16, xh16 = (torch.empty(M, D, dtype=torch.float16, device=dev) for _ in range(2))
    rstd = torch.empty(M, device=dev)
    _call("eend_layernorm_train_f16", x, gam, bet, 1e-5, y16, xh16, rstd, M)
    want = F.layer_norm(x.double(), (D,), gam.double(), bet.double(), 1e-5)
    assert float((y16.double() - want).abs().max()) < 4e-3
    xd = x.double().requires_grad_(True)
    gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    out = F.layer_norm(xd, (D,), gd, bd, 1e-5)
    for g_is_bf16 in (False, True):
        go = torch.randn(M, D, generator=g) * 1e-3
        gin = go.to(torch.bfloat16).to(dev) if g_is_bf16 else go.to(dev)
        gx, ggam, gbet = torch.autograd.grad(out, [xd, gd, bd], gin.double(), retain_graph=True)
        res0 = torch.randn(M, D, generator=g).to(dev) * 1e-3
        ds32 = res0.clone()
        ds16 = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        dg, db, dbias = (torch.empty(D, device=dev) for _ in range(3))
        _call("eend_layernorm_bwd2_f32", gin, 1 if g_is_bf16 else 0, xh16, rstd, gam, ds32, 1, ds16, 0.5, _ws(dev), 8 * 1024 * 1024, dg, db,
              dbias, M, None)
        tol = 3e-3 * float(gx.abs().max())
        assert float((ds32.double() - res0.double() - gx).abs().max()) < tol                    # accumulated into the residual stream
        assert float((ds16.double() - 0.5 * gx).abs().max()) < 0.5 * tol + 4e-3 * float(gx.abs().max())
        assert float((dg.double() - ggam).abs().max()) < 2e-3 * float(ggam.abs().max())
        assert float((db.double() - gbet).abs().max()) < 1e-4 * float(gbet.abs().max()) + 1e-9
        assert float((dbias.double() - 0.5 * gx.sum(0)).abs().max()) < 2e-3 * float((0.5 * gx).abs().sum(0).max())


def test_resgrad_cast(hip_lib, dev):
    M = 2051
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(M, D, generator=g) * 1e-4).to(dev)
    spec = _spec(0.1, 5)
    ds16 = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    dbias = torch.empty(D, device=dev)
    _call("eend_resgrad_cast_bf16", x, ds16, 0.5, _ws(dev), 8 * 1024 * 1024, dbias, M, ctypes.byref(spec))
    keep = ds16 != 0
    assert abs(float(keep.float().mean()) - 0.9) < 5e-3
    want = 0.5 * x.double() / 0.9
    assert float((ds16.double()[keep] - want[keep]).abs().max()) < 5e-3 * float(want.abs().max())
    # the column sums are taken over the f32 values before the bf16 rounding of ds16
    assert float((dbias.double() - (want * keep).sum(0)).abs().max()) < 1e-4 * float((want * keep).abs().sum(0).max())


@pytest.mark.parametrize("nseq,Tv,k", [(3, 300, 16), (2, 500, 16), (2, 130, 7)])
def test_conv_module_train_forward_backward(hip_lib, dev, nseq, Tv, k):
    """GLU -> causal depthwise conv -> BatchNorm (batch statistics over the nseq*Tv valid frames, 2-rank merge) -> swish,
    forward and backward, against fp64 autograd over the same f16-rounded input."""
    from fs_eend_amd import ops
    from oracle import bn_sync_ref as BR
    Tp = ops.frames_pad(Tv)
    M = nseq * Tp
    g = torch.Generator().manual_seed(nseq * 100 + Tv)
    P = torch.randn(M, 2 * D, generator=g).to(torch.float16).to(dev)
    w = (torch.randn(D, k, generator=g) * 0.3).to(dev)
    gam, bet = (1 + 0.2 * torch.randn(D, generator=g)).to(dev), (0.1 * torch.randn(D, generator=g)).to(dev)
    c16 = torch.full((M, D), float("nan"), dtype=torch.float16, device=dev)
    _call("eend_glu_dwconv_f16", P, w, c16, nseq, Tp, Tv, k)
    # fp64 restatement on the valid frames
    Pd = P.double().view(nseq, Tp, 2 * D)[:, :Tv].clone().requires_grad_(True)
    wd = w.double().clone().requires_grad_(True)
    gd, bd = gam.double().clone().requires_grad_(True), bet.double().clone().requires_grad_(True)
    u = Pd[..., :D] * torch.sigmoid(Pd[..., D:])
    c = F.conv1d(F.pad(u.transpose(1, 2), (k - 1, 0)), wd[:, None, :], groups=D).transpose(1, 2)        # (nseq, Tv, D)
    got_c = c16.view(nseq, Tp, D)
    assert float((got_c[:, :Tv].double() - c.detach()).abs().max()) < 4e-3 * float(c.abs().max())
    assert float(got_c[:, Tv:].abs().max()) == 0.0 if Tp > Tv else True
    stats = torch.empty(513, device=dev)
    _call("eend_bn_batch_stats_f16", c16, _ws(dev), 8 * 1024 * 1024, stats, nseq, Tp, Tv)
    flat = got_c[:, :Tv].double().reshape(-1, D)
    ref = BR.local_stats(flat.cpu())
    assert float((stats.cpu().double() - ref).abs().max()) < 1e-3 * float(ref[:512].abs().max()) and stats[512].item() == nseq * Tv
    # merge with a second (synthetic) rank, and alone
    other = BR.local_stats(torch.randn(333, D, generator=g) * 2 + 0.5).float()
    table = torch.stack([stats.cpu(), other]).to(dev)
    mean, var, nout = torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(1, device=dev)
    rm, rv = torch.zeros(D, device=dev), torch.ones(D, device=dev)
    _call("eend_bn_merge_f32", table, 2, mean, var, nout, rm, rv, 0.1)
    m_ref, v_ref, n_ref, vu_ref = BR.merge(table.cpu())
    assert float((mean.cpu().double() - m_ref).abs().max()) < 1e-5 and float((var.cpu().double() - v_ref).abs().max()) < 1e-4
    assert nout.item() == n_ref
    assert float((rm.cpu().double() - 0.1 * m_ref).abs().max()) < 1e-5 and float((rv.cpu().double() - (0.9 + 0.1 * vu_ref)).abs().max()) < 1e-4
    _call("eend_bn_merge_f32", stats.view(1, -1), 1, mean, var, nout, None, None, 0.1)
    mu, vr = flat.mean(0), flat.var(0, unbiased=False)
    assert float((mean.double() - mu).abs().max()) < 1e-4 and float((var.double() - vr).abs().max()) < 1e-3 * float(vr.max())
    s16 = torch.empty(M, D, dtype=torch.float16, device=dev)
    _call("eend_bn_swish_f16", c16, mean, var, 1e-5, gam, bet, s16, M)
    cm, cv = c.mean((0, 1)), c.var((0, 1), unbiased=False)
    y = gd * (c - cm) / torch.sqrt(cv + 1e-5) + bd
    s = y * torch.sigmoid(y)
    assert float((s16.view(nseq, Tp, D)[:, :Tv].double() - s.detach()).abs().max()) < 1e-2
    # backward
    ds = (torch.randn(nseq, Tp, D, generator=g) * 1e-3).to(torch.bfloat16).to(dev)
    gP, gw, gg, gb = torch.autograd.grad(s, [Pd, wd, gd, bd], ds[:, :Tv].double())
    sums, dgam, dbet = torch.empty(512, device=dev), torch.empty(D, device=dev), torch.empty(D, device=dev)
    dsw = ds.clone().view(M, D)
    _call("eend_bn_swish_bwd_stats_bf16", dsw, c16, mean, var, 1e-5, gam, bet, _ws(dev), 8 * 1024 * 1024, sums, dgam, dbet, nseq, Tp, Tv)
    assert float((dgam.double() - gg).abs().max()) < 5e-3 * float(gg.abs().max())
    assert float((dbet.double() - gb).abs().max()) < 5e-3 * float(gb.abs().max())
    _call("eend_bn_swish_bwd_apply_bf16", dsw, c16, mean, var, 1e-5, gam, bet, sums, nout, nseq, Tp, Tv)
    dP = torch.full((M, 2 * D), float("nan"), dtype=torch.bfloat16, device=dev)
    dw = torch.empty(D, k, device=dev)
    _call("eend_dwconv_glu_bwd_bf16", dsw, P, w, dP, _ws(dev), 8 * 1024 * 1024, dw, nseq, Tp, Tv, k)
    got = dP.view(nseq, Tp, 2 * D)
    err = float((got[:, :Tv].double() - gP).norm() / gP.norm())
    assert err < 8e-3, err
    assert float((got[:, :Tv].double() - gP).abs().max()) < 3e-2 * float(gP.abs().max())
    if Tp > Tv:
        assert float(got[:, Tv:].float().abs().max()) == 0.0
    assert float((dw.double() - gw).norm() / gw.norm()) < 5e-3


def test_retention_train_outputs_consistent_with_the_forward(hip_lib, dev):
    """The extra outputs of the training forward: ctx = swish(g) * rhat (what the inference kernel writes), rhat is a
    per-head normalised row, and rc = rstd * detached scale reproduces d out / d (q . prefix) numerically."""
    from fs_eend_amd import ops
    from oracle import ls_eend_ref as R
    nseq, H, L, Tv = 3, 4, 100, 300
    Tp = ops.frames_pad(Tv)
    M = nseq * Tp
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, D, generator=g).to(torch.float16).to(dev)
    w = (torch.randn(4 * D, D, generator=g) * 0.08).to(torch.float16).to(dev)
    b = (torch.randn(4 * D, generator=g) * 0.05).to(dev)
    f16 = lambda *s: torch.empty(*s, dtype=torch.float16, device=dev)
    q, k, kt, vt = (f16(M * D) for _ in range(4))
    gate = f16(M, D)
    ops.retention_proj(x, w, b, q, k, kt, vt, gate, nseq, Tp, H)
    nc = (Tp + L - 1) // L                                 # the inference wrapper sizes its workspace for every slab chunk
    st, cs, se = f16(nseq * H * nc * 2 * 4096), torch.empty(nseq * H * nc, device=dev), torch.empty(nseq * H * nc, device=dev)
    kv = torch.empty(nseq * H * nc * 4096, device=dev)
    ctx, rhat, rc = torch.zeros(M, D, dtype=torch.float16, device=dev), torch.zeros(M, D, dtype=torch.float16, device=dev), torch.zeros(M, H, device=dev)
    _call("eend_retention_chunk_train_f16", q, k, kt, vt, gate, ctx, rhat, rc, st, kv, cs, se, nseq, H, Tp, L, D, D, 1e-6, Tv)
    o_inf = torch.zeros(M, D, dtype=torch.float16, device=dev)
    ops.retention_chunk(q, k, kt, vt, gate, o_inf, st, cs, se, nseq, H, Tp, L, 1e-6, t_valid=Tv)
    assert torch.equal(ctx, o_inf)                                              # same forward values as the inference entry
    gg = gate.double()
    v = slice(0, Tv)
    rh = rhat.view(nseq, Tp, D)[:, v].double()
    assert float((ctx.view(nseq, Tp, D)[:, v].double() - gg.view(nseq, Tp, D)[:, v] * torch.sigmoid(gg.view(nseq, Tp, D)[:, v]) * rh).abs().max()) < 1.5e-2
    rh4 = rh.view(nseq, Tv, H, 64)
    assert float(rh4.mean(-1).abs().max()) < 2e-3 and float(((rh4 ** 2).mean(-1) - 1).abs().max()) < 2e-2
    # oracle: out = retention_chunk(q, k, v); rhat = LN(out); rc = rstd(out) * out / (q . prefix-state)
    qh = q.view(nseq, H, Tp, 64)[:, :, v].double().cpu()
    kh = k.view(nseq, H, Tp, 64)[:, :, v].double().cpu()
    vh = vt.view(nseq, H, 64, Tp)[:, :, :, v].double().cpu().permute(0, 3, 1, 2).reshape(nseq, Tv, H * 64)
    out = R.retention_chunk(qh, kh, vh, L)                                       # (nseq, Tv, H, 64)
    mu, var = out.mean(-1, keepdim=True), out.var(-1, unbiased=False, keepdim=True)
    assert float((rh4.cpu() - (out - mu) / torch.sqrt(var + 1e-6)).abs().max()) < 2e-2
    causal = torch.tril(torch.ones(Tv, Tv, dtype=torch.float64))
    raw = ((qh @ kh.transpose(-1, -2)) * causal) @ vh.view(nseq, Tv, H, 64).transpose(1, 2)       # (nseq, H, Tv, 64)
    c_t = (out.transpose(1, 2) * raw).sum(-1) / (raw * raw).sum(-1)                              # least-squares scalar per row
    want_rc = (c_t / torch.sqrt(var.squeeze(-1).transpose(1, 2) + 1e-6)).transpose(1, 2)        # (nseq, Tv, H)
    got_rc = rc.view(nseq, Tp, H)[:, v].double().cpu()
    assert float(((got_rc - want_rc) / want_rc).abs().max()) < 1e-2


@pytest.mark.parametrize("M,Fh,prenorm,pdrop", [(384, 1024, 1, 0.0), (777, 1024, 1, 0.25), (1000, 1024, 0, 0.25), (70001, 1024, 0, 0.0), (130, 64, 1, 0.1)])
def test_ffn_swish_train_fused(hip_lib, dev, M, Fh, prenorm, pdrop):
    """eend_ffn_swish_train_f16 (ffn.hip MODE 3 with Swish, round 5): the Macaron half-step FFN in one launch == eend_linear_f16 +
    eend_swish_dropout_f16 + eend_linear_res_scale_ln_train_f16 (pre-norm join) / eend_linear_res_ln_train_f16 (block-final LayerNorm):
    saved pre-activation z and dropped activation a, residual stream, LayerNorm output, normalised rows, 1/sigma."""
    from fs_eend_amd import ops
    g = torch.Generator().manual_seed(5)
    r = lambda *s_, sc=1.0: (torch.randn(*s_, generator=g) * sc).to(dev)
    x = r(M, 256).half()
    w1, b1 = r(Fh, 256, sc=1 / 16).half(), r(Fh, sc=0.1)
    w2, b2 = r(256, Fh, sc=1 / math.sqrt(Fh)).half(), r(256, sc=0.1)
    res = r(M, 256)
    gm, be = 1 + r(256, sc=0.2), r(256, sc=0.1)
    s1, s2 = (_spec(pdrop, 11), _spec(pdrop, 12)) if pdrop else (None, None)
    r1, r2 = (ctypes.byref(s1), ctypes.byref(s2)) if pdrop else (None, None)
    nan = lambda *sh, dt=torch.float16: torch.full(sh, float("nan"), dtype=dt, device=dev)
    o32, o16, z, a, xh, rs = nan(M, 256, dt=torch.float32), nan(M, 256), nan(M, Fh), nan(M, Fh), nan(M, 256), nan(M, dt=torch.float32)
    _call("eend_ffn_swish_train_f16", x, 256, w1, b1, w2, b2, res, 0.5, gm, be, 1e-5, o32, o16, z, a, xh, rs, M, Fh, prenorm, r1, r2)
    torch.cuda.synchronize()
    for t in (o32, o16, z, a, xh, rs):
        assert torch.isfinite(t).all()
    # torch on the saved z (what the backward differentiates at)
    zr = x.float() @ w1.float().t() + b1
    assert float((z.float() - zr).abs().max()) < 4e-3
    sw = z.float() * torch.sigmoid(z.float())
    keep = (a != 0) | (sw.abs() < 1e-4)
    if pdrop:
        assert abs(float((a != 0).float().mean()) - (1 - pdrop)) < 1e-2
        assert float((a.float()[a != 0] - sw[a != 0] / (1 - pdrop)).abs().max()) < 6e-3
    else:
        assert float((a.float() - sw).abs().max()) < 3e-3
    if Fh % 128 == 0:                                                 # the launches it replaces, same dropout specs -> same masks
        z2, a2 = torch.empty_like(z), torch.empty_like(a)
        p32, p16, pxh, prs = torch.empty_like(o32), torch.empty_like(o16), torch.empty_like(xh), torch.empty_like(rs)
        ops.linear(x, w1, b1, z2)
        _call("eend_swish_dropout_f16", z2, a2, M, Fh, r1)
        name = "eend_linear_res_scale_ln_train_f16" if prenorm else "eend_linear_res_ln_train_f16"
        _call(name, a2, Fh, w2, Fh, b2, res, 0.5, gm, be, 1e-5, p32, p16, pxh, prs, M, Fh, r2)
        assert float((z.float() - z2.float()).abs().max()) < 4e-3
        big = sw.abs() > 1e-2                                          # away from swish's zero the dropout zero patterns agree exactly
        assert ((a == 0) == (a2 == 0))[big].all()
        assert float((a.float() - a2.float()).abs().max()) < 8e-3
        assert float((o32 - p32).abs().max()) < 3e-3 and float((o16.float() - p16.float()).abs().max()) < 6e-3
        assert float((xh.float() - pxh.float()).abs().max()) < 6e-3 and float(((rs - prs).abs() / prs).max()) < 2e-3
    # the operator itself, from the kernel's own a (masks included)
    y = a.float() @ w2.float().t() + b2
    if pdrop:                                                         # output dropout: infer the mask from the two-launch comparison above
        return
    y = y * 0.5 + res
    ln = F.layer_norm(y, (256,), gm, be, 1e-5)
    assert float((o32 - (y if prenorm else ln)).abs().max()) < 3e-3
    assert float((o16.float() - ln).abs().max()) < 6e-3
    mu, var = y.mean(-1, keepdim=True), y.var(-1, unbiased=False, keepdim=True)
    assert float((xh.float() - (y - mu) / torch.sqrt(var + 1e-5)).abs().max()) < 6e-3
