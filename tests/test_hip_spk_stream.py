"""attnout_spk_stream (spk_stream.hip): LN11(A Wo1^T + bo1 + res) + speaker-axis MHA in one launch, against a torch fp32
restatement of the same operator and against the un-fused launches (linear_res16_ln + linear + spk_attn)."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

ops = importlib.import_module("fs-eend_amd.ops")
_lib = importlib.import_module("fs-eend_amd.lib")


def _inputs(B, C, Tp, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    M = B * C * Tp
    dev = "cuda"
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    a = r(M, 256).half()
    res = r(M, 256).half()
    wo = r(256, 256, sc=1 / 16).half()
    win = r(768, 256, sc=1 / 8).half()
    bo = r(256, sc=0.1); g1 = 1 + r(256, sc=0.1); be1 = r(256, sc=0.1); bin_ = r(768, sc=0.3)
    return a, res, wo, win, bo, g1, be1, bin_


def _torch_ref(a, res, wo, win, bo, g1, be1, bin_, B, C, Tp):
    x = torch.nn.functional.layer_norm(a.float() @ wo.float().T + bo + res.float(), (256,), g1, be1, 1e-5)
    x16 = x.half()
    qkv = x16.float() @ win.float().T + bin_
    q, k, v = qkv.split(256, dim=1)
    sh = lambda t: t.view(B, C, Tp, 4, 64).permute(0, 2, 3, 1, 4)      # B, Tp, H, C, dh
    p = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) * 0.125, dim=-1)
    o = (p @ sh(v)).permute(0, 3, 1, 2, 4).reshape(B * C * Tp, 256)
    return x, o


@pytest.mark.parametrize("B,C,Tp", [(2, 6, 64), (3, 6, 512), (1, 3, 128), (2, 12, 64), (5, 12, 256), (70, 6, 96),
                                    # slot counts that leave phantom positions in the tiling (masked keys, rows never stored)
                                    (2, 1, 64), (2, 2, 128), (3, 4, 64), (2, 5, 96), (2, 7, 64), (2, 8, 64), (1, 9, 128), (3, 10, 512), (2, 11, 64)])
def test_attnout_spk_stream_vs_torch(B, C, Tp):
    a, res, wo, win, bo, g1, be1, bin_ = _inputs(B, C, Tp, 11 + C)
    assert ops.spk_stream_ok(C, Tp)
    ws = ops.spk_stream_pack(wo, win)
    x16 = torch.empty_like(res); o16 = torch.empty_like(a)
    ops.attnout_spk_stream(a, ws, bo, res, g1, be1, 1e-5, x16, bin_, o16, B, C, Tp)
    torch.cuda.synchronize()
    xr, orf = _torch_ref(a, res, wo, win, bo, g1, be1, bin_, B, C, Tp)
    assert torch.isfinite(o16).all() and torch.isfinite(x16).all()
    assert (x16.float() - xr).abs().max().item() < 2e-2
    assert (o16.float() - orf).abs().max().item() < 2e-2
    # the un-fused launches
    x2 = torch.empty_like(res); o2 = torch.empty_like(a)
    ops.linear_res16_ln(a, wo, bo, res, g1, be1, None, x2, 1e-5)
    qkv2 = torch.empty(a.shape[0], 768, dtype=a.dtype, device=a.device)
    ops.linear(x2, win, bin_, qkv2)
    ops.spk_attn(qkv2, o2, B, C, Tp, 4)
    torch.cuda.synchronize()
    assert (x16.float() - x2.float()).abs().max().item() < 1e-2
    assert (o16.float() - o2.float()).abs().max().item() < 2e-2


def test_attnout_spk_stream_in_place():
    B, C, Tp = 4, 6, 128
    a, res, wo, win, bo, g1, be1, bin_ = _inputs(B, C, Tp, 5)
    ws = ops.spk_stream_pack(wo, win)
    x16 = torch.empty_like(res); o16 = torch.empty_like(a)
    ops.attnout_spk_stream(a, ws, bo, res, g1, be1, 1e-5, x16, bin_, o16, B, C, Tp)
    a2, r2 = a.clone(), res.clone()
    ops.attnout_spk_stream(a2, ws, bo, r2, g1, be1, 1e-5, r2, bin_, a2, B, C, Tp)      # x over res, O over A
    torch.cuda.synchronize()
    assert torch.equal(r2, x16) and torch.equal(a2, o16)


def test_attnout_spk_stream_unsupported_shapes():
    assert not ops.spk_stream_ok(13, 512) and not ops.spk_stream_ok(0, 512) and not ops.spk_stream_ok(6, 500) and not ops.spk_stream_ok(3, 96)
    B, C, Tp = 1, 6, 80
    a, res, wo, win, bo, g1, be1, bin_ = _inputs(B, C, Tp, 3)
    ws = ops.spk_stream_pack(wo, win)
    with pytest.raises(_lib.EendHipError):
        ops.attnout_spk_stream(a, ws, bo, res, g1, be1, 1e-5, torch.empty_like(res), bin_, torch.empty_like(a), B, C, Tp)


# ---- the f32-residual form (round 5; LS-EEND's decoder: merge_retnet_layer.py:301-306 on the f32 stream of DESIGN 4)
@pytest.mark.parametrize("B,C,Tp", [(2, 6, 64), (3, 10, 512), (2, 12, 64), (1, 3, 128), (3, 4, 64), (2, 7, 64), (1, 9, 128), (2, 1, 64),
                                    (16, 10, 2048)])
def test_attnout_spk_stream_res32_vs_torch(B, C, Tp):
    a, res16, wo, win, bo, g1, be1, bin_ = _inputs(B, C, Tp, 31 + C)
    g = torch.Generator(device="cpu").manual_seed(77 + C)
    res = torch.randn(B * C * Tp, 256, generator=g).cuda()                    # a genuinely f32 residual
    ws = ops.spk_stream_pack(wo, win)
    x32 = torch.full_like(res, float("nan")); o16 = torch.full_like(a, float("nan"))
    ops.attnout_spk_stream_res32(a, ws, bo, res, g1, be1, 1e-5, x32, bin_, o16, B, C, Tp)
    torch.cuda.synchronize()
    xr, orf = _torch_ref(a, res, wo, win, bo, g1, be1, bin_, B, C, Tp)
    assert torch.isfinite(o16).all() and torch.isfinite(x32).all()
    assert (x32 - xr).abs().max().item() < 2e-4                               # f32 rows: accumulation order only
    assert (o16.float() - orf).abs().max().item() < 2e-2
    # the un-fused launches
    x2 = torch.empty_like(res); x2h = torch.empty_like(a); o2 = torch.empty_like(a)
    ops.linear_res_ln(a, wo, bo, res, g1, be1, x2, x2h, 1e-5)
    qkv2 = torch.empty(a.shape[0], 768, dtype=a.dtype, device=a.device)
    ops.linear(x2h, win, bin_, qkv2)
    ops.spk_attn(qkv2, o2, B, C, Tp, 4)
    torch.cuda.synchronize()
    assert (x32 - x2).abs().max().item() < 2e-4
    assert (o16.float() - o2.float()).abs().max().item() < 2e-2
    # in place, as the model calls it: x over res, O over A
    a2, r2 = a.clone(), res.clone()
    ops.attnout_spk_stream_res32(a2, ws, bo, r2, g1, be1, 1e-5, r2, bin_, a2, B, C, Tp)
    torch.cuda.synchronize()
    assert torch.equal(r2, x32) and torch.equal(a2, o16)


def test_attnout_spk_stream_res32_rejects_mixed_forms():
    B, C, Tp = 1, 6, 64
    a, res16, wo, win, bo, g1, be1, bin_ = _inputs(B, C, Tp, 3)
    ws = ops.spk_stream_pack(wo, win)
    with pytest.raises(_lib.EendHipError):                                    # unsupported padded length
        ops.attnout_spk_stream_res32(a[:6 * 40], ws, bo, res16[:6 * 40].float(), g1, be1, 1e-5, torch.empty(6 * 40, 256, device="cuda"), bin_,
                                     torch.empty_like(a[:6 * 40]), 1, 6, 40)
    with pytest.raises(_lib.EendHipError):                                    # dtype check: an f16 residual is the other entry's
        ops.attnout_spk_stream_res32(a, ws, bo, res16, g1, be1, 1e-5, res16, bin_, torch.empty_like(a), B, C, Tp)
