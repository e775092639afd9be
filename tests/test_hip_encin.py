"""encoder_input (encin.hip): pad_sequence + BatchNorm + input projection + LayerNorm in one launch, against the two launches it
replaces (gather_bn_cast_pad + linear_res_ln) and against a torch fp32 restatement."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

ops = importlib.import_module("fs-eend_amd.ops")
_lib = importlib.import_module("fs-eend_amd.lib")


def _case(B, T, Fin, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    dev = "cuda"
    lens = [max(1, T - (i * 37) % max(1, T // 2)) if ragged else T for i in range(B)]
    if ragged:
        lens[0] = T
    src = [(torch.randn(l, Fin, generator=g) * 2 - 1).to(dev) for l in lens]
    bn = tuple(t.to(dev) for t in (1 + 0.1 * torch.randn(Fin, generator=g), 0.1 * torch.randn(Fin, generator=g),
                                   0.3 * torch.randn(Fin, generator=g), 0.5 + torch.rand(Fin, generator=g)))
    Kp = (Fin + 63) // 64 * 64
    w = torch.zeros(256, Kp)
    w[:, :Fin] = torch.randn(256, Fin, generator=g) / 16
    bias = 0.1 * torch.randn(256, generator=g); gamma = 1 + 0.1 * torch.randn(256, generator=g); beta = 0.1 * torch.randn(256, generator=g)
    return src, lens, bn, w.to(dev).half(), bias.to(dev), gamma.to(dev), beta.to(dev)


@pytest.mark.parametrize("B,T,Fin", [(1, 32, 345), (3, 100, 345), (5, 500, 345), (64, 500, 345), (2, 75, 384), (4, 130, 321), (7, 257, 352)])
def test_encoder_input_vs_pair_and_torch(B, T, Fin):
    src, lens, bn, w, bias, gamma, beta = _case(B, T, Fin, 3 + B)
    Tp = ops.frames_pad(T)
    assert ops.encoder_input_ok(src, Tp, w)
    o16 = torch.full((B * Tp, 256), float("nan"), dtype=torch.float16, device="cuda")
    o32 = torch.full((B * Tp, 256), float("nan"), device="cuda")
    ops.encoder_input(src, bn, w, bias, gamma, beta, o32, o16, T, Tp, -1.0)
    # the pair it replaces
    x16 = torch.zeros(B * Tp, w.shape[1], dtype=torch.float16, device="cuda")
    p16 = torch.empty_like(o16); p32 = torch.empty_like(o32)
    ops.gather_bn_cast_pad(src, bn, x16, T, Tp, -1.0, True)
    ops.linear_res_ln(x16, w, bias, None, gamma, beta, p32, p16)
    torch.cuda.synchronize()
    assert torch.isfinite(o16).all() and torch.isfinite(o32).all()
    assert (o16.float() - p16.float()).abs().max().item() < 4e-3
    assert (o32 - p32).abs().max().item() < 2e-3
    # torch fp32 on the same f16 operands
    x = torch.full((B, Tp, Fin), -1.0, device="cuda")
    for i, s_ in enumerate(src):
        x[i, :lens[i]] = s_
    scl = bn[0] / torch.sqrt(bn[3] + 1e-5)
    x = x * scl + (bn[1] - bn[2] * scl)
    x[:, T:] = 0
    want = torch.nn.functional.layer_norm(x.half().float() @ w[:, :Fin].float().T + bias, (256,), gamma, beta, 1e-5).view(B * Tp, 256)
    assert (o32 - want).abs().max().item() < 2e-3
    o16b = torch.empty_like(o16)
    ops.encoder_input(src, bn, w, bias, gamma, beta, None, o16b, T, Tp, -1.0)          # without the f32 output
    torch.cuda.synchronize()
    assert torch.equal(o16b, o16)


def test_encoder_input_unsupported():
    src, lens, bn, w, bias, gamma, beta = _case(2, 64, 200, 1)
    assert not ops.encoder_input_ok(src, 64, w)
    with pytest.raises(_lib.EendHipError):
        ops.encoder_input(src, bn, w, bias, gamma, beta, None, torch.empty(128, 256, dtype=torch.float16, device="cuda"), 64, 64, -1.0)
