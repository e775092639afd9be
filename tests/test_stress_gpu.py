"""GPU: race / determinism screen for the multi-wave fused kernels at the BASELINE problem size.
A missing barrier or an early LDS read shows up as run-to-run differences long before it shows up
in a tolerance check, so every kernel is run several times on the same inputs and the outputs must
be bitwise identical; sampled rows are also checked against plain torch fp32."""
import pytest
import torch

pytestmark = pytest.mark.gpu
F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32


def rnd(shape, dev, seed, dtype=F32, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


def test_ffn_fused_full_size_deterministic(hip_lib, dev):
    from fs_eend_amd import ops
    M, Fh = 196608, 2048
    x = rnd((M, 256), dev, 1, F16)
    w1, b1 = rnd((Fh, 256), dev, 2, F16, 0.08), rnd((Fh,), dev, 3) * 0.3
    w2, b2 = rnd((256, Fh), dev, 4, F16, 0.04), rnd((256,), dev, 5) * 0.3
    res = rnd((M, 256), dev, 6)
    g, be = rnd((256,), dev, 7) * 0.2 + 1, rnd((256,), dev, 8) * 0.1
    outs = []
    for _ in range(4):
        o32 = torch.empty((M, 256), dtype=F32, device=dev)
        o16 = torch.empty((M, 256), dtype=F16, device=dev)
        ops.ffn_fused(x, w1, b1, w2, b2, res, g, be, o32, o16)
        outs.append((o32, o16))
    torch.cuda.synchronize()
    for o32, o16 in outs[1:]:
        assert torch.equal(o32, outs[0][0]) and torch.equal(o16, outs[0][1]), "fused FFN is not deterministic (race?)"
    rows = torch.arange(0, M, 997, device=dev)
    h = (x[rows].float() @ w1.float().t() + b1).relu().to(F16).float()
    want = torch.nn.functional.layer_norm(h @ w2.float().t() + b2 + res[rows], (256,), g, be, 1e-5)
    assert (outs[0][0][rows] - want).abs().max().item() < 2e-3


def test_attention_full_size_deterministic(hip_lib, dev):
    from fs_eend_amd import ops
    nseq, H, Tp = 384, 4, 512
    q, k, v = (rnd((nseq, H, Tp, 64), dev, s, BF16) for s in (11, 12, 13))
    vt = v.transpose(-1, -2).contiguous()
    outs = []
    for _ in range(4):
        o = torch.empty((nseq * Tp, 256), dtype=F16, device=dev)
        ops.attn_causal(q.view(-1), k.view(-1), vt.view(-1), o, nseq, H, Tp, 0, Tp)
        outs.append(o)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "attention is not deterministic (race?)"
    for s in (0, 191, 383):                               # sampled sequences against torch
        qs, ks, vs = q[s].float(), k[s].float(), v[s].float()
        sc = (qs @ ks.transpose(-1, -2) / 8.0).masked_fill(~torch.tril(torch.ones(Tp, Tp, dtype=torch.bool, device=dev)), float("-inf"))
        want = (torch.softmax(sc, -1) @ vs).permute(1, 0, 2).reshape(Tp, 256)
        assert (outs[0].view(nseq, Tp, 256)[s].float() - want).abs().max().item() < 2e-2


def test_proj_full_size_deterministic(hip_lib, dev):
    from fs_eend_amd import ops
    nseq, Tp, H = 384, 512, 4
    M = nseq * Tp
    a, w, b = rnd((M, 256), dev, 21, F16), rnd((768, 256), dev, 22, F16, 0.1), rnd((768,), dev, 23)
    outs = []
    for _ in range(3):
        q = torch.empty((M * 256,), dtype=BF16, device=dev)
        k, vt = torch.empty_like(q), torch.empty_like(q)
        plain = torch.empty((M, 768), dtype=F16, device=dev)
        ops.inproj_heads(a, w, b, q, k, vt, nseq, Tp, H)
        ops.linear(a, w, b, plain)
        outs.append((q, k, vt, plain))
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert all(torch.equal(x, y) for x, y in zip(o, outs[0])), "projection kernel is not deterministic (race?)"
    rows = torch.arange(0, M, 1013, device=dev)
    want = a[rows].float() @ w.float().t() + b
    assert (outs[0][3][rows].float() - want).abs().max().item() < 4e-2 * want.abs().max().item() + 1e-2
    seq, t = rows // Tp, rows % Tp
    gotq = outs[0][0].view(nseq, H, Tp, 64)[seq, :, t].reshape(-1, 256).float()
    gotv = outs[0][2].view(nseq, H, 64, Tp)[seq, :, :, t].reshape(-1, 256).float()
    assert (gotq - want[:, :256]).abs().max().item() < 5e-2
    assert (gotv - want[:, 512:]).abs().max().item() < 5e-2


def test_model_forward_deterministic(hip_lib, dev):
    """Whole FS-EEND forward at B=64, T=500, C=6: repeated runs are bitwise identical."""
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    torch.manual_seed(0)
    m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, n_units=256, n_heads=4, enc_n_layers=4,
                                       dec_n_layers=2, dropout=0.1, has_mask=True, max_seqlen=500,
                                       dec_dim_feedforward=2048).eval().to(dev)
    g = torch.Generator().manual_seed(5)
    src = [(torch.randn(500, 345, generator=g) * 2 - 3).to(dev) for _ in range(64)]
    a = torch.stack(m.test(src, [500] * 64, 6)[0]).clone()
    for _ in range(3):
        b = torch.stack(m.test(src, [500] * 64, 6)[0])
        assert torch.equal(a, b)
