"""GPU: the skinny (M <= 16 rows) linear kernels of csrc/skinny.hip -- the path every nn.Linear of the frame-by-frame
streaming steps takes -- against plain fp32 torch on the same f16 operands, every epilogue they stand in for, and
against the tiled MFMA GEMM on the same rows (M = 17 takes the GEMM: its first rows must agree)."""
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu
F16, F32 = torch.float16, torch.float32


def _mk(dev, M, N, K, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = (torch.randn(M, K, generator=g)).to(dev).to(F16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(F16)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    return a, w, b


@pytest.mark.parametrize("M", [1, 2, 3, 6, 10, 13, 16])
@pytest.mark.parametrize("N,K", [(768, 256), (2048, 256), (1024, 256), (256, 512), (200, 256)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_skinny_linear(hip_lib, dev, M, N, K, act):
    from fs_eend_amd import ops
    a, w, b = _mk(dev, M, N, K, M * 1000 + N + act)
    out = torch.empty(M, N, dtype=F16, device=dev)
    ops.linear(a, w, b, out, act=act)
    y = a.float() @ w.float().t() + b
    want = (y, torch.relu(y), Fn.silu(y))[act]
    assert (out.float() - want).abs().max() < 2e-3 + 1e-3 * want.abs().max()
    # the tiled GEMM on 17 rows (first M rows identical inputs)
    a17 = torch.cat([a, a[:1].expand(17 - M, K)]).contiguous()
    o17 = torch.empty(17, N, dtype=F16, device=dev)
    if N % 128 == 0:
        ops.linear(a17, w, b, o17, act=act)
        assert (out.float() - o17[:M].float()).abs().max() < 2e-3 + 1e-3 * want.abs().max()


@pytest.mark.parametrize("M", [1, 4, 10, 16])
@pytest.mark.parametrize("K", [256, 1024, 2048, 4864])
def test_skinny_res_family(hip_lib, dev, M, K):
    from fs_eend_amd import ops
    a, w, b = _mk(dev, M, 256, K, M + K)
    g = torch.Generator().manual_seed(K)
    res = torch.randn(M, 256, generator=g).to(dev)
    gm = (1 + 0.2 * torch.randn(256, generator=g)).to(dev)
    be = (0.1 * torch.randn(256, generator=g)).to(dev)
    v = (a.float() @ w.float().t() + b) * 0.5 + res
    # residual + scale (StreamingConv1d, conformer half-step)
    o32, o16 = torch.empty(M, 256, device=dev), torch.empty(M, 256, dtype=F16, device=dev)
    ops.linear_res_scale(a, w, b, res, 0.5, o32, o16)
    assert (o32 - v).abs().max() < 2e-3 and (o16.float() - v).abs().max() < 4e-3
    # + LayerNorm on both outputs (post-norm transformer layers); in place on the residual stream
    stream = res.clone()
    ops.linear_res_ln(a, w, b, stream, gm, be, stream, o16, 1e-5, alpha=0.5)
    want = Fn.layer_norm(v, (256,), gm, be, 1e-5)
    assert (stream - want).abs().max() < 3e-3 and (o16.float() - want).abs().max() < 6e-3
    # un-normalised stream + normalised f16 copy (conformer modules)
    stream = res.clone()
    ops.linear_res_scale_ln16(a, w, b, stream, 0.5, gm, be, stream, o16, 1e-5)
    assert (stream - v).abs().max() < 2e-3 and (o16.float() - want).abs().max() < 6e-3


@pytest.mark.parametrize("M", [1, 5, 16])
def test_skinny_glu(hip_lib, dev, M):
    from fs_eend_amd import ops
    a, w, b = _mk(dev, M, 512, 256, 77 + M)                       # rows interleaved: 2n value, 2n+1 gate
    out = torch.empty(M, 256, dtype=F16, device=dev)
    ops.linear_glu(a, w, b, out)
    y = a.float() @ w.float().t() + b
    want = y[:, 0::2] * torch.sigmoid(y[:, 1::2])
    assert (out.float() - want).abs().max() < 3e-3


def test_skinny_switch_matches_gemm_semantics(hip_lib, dev):
    """rows 0..15 through the skinny path == the same rows inside a 32-row GEMM call (tolerance: f32 summation order)."""
    from fs_eend_amd import ops
    a, w, b = _mk(dev, 32, 256, 2048, 5)
    res = torch.randn(32, 256, device=dev)
    gm, be = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    o32a, o16a = torch.empty(32, 256, device=dev), torch.empty(32, 256, dtype=F16, device=dev)
    ops.linear_res_ln(a, w, b, res, gm, be, o32a, o16a)
    o32b, o16b = torch.empty(16, 256, device=dev), torch.empty(16, 256, dtype=F16, device=dev)
    ops.linear_res_ln(a[:16], w, b, res[:16], gm, be, o32b, o16b)
    assert (o32a[:16] - o32b).abs().max() < 1e-3
