"""conv1d_l2norm_stream (conv_stream.hip) against conv1d_l2norm (the implicit GEMM it replaces) and torch fp32."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

ops = importlib.import_module("fs-eend_amd.ops")
_lib = importlib.import_module("fs-eend_amd.lib")


@pytest.mark.parametrize("nseq,Tp,ktaps,pad,lens", [(1, 64, 19, 9, [64]), (3, 128, 19, 9, [128, 77, 1]), (2, 192, 19, 9, [150, 192]),
                                                    (64, 512, 19, 9, None), (2, 256, 7, 3, [200, 256]), (1, 320, 24, 0, [300])])
def test_conv_stream_vs_gemm_and_torch(nseq, Tp, ktaps, pad, lens):
    g = torch.Generator().manual_seed(nseq * 31 + ktaps)
    dev = "cuda"
    x = torch.randn(nseq * Tp, 256, generator=g).to(dev).half()
    w = (torch.randn(256, 256, ktaps, generator=g) / 40).to(dev)                  # conv.weight [o][i][tap]
    wr = w.permute(0, 2, 1).reshape(256, ktaps * 256).half().contiguous()
    bias = (torch.randn(256, generator=g) * 0.1).to(dev)
    lens = lens or [500 - (i * 13) % 200 for i in range(nseq)]
    il = torch.tensor(lens, dtype=torch.int32, device=dev)
    assert ops.conv_stream_ok(256, ktaps, pad)
    ws = ops.conv_stream_pack(wr, ktaps)
    o32 = torch.full((nseq * Tp, 256), float("nan"), device=dev); o16 = torch.full((nseq * Tp, 256), float("nan"), dtype=torch.float16, device=dev)
    ops.conv1d_l2norm_stream(x, ws, bias, il, o32, o16, nseq, Tp, ktaps, pad)
    r32 = torch.empty_like(o32); r16 = torch.empty_like(o16)
    ops.conv1d_l2norm(x, wr, bias, il, r32, r16, nseq, Tp, 256, ktaps, pad)
    torch.cuda.synchronize()
    assert torch.isfinite(o32).all() and torch.isfinite(o16).all()
    assert (o32 - r32).abs().max().item() < 2e-4
    assert (o16.float() - r16.float()).abs().max().item() < 1e-3
    # torch fp32 on the same f16 operands (only sequences of the small cases)
    if nseq <= 3:
        xs = x.float().view(nseq, Tp, 256).clone()
        for i, l in enumerate(lens):
            xs[i, l:] = 0
        y = torch.nn.functional.conv1d(xs.transpose(1, 2), wr.float().view(256, ktaps, 256).permute(0, 2, 1), bias, padding=pad)
        if y.shape[2] < Tp:                                       # pad < (ktaps - 1) / 2: the tail frames see zeros beyond Tp
            y = torch.nn.functional.conv1d(torch.nn.functional.pad(xs.transpose(1, 2), (pad, ktaps - 1 - pad)),
                                           wr.float().view(256, ktaps, 256).permute(0, 2, 1), bias)
        y = y[:, :, :Tp].transpose(1, 2).reshape(nseq * Tp, 256)
        want = y / y.norm(dim=1, keepdim=True)
        assert (o32 - want).abs().max().item() < 2e-4


def test_conv_stream_unsupported():
    assert not ops.conv_stream_ok(128, 19, 9) and not ops.conv_stream_ok(256, 25, 9) and not ops.conv_stream_ok(256, 19, 19)
