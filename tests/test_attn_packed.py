"""GPU: the packed in-projection + causal attention kernel (csrc/attn_stream.hip, Tp = 64 m <= 512) against the general two-kernel path
(eend_inproj_heads_bf16 -> eend_attn_causal_bf16, longer or ragged chunk lengths) and against fp32 torch on the same f16 operands."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32


def _case(dev, nseq, Tp, seed):
    from fs_eend_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(nseq * Tp, 256, generator=g).to(dev).to(F16)
    w = (torch.randn(768, 256, generator=g) / 16)
    b = torch.randn(768, generator=g) * 0.1
    w[:256] *= ops.QSCALE_LOG2
    b[:256] *= ops.QSCALE_LOG2
    return x, w.to(dev).to(F16), b.to(dev)


def _two_kernel(x, w, b, nseq, Tp, delay, kv_len):
    from fs_eend_amd import ops
    dev = x.device
    q, k, vt = (torch.empty(nseq * Tp * 256, dtype=BF16, device=dev) for _ in range(3))
    o = torch.empty(nseq * Tp, 256, dtype=F16, device=dev)
    ops.inproj_heads(x, w, b, q, k, vt, nseq, Tp, 4)
    ops.attn_causal(q, k, vt, o, nseq, 4, Tp, delay, kv_len, scale=ops.LN2)
    return o


def _fp32(x, w, b, nseq, Tp, delay, kv_len):
    dev = x.device
    y = x.float() @ w.float().t() + b
    q, k, v = (y[:, i * 256:(i + 1) * 256].view(nseq, Tp, 4, 64).transpose(1, 2) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * math.log(2.0)
    i = torch.arange(Tp, device=dev)[:, None]
    j = torch.arange(Tp, device=dev)[None, :]
    ok = ((j - i) <= delay) & (j < kv_len)
    return (torch.softmax(s.masked_fill(~ok, float("-inf")), -1) @ v).transpose(1, 2).reshape(nseq * Tp, 256)


@pytest.mark.parametrize("nseq,Tp,delay,kv_len", [(1, 64, 0, 64), (3, 128, 0, 100), (2, 192, 2, 192), (16, 256, 0, 250), (2, 448, 1000, 448),
                                                  (3, 320, 0, 300), (2, 640, 0, 600), (1, 1024, 0, 1000)])
def test_general_path_other_chunk_lengths(hip_lib, dev, nseq, Tp, delay, kv_len):
    """in-projection + resident (<= 512) / tiled attention: the path fs_model takes for every chunk length but 512."""
    x, w, b = _case(dev, nseq, Tp, nseq * 7 + Tp)
    o = _two_kernel(x, w, b, nseq, Tp, delay, kv_len)
    want = _fp32(x, w, b, nseq, Tp, delay, kv_len)
    assert torch.isfinite(o).all()
    assert (o.float() - want).abs().max().item() < 2e-2


@pytest.mark.parametrize("nseq,delay,kv_len,Tp", [(1, 0, 512, 512), (8, 0, 500, 512), (5, 3, 470, 512), (40, 0, 500, 512), (2, 1000, 512, 512),
                                                   (300, 0, 500, 512),
                                                   # round 6: shorter padded lengths on the same kernel (block slots beyond Tp stay empty)
                                                   (7, 0, 300, 320), (3, 0, 64, 64), (9, 2, 100, 128), (12, 0, 440, 448), (4, 1000, 380, 384),
                                                   (5, 0, 192, 192), (130, 0, 250, 256)])
def test_inproj_attn_packed(hip_lib, dev, nseq, delay, kv_len, Tp):
    """attn_stream.hip (token-owning waves, packed weights, Q in registers) against fp32 torch and against the two-kernel path."""
    from fs_eend_amd import ops
    x, w, b = _case(dev, nseq, Tp, nseq * 11 + delay)
    wp = ops.inproj_attn_pack(w)
    o = torch.full((nseq * Tp, 256), float("nan"), dtype=F16, device=dev)
    ops.inproj_attn_causal_packed(x, wp, b, o, nseq, 4, Tp, delay, kv_len)
    o1 = _two_kernel(x, w, b, nseq, Tp, delay, kv_len)
    assert torch.isfinite(o).all()
    e_old = (o.float() - o1.float()).abs().max().item()
    if nseq <= 40:
        want = _fp32(x, w, b, nseq, Tp, delay, kv_len)
        e_ref = (o.float() - want).abs().max().item()
        e_ref_old = (o1.float() - want).abs().max().item()
        print(f"packed vs fp32 {e_ref:.2e} (two kernels vs fp32 {e_ref_old:.2e}); packed vs two kernels {e_old:.2e}")
        assert e_ref < 2e-2
    # two different bf16 roundings of K (here without the key bias, which cancels in the softmax) and of Q
    assert e_old < 3e-2


def test_inproj_attn_packed_rejects_other_lengths(hip_lib, dev):
    """windows beyond the 512 frames the LDS tiles hold (and lengths that are not a multiple of 64): EEND_EINVAL, the caller keeps the two kernels"""
    from fs_eend_amd import ops, lib as _lib
    for Tp in (576, 1024, 96):
        x, w, b = _case(dev, 2, Tp, 1)
        wp = ops.inproj_attn_pack(w)
        with pytest.raises(_lib.EendHipError):
            ops.inproj_attn_causal_packed(x, wp, b, torch.empty_like(x), 2, 4, Tp, 0, Tp)


@pytest.mark.parametrize("B,T,C", [(3, 130, 3), (5, 500, 6), (2, 512, 10), (2, 700, 4)])
def test_model_chunk_lengths_agree_with_oracle(hip_lib, dev, B, T, C):
    """model.test through both attention forms (Tp = 512: packed kernel; other lengths: in-projection + resident / tiled attention)
    against the fp32 oracle, ragged lengths included."""
    from fs_eend_amd import fs_model as FM
    from oracle import fs_eend_ref as R
    torch.manual_seed(3)
    cfg = dict(n_units=256, n_heads=4, enc_n_layers=2, dec_n_layers=2, dropout=0.1, has_mask=True, max_seqlen=500,
               dec_dim_feedforward=512, conv_delay=9, mask_delay=0, decom_kernel_size=64)
    m = FM.OnlineTransformerDADiarization(n_speakers=None, in_size=345, **cfg).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B * 100 + T)
    lens = [T - (i * 7) % max(1, T // 3) for i in range(B)]
    src = [torch.randn(l, 345, generator=g) * 2 - 3 for l in lens]
    want = R.fs_test(src, lens, sd, n_heads=4, enc_n_layers=2, dec_n_layers=2, max_nspks=C)
    m = m.to(dev)
    got = m.test([x.to(dev) for x in src], lens, C)
    for x, y in zip(got[0], want[0]):
        assert x.shape == y.shape and torch.isfinite(x).all()
        assert (x.cpu() - y).abs().max() < 1e-3, float((x.cpu() - y).abs().max())


@pytest.mark.parametrize("nseq,delay,kv_len,Tp", [(3, 0, 1000, 1024), (2, 0, 600, 640), (5, 3, 1500, 1536), (2, 0, 2040, 2048), (1, 5000, 1000, 1024),
                                                   (2, 700, 1100, 1152), (40, 0, 1024, 1024), (2, 0, 520, 1024), (1, 0, 4000, 4096), (3, 600, 300, 1024)])
def test_inproj_attn_long(hip_lib, dev, nseq, delay, kv_len, Tp):
    """Windows of more than 512 frames on attn_stream.hip's (query group, key group) items + combine pass, against fp32 torch and the
    two-kernel path (in-projection, tiled attn.hip); covers look-ahead masks that reach later key groups, key groups cut by kv_len
    (rows without a key in reach of an item) and a last group shorter than 512 frames."""
    from fs_eend_amd import ops
    x, w, b = _case(dev, nseq, Tp, nseq * 13 + delay + Tp)
    wp = ops.inproj_attn_pack(w)
    need = ops.inproj_attn_long_scratch(nseq, Tp, delay, kv_len)
    assert need is not None
    part = torch.full((need[0],), float("nan"), dtype=F16, device=dev)
    lse = torch.full((need[1],), float("nan"), dtype=F32, device=dev)
    o = torch.full((nseq * Tp, 256), float("nan"), dtype=F16, device=dev)
    ops.inproj_attn_causal_long(x, wp, b, o, part, lse, nseq, 4, Tp, delay, kv_len)
    o1 = _two_kernel(x, w, b, nseq, Tp, delay, kv_len)
    assert torch.isfinite(o).all()
    e_old = (o.float() - o1.float()).abs().max().item()
    if nseq * Tp <= 8192:
        want = _fp32(x, w, b, nseq, Tp, delay, kv_len)
        e_ref = (o.float() - want).abs().max().item()
        e_ref_old = (o1.float() - want).abs().max().item()
        print(f"long vs fp32 {e_ref:.2e} (two kernels vs fp32 {e_ref_old:.2e}); long vs two kernels {e_old:.2e}")
        assert e_ref < 2e-2
    assert e_old < 3e-2


def test_inproj_attn_long_scratch_shapes(hip_lib, dev):
    """the scratch query is the shape predicate: windows <= 512 frames, ragged lengths and more than 64 items are not covered"""
    from fs_eend_amd import ops
    assert ops.inproj_attn_long_scratch(4, 512) is None
    assert ops.inproj_attn_long_scratch(4, 1000) is None
    assert ops.inproj_attn_long_scratch(4, 1024, 0, 0) is None
    assert ops.inproj_attn_long_scratch(4, 1024) == (1 * 4 * 512 * 256, 3 * 4 * 4 * 512)
    assert ops.inproj_attn_long_scratch(4, 1024, 10000) == (2 * 4 * 512 * 256, 4 * 4 * 4 * 512)
    assert ops.inproj_attn_long_scratch(1, 4096) is not None               # 36 items
    assert ops.inproj_attn_long_scratch(1, 4096, 10000) is not None        # 64 items
    assert ops.inproj_attn_long_scratch(1, 4608, 10000) is None            # 81 items
    assert ops.inproj_attn_long_scratch(1, 6144) is None                   # 78 items (and 11 key groups for the last query group)
