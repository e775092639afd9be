"""GPU: every backward / optimiser kernel of the training step, one at a time, through the C-ABI, against a plain
PyTorch fp32 (autograd) reference of the same op on the same device.

Tolerances: gradient tensors travel as bf16 (8 significant bits) and are accumulated in fp32, so element-wise
comparisons use a tolerance relative to the tensor's scale (`rel`), reductions over many rows much tighter ones.
"""
import math

import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu

F16, BF16, F32, I32 = torch.float16, torch.bfloat16, torch.float32, torch.int32


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def relnorm(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def T(hip_lib, dev):
    from fs_eend_amd import train
    return train


@pytest.fixture(scope="module")
def ws(dev):
    return torch.empty(8 * 1024 * 1024, dtype=F32, device=dev)


def g(dev, seed):
    return torch.Generator(device=dev).manual_seed(seed)


@pytest.mark.parametrize("M,N,K,bias", [(300, 256, 256, True), (1000, 384, 256, False), (515, 2048, 256, True), (129, 128, 768, False)])
def test_gemm_bf16(T, dev, M, N, K, bias):
    gen = g(dev, M + N)
    A = torch.randn(M, K, device=dev, generator=gen).to(BF16)
    W = (torch.randn(N, K, device=dev, generator=gen) / math.sqrt(K)).to(BF16)
    b = torch.randn(N, device=dev, generator=gen) if bias else None
    out = torch.full((M, N), 7.0, dtype=BF16, device=dev)
    T._call("eend_gemm_bf16", A, K, W, K, b, out, N, M, N, K)
    want = A.float() @ W.float().t() + (b if bias else 0)
    assert rel(out, want) < 6e-3


@pytest.mark.parametrize("M,F", [(300, 2048), (64, 256), (1000, 512)])
def test_gemm_relu_bwd(T, dev, M, F):
    gen = g(dev, M)
    dy = (torch.randn(M, 256, device=dev, generator=gen) * 1e-4).to(BF16)
    w2t = (torch.randn(F, 256, device=dev, generator=gen) / 16).to(BF16)        # [F][256] = W2^T
    act = torch.relu(torch.randn(M, F, device=dev, generator=gen)).to(F16)
    out = torch.full((M, F), 3.0, dtype=BF16, device=dev)
    T._call("eend_gemm_relu_bwd_bf16", dy, 256, w2t, 256, act, F, out, F, M, F, 256, 1.0)
    want = (dy.float() @ w2t.float().t()) * (act > 0)
    assert rel(out, want) < 6e-3
    assert (out[act == 0] == 0).all()



def _stream_of(T, dev, a, b, F):
    """eend_ffn_train_stream_pack of two 16-bit matrices a [F][256], b [256][F] (f16 forward operands / bf16 transposed copies)."""
    from fs_eend_amd import lib as L
    n = L.load().eend_ffn_train_stream_elems(F)
    assert n == 2 * (F // 32) * 8192
    out = torch.empty(n, dtype=a.dtype, device=dev)
    T._call("eend_ffn_train_stream_pack", a, b, out, F)
    return out


def _to_blocked(t, pad=0.0):
    """row-major [M][F] -> the blocked layout of the stream entries, [ceil(M/16)*16][F] storage (padding rows = pad)."""
    M, F = t.shape
    Mp = (M + 15) // 16 * 16
    tp = torch.full((Mp, F), pad, dtype=t.dtype, device=t.device)
    tp[:M] = t
    return tp.view(Mp // 16, 16, F // 32, 32).permute(0, 2, 1, 3).contiguous().view(Mp, F)


def _from_blocked(tb, M):
    Mp, F = tb.shape
    return tb.view(Mp // 16, F // 32, 16, 32).permute(0, 2, 1, 3).contiguous().view(Mp, F)[:M]


def _ffn_train_call(T, dev, impl, x, w1, b1, w2, b2, res, gm, be, o32, o16, hid, xh, rs, M, F, r1, r2):
    if impl == "stream":
        hb = torch.full(((M + 15) // 16 * 16, F), float("nan"), dtype=F16, device=dev)
        T._call("eend_ffn_train_stream_f16", x, 256, _stream_of(T, dev, w1, w2, F), b1, b2, res, 1.0, gm, be, 1e-5, o32, o16, hb, xh, rs, M, F, r1, r2)
        assert torch.isfinite(hb).all()                      # the padding rows of the last block too: the weight gradient multiplies them by zero rows
        hid.copy_(_from_blocked(hb, M))
    else:
        T._call("eend_ffn_train_f16", x, 256, w1, b1, w2, b2, res, 1.0, gm, be, 1e-5, o32, o16, hid, xh, rs, M, F, r1, r2)


def _ffn_bwd_call(T, dev, impl, dy, w2t, act, w1t, scale, dh, gout, M, F):
    if impl == "stream":
        db = torch.full(((M + 15) // 16 * 16, F), float("nan"), dtype=BF16, device=dev)
        T._call("eend_ffn_bwd_data_stream_bf16", dy, 256, _stream_of(T, dev, w2t, w1t, F), _to_blocked(act), scale, db, gout, M, F)
        assert torch.isfinite(db.float()).all()
        dh.copy_(_from_blocked(db, M))
    else:
        T._call("eend_ffn_bwd_data_bf16", dy, 256, w2t, act, w1t, scale, dh, gout, M, F)

@pytest.mark.parametrize("impl", ["fused", "stream"])
@pytest.mark.parametrize("M,F,scale", [(1000, 1024, 1.0 / 0.75), (300, 2048, 1.0), (70001, 2048, 1.0 / 0.9), (64, 64, 1.0)])
def test_ffn_bwd_data_fused(T, dev, M, F, scale, impl):
    """eend_ffn_bwd_data_bf16 (ffn.hip MODE 4, round 5) == eend_gemm_relu_bwd_bf16 + eend_gemm_acc_bf16: dH bit-comparable (same bf16
    operands, f32 accumulation in a different k order), the residual-gradient stream accumulated in place; both == torch fp32.
    impl "stream": the same operator on the packed weight stream (eend_ffn_bwd_data_stream_bf16, ffn_train_stream.hip, round 6)."""
    gen = g(dev, M + F)
    dy = (torch.randn(M, 256, device=dev, generator=gen) * 1e-4).to(BF16)
    w2t = (torch.randn(F, 256, device=dev, generator=gen) / 16).to(BF16)        # [F][256] = W2^T
    w1t = (torch.randn(256, F, device=dev, generator=gen) / math.sqrt(F)).to(BF16)   # [256][F] = W1^T
    act = torch.relu(torch.randn(M, F, device=dev, generator=gen)).to(F16)
    g32 = torch.randn(M, 256, device=dev, generator=gen) * 1e-4
    dh = torch.full((M, F), 3.0, dtype=BF16, device=dev)
    gout = g32.clone()
    _ffn_bwd_call(T, dev, impl, dy, w2t, act, w1t, scale, dh, gout, M, F)
    torch.cuda.synchronize()
    want_dh = (dy.float() @ w2t.float().t()) * scale * (act > 0)
    assert torch.isfinite(dh.float()).all() and torch.isfinite(gout).all()
    assert rel(dh, want_dh) < 6e-3
    assert (dh[act == 0] == 0).all()
    want_g = dh.float() @ w1t.float().t() + g32                                   # from the kernel's own (rounded) dH
    assert rel(gout - g32, want_g - g32) < 3e-3
    if F % 128 == 0:                                                              # the two launches it replaces
        dh2 = torch.empty(M, F, dtype=BF16, device=dev)
        g2 = g32.clone()
        T._call("eend_gemm_relu_bwd_bf16", dy, 256, w2t, 256, act, F, dh2, F, M, F, 256, scale)
        T._call("eend_gemm_acc_bf16", dh2, F, w1t, F, g2, 1.0, g2, None, M, F)
        assert rel(dh, dh2) < 5e-3                                                # a value may round to the neighbouring bf16
        assert rel(gout - g32, g2 - g32) < 5e-3


def test_fused_ffn_entries_reject_what_they_do_not_cover(T, dev):
    """EEND_EINVAL (never a silent wrong answer) outside the fused training FFN's shapes: the host then keeps the GEMM launches."""
    from fs_eend_amd.lib import EendHipError
    M, F = 128, 96                                                                # F not a multiple of the 64-unit chunk
    x = torch.zeros(M, 256, dtype=F16, device=dev)
    w1, b1 = torch.zeros(F, 256, dtype=F16, device=dev), torch.zeros(F, device=dev)
    w2, b2 = torch.zeros(256, F, dtype=F16, device=dev), torch.zeros(256, device=dev)
    res, gm = torch.zeros(M, 256, device=dev), torch.ones(256, device=dev)
    o32, o16, hid, xh, rs = torch.empty(M, 256, device=dev), torch.empty(M, 256, dtype=F16, device=dev), torch.empty(M, F, dtype=F16, device=dev), torch.empty(M, 256, dtype=F16, device=dev), torch.empty(M, device=dev)
    with pytest.raises(EendHipError):
        T._call("eend_ffn_train_f16", x, 256, w1, b1, w2, b2, res, 1.0, gm, b2, 1e-5, o32, o16, hid, xh, rs, M, F, None, None)
    with pytest.raises(EendHipError):                                             # missing statistics buffers
        T._call("eend_ffn_train_f16", x, 256, w1, b1, w2, b2, res, 1.0, gm, b2, 1e-5, o32, o16, hid, None, None, M, 64, None, None)
    dy = torch.zeros(M, 256, dtype=BF16, device=dev)
    with pytest.raises(EendHipError):
        T._call("eend_ffn_bwd_data_bf16", dy, 256, w1.to(BF16), hid, w2.to(BF16), 1.0, torch.empty(M, F, dtype=BF16, device=dev), res, M, F)
    with pytest.raises(EendHipError):                                             # no residual-gradient stream
        T._call("eend_ffn_bwd_data_bf16", dy, 256, w1.to(BF16), hid, w2.to(BF16), 1.0, torch.empty(M, 64, dtype=BF16, device=dev), None, M, 64)


@pytest.mark.parametrize("M,K", [(300, 256), (777, 768), (200, 2048)])
def test_gemm_acc(T, dev, M, K):
    gen = g(dev, K)
    A = (torch.randn(M, K, device=dev, generator=gen) * 1e-3).to(BF16)
    W = (torch.randn(256, K, device=dev, generator=gen) / math.sqrt(K)).to(BF16)
    res = torch.randn(M, 256, device=dev, generator=gen) * 1e-3
    out = res.clone()
    T._call("eend_gemm_acc_bf16", A, K, W, K, out, 1.0, out, None, M, K)          # in place, as the trainer uses it
    want = A.float() @ W.float().t() + res
    assert rel(out, want) < 2e-3


@pytest.mark.parametrize("M,N,K,f16", [(1000, 256, 256, True), (4133, 256, 2048, True), (4133, 2048, 256, False), (70, 768, 256, True),
                                       (50000, 256, 384, True),
                                       # >= 131072 tokens with 256-aligned shapes: the 256 x 256 output tile (ragged last step)
                                       (131072 + 37, 256, 512, True), (140000, 512, 256, False), (131072, 256, 256, True)])
def test_wgrad(T, dev, ws, M, N, K, f16):
    gen = g(dev, M + K)
    dy = (torch.randn(M, N, device=dev, generator=gen) * 1e-5).to(BF16)
    x = torch.randn(M, K, device=dev, generator=gen).to(F16 if f16 else BF16)
    out = torch.full((N, K), 5.0, dtype=F32, device=dev)
    T._call("eend_wgrad_bf16", dy, N, x, K, 1 if f16 else 0, M, N, K, ws, ws.numel(), out, K, K, 1.0, 0)
    want = dy.float().t() @ x.to(BF16).float()
    assert relnorm(out, want) < 2e-3, relnorm(out, want)
    assert rel(out, want) < 1e-2
    # the same with the bias gradient on the side (column sums of dy from the staging registers of the k-tile-0 workgroups)
    out2 = torch.full((N, K), 7.0, dtype=F32, device=dev)
    bias = torch.full((N,), 3.0, dtype=F32, device=dev)
    T._call("eend_wgrad_bias_bf16", dy, N, x, K, 1 if f16 else 0, M, N, K, ws, ws.numel(), out2, K, K, bias, 1.0, 0)
    assert relnorm(out2, want) < 2e-3 and rel(out2, out) < 1e-4        # (a different split plan: not bit-identical to `out`)
    wantb = dy.double().sum(0)
    assert float((bias.double() - wantb).abs().max()) < 2e-5 * float(dy.double().abs().sum(0).max()), float((bias.double() - wantb).abs().max())
    # operands that are column blocks of wider tensors (lda > N, ldb > K), as the LS trainer passes them
    if K == 512:
        wide_dy = torch.zeros(M, 4 * N, dtype=BF16, device=dev)
        wide_dy[:, 2 * N:3 * N] = dy
        wide_x = torch.zeros(M, 2 * K, dtype=x.dtype, device=dev)
        wide_x[:, K:] = x
        out3 = torch.full((N, K), 1.0, dtype=F32, device=dev)
        T._call("eend_wgrad_bf16", wide_dy[:, 2 * N:], 4 * N, wide_x[:, K:], 2 * K, 1 if f16 else 0, M, N, K, ws, ws.numel(), out3, K, K, 1.0, 0)
        assert (out3 == out).all()
    # an operand in the blocked layout of the stream FFN entries ([M/16][F/32][16][32], x_is_f16 & 2: X, & 4: dY), incl. a partial last
    # block whose padding rows hold finite garbage (they meet zero rows of the row-major operand): bit-identical to the row-major call
    if K % 32 == 0 and N % 32 == 0:
        xb = _to_blocked(x, pad=3.0)
        out4 = torch.full((N, K), 1.0, dtype=F32, device=dev)
        T._call("eend_wgrad_bf16", dy, N, xb, K, (1 if f16 else 0) | 2, M, N, K, ws, ws.numel(), out4, K, K, 1.0, 0)
        assert (out4 == out).all()
        dyb = _to_blocked(dy, pad=1e-5)
        out5 = torch.full((N, K), 1.0, dtype=F32, device=dev)
        bias5 = torch.full((N,), 3.0, dtype=F32, device=dev)
        T._call("eend_wgrad_bias_bf16", dyb, N, x, K, (1 if f16 else 0) | 4, M, N, K, ws, ws.numel(), out5, K, K, bias5, 1.0, 0)
        assert (out5 == out2).all()
        if M % 16 == 0:                                           # (the bias column sums see the padding rows of a blocked dY: whole blocks only)
            assert (bias5 == bias).all()
    # transpose-detecting: an asymmetric case is already covered (N != K); accumulate + narrow destination
    if K == 384:
        dst = torch.ones(N, 345, dtype=F32, device=dev)
        T._call("eend_wgrad_bf16", dy, N, x, K, 1, M, N, K, ws, ws.numel(), dst, 345, 345, 2.0, 1)
        assert relnorm(dst, 2.0 * want[:, :345] + 1.0) < 2e-3


def test_colsum(T, dev, ws):
    gen = g(dev, 5)
    for M, N, bf in [(1000, 256, True), (3333, 768, True), (500, 2048, False)]:
        y = torch.randn(M, N, device=dev, generator=gen).to(BF16 if bf else F16)
        out = torch.zeros(N, dtype=F32, device=dev)
        T._call("eend_colsum_f32", y, N, M, N, 1 if bf else 0, ws, ws.numel(), out, 1.0, 0)
        assert rel(out, y.float().sum(0)) < 1e-5


def _conv_case(dev, seed, nseq=3, Tp=128, lens=(100, 128, 37)):
    gen = g(dev, seed)
    x = torch.randn(nseq, Tp, 256, device=dev, generator=gen)
    w = torch.randn(256, 256, 19, device=dev, generator=gen) / 70
    dy = torch.randn(nseq, Tp, 256, device=dev, generator=gen) * 1e-4
    return x, w, dy, list(lens)


@pytest.mark.parametrize("big", [False, True])
def test_conv1d_grads(T, dev, ws, big):
    # big: >= 16384 token rows -> the 256 x 256 output tile of wgrad.hip (one tap x all 256 input channels per k-tile)
    x, w, dy, lens = _conv_case(dev, 12, nseq=36, Tp=512, lens=tuple(512 - 13 * (i % 7) - (300 if i == 5 else 0) for i in range(36))) if big \
        else _conv_case(dev, 11)
    nseq, Tp = x.shape[0], x.shape[1]
    Tmax = max(lens)
    x16 = x.to(F16)
    dy[:, Tmax:] = 0
    dy16 = dy.to(BF16)
    il = torch.tensor(lens, dtype=I32, device=dev)
    tl = torch.full((nseq,), Tmax, dtype=I32, device=dev)
    # reference: truncate to ilen (zero beyond), conv over the Tmax frames
    xr = x16.float().clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    mask = (torch.arange(Tp, device=dev)[None, :] < il[:, None]).float()[..., None]
    y = Fn.conv1d((xr * mask)[:, :Tmax].transpose(1, 2), wr, padding=9).transpose(1, 2)
    (y * dy16.float()[:, :Tmax]).sum().backward()
    # weight gradient
    tmp = torch.empty(256 * 19 * 256, dtype=F32, device=dev)
    gw = torch.zeros(256, 256, 19, dtype=F32, device=dev)
    T._call("eend_conv1d_wgrad_bf16", dy16.view(-1, 256), x16.view(-1, 256), il, nseq, Tp, 256, 19, 9, ws, ws.numel(), tmp, gw)
    assert relnorm(gw, wr.grad) < 3e-3, relnorm(gw, wr.grad)
    # data gradient
    wd = w.permute(1, 2, 0).flip(1).reshape(256, 19 * 256).to(BF16).contiguous()      # [ci][tap'][co], tap' = 18 - tap
    gx = torch.full((nseq * Tp, 256), 9.0, dtype=F32, device=dev)
    T._call("eend_conv1d_dgrad_bf16", dy16.view(-1, 256), wd, tl, il, gx, nseq, Tp, 256, 19, 9)
    want = xr.grad
    assert relnorm(gx.view(nseq, Tp, 256), want) < 4e-3, relnorm(gx.view(nseq, Tp, 256), want)
    for b, l in enumerate(lens):
        assert (gx.view(nseq, Tp, 256)[b, l:] == 0).all()


def test_linear_res_ln_train(T, dev):
    from fs_eend_amd import ops
    gen = g(dev, 3)
    M, K = 333, 256
    a = torch.randn(M, K, device=dev, generator=gen).to(F16)
    w = (torch.randn(256, K, device=dev, generator=gen) / 16).to(F16)
    b = torch.randn(256, device=dev, generator=gen) * 0.1
    res = torch.randn(M, 256, device=dev, generator=gen)
    gm = 1 + 0.2 * torch.randn(256, device=dev, generator=gen)
    be = 0.1 * torch.randn(256, device=dev, generator=gen)
    o32, o16 = torch.empty(M, 256, device=dev), torch.empty(M, 256, dtype=F16, device=dev)
    xh, rs = torch.empty(M, 256, dtype=F16, device=dev), torch.empty(M, device=dev)
    T._call("eend_linear_res_ln_train_f16", a, K, w, K, b, res, 1.0, gm, be, 1e-5, o32, o16, xh, rs, M, K, None)
    s = a.float() @ w.float().t() + b + res
    mu, var = s.mean(-1, keepdim=True), s.var(-1, unbiased=False, keepdim=True)
    xhat = (s - mu) / torch.sqrt(var + 1e-5)
    assert (o32 - (xhat * gm + be)).abs().max() < 2e-4
    assert (xh.float() - xhat).abs().max() < 3e-3
    assert rel(rs, 1 / torch.sqrt(var + 1e-5).squeeze(-1)) < 1e-4
    r32, r16 = torch.empty_like(o32), torch.empty_like(o16)
    ops.linear_res_ln(a, w, b, res, gm, be, r32, r16)
    assert torch.equal(r32, o32) and torch.equal(r16, o16)        # same values as the inference epilogue


def test_layernorm_bwd(T, dev, ws):
    gen = g(dev, 4)
    M = 3001
    s = torch.randn(M, 256, device=dev, generator=gen) * 2 + 0.3
    gm = 1 + 0.2 * torch.randn(256, device=dev, generator=gen)
    be = torch.zeros(256, device=dev)
    gy = torch.randn(M, 256, device=dev, generator=gen) * 1e-5
    sr, gr, br = s.clone().requires_grad_(True), gm.clone().requires_grad_(True), be.clone().requires_grad_(True)
    (Fn.layer_norm(sr, (256,), gr, br, 1e-5) * gy).sum().backward()
    mu, var = s.mean(-1, keepdim=True), s.var(-1, unbiased=False, keepdim=True)
    xh = ((s - mu) / torch.sqrt(var + 1e-5)).to(F16)
    rs = (1 / torch.sqrt(var + 1e-5)).squeeze(-1).contiguous()
    gbuf = gy.clone()
    d16 = torch.empty(M, 256, dtype=BF16, device=dev)
    dg, db, dbias = torch.empty(256, device=dev), torch.empty(256, device=dev), torch.empty(256, device=dev)
    T._call("eend_layernorm_bwd_f32", gbuf, xh, rs, gm, gbuf, d16, ws, ws.numel(), dg, db, dbias, M, None)
    assert rel(dbias, sr.grad.sum(0)) < 2e-3
    assert rel(gbuf, sr.grad) < 2e-3
    assert rel(d16, sr.grad) < 6e-3
    assert rel(dg, gr.grad) < 2e-3 and rel(db, br.grad) < 1e-4


@pytest.mark.parametrize("M", [4133, 140000])
def test_wgrad_grouped(T, dev, ws, M):
    """eend_wgrad_bias_grouped_bf16 (round 6): the weight and bias gradients of four linear layers sharing one input (the q / k / v / g
    projections of a retention module), destinations equally spaced in one buffer, == four eend_wgrad_bias_bf16 calls on the column
    blocks of dY; the floats between the destinations stay untouched."""
    gen = g(dev, M)
    dy = (torch.randn(M, 1024, device=dev, generator=gen) * 1e-5).to(BF16)
    x = torch.randn(M, 256, device=dev, generator=gen).to(F16)
    stride = 256 * 256 + 256 + 64                                  # weight, bias, and a gap that must survive
    buf = torch.full((4 * stride,), 7.0, dtype=F32, device=dev)
    T._call("eend_wgrad_bias_grouped_bf16", dy, 1024, x, 256, 1, M, 1024, 256, ws, ws.numel(), buf, buf[256 * 256:], 256, stride, 1.0)
    torch.cuda.synchronize()
    for j in range(4):
        w1 = torch.empty(256, 256, dtype=F32, device=dev)
        b1 = torch.empty(256, dtype=F32, device=dev)
        T._call("eend_wgrad_bias_bf16", dy[:, j * 256:(j + 1) * 256], 1024, x, 256, 1, M, 256, 256, ws, ws.numel(), w1, 256, 256, b1, 1.0, 0)
        got_w = buf[j * stride:j * stride + 65536].view(256, 256)
        got_b = buf[j * stride + 65536:j * stride + 65536 + 256]
        assert relnorm(got_w, w1) < 1e-5 and rel(got_w, w1) < 1e-3, j
        assert rel(got_b, b1) < 1e-4, j
        assert (buf[j * stride + 65536 + 256:(j + 1) * stride] == 7.0).all()
    want = dy.float().t() @ x.to(BF16).float()
    assert relnorm(buf[:65536].view(256, 256), want[:256]) < 2e-3


@pytest.mark.parametrize("M,K,pdrop", [(3001, 768, 0.25), (777, 256, 0.0), (70001, 768, 0.1), (64, 1024, 0.0)])
def test_gemm_acc_layernorm_bwd_fused(T, dev, ws, M, K, pdrop):
    """eend_gemm_acc_lnbwd_bf16 (round 6: the LayerNorm backward of a post-norm site in the epilogue of the data-gradient GEMM that
    produces its input gradient) == eend_gemm_acc_bf16 followed by eend_layernorm_bwd_f32: same masks, same partial-sum semantics; in
    place on the f32 gradient stream."""
    import ctypes
    gen = g(dev, M + K)
    A = (torch.randn(M, K, device=dev, generator=gen) * 1e-4).to(BF16)
    W = (torch.randn(256, K, device=dev, generator=gen) / math.sqrt(K)).to(BF16)
    g0 = torch.randn(M, 256, device=dev, generator=gen) * 1e-4
    xh = torch.randn(M, 256, device=dev, generator=gen).to(F16)
    rs = 0.5 + torch.rand(M, device=dev, generator=gen)
    gm = 1 + 0.2 * torch.randn(256, device=dev, generator=gen)
    if pdrop > 0:
        spec, _d = _drop_spec(pdrop, 64, site=3)
        dr = ctypes.byref(spec)
    else:
        dr = None
    # the two launches
    g1 = g0.clone()
    T._call("eend_gemm_acc_bf16", A, K, W, K, g1, 1.0, g1, None, M, K)
    gsum = g1.clone()
    d16a = torch.empty(M, 256, dtype=BF16, device=dev)
    dga, dba, dbia = (torch.empty(256, device=dev) for _ in range(3))
    T._call("eend_layernorm_bwd_f32", g1, xh, rs, gm, g1, d16a, ws, ws.numel(), dga, dba, dbia, M, dr)
    # one launch, in place
    g2 = g0.clone()
    d16b = torch.full((M, 256), float("nan"), dtype=BF16, device=dev)
    dgb, dbb, dbib = (torch.full((256,), float("nan"), device=dev) for _ in range(3))
    T._call("eend_gemm_acc_lnbwd_bf16", A, K, W, K, g2, xh, rs, gm, g2, d16b, ws, ws.numel(), dgb, dbb, dbib, M, K, dr)
    torch.cuda.synchronize()
    assert torch.isfinite(g2).all() and torch.isfinite(d16b.float()).all()
    # dz cancels its two largest components: compare on the scale of its input gradient
    scale = float((gsum * gm).abs().max() * rs.max())
    assert float((g2 - g1).abs().max()) < 2e-5 * scale
    assert ((d16b == 0) == (d16a == 0)).all()
    assert float((d16b.float() - d16a.float()).abs().max()) < 1e-2 * float(d16a.float().abs().max())
    assert rel(dgb, dga) < 1e-4 and rel(dbb, dba) < 1e-4 and rel(dbib, dbia) < 1e-3
    # against float64 from the same operands
    g64 = A.double() @ W.double().t() + g0.double()
    dd = g64 * gm.double()
    c1, c2 = dd.mean(-1, keepdim=True), (dd * xh.double()).mean(-1, keepdim=True)
    dz = rs.double()[:, None] * (dd - c1 - xh.double() * c2)
    assert float((g2.double() - dz).abs().max()) < 1e-4 * scale
    assert rel(dgb, (g64 * xh.double()).sum(0).float()) < 2e-3 and rel(dbb, g64.sum(0).float()) < 2e-3
    # without the optional bias gradient
    g3 = g0.clone()
    T._call("eend_gemm_acc_lnbwd_bf16", A, K, W, K, g3, xh, rs, gm, g3, d16b, ws, ws.numel(), dgb, dbb, None, M, K, dr)
    assert torch.equal(g3, g2)


def want_mask_sum(dgen, grad, rows):
    return dgen.rows(grad, 5, rows).sum(0)


def test_sublayer_dropout_fwd_and_ln_bwd(T, dev, ws):
    """dropout1/dropout2 (FS model :147; merge_tfm_encoder.py:385,394,399): y = LN(res + drop(a W^T + b)); the backward
    masks the branch gradient (bf16) and leaves the residual-stream gradient (f32) alone."""
    import ctypes
    gen = g(dev, 31)
    M, K = 777, 256
    spec, dgen = _drop_spec(0.25, 64, site=5)
    a = torch.randn(M, K, device=dev, generator=gen).to(F16)
    w = (torch.randn(256, K, device=dev, generator=gen) / 16).to(F16)
    b = torch.randn(256, device=dev, generator=gen) * 0.1
    res = torch.randn(M, 256, device=dev, generator=gen)
    gm = 1 + 0.2 * torch.randn(256, device=dev, generator=gen)
    be = 0.1 * torch.randn(256, device=dev, generator=gen)
    o32, o16 = torch.empty(M, 256, device=dev), torch.empty(M, 256, dtype=F16, device=dev)
    xh, rs = torch.empty(M, 256, dtype=F16, device=dev), torch.empty(M, device=dev)
    T._call("eend_linear_res_ln_train_f16", a, K, w, K, b, res, 1.0, gm, be, 1e-5, o32, o16, xh, rs, M, K, ctypes.byref(spec))
    rows = torch.arange(M, device=dev)
    branch = dgen.rows(a.float() @ w.float().t() + b, 5, rows)
    sref = (branch + res).requires_grad_(True)
    y = Fn.layer_norm(sref, (256,), gm, be, 1e-5)
    assert (o32 - y.detach()).abs().max() < 3e-4
    gy = torch.randn(M, 256, device=dev, generator=gen) * 1e-5
    (y * gy).sum().backward()
    gbuf = gy.clone()
    d16 = torch.empty(M, 256, dtype=BF16, device=dev)
    dg, db = torch.empty(256, device=dev), torch.empty(256, device=dev)
    dbias = torch.empty(256, device=dev)
    T._call("eend_layernorm_bwd_f32", gbuf, xh, rs, gm, gbuf, d16, ws, ws.numel(), dg, db, dbias, M, ctypes.byref(spec))
    assert rel(dbias, want_mask_sum(dgen, sref.grad, rows)) < 3e-3
    assert rel(gbuf, sref.grad) < 3e-3                               # residual stream: un-masked
    want = dgen.rows(sref.grad, 5, rows)                             # branch: masked and scaled
    assert rel(d16, want) < 6e-3
    assert ((d16 == 0) == (want == 0)).all()


def test_ffn_hidden_dropout_fwd_bwd(T, dev):
    """the FFN's inner dropout (merge_tfm_encoder.py:398,613): h = drop(relu(x W1^T + b1)); backward: dz = scale * (dy W2)
    where h != 0."""
    import ctypes
    gen = g(dev, 32)
    M, F = 1000, 1024
    spec, dgen = _drop_spec(0.25, 64, site=4)
    x = torch.randn(M, 256, device=dev, generator=gen).to(F16)
    w1 = (torch.randn(F, 256, device=dev, generator=gen) / 16).to(F16)
    b1 = torch.randn(F, device=dev, generator=gen) * 0.1
    h = torch.empty(M, F, dtype=F16, device=dev)
    T._call("eend_linear_relu_train_f16", x, 256, w1, 256, b1, h, F, M, F, 256, ctypes.byref(spec))
    z = x.float() @ w1.float().t() + b1
    want = dgen.rows(torch.relu(z), 4, torch.arange(M, device=dev))
    assert rel(h, want) < 3e-3
    sure = z.abs() > 1e-2                                            # away from the ReLU kink the zero patterns agree
    assert ((h == 0) == (want == 0))[sure].all()
    dy = (torch.randn(M, 256, device=dev, generator=gen) * 1e-4).to(BF16)
    w2t = (torch.randn(F, 256, device=dev, generator=gen) / 16).to(BF16)
    out = torch.empty(M, F, dtype=BF16, device=dev)
    T._call("eend_gemm_relu_bwd_bf16", dy, 256, w2t, 256, h, F, out, F, M, F, 256, spec.scale)
    wantg = (dy.float() @ w2t.float().t()) * (h != 0) * spec.scale
    assert rel(out, wantg) < 6e-3
    assert (out[h == 0] == 0).all()


@pytest.mark.parametrize("impl", ["fused", "stream"])
@pytest.mark.parametrize("M,F,pdrop", [(1000, 1024, 0.25), (300, 2048, 0.0), (70001, 2048, 0.1), (64, 64, 0.25)])
def test_ffn_train_fused(T, dev, M, F, pdrop, impl):
    """eend_ffn_train_f16 (ffn.hip MODE 3, round 5): the FFN of a post-norm block in one launch == eend_linear_relu_train_f16 +
    eend_linear_res_ln_train_f16 with the same two dropout sites (same masks), and == torch on the kernels' masks: saved hidden activation
    (its zeros are the ReLU-and-dropout mask of the backward), LayerNorm output, normalised rows, 1/sigma; in place on the residual stream.
    impl "stream": the same operator on the packed weight stream (eend_ffn_train_stream_f16, ffn_train_stream.hip, round 6)."""
    import ctypes
    gen = g(dev, 33)
    x = torch.randn(M, 256, device=dev, generator=gen).to(F16)
    w1 = (torch.randn(F, 256, device=dev, generator=gen) / 16).to(F16)
    b1 = torch.randn(F, device=dev, generator=gen) * 0.1
    w2 = (torch.randn(256, F, device=dev, generator=gen) / math.sqrt(F)).to(F16)
    b2 = torch.randn(256, device=dev, generator=gen) * 0.1
    res = torch.randn(M, 256, device=dev, generator=gen)
    gm = 1 + 0.2 * torch.randn(256, device=dev, generator=gen)
    be = 0.1 * torch.randn(256, device=dev, generator=gen)
    if pdrop > 0:
        s1, d1 = _drop_spec(pdrop, 64, site=4)
        s2, d2 = _drop_spec(pdrop, 64, site=5)
        r1, r2 = ctypes.byref(s1), ctypes.byref(s2)
    else:
        r1 = r2 = None
    nan16 = lambda *sh: torch.full(sh, float("nan"), dtype=F16, device=dev)
    o32, o16, hid, xh, rs = torch.full((M, 256), float("nan"), device=dev), nan16(M, 256), nan16(M, F), nan16(M, 256), torch.full((M,), float("nan"), device=dev)
    _ffn_train_call(T, dev, impl, x, w1, b1, w2, b2, res, gm, be, o32, o16, hid, xh, rs, M, F, r1, r2)
    torch.cuda.synchronize()
    for t in (o32, o16, hid, xh, rs):
        assert torch.isfinite(t).all()
    z = x.float() @ w1.float().t() + b1
    sure = z.abs() > 1e-2                                            # away from the ReLU kink the zero patterns agree
    if F % 128 == 0:                                                 # the two launches it replaces (the tiled GEMM wants N % 128 == 0)
        h2 = torch.empty(M, F, dtype=F16, device=dev)
        p32, p16, pxh, prs = torch.empty(M, 256, device=dev), torch.empty(M, 256, dtype=F16, device=dev), torch.empty(M, 256, dtype=F16, device=dev), torch.empty(M, device=dev)
        T._call("eend_linear_relu_train_f16", x, 256, w1, 256, b1, h2, F, M, F, 256, r1)
        T._call("eend_linear_res_ln_train_f16", h2, F, w2, F, b2, res, 1.0, gm, be, 1e-5, p32, p16, pxh, prs, M, F, r2)
        assert ((hid == 0) == (h2 == 0))[sure].all()
        assert rel(hid, h2) < 2e-3                                   # (bias position in the f32 sum differs: a unit may round to the neighbouring f16)
        assert (o32 - p32).abs().max() < 3e-3 and (xh.float() - pxh.float()).abs().max() < 4e-3
        assert ((rs - prs).abs() / prs).max() < 2e-3
    # torch on the kernels' masks
    rows = torch.arange(M, device=dev)
    hh = torch.relu(z)
    if pdrop > 0:
        hh = d1.rows(hh, 4, rows)
    hh = hh.to(F16).float()
    y = hh @ w2.float().t() + b2
    if pdrop > 0:
        y = d2.rows(y, 5, rows)
    y = y + res
    want = Fn.layer_norm(y, (256,), gm, be, 1e-5)
    assert rel(hid, hh) < 3e-3
    assert ((hid == 0) == (hh == 0))[sure].all()
    assert (o32 - want).abs().max() < 3e-3 and (o16.float() - want).abs().max() < 6e-3
    mu, var = y.mean(-1, keepdim=True), y.var(-1, unbiased=False, keepdim=True)
    assert (xh.float() - (y - mu) / torch.sqrt(var + 1e-5)).abs().max() < 6e-3
    assert ((rs - 1 / torch.sqrt(var.squeeze(-1) + 1e-5)).abs() * torch.sqrt(var.squeeze(-1) + 1e-5)).max() < 2e-3
    # in place on the residual stream (out_f32 = res), as the training step calls it; deterministic
    res2 = res.clone()
    q16, qh, qxh, qrs = torch.empty_like(o16), torch.empty_like(hid), torch.empty_like(xh), torch.empty_like(rs)
    _ffn_train_call(T, dev, impl, x, w1, b1, w2, b2, res2, gm, be, res2, q16, qh, qxh, qrs, M, F, r1, r2)
    torch.cuda.synchronize()
    assert torch.equal(res2, o32) and torch.equal(q16, o16) and torch.equal(qh, hid) and torch.equal(qxh, xh) and torch.equal(qrs, rs)


def _attn_ref(q, k, v, delay, kv_len, scale, pdrop=None):
    """q,k,v (n,H,T,64) fp32 -> output (n,T,256), with the index-predicate mask."""
    Tq = q.shape[2]
    i = torch.arange(Tq, device=q.device)[:, None]
    j = torch.arange(Tq, device=q.device)[None, :]
    ok = ((j - i) <= delay) & (j < kv_len)
    s = (q @ k.transpose(-1, -2)) * scale
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, -1)
    if pdrop is not None:
        p = pdrop(p)
    o = p @ v
    return o.transpose(1, 2).reshape(q.shape[0], Tq, 256)


def _drop_spec(p_drop, Tp, site=3):
    """(ctypes eend_dropout or None, matching oracle mask generator or None)"""
    if not p_drop:
        return None, None
    from fs_eend_amd import lib as L
    from oracle import dropout_ref as DR
    d = DR.HashDropout(p_drop, 42, 7, Tp)
    return L.Dropout(DR.site_seed(d.base, site), d.thresh24, d.scale), d


@pytest.mark.parametrize("p_drop", [0.0, 0.2])
@pytest.mark.parametrize("nseq,Tv,delay,prescaled", [(3, 100, 0, True), (2, 500, 0, True), (2, 192, 2, False), (1, 640, 0, True),
                                                     (2, 130, 1000, True)])
def test_attn_fwd_lse_and_bwd(T, dev, nseq, Tv, delay, prescaled, p_drop):
    import ctypes
    from fs_eend_amd import ops
    spec, dgen = _drop_spec(p_drop, ops.frames_pad(Tv))
    dref = None if spec is None else ctypes.byref(spec)
    pdrop = None if spec is None else (lambda p: dgen.attn(p, 3))
    gen = g(dev, Tv + delay)
    Tp = ops.frames_pad(Tv)
    H = 4
    qt_ = torch.randn(nseq, H, Tp, 64, device=dev, generator=gen)
    k_ = torch.randn(nseq, H, Tp, 64, device=dev, generator=gen)
    v_ = torch.randn(nseq, H, Tp, 64, device=dev, generator=gen)
    c = ops.QSCALE_LOG2 if prescaled else 1.0
    q16 = (qt_ * c).to(BF16)
    k16, v16 = k_.to(BF16), v_.to(BF16)
    q_true = (q16.float() / c).requires_grad_(True)
    kr, vr = k16.float().requires_grad_(True), v16.float().requires_grad_(True)
    o_ref = _attn_ref(q_true, kr, vr, delay, Tv, 0.125, pdrop)
    dO = torch.randn(nseq, Tp, 256, device=dev, generator=gen) * 1e-4
    dO[:, Tv:] = 0
    dO16 = dO.to(BF16)
    (o_ref * dO16.float()).sum().backward()
    O = torch.empty(nseq * Tp, 256, dtype=F16, device=dev)
    lse = torch.empty(nseq * H * Tp, dtype=F32, device=dev)
    qT, kT, vT = (t.transpose(-1, -2).contiguous() for t in (q16, k16, v16))
    scale = ops.LN2 if prescaled else 0.125
    T._call("eend_attn_causal_lse_bf16", q16, k16, vT, O, lse, nseq, H, Tp, 256, delay, Tv, scale, dref)
    assert (O.view(nseq, Tp, 256)[:, :Tv].float() - o_ref[:, :Tv]).abs().max() < 2e-2
    # lse check (log2 domain)
    s2 = (q_true.detach() @ kr.detach().transpose(-1, -2)) * 0.125 * math.log2(math.e)
    i = torch.arange(Tp, device=dev)[:, None]
    j = torch.arange(Tp, device=dev)[None, :]
    ok = ((j - i) <= delay) & (j < Tv)
    want_lse = torch.logsumexp(s2.masked_fill(~ok, float("-inf")) * math.log(2), -1) / math.log(2)
    got_lse = lse.view(nseq, H, Tp)
    assert (got_lse[:, :, :Tv] - want_lse[:, :, :Tv]).abs().max() < 2e-2
    dot_ws = torch.empty(nseq * Tp * 256, dtype=BF16, device=dev)
    dh_ws = torch.empty(nseq * H * Tp, dtype=F32, device=dev)
    dqkv = torch.full((nseq * Tp, 768), 3.0, dtype=BF16, device=dev)
    sl = 1.0 if prescaled else 0.125 * math.log2(math.e)
    sk = ops.LN2 if prescaled else 0.125
    T._call("eend_attn_causal_bwd_bf16", q16, qT, k16, kT, v16, dO16.view(-1, 256), 256, O, 256, lse, dot_ws, dh_ws, dqkv, 768, nseq, H, Tp,
            delay, Tv, Tv, sl, 0.125, sk, dref)
    d = dqkv.view(nseq, Tp, 3, H, 64).float()
    for idx, (name, ref) in enumerate((("dq", q_true.grad), ("dk", kr.grad), ("dv", vr.grad))):
        got = d[:, :, idx].permute(0, 2, 1, 3)                    # (n,H,Tp,64)
        e = relnorm(got[:, :, :Tv], ref[:, :, :Tv])
        assert e < 1.5e-2, (name, e)
        assert rel(got[:, :, :Tv], ref[:, :, :Tv]) < 3e-2, name
        assert (got[:, :, Tv:] == 0).all(), name + " pad rows"


@pytest.mark.parametrize("p_drop", [0.0, 0.2])
@pytest.mark.parametrize("C", [1, 3, 6, 10])
def test_spk_attn_fwd_bwd(T, dev, C, p_drop):
    import ctypes
    gen = g(dev, C)
    B, Tp = 2, 64
    spec, dgen = _drop_spec(p_drop, Tp, site=2)
    dref = None if spec is None else ctypes.byref(spec)
    rows = B * C * Tp
    qkv = torch.randn(rows, 768, device=dev, generator=gen).to(F16)
    dO = (torch.randn(rows, 256, device=dev, generator=gen) * 1e-4).to(BF16)
    x = qkv.float().view(B, C, Tp, 3, 4, 64).requires_grad_(True)
    q, k, v = (x[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))            # (B,Tp,H,C,64)
    p = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, -1)
    if dgen is not None:                                          # (B,Tp,H,C,C) -> the oracle's (B*T, H, C, C) with T = Tp
        p = dgen.spk(p.reshape(B * Tp, 4, C, C), 2, B, Tp).reshape(B, Tp, 4, C, C)
    o = (p @ v).permute(0, 3, 1, 2, 4).reshape(rows, 256)
    (o * dO.float()).sum().backward()
    ofwd = torch.empty(rows, 256, dtype=F16, device=dev)
    T._call("eend_spk_attn_train_f16", qkv, ofwd, B, C, Tp, 4, 0.125, dref)
    assert relnorm(ofwd, o.detach()) < 3e-3
    out = torch.empty(rows, 768, dtype=BF16, device=dev)
    T._call("eend_spk_attn_bwd_bf16", qkv, dO, out, B, C, Tp, 4, 0.125, dref)
    want = x.grad.reshape(rows, 768)
    assert relnorm(out, want) < 6e-3, relnorm(out, want)


def test_head_bce(T, dev, ws):
    gen = g(dev, 6)
    B, Tv, Tp, C = 3, 100, 128, 5
    emb = Fn.normalize(torch.randn(B, Tp, 256, device=dev, generator=gen), dim=-1)
    attr = torch.randn(B * C * Tp, 256, device=dev, generator=gen) * 3
    ilens, ncols = [100, 77, 50], [5, 3, 4]
    lab = (torch.rand(B, Tv, C, device=dev, generator=gen) < 0.3).float()
    il, nc = torch.tensor(ilens, dtype=I32, device=dev), torch.tensor(ncols, dtype=I32, device=dev)
    n_frames = sum(ilens)
    er = emb.clone().requires_grad_(True)
    ar = attr.clone().requires_grad_(True)
    a4 = ar.view(B, C, Tp, 256)
    an = a4 / a4.norm(dim=-1, keepdim=True)
    logit = (er[:, None] * an).sum(-1).permute(0, 2, 1)                            # (B,Tp,C)
    loss = 0
    for b in range(B):
        y, t = logit[b, :ilens[b], :ncols[b]], lab[b, :ilens[b], :ncols[b]]
        loss = loss + Fn.binary_cross_entropy_with_logits(y, t) * ilens[b]
    loss = loss / n_frames
    loss.backward()
    logits = torch.zeros(B, Tv, C, device=dev)
    da, de = torch.full((B * C * Tp, 256), 7.0, device=dev), torch.full((B * Tp, 256), 7.0, device=dev)
    lo = torch.zeros(1, device=dev)
    T._call("eend_head_bce_f32", emb, attr, lab, il, nc, 1.0 / n_frames, None, logits, da, de, ws, ws.numel(), lo, B, Tv, Tp, C)
    assert abs(lo.item() - loss.item()) < 1e-5
    assert (logits - logit[:, :Tv].detach()).abs().max() < 1e-5
    assert rel(da, ar.grad) < 1e-4 and rel(de.view(B, Tp, 256), er.grad) < 1e-4
    assert (da.view(B, C, Tp, 256)[:, :, Tv:] == 0).all() and (de.view(B, Tp, 256)[:, Tv:] == 0).all()
    # external d loss / d logits (the autograd drop-in path): same gradients from the gradient autograd would hand over
    lg = logit[:, :Tv].detach().clone().requires_grad_(True)
    l2 = sum(Fn.binary_cross_entropy_with_logits(lg[b, :ilens[b], :ncols[b]], lab[b, :ilens[b], :ncols[b]]) * ilens[b] for b in range(B)) / n_frames
    l2.backward()
    da2, de2 = torch.full_like(da, 3.0), torch.full_like(de, 3.0)
    T._call("eend_head_bce_f32", emb, attr, None, None, None, 0.0, lg.grad.contiguous(), None, da2, de2, ws, ws.numel(), lo, B, Tv, Tp, C)
    assert rel(da2, ar.grad) < 1e-4 and rel(de2.view(B, Tp, 256), er.grad) < 1e-4


def test_l2norm_bwd(T, dev):
    gen = g(dev, 7)
    B, Tv, Tp = 2, 50, 64
    y = torch.randn(B * Tp, 256, device=dev, generator=gen) * 2
    yr = y.clone().requires_grad_(True)
    dy = torch.randn(B * Tp, 256, device=dev, generator=gen) * 1e-4
    dy.view(B, Tp, 256)[:, Tv:] = 0
    e = yr / yr.norm(dim=-1, keepdim=True)
    (e * dy).sum().backward()
    out = torch.empty(B * Tp, 256, dtype=BF16, device=dev)
    T._call("eend_l2norm_bwd_bf16", e.detach().contiguous(), dy, (1 / y.norm(dim=-1)).contiguous(), out, B, Tv, Tp)
    assert rel(out, yr.grad) < 6e-3


def test_convert_bwd_and_const(T, dev, ws):
    gen = g(dev, 8)
    B, Tp, C = 3, 64, 6
    g0 = torch.randn(B * C * Tp, 256, device=dev, generator=gen) * 1e-4
    gsum = torch.empty(B * Tp, 256, dtype=BF16, device=dev)
    dpc = torch.empty(C, 256, device=dev)
    T._call("eend_convert_fanout_bwd_f32", g0, gsum, ws, ws.numel(), dpc, B, Tp, C)
    g4 = g0.view(B, C, Tp, 256)
    assert rel(gsum.view(B, Tp, 256), g4.sum(1)) < 6e-3
    assert rel(dpc, g4.sum((0, 2))) < 1e-4
    W = torch.randn(256, 512, device=dev, generator=gen) / 20
    bias = torch.randn(256, device=dev, generator=gen)
    pe = torch.randn(50, 256, device=dev, generator=gen)
    pc = torch.empty(C, 256, device=dev)
    T._call("eend_convert_const_f32", 0, W, bias, pe, pc, None, None, None, C)
    assert rel(pc, pe[:C] @ W[:, 256:].t() + bias) < 1e-5
    dW = torch.zeros(256, 512, device=dev)
    dbias = torch.zeros(256, device=dev)
    T._call("eend_convert_const_f32", 1, None, None, pe, None, dpc, dW, dbias, C)
    assert rel(dW[:, 256:], dpc.t() @ pe[:C]) < 1e-5 and (dW[:, :256] == 0).all()
    assert rel(dbias, dpc.sum(0)) < 1e-5


def test_bn_train_stats_and_bwd(T, dev, ws):
    gen = g(dev, 9)
    F, Tp = 345, 128
    lens = [100, 64, 7]
    Tv = max(lens)
    xs = [torch.randn(l, F, device=dev, generator=gen) * 2 - 3 for l in lens]
    B = len(xs)
    ptrs = torch.tensor([x.data_ptr() for x in xs], dtype=torch.int64, device=dev)
    ln = torch.tensor(lens, dtype=I32, device=dev)
    mean, var = torch.empty(F, device=dev), torch.empty(F, device=dev)
    rm, rv = torch.randn(F, device=dev, generator=gen), torch.rand(F, device=dev, generator=gen) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    T._call("eend_bn_train_stats_f32", ptrs, ln, -1.0, ws, ws.numel(), mean, var, rm, rv, 0.1, B, Tv, F)
    xp = torch.nn.utils.rnn.pad_sequence(xs, batch_first=True, padding_value=-1.0)             # (B,Tv,F)
    flat = xp.reshape(-1, F)
    assert (mean - flat.mean(0)).abs().max() < 1e-5 and rel(var, flat.var(0, unbiased=False)) < 1e-5
    n = flat.shape[0]
    assert (rm - (0.9 * rm0 + 0.1 * flat.mean(0))).abs().max() < 1e-5
    assert rel(rv, 0.9 * rv0 + 0.1 * flat.var(0, unbiased=True)) < 1e-5
    dy = torch.zeros(B * Tp, 384, device=dev)
    dy.view(B, Tp, 384)[:, :Tv, :F] = torch.randn(B, Tv, F, device=dev, generator=gen) * 1e-4
    dy16 = dy.to(BF16)
    dg, db = torch.empty(F, device=dev), torch.empty(F, device=dev)
    T._call("eend_bn_bwd_f32", ptrs, ln, -1.0, mean, var, 1e-5, dy16, 384, ws, ws.numel(), dg, db, B, Tv, Tp, F)
    xhat = (xp - flat.mean(0)) / torch.sqrt(flat.var(0, unbiased=False) + 1e-5)
    d3 = dy16.float().view(B, Tp, 384)[:, :Tv, :F]
    assert rel(dg, (d3 * xhat).sum((0, 1))) < 1e-4 and rel(db, d3.sum((0, 1))) < 1e-4


def test_emb_consistency_bwd(T, dev):
    from oracle import fs_eend_ref as R                         # checker only
    gen = g(dev, 10)
    B, Tv, Tp, C = 2, 150, 192, 4
    y = torch.randn(B, Tp, 256, device=dev, generator=gen)
    yr = y.clone().requires_grad_(True)
    e = yr / yr.norm(dim=-1, keepdim=True)
    lab = (torch.rand(B, Tv, C, device=dev, generator=gen) < 0.4).float()
    loss = R.emb_consistency_loss(e[:, :Tv], lab)
    loss.backward()
    e_det = e.detach().contiguous()
    de = torch.zeros(B * Tp, 256, device=dev)
    T._call("eend_emb_consistency_bwd_f16", e_det.to(F16), lab, None, 0.0, de, B, Tv, Tp, 256, C)
    # compare after the L2-norm projection (the only way this gradient is consumed)
    inv = (1 / y.norm(dim=-1)).reshape(-1).contiguous()
    out = torch.empty(B * Tp, 256, dtype=BF16, device=dev)
    T._call("eend_l2norm_bwd_bf16", e_det.view(-1, 256), de, inv, out, B, Tv, Tp)
    want = yr.grad.view(-1, 256)
    assert relnorm(out, want) < 1e-2, relnorm(out, want)


def test_adam_and_sumsq(T, dev, ws):
    gen = g(dev, 12)
    n = 100003
    p0 = torch.randn(n, device=dev, generator=gen)
    p = p0.clone()
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1.0, betas=(0.9, 0.98), eps=1e-9)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    hp = torch.zeros(4, device=dev)
    ss = torch.zeros(1, device=dev)
    for step in range(1, 4):
        gr = torch.randn(n, device=dev, generator=gen) * (0.5 if step == 1 else 0.001)
        T._call("eend_grad_sumsq_f32", gr, n, ws, ws.numel(), ss)
        assert abs(ss.item() - (gr.double() ** 2).sum().item()) < 1e-4 * ss.item()
        lr = 1e-3 * step
        ref.grad = gr.clone()
        total = torch.nn.utils.clip_grad_norm_([ref], 5.0)
        for gq in opt.param_groups:
            gq["lr"] = lr
        opt.step()
        hp.copy_(torch.tensor([lr, 1 - 0.9 ** step, 1 - 0.98 ** step, 5.0]))
        T._call("eend_adam_step_f32", p, gr, m, v, n, hp, ss, 0.9, 0.98, 1e-9)
        assert (p - ref.detach()).abs().max() < 2e-6, step


def test_prep_weights_table(T, dev):
    from fs_eend_amd import lib as L
    gen = g(dev, 13)
    src = torch.randn(256 * 19 * 40 + 7, device=dev, generator=gen)
    w = src[7:7 + 40 * 256 * 19].view(40, 256, 19)                          # (co, ci, tap)
    d1 = torch.zeros(256, 19 * 40, dtype=BF16, device=dev)                   # [ci][tap'][co] = w[co][ci][18 - tap']
    d2 = torch.zeros(40, 384, dtype=F16, device=dev)                          # first 300 of a 4864-wide row, first 8 rows x 2
    ents = []
    for dst, dims, strides, off, dt, cpad, ns, sc in ((d1, (256, 19, 40), (19, -1, 256 * 19), 7 + 18, 1, 40, 0, 1.0),
                                                       (d2, (40, 1, 300), (256 * 19, 0, 1), 7, 0, 384, 8, 2.0)):
        e = L.PrepEntry()
        e.src, e.off, e.dst = src.data_ptr(), off, dst.data_ptr()
        e.A, e.B, e.C, e.Cpad = dims[0], dims[1], dims[2], cpad
        e.sa, e.sb, e.sc = strides
        e.dtype, e.nscale, e.scale, e.reserved = dt, ns, sc, 0
        ents.append(e)
    arr = (L.PrepEntry * 2)(*ents)
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    T._call("eend_prep_weights", tab, 2)
    want1 = w.permute(1, 2, 0).flip(1).reshape(256, 19 * 40)
    assert torch.equal(d1, want1.to(BF16))
    flat = w.reshape(40, -1)[:, :300].clone()
    flat[:8] *= 2.0
    assert torch.equal(d2[:, :300], flat.to(F16)) and (d2[:, 300:] == 0).all()


@pytest.mark.parametrize("M,K", [(192, 768), (1000, 1024), (4096 + 48, 256), (131, 768), (50000, 768), (30000, 2048)])
def test_gemm_acc_stream(hip_lib, dev, M, K):
    """g += A Wt^T on the packed weight stream (gemm_acc_stream.hip) against float64 and against eend_gemm_acc_bf16 on the same operands;
    ragged M (rows beyond M neither read into the result nor written), both tile heights"""
    from fs_eend_amd import train as T, lib as L
    g_ = torch.Generator().manual_seed(M + K)
    a = (torch.randn(M, K, generator=g_) * 0.5).to(dev).to(BF16)
    wt = (torch.randn(256, K, generator=g_) / 16).to(dev).to(BF16)
    g0 = torch.randn(M, 256, generator=g_).to(dev)
    n = L.load().eend_gemm_acc_stream_elems(K)
    assert n > 0 and L.load().eend_gemm_acc_stream_ok(M, K, K)
    ws = torch.empty(n, dtype=BF16, device=dev)
    T._call("eend_gemm_acc_stream_pack_bf16", wt, K, ws, K)
    pad = torch.full((M + 256, 256), float("nan"), dtype=torch.float32, device=dev)
    pad[:M] = g0
    T._call("eend_gemm_acc_stream_bf16", a, K, ws, pad, M, K)
    want = g0.double() + a.double() @ wt.double().t()
    assert torch.isnan(pad[M:]).all()
    err = (pad[:M].double() - want).abs().max().item()
    g1 = g0.clone()
    T._call("eend_gemm_acc_bf16", a, K, wt, K, g1, 1.0, g1, None, M, K)
    err_old = (g1.double() - want).abs().max().item()
    print(f"stream vs float64 {err:.2e} (tiled GEMM {err_old:.2e})")
    assert err < 2e-4 * (K / 256) ** 0.5 + 1e-5
    assert not L.load().eend_gemm_acc_stream_ok(M, 320, 320) and not L.load().eend_gemm_acc_stream_ok(M, 4096, 4096)


@pytest.mark.parametrize("p_drop", [0.0, 0.2])
@pytest.mark.parametrize("nseq,Tp,Tv,delay", [(3, 128, 100, 0), (2, 512, 500, 0), (2, 192, 192, 2), (5, 320, 300, 0), (2, 512, 470, 1000), (40, 512, 500, 0)])
def test_inproj_attn_train_fused(T, dev, nseq, Tp, Tv, delay, p_drop):
    """eend_inproj_attn_train_bf16 (attn_stream.hip TRAIN: in-projection + attention of the training forward in one launch) against the two
    launches it replaces: the same bf16 Q / K / V head rows and lse for the backward, the same dropout masks on the probabilities, context
    rows within the rounding of two different bf16 score paths"""
    import ctypes
    from fs_eend_amd import ops, lib as L
    gen = g(dev, nseq * 7 + Tp + delay)
    M, n = nseq * Tp, nseq * Tp * 256
    x = torch.randn(M, 256, device=dev, generator=gen).to(F16)
    w = (torch.randn(768, 256, device=dev, generator=gen) / 16)
    b = torch.randn(768, device=dev, generator=gen) * 0.1
    w[:256] *= ops.QSCALE_LOG2
    b[:256] *= ops.QSCALE_LOG2
    w = w.to(F16)
    spec, _d = _drop_spec(p_drop, Tp, site=2)
    dr = ctypes.byref(spec) if spec is not None else None
    q1, k1, v1, vt1 = (torch.empty(n, dtype=BF16, device=dev) for _ in range(4))
    ctx1 = torch.empty(M, 256, dtype=F16, device=dev)
    lse1 = torch.empty(nseq * 4 * Tp, device=dev)
    T._call("eend_inproj_heads_train_bf16", x, 256, w, b, q1, None, k1, None, v1, vt1, nseq, Tp, 4)
    T._call("eend_attn_causal_lse_bf16", q1, k1, vt1, ctx1, lse1, nseq, 4, Tp, 256, delay, Tv, ops.LN2, dr)
    wp = ops.inproj_attn_pack(w)
    q2, k2, v2 = (torch.full((n,), float("nan"), dtype=BF16, device=dev) for _ in range(3))
    ctx2 = torch.full((M, 256), float("nan"), dtype=F16, device=dev)
    lse2 = torch.full((nseq * 4 * Tp,), float("nan"), device=dev)
    T._call("eend_inproj_attn_train_bf16", x, 256, wp, b, ctx2, 256, q2, k2, v2, lse2, nseq, 4, Tp, delay, Tv, dr)
    torch.cuda.synchronize()
    for a, c, name in ((q1, q2, "q"), (k1, k2, "k"), (v1, v2, "v")):
        assert torch.isfinite(c.float()).all(), name
        assert float((a.float() - c.float()).abs().max()) < 3e-2, name            # one bf16 ulp at |y| <= 4 between two accumulation orders
    assert torch.isfinite(lse2).all() and torch.isfinite(ctx2.float()).all()
    assert float((lse1 - lse2).abs().max()) < 3e-2
    assert float((ctx1.float() - ctx2.float()).abs().max()) < 3e-2
    if p_drop:
        # the same mask: a probability row that the two-launch path dropped entirely from a context feature is dropped here too (spot check
        # through the context of the first frames, whose single visible key is either kept or not)
        first = ctx2.view(nseq, Tp, 4, 64)[:, 0].float().abs().sum(-1) == 0
        first1 = ctx1.view(nseq, Tp, 4, 64)[:, 0].float().abs().sum(-1) == 0
        if delay == 0:
            assert torch.equal(first, first1)
    with pytest.raises(L.EendHipError):
        T._call("eend_inproj_attn_train_bf16", x, 256, wp, b, ctx2, 256, q2, k2, v2, lse2, 1, 4, 576, delay, 500, dr)
