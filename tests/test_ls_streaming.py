"""GPU: LS-EEND frame-by-frame streaming through the one-step API, driven exactly as the
reference's LS-EEND/streaming_infer_dia.py:52-97 drives it, against (a) the reference's own
streaming logits (golden) and (b) the one-step kernels' fp32 oracles."""
import pytest
import torch

from oracle import fixtures as FX
from oracle import ls_eend_ref as R
from tests.helpers import build_ls_mirror, max_abs

pytestmark = pytest.mark.gpu
F16, F32 = torch.float16, torch.float32


def rnd(shape, dev, seed, dtype=F32, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


def test_retention_step_kernel(hip_lib, dev):
    from fs_eend_amd import ops
    N, H, D = 5, 4, 256
    kv = torch.zeros(N, H, 64, 64, dtype=F32, device=dev)
    s_a, s_b = torch.zeros(H, device=dev), torch.empty(H, device=dev)
    state = {}
    for t in range(6):
        qkvg = rnd((N, 4 * D), dev, 100 + t, F16)
        out = torch.empty((N, D), dtype=F16, device=dev)
        ops.retention_step(qkvg, kv, s_a, s_b, out, N, H, 1e-6)
        s_a, s_b = s_b, s_a
        x = qkvg.float().cpu()
        qh = x[:, :D].reshape(N, 1, H, 64).transpose(1, 2)
        kh = x[:, D:2 * D].reshape(N, 1, H, 64).transpose(1, 2)
        o = R.retention_step(qh, kh, x[:, 2 * D:3 * D].reshape(N, 1, D), state)[:, None]        # (N,1,H,64)
        o = R.layer_norm(o, None, None, 1e-6).reshape(N, D)
        want = R.swish(x[:, 3 * D:]) * o
        assert max_abs(out, want) < 6e-3, f"step {t}"
        assert max_abs(kv, state["prev_key_value"]) < 1e-5
        assert max_abs(s_a, state["scale"]) == 0


def test_dwconv_step_kernel(hip_lib, dev):
    from fs_eend_amd import ops
    B, D, k = 3, 256, 16
    w = rnd((D, k), dev, 120) * 0.3
    bn = (rnd((D,), dev, 121) * 0.2 + 1, rnd((D,), dev, 122) * 0.1, rnd((D,), dev, 123) * 0.1, rnd((D,), dev, 124).abs() + 0.5)
    cache = torch.zeros(B, D, k - 1, dtype=F32, device=dev)
    ref_cache = torch.zeros(B, D, k - 1)
    for t in range(20):
        x = rnd((B, D), dev, 130 + t, F16)
        out = torch.empty((B, D), dtype=F16, device=dev)
        ops.dwconv_step(x, cache, w, bn, out, 1e-5)
        xp = torch.cat([ref_cache, x.float().cpu()[:, :, None]], dim=2)
        ref_cache = xp[:, :, 1:]
        y = (xp * w.cpu()).sum(-1)
        y = (y - bn[2].cpu()) / torch.sqrt(bn[3].cpu() + 1e-5) * bn[0].cpu() + bn[1].cpu()
        assert max_abs(out, y * torch.sigmoid(y)) < 3e-3
        assert max_abs(cache, ref_cache) < 1e-6


def _drive(m, src, C, dev):
    """LS-EEND/streaming_infer_dia.py:52-97 with the product's model / StreamingConv1d."""
    from fs_eend_amd.ls_model import StreamingConv1d
    scnn = StreamingConv1d(m.n_units, m.n_units, kernel_size=2 * m.delay + 1).to(dev)
    scnn.conv.load_state_dict(m.cnn.state_dict())
    scnn.eval()
    n_enc, n_dec = len(m.enc.encoder.layers), len(m.dec.layers)
    ret_states = [dict() for _ in range(n_enc)]
    caches = [torch.zeros(1, m.n_units, m.enc.encoder._conv_kernel_size - 1, device=dev) for _ in range(n_enc)]
    dec_states = [dict() for _ in range(n_dec)]
    scnn.buffer.clear()
    scnn.t = 0
    preds, dec_t = [], 0

    def step(emb_t, dec_t):
        e = scnn(emb_t.transpose(1, 2))
        if e is None:
            return None, dec_t
        e = e.transpose(1, 2)
        e = e / torch.norm(e, dim=-1, keepdim=True)
        a = m.dec.forward_one_step(e, dec_t, C, dec_states)
        a = a / torch.norm(a, dim=-1, keepdim=True)
        return torch.matmul(e.unsqueeze(-2), a.transpose(-1, -2)).squeeze(-2), dec_t + 1

    for t in range(src.shape[0]):
        e = m.enc.forward_one_step(src[t:t + 1].unsqueeze(0), t, ret_states, caches)
        y, dec_t = step(e, dec_t)
        if y is not None:
            preds.append(y)
    for _ in range(m.delay):
        y, dec_t = step(torch.zeros(1, 1, m.n_units, device=dev), dec_t)
        if y is not None:
            preds.append(y)
    return torch.cat(preds, dim=1).squeeze(0), ret_states


def test_ls_streaming_vs_reference_streaming(hip_lib, dev):
    meta, arr = FX.load_case("ls_stream_T120")
    m = build_ls_mirror(meta).to(dev)
    src = FX.make_src([meta["T"]], meta["in_size"], meta["xseed"])[0].to(dev)
    ys, states = _drive(m, src, meta["C"], dev)
    assert ys.shape == arr["stream_logits"].shape
    err = max_abs(ys, arr["stream_logits"])
    print(f"streaming logits vs reference streaming: {err:.2e}")
    assert err < 1e-3
    # O(1) state: (B,H,64,64) f32 + (H,) per layer, whatever the stream length
    assert states[0]["prev_key_value"].shape == (1, 4, 64, 64) and states[0]["scale"].shape == (4,)
    assert float(states[0]["scale"][0]) == meta["T"]
    # the batch (chunk-recurrent) HIP forward agrees with HIP streaming as well as the reference's two forms do
    batch = m.test([src], [meta["T"]], meta["C"])[0][0]
    assert max_abs(ys, batch.cpu()) < 5e-3


@pytest.mark.parametrize("use_graph", [False, True])
def test_ls_stream_session(hip_lib, dev, use_graph):
    """The device-resident streaming session (three hipGraph replays per frame) against the reference's streaming
    logits and against the eager one-step driver."""
    from fs_eend_amd.ls_stream import LsStreamSession
    meta, arr = FX.load_case("ls_stream_T120")
    m = build_ls_mirror(meta).to(dev)
    src = FX.make_src([meta["T"]], meta["in_size"], meta["xseed"])[0].to(dev)
    want, _ = _drive(m, src, meta["C"], dev)
    sess = LsStreamSession(m, meta["C"], batch=1, use_graph=use_graph)
    for rep in range(2):                                   # a session is reusable after reset()
        ys = []
        for t in range(src.shape[0]):
            y = sess.push(src[t:t + 1])
            assert (y is None) == (t < m.delay)
            if y is not None:
                ys.append(y)
        ys += sess.flush()
        ys = torch.cat(ys, dim=1).squeeze(0)
        assert ys.shape == arr["stream_logits"].shape
        assert max_abs(ys, arr["stream_logits"]) < 1e-3
        assert max_abs(ys, want.cpu()) < 3e-4
        assert float(sess.enc_states[0]["scale"][0]) == meta["T"]
        sess.reset()


def test_stream_sessions_survive_a_weight_refresh(hip_lib, dev):
    """the captured graphs point into the model's operand copies; after load_state_dict (which rebuilds them)
    the sessions must capture again instead of replaying over freed memory -- same weights reloaded mid-stream must not
    change a single output bit, for both flavours."""
    import gc
    from fs_eend_amd.fs_stream import FsStreamSession, StreamingTransformerEDADiarization, copy_params_from_masked_to_streaming
    from fs_eend_amd.ls_stream import LsStreamSession
    from tests.helpers import build_fs_mirror
    meta, _ = FX.load_case("ls_stream_T120")
    m = build_ls_mirror(meta).to(dev)
    src = FX.make_src([meta["T"]], meta["in_size"], meta["xseed"])[0].to(dev)

    def run_ls(refresh):
        s = LsStreamSession(m, meta["C"])
        out = []
        for t in range(60):
            if refresh and t == 30:
                m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
                gc.collect()
                torch.cuda.empty_cache()
                junk = torch.full((64 * 1024 * 1024,), float("nan"), device=dev)      # scribble over whatever was freed
                del junk
            y = s.push(src[t])
            if y is not None:
                out.append(y)
        return torch.cat(out)

    assert torch.equal(run_ls(False), run_ls(True))
    fmeta, _ = FX.load_case("fs_stream_T60")
    fm = build_fs_mirror(fmeta).to(dev)
    sm = StreamingTransformerEDADiarization(in_size=fmeta["in_size"], **fmeta["cfg"]).eval().to(dev)
    copy_params_from_masked_to_streaming(fm, sm)
    fsrc = FX.make_src([fmeta["T"]], fmeta["in_size"], fmeta["xseed"])[0].to(dev)

    def run_fs(refresh):
        s = FsStreamSession(sm, fmeta["C"], cap=64)
        out = []
        for t in range(50):
            if refresh and t == 25:
                sm.load_state_dict({k: v.clone() for k, v in sm.state_dict().items()})
                gc.collect()
                torch.cuda.empty_cache()
                junk = torch.full((64 * 1024 * 1024,), float("nan"), device=dev)
                del junk
            y = s.push(fsrc[t])
            if y is not None:
                out.append(y)
        return torch.cat(out)

    assert torch.equal(run_fs(False), run_fs(True))


@pytest.mark.parametrize("N,C", [(10, 10), (1, 1), (16, 8), (6, 3), (640, 10), (64, 1), (200, 10), (33, 3), (17, 1)])
def test_f32_frame_step_entries_vs_torch(hip_lib, dev, N, C):
    from fs_eend_amd import ops
    """The f32 pieces of the all-f32 decoder frame step (f32 activations AND weights) against plain torch fp32:
    linear (+ReLU), linear + residual + LayerNorm (K = 256 and the split-K K = 2048 path), speaker-axis attention."""
    g = torch.Generator().manual_seed(100 + N)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x, w1, b1 = rn(N, 256), rn(2048, 256, sc=0.06), rn(2048, sc=0.1)
    h = torch.empty(N, 2048, device=dev)
    ops.linear_step_f32(x, w1, b1, h, act=ops.ACT_RELU)
    want_h = torch.relu(x.double() @ w1.double().t() + b1.double())
    assert float((h.double() - want_h).abs().max()) < 2e-6 * float(want_h.abs().max() + 1)
    w2, b2, res = rn(256, 2048, sc=0.02), rn(256, sc=0.1), rn(N, 256)
    gam, bet = rn(256, sc=0.3) + 1.0, rn(256, sc=0.1)
    out = res.clone()
    o16 = torch.empty(N, 256, dtype=torch.float16, device=dev)
    ops.linear_res_ln_step_f32(h, w2, b2, out, gam, bet, out, eps=1e-5, alpha=0.5, out16=o16)      # in place on the residual stream
    pre = (want_h @ w2.double().t() + b2.double()) * 0.5 + res.double()
    want = torch.nn.functional.layer_norm(pre, (256,), gam.double(), bet.double(), 1e-5)
    assert float((out.double() - want).abs().max()) < 5e-6 * float(want.abs().max())
    assert float((o16.double() - want).abs().max()) < 2e-3 * float(want.abs().max())
    # pre-norm join: un-normalised stream + LayerNorm of it (f32 and f16 copies); plain row LayerNorm
    stream = res.clone()
    ln32, ln16 = torch.empty(N, 256, device=dev), torch.empty(N, 256, dtype=torch.float16, device=dev)
    ops.linear_res_scale_ln_step_f32(h, w2, b2, stream, 0.5, gam, bet, stream, ln_out32=ln32, ln_out16=ln16, eps=1e-5)
    assert float((stream.double() - pre).abs().max()) < 5e-6 * float(pre.abs().max())
    assert float((ln32.double() - want).abs().max()) < 5e-6 * float(want.abs().max())
    assert float((ln16.double() - want).abs().max()) < 2e-3 * float(want.abs().max())
    ln_only = torch.empty(N, 256, device=dev)
    ops.layernorm_rows_f32(stream, gam, bet, ln_only, 1e-5)
    assert torch.equal(ln_only, ln32)
    wq, bq = rn(768, 256, sc=0.06), rn(768, sc=0.1)
    qkv = torch.empty(N, 768, device=dev)
    ops.linear_step_f32(x, wq, bq, qkv)
    assert float((qkv.double() - (x.double() @ wq.double().t() + bq.double())).abs().max()) < 1e-5
    B = N // C
    att = torch.empty(N, 256, device=dev)
    ops.spk_attn_step_f32(qkv, att, B, C)
    q, k, v = (t.double().view(B, C, 4, 64).transpose(1, 2) for t in qkv.split(256, dim=1))
    p = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    want_a = (p @ v).transpose(1, 2).reshape(N, 256)
    assert float((att.double() - want_a).abs().max()) < 1e-5


@pytest.mark.parametrize("nstreams", [3, 20])
def test_ls_stream_session_several_streams_match_single_streams(hip_lib, dev, nstreams):
    """A session with several concurrent streams (3 x C = 30 and 20 x C = 200 rows per frame: several 16-row groups of the
    all-f32 frame steps, round 4) gives every stream what a one-stream session gives it -- the same f32 arithmetic per row, so
    to fp32 summation-order level -- and the streams do not leak into each other (stream 0 is fed the same input in both)."""
    from fs_eend_amd.ls_stream import LsStreamSession
    meta, arr = FX.load_case("ls_stream_T120")
    m = build_ls_mirror(meta).to(dev)
    C, T = meta["C"], 70
    g = torch.Generator().manual_seed(nstreams)
    base = FX.make_src([meta["T"]], meta["in_size"], meta["xseed"])[0][:T]
    srcs = torch.stack([base] + [base + 0.3 * torch.randn(base.shape, generator=g) for _ in range(nstreams - 1)]).to(dev)   # (S, T, F)
    multi = LsStreamSession(m, C, batch=nstreams)
    ys = []
    for t in range(T):
        y = multi.push(srcs[:, t])
        if y is not None:
            ys.append(y)
    ys = torch.cat(ys, dim=1)                               # (S, frames, C)
    for s_ in (0, nstreams - 1):
        one = LsStreamSession(m, C, batch=1)
        yo = []
        for t in range(T):
            y = one.push(srcs[s_:s_ + 1, t])
            if y is not None:
                yo.append(y)
        yo = torch.cat(yo, dim=1)
        assert yo.shape[1] == ys.shape[1]
        assert max_abs(ys[s_], yo[0].cpu()) < 1e-4, (s_, max_abs(ys[s_], yo[0].cpu()))
    assert max_abs(ys[0], arr["stream_logits"][:ys.shape[1]]) < 1e-3          # stream 0 is the golden input
