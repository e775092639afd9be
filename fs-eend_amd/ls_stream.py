"""Frame-by-frame LS-EEND on MI355X: the one-step API the reference's streaming driver calls
(LS-EEND/streaming_infer_dia.py:52-97):

    model.enc.forward_one_step(x_t, t, ret_states, conv_caches) -> (B,1,D)      (LS model :291-293)
    StreamingConv1d(...)(emb_t.transpose(1,2)) -> (B,D,1) | None                 (LS model :151-186)
    model.dec.forward_one_step(emb_t, t, max_nspks, ret_states) -> (B,1,C,D)     (LS model :235-243)

State is O(1) per stream and stays in HBM in the driver's own containers: each
``ret_states[i]`` dict holds ``prev_key_value`` (N,H,64,64) f32 and ``scale`` (H,) f32 exactly as
the reference's incremental_state does (retention.py:141-142), each ``conv_caches[i]`` is the
driver's (B,D,k-1) f32 tensor, shifted in place.
"""
from collections import deque

import os

import torch
import torch.nn as nn

from . import ops
from .lib import EendHipError

F16, F32 = torch.float16, torch.float32


def _ret_state(state: dict, N: int, H: int, dev):
    """Fetch (or lazily create, like the reference's `"prev_key_value" in state` branch) the state."""
    if "prev_key_value" not in state:
        state["prev_key_value"] = torch.zeros(N, H, 64, 64, dtype=F32, device=dev)
        state["scale"] = torch.zeros(H, dtype=F32, device=dev)       # scale_0 = 0 -> first frame gets scale 1
        state["_scale_next"] = torch.empty(H, dtype=F32, device=dev)
    kv = state["prev_key_value"]
    if kv.shape != (N, H, 64, 64) or kv.dtype != F32 or not kv.is_contiguous():
        raise EendHipError("retention state has an unexpected shape/dtype")
    if "_scale_next" not in state:
        state["_scale_next"] = torch.empty(H, dtype=F32, device=dev)
    return kv, state["scale"].to(F32).contiguous(), state["_scale_next"]


F32_PROJ = os.environ.get("EEND_STREAM_RET_F32", "1") != "0"      # f32 retention projections in the frame steps (A/B switch)
DEC_F32 = os.environ.get("EEND_STREAM_DEC_F32", "1") != "0"       # all-f32 decoder frame step when a frame has <= 16 rows


def _ret_step(x16, x32, ln, Wd, state, N, H, scratch, out32=None):
    """One frame of MultiScaleRetention (retention.py:126-144) on N rows.  The q / k / v / g projections run in full f32 from
    the f32 residual stream (x32, through LayerNorm `ln` where the block is pre-norm): the recurrence amplifies their
    rounding with the stream position (see csrc/stream.hip ret_proj_step_kernel)."""
    o16 = scratch["o16"][:N]
    kv, s_in, s_out = _ret_state(state, N, H, x16.device)
    if F32_PROJ:
        q32 = scratch["qkvg32"][:N]
        ops.retention_proj_step(x32, ln, Wd["wqkvg32"], Wd["bqkvg"], q32, N)
        ops.retention_step_f32(q32, kv, s_in, s_out, o16, N, H, Wd["gn_eps"], out32=out32)
    else:
        qkvg = scratch["qkvg"][:N]
        ops.linear(x16, Wd["wqkvg"], Wd["bqkvg"], qkvg)
        ops.retention_step(qkvg, kv, s_in, s_out, o16, N, H, Wd["gn_eps"])
    if state.get("_static"):
        s_in.copy_(s_out)             # graph-captured sessions: fixed buffers, the new scale is copied back (4 floats)
    else:
        state["scale"], state["_scale_next"] = s_out, s_in            # ping-pong
    return o16


def _scratch(owner, key, N, D, F):
    sc = owner._step_scratch.get(key)
    if sc is None or sc["N"] < N or sc["F"] < F:
        dev = owner.cnn.weight.device
        e = lambda *s, dt=F16: torch.empty(*s, dtype=dt, device=dev)
        sc = dict(N=N, F=F, xin16=torch.zeros(N, owner._prepare()["Fin_pad"], dtype=F16, device=dev),
                  h32=e(N, D, dt=F32), h16=e(N, D), x16=e(N, D), qkvg=e(N, 4 * D), qkvg32=e(N, 4 * D, dt=F32), o16=e(N, D), glu16=e(N, D),
                  dw16=e(N, D), ff16=e(N * F), qkv16=e(N, 3 * D))
        # buffers of the f32 frame steps (<= 16 rows): always present -- the scratch is cached on the model and a later, smaller
        # session reuses what a larger one allocated
        n32 = min(N, ops.STEP_F32_MAX_ROWS)
        sc.update(o32=e(n32, D, dt=F32), qkv32=e(n32, 3 * D, dt=F32), ff32=e(n32 * F, dt=F32),
                  xin32=torch.zeros(n32, owner._prepare()["Fin_pad"], dtype=F32, device=dev))
        owner._step_scratch[key] = sc
    return sc


@torch.no_grad()
def enc_step(owner, x_t, t, ret_states, conv_caches):
    """ConformerEncoder.forward_one_step (conformer/encoder.py:223-228)."""
    P = owner._prepare()
    dev = owner.cnn.weight.device
    D, H = owner.n_units, owner._n_heads
    B = x_t.shape[0]
    x = x_t.to(device=dev, dtype=F32).reshape(B, 1, -1).contiguous()
    F = P["blocks"][0]["w1a"].shape[0] if P["blocks"] else 0
    sc = _scratch(owner, "enc", B, D, F)
    xin16, h32, h16, x16 = sc["xin16"][:B], sc["h32"][:B], sc["h16"][:B], sc["x16"][:B]
    if F32_PROJ and DEC_F32 and B <= ops.STEP_F32_MAX_ROWS:
        # the input projection in f32: the raw log-mel features are O(10), so their f16 rounding is the largest single operand
        # error of the encoder step (5e-4 on the worst logit of the one-hour stream, emulated on the oracle)
        xin32 = sc["xin32"][:B]
        xin32[:, :x.shape[-1]].copy_(x.view(B, -1))
        ops.linear_res_ln_step_f32(xin32, P["in.w32"], P["in.b"], None, P["in.g"], P["in.beta"], h32, P["in.eps"], out16=h16)
    else:
        ops.bn_cast_pad(x, None, xin16, 1, 1, False)
        ops.linear_res_ln(xin16, P["in.w"], P["in.b"], None, P["in.g"], P["in.beta"], h32, h16, P["in.eps"])
    nb = len(P["blocks"])
    f32_ffn = F32_PROJ and DEC_F32 and B <= ops.STEP_F32_MAX_ROWS     # the two half-step FFNs of every block in f32 (weights included)
    xn32 = sc["o32"][:B] if f32_ffn else None                          # LN(x) in f32: the FFN input
    for i, Bk in enumerate(P["blocks"]):
        Fi = Bk["w1a"].shape[0]
        if f32_ffn:
            ff32 = sc["ff32"][:B * Fi].view(B, Fi)
            if i == 0:
                ops.layernorm_rows_f32(h32, Bk["lna"][0], Bk["lna"][1], xn32, Bk["lna"][2])
            ops.linear_step_f32(xn32, Bk["w1a32"], Bk["b1a"], ff32, act=ops.ACT_SWISH)
            ops.linear_res_scale_ln_step_f32(ff32, Bk["w2a32"], Bk["b2a"], h32, Bk["fa"], Bk["lnb"][0], Bk["lnb"][1], h32,
                                             ln_out16=x16, eps=Bk["lnb"][2])
        else:
            if i == 0:
                ops.layernorm_f16(h32, Bk["lna"][0], Bk["lna"][1], x16, Bk["lna"][2])
            ff = sc["ff16"][:B * Fi].view(B, Fi)
            ops.linear(x16, Bk["w1a"], Bk["b1a"], ff, act=ops.ACT_SWISH)
            ops.linear_res_scale_ln16(ff, Bk["w2a"], Bk["b2a"], h32, Bk["fa"], Bk["lnb"][0], Bk["lnb"][1], h32, x16, Bk["lnb"][2])
        o16 = _ret_step(x16, h32, Bk["lnb"], Bk, ret_states[i], B, H, sc)
        ops.linear_res_scale_ln16(o16, Bk["wo"], Bk["bo"], h32, 1.0, Bk["lnc"][0], Bk["lnc"][1], h32, x16, Bk["lnc"][2])
        glu, dw = sc["glu16"][:B], sc["dw16"][:B]
        ops.linear_glu(x16, Bk["pw1"], Bk["pb1"], glu)
        cache = conv_caches[i]
        if cache.dtype != F32 or not cache.is_contiguous() or cache.device != dev:
            raise EendHipError("conv cache must be a contiguous f32 GPU tensor (B, D, k-1)")
        ops.dwconv_step(glu, cache, Bk["dw"], Bk["bn"], dw, Bk["bn_eps"])      # cache shifted in place
        ops.linear_res_scale_ln16(dw, Bk["pw2"], Bk["pb2"], h32, 1.0, Bk["lnd"][0], Bk["lnd"][1], h32, x16, Bk["lnd"][2])
        if f32_ffn:
            ops.layernorm_rows_f32(h32, Bk["lnd"][0], Bk["lnd"][1], xn32, Bk["lnd"][2])
            ops.linear_step_f32(xn32, Bk["w1b32"], Bk["b1b"], ff32, act=ops.ACT_SWISH)
            ops.linear_res_ln_step_f32(ff32, Bk["w2b32"], Bk["b2b"], h32, Bk["lne"][0], Bk["lne"][1], h32, Bk["lne"][2], alpha=Bk["fb"],
                                       out16=h16)
            if i + 1 < nb:
                nx = P["blocks"][i + 1]["lna"]
                ops.layernorm_rows_f32(h32, nx[0], nx[1], xn32, nx[2])
            continue
        ops.linear(x16, Bk["w1b"], Bk["b1b"], ff, act=ops.ACT_SWISH)
        ops.linear_res_ln(ff, Bk["w2b"], Bk["b2b"], h32, Bk["lne"][0], Bk["lne"][1], h32, h16, Bk["lne"][2], alpha=Bk["fb"])
        if i + 1 < nb:
            nx = P["blocks"][i + 1]["lna"]
            ops.layernorm_f16(h32, nx[0], nx[1], x16, nx[2])
    return h32.view(B, 1, D).clone()


@torch.no_grad()
def dec_step(owner, emb_t, t, max_nspks, ret_states):
    """MaskedTransformerDecoderModel.forward_one_step (LS model :235-243)."""
    P = owner._prepare()
    dev = owner.cnn.weight.device
    D, H, C = owner.n_units, owner._n_heads, max_nspks
    B = emb_t.shape[0]
    N = B * C
    F = P["dec.layers"][0]["w1"].shape[0] if P["dec.layers"] else 0
    sc = _scratch(owner, "dec", N, D, F)
    e32 = emb_t.to(device=dev, dtype=F32).reshape(B, D).contiguous()
    a32, a16 = sc["h32"][:N], sc["h16"][:N]
    if F32_PROJ:
        ops.convert_fanout_step_f32(e32, P["convert.w32"], owner._convert_const(C), a32, a16, B, C)
    else:
        ops.convert_fanout(e32.to(F16), P["convert.w1"], owner._convert_const(C), a32, a16, B, 1, C)
    if F32_PROJ and DEC_F32 and N <= ops.STEP_F32_MAX_ROWS:
        # one frame x <= 16 slots: the whole layer in f32 (weights included) -- the f16 operand rounding of the linears in
        # front of the retention's per-head LayerNorm was the heavy tail of the one-hour stream
        o32, qkv32 = sc["o32"][:N], sc["qkv32"][:N]
        for i, Ld in enumerate(P["dec.layers"]):
            Fi = Ld["w1_32"].shape[0]
            ff32 = sc["ff32"][:N * Fi].view(N, Fi)
            _ret_step(a16, a32, None, Ld, ret_states[i], N, H, sc, out32=o32)
            ops.linear_res_ln_step_f32(o32, Ld["out1_w32"], Ld["out1_b"], a32, Ld["g11"], Ld["be11"], a32, Ld["eps11"])
            ops.linear_step_f32(a32, Ld["in2_w32"], Ld["in2_b"], qkv32)
            ops.spk_attn_step_f32(qkv32, o32, B, C)
            ops.linear_res_ln_step_f32(o32, Ld["out2_w32"], Ld["out2_b"], a32, Ld["g21"], Ld["be21"], a32, Ld["eps21"])
            ops.linear_step_f32(a32, Ld["w1_32"], Ld["b1"], ff32, act=ops.ACT_RELU)
            ops.linear_res_ln_step_f32(ff32, Ld["w2_32"], Ld["b2"], a32, Ld["g22"], Ld["be22"], a32, Ld["eps22"])
        return a32.view(B, 1, C, D).clone()
    for i, Ld in enumerate(P["dec.layers"]):
        Fi = Ld["w1"].shape[0]
        ff = sc["ff16"][:N * Fi].view(N, Fi)
        o16 = _ret_step(a16, a32, None, Ld, ret_states[i], N, H, sc)
        ops.linear_res_ln(o16, Ld["out1_w"], Ld["out1_b"], a32, Ld["g11"], Ld["be11"], a32, a16, Ld["eps11"])
        qkv = sc["qkv16"][:N]
        ops.linear(a16, Ld["in2_w"], Ld["in2_b"], qkv)
        ops.spk_attn(qkv, o16, B, C, 1, H)
        ops.linear_res_ln(o16, Ld["out2_w"], Ld["out2_b"], a32, Ld["g21"], Ld["be21"], a32, a16, Ld["eps21"])
        ops.linear(a16, Ld["w1"], Ld["b1"], ff, relu=True)
        ops.linear_res_ln(ff, Ld["w2"], Ld["b2"], a32, Ld["g22"], Ld["be22"], a32, a16, Ld["eps22"])
    return a32.view(B, 1, C, D).clone()


class StreamingConv1d(nn.Module):
    """Ring-buffered look-ahead conv (reference LS model :151-186): emits from the (k//2+1)-th push,
    zero left padding.  ``.conv`` holds the parameters (the driver copies model.cnn into it),
    ``.buffer`` / ``.t`` are reset by the driver exactly like the reference's."""

    F32_WINDOW = True       # LS frame steps: f32 window and weights for <= 16 rows; the FS flavour keeps the f16 operands

    def __init__(self, in_channels, out_channels, kernel_size=19):
        super().__init__()
        if out_channels != 256 or in_channels % 64:
            raise NotImplementedError("HIP path is specialised for 256 output channels")
        self.kernel_size = kernel_size
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, padding=0)
        self.buffer = deque(maxlen=kernel_size)
        self.center = kernel_size // 2
        self.t = 0
        self._w = self._w_key = None

    def _weights(self):
        key = (self.conv.weight.data_ptr(), self.conv.weight._version, self.conv.bias._version)
        if self._w is None or key != self._w_key:
            w = self.conv.weight.detach()
            if not w.is_cuda:
                raise EendHipError("StreamingConv1d parameters must live on the GPU: no CPU fallback")
            w2 = w.permute(0, 2, 1).reshape(w.shape[0], -1)
            self._w = (w2.to(F16).contiguous(), self.conv.bias.detach().to(F32).contiguous(), w2.to(F32).contiguous())
            self._w_key = key
        return self._w

    @torch.no_grad()
    def forward(self, x_t):
        """x_t (B, C, 1) -> (B, C_out, 1) or None while the look-ahead is not filled."""
        self.t += 1
        self.buffer.append(x_t)
        left = self.kernel_size - len(self.buffer)
        frames = [torch.zeros_like(x_t)] * left + list(self.buffer)
        win = torch.cat(frames, dim=2)                                   # (B, C, k)
        B = win.shape[0]
        wr, bias, w32 = self._weights()
        out = torch.empty(B, wr.shape[0], dtype=F32, device=win.device)
        if self.F32_WINDOW and F32_PROJ and DEC_F32 and B <= ops.STEP_F32_MAX_ROWS:       # as LsStreamSession
            win32 = win.permute(0, 2, 1).reshape(B, -1).to(F32).contiguous()  # [b][tap*C + c]
            ops.linear_step_f32(win32, w32, bias, out)
        else:
            win16 = win.permute(0, 2, 1).reshape(B, -1).to(F16).contiguous()
            ops.linear_res_scale(win16, wr, bias, None, 1.0, out, None)
        return out.unsqueeze(-1) if self.t >= self.center + 1 else None


class LsStreamSession:
    """Frame-by-frame LS-EEND with every piece of state resident in fixed HBM buffers and the per-frame work replayed
    from three captured hipGraphs (LS-EEND/streaming_infer_dia.py:52-97 is the procedure reproduced):

        G_enc  : Conformer-retention encoder one-step (retention states, depthwise-conv caches updated in place)
        G_conv : push the frame into the 19-frame look-ahead window, Conv1d + L2 norm of the window's centre frame
        G_dec  : attractor decoder one-step (retention states in place), attractor L2 norm, embedding . attractor head

    `push(x_t)` returns the logits of frame t - conv_delay (B, 1, C), or None during the first conv_delay frames;
    `flush()` feeds conv_delay zero embeddings like the reference driver (:91-95).  The eager one-step API
    (`enc.forward_one_step` / `dec.forward_one_step`) issues ~70 launches per frame from Python; a replay is three
    graph launches, so the per-frame cost is the kernels' own few hundred microseconds.
    """

    def __init__(self, model, max_nspks: int, batch: int = 1, use_graph: bool = True):
        self.m, self.C, self.B = model, max_nspks, batch
        P = model._prepare()
        dev = model.cnn.weight.device
        self.dev = dev
        D, H = model.n_units, model._n_heads
        self.D, self.k, self.center = D, P["cnn.k"], P["cnn.k"] // 2
        nb, nd = len(P["blocks"]), len(P["dec.layers"])
        K1 = model.enc.encoder._conv_kernel_size - 1

        def ret(N):
            return dict(prev_key_value=torch.zeros(N, H, 64, 64, dtype=F32, device=dev), scale=torch.zeros(H, dtype=F32, device=dev),
                        _scale_next=torch.zeros(H, dtype=F32, device=dev), _static=True)

        self.enc_states = [ret(batch) for _ in range(nb)]
        self.dec_states = [ret(batch * max_nspks) for _ in range(nd)]
        self.caches = [torch.zeros(batch, D, K1, dtype=F32, device=dev) for _ in range(nb)]
        self.x_in = torch.zeros(batch, 1, model._in_size, dtype=F32, device=dev)
        self.enc_out = torch.zeros(batch, 1, D, dtype=F32, device=dev)
        self.win16 = torch.zeros(batch, 64, D, dtype=F16, device=dev)            # frames 0..k-1 = the look-ahead window
        self._shift = torch.zeros(batch, self.k - 1, D, dtype=F16, device=dev)
        # <= 16 streams: the look-ahead conv and the embedding normalisation in f32 too (window, weights, output): the embedding
        # is the decoder's input, whose rounding the decoder retention amplifies
        self.f32_conv = F32_PROJ and DEC_F32 and batch <= ops.STEP_F32_MAX_ROWS
        self.win32 = torch.zeros(batch, self.k, D, dtype=F32, device=dev)
        self._shift32 = torch.zeros(batch, self.k - 1, D, dtype=F32, device=dev)
        self.y32 = torch.zeros(batch, D, dtype=F32, device=dev)
        self.emb32 = torch.zeros(batch * 64, D, dtype=F32, device=dev)
        self.emb16 = torch.zeros(batch * 64, D, dtype=F16, device=dev)
        self.emb_t = torch.zeros(batch, 1, D, dtype=F32, device=dev)
        self.klen = torch.full((batch,), self.k, dtype=torch.int32, device=dev)
        self.attr = torch.zeros(batch, 1, max_nspks, D, dtype=F32, device=dev)
        self.logits = torch.zeros(batch, 1, max_nspks, dtype=F32, device=dev)
        self.t = 0
        self._graphs = None
        self._stateful = ([s["prev_key_value"] for s in self.enc_states + self.dec_states] +
                          [s[k_] for s in self.enc_states + self.dec_states for k_ in ("scale", "_scale_next")] + self.caches +
                          [self.win16, self.win32, self.enc_out])
        if use_graph:
            self._capture()

    # ---- the three stages (eager bodies; captured once)
    def _enc(self):
        self.enc_out.copy_(enc_step(self.m, self.x_in, self.t, self.enc_states, self.caches))

    def _conv(self):
        P = self.m._prepare()
        k = self.k
        if self.f32_conv:
            self._shift32.copy_(self.win32[:, 1:k])
            self.win32[:, :k - 1].copy_(self._shift32)
            self.win32[:, k - 1:k].copy_(self.enc_out)
            ops.linear_step_f32(self.win32.view(self.B, k * self.D), P["cnn.w32"], P["cnn.b"], self.y32)
            ops.l2norm_rows_f32(self.y32, self.emb_t.view(self.B, self.D))
            return
        self._shift.copy_(self.win16[:, 1:k])
        self.win16[:, :k - 1].copy_(self._shift)
        self.win16[:, k - 1:k].copy_(self.enc_out)                                  # f32 -> f16 (as the batch path's operand)
        ops.conv1d_l2norm(self.win16.view(-1, self.D), P["cnn.w"], P["cnn.b"], self.klen, self.emb32, self.emb16, self.B, 64, self.D,
                          k, self.center)
        self.emb_t.copy_(self.emb32.view(self.B, 64, self.D)[:, self.center:self.center + 1])

    def _dec(self):
        a = dec_step(self.m, self.emb_t, self.t, self.C, self.dec_states)          # (B,1,C,D) un-normalised attractors
        a32 = a.reshape(self.B * self.C, self.D)
        ops.head_l2dot(self.emb_t.view(self.B, self.D), a32, self.attr.view(self.B, 1, self.C, self.D), self.logits, self.B, 1, 1,
                       self.C, self.D)

    def reset(self):
        for t_ in self._stateful:
            t_.zero_()
        self.t = 0

    def _capture(self):
        self._P_captured = self.m._prepare()             # the operand set the recorded graphs point into
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                                                # warm-up: workspaces, operand caches
            self._enc(); self._conv(); self._dec()
        torch.cuda.current_stream().wait_stream(s)
        gs = []
        for fn in (self._enc, self._conv, self._dec):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            gs.append(g)
        self._graphs = gs
        self.reset()

    def _run(self, i):
        if self._graphs is not None:
            self._graphs[i].replay()
        else:
            (self._enc, self._conv, self._dec)[i]()

    def _check_weights(self):
        """The captured graphs hold raw pointers into the model's operand copies: re-capture (keeping the streaming state)
        when the weights were refreshed since -- load_state_dict, .to(), an optimiser step."""
        P = self.m._prep
        if P is None or (self.t & 255) == 0:
            P = self.m._prepare()
        if P is not getattr(self, "_P_captured", None):
            # (_P_captured is set by _capture() itself, at the time the graphs are recorded: a refresh between construction and
            # the first push re-captures like any later one -- ADVICE r03)
            if self._graphs is None:
                self._P_captured = P
            else:
                keep = [t_.clone() for t_ in self._stateful]
                t = self.t
                self._capture()
                for dst, src in zip(self._stateful, keep):
                    dst.copy_(src)
                self.t = t

    @torch.no_grad()
    def push(self, x_t):
        """x_t (B, 1, in_size) or (B, in_size) features of the next frame -> logits (B, 1, C) of frame t - delay, or None."""
        self._check_weights()
        self.x_in.copy_(x_t.reshape(self.B, 1, -1))
        self._run(0)
        return self._emit()

    def _emit(self):
        self._run(1)
        self.t += 1
        if self.t < self.center + 1:
            return None
        self._run(2)
        return self.logits.clone()

    @torch.no_grad()
    def flush(self):
        """the last conv_delay frames: zero embeddings through the look-ahead window (reference driver :91-95)"""
        out = []
        for _ in range(self.center):
            self.enc_out.zero_()
            y = self._emit()
            if y is not None:
                out.append(y)
        return out
