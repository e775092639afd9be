"""Frame-by-frame FS-EEND on MI355X.

Drop-in for ``nnet.model.streaming_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm``,
``nnet.modules.streaming_tfm`` and ``nnet.utils.copy_params`` of the reference
(FS-EEND/streaming_infer_dia.py:11-16 imports them): same class names, constructor arguments
and attribute tree (``enc.{bn,proj,proj_norm,layers[i].self_attn.attention,...}``,
``cnn.conv``, ``dec.{pos_enc,convert,layers[i].{temp_attn.attention,spk_attn,...}}``), so
``copy_params_from_masked_to_streaming`` and checkpoints work unchanged.

``test(x_t)`` is one frame in -> one (conv_delay-delayed) frame of logits out.  The reference's
growing K/V cache (streaming_tfm.py:118-127) is kept as *projected* K/V in HBM
(f16 (N,H,cap,64) per layer, capacity doubled on demand) -- same arithmetic, O(t) per frame.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .fs_model import PositionalEncoding, _f16, _f32
from .lib import EendHipError
from .ls_stream import StreamingConv1d as _LsStreamingConv1d


class StreamingConv1d(_LsStreamingConv1d):
    """The same ring-buffer conv as the LS flavour's (reference streaming_tfm.py:141-167), f16 MFMA operands."""
    F32_WINDOW = False

F16, F32 = torch.float16, torch.float32


class IncrementalSelfAttention(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.attention = nn.MultiheadAttention(embed_dim=d_model, num_heads=nhead, batch_first=True)


class StreamingTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation=None):
        super().__init__()
        self.self_attn = IncrementalSelfAttention(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)


class StreamingEmbeddingEncoder(nn.Module):
    def __init__(self, in_size, d_model, nhead, num_layers, dim_feedforward=2048, dropout=0.1, activation=F.relu):
        super().__init__()
        self.bn = nn.BatchNorm1d(in_size)
        self.proj = nn.Linear(in_size, d_model)
        self.proj_norm = nn.LayerNorm(d_model)
        self.layers = nn.ModuleList([StreamingTransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation)
                                     for _ in range(num_layers)])
        self.cache = [{} for _ in range(num_layers)]
        self.proj.bias.data.zero_()
        self.proj.weight.data.uniform_(-0.1, 0.1)


class StreamingAttractorDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation=F.relu):
        super().__init__()
        self.temp_attn = IncrementalSelfAttention(d_model, nhead)
        self.spk_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)


class StreamingAttractorDecoder(nn.Module):
    def __init__(self, d_model, nhead, num_layers, dim_feedforward=2048, dropout=0.1, activation=F.relu):
        super().__init__()
        self.pos_enc = PositionalEncoding(d_model, dropout)
        self.convert = nn.Linear(2 * d_model, d_model)
        self.layers = nn.ModuleList([StreamingAttractorDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation)
                                     for _ in range(num_layers)])
        self.cache = [{} for _ in range(num_layers)]


class _KvCache:
    """Projected K/V history of one layer: f16 (N,H,cap,64) x2, doubled when full."""

    def __init__(self, N, H, dev, cap=512):
        self.N, self.H, self.cap, self.t = N, H, cap, 0
        self.k = torch.empty(N, H, cap, 64, dtype=F16, device=dev)
        self.v = torch.empty(N, H, cap, 64, dtype=F16, device=dev)

    def ensure_room(self):
        if self.t + 1 >= self.cap:
            for name in ("k", "v"):
                old = getattr(self, name)
                new = torch.empty(self.N, self.H, 2 * self.cap, 64, dtype=F16, device=old.device)
                new[:, :, :self.t] = old[:, :, :self.t]
                setattr(self, name, new)
            self.cap *= 2


class StreamingTransformerEDADiarization(nn.Module):
    """reference FS-EEND/nnet/model/streaming_tfm_enc_...l2norm.py:9-60."""

    def __init__(self, in_size, n_units, n_heads, enc_n_layers, dec_n_layers, dropout, has_mask, max_seqlen,
                 dec_dim_feedforward, conv_delay=9, mask_delay=0, decom_kernel_size=64):
        super().__init__()
        if n_units != 256 or n_heads != 4:
            raise NotImplementedError("HIP kernels are specialised for n_units=256, n_heads=4")
        self.delay = conv_delay
        self.n_units = n_units
        self._H, self._in_size = n_heads, in_size
        # NB the reference builds the streaming encoder with dim_feedforward=dec_dim_feedforward (:24)
        self.enc = StreamingEmbeddingEncoder(in_size, n_units, n_heads, enc_n_layers, dim_feedforward=dec_dim_feedforward,
                                             dropout=dropout)
        self.cnn = StreamingConv1d(n_units, n_units, kernel_size=2 * conv_delay + 1)
        self.dec = StreamingAttractorDecoder(n_units, n_heads, dec_n_layers, dim_feedforward=dec_dim_feedforward,
                                             dropout=dropout)
        self._prep = self._prep_key = None
        self._pc, self._sc = {}, None
        self._enc_kv, self._dec_kv = None, None
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.refresh_weights())

    def refresh_weights(self):
        """Drop the cached operand copies (needed after writes through `.data`, which do not bump _version)."""
        self._prep = None

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._prep = None
        return out

    # ------------------------------------------------------------------ weights
    def _fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _prepare(self):
        key = self._fingerprint()
        if self._prep is not None and key == self._prep_key:
            return self._prep
        dev = self.cnn.conv.weight.device
        if dev.type != "cuda":
            raise EendHipError("model parameters must live on the GPU: the HIP path has no CPU fallback")
        D, e = self.n_units, self.enc
        P = {}
        Fin_pad = (self._in_size + 63) // 64 * 64
        w = torch.zeros(D, Fin_pad, dtype=F16, device=dev)
        w[:, :self._in_size] = e.proj.weight.detach().to(F16)
        P["in.w"], P["in.b"] = w, _f32(e.proj.bias)
        P["in.g"], P["in.beta"], P["in.eps"] = _f32(e.proj_norm.weight), _f32(e.proj_norm.bias), e.proj_norm.eps
        P["bn"] = tuple(_f32(t) for t in (e.bn.weight, e.bn.bias, e.bn.running_mean, e.bn.running_var))
        P["bn.eps"], P["Fin_pad"] = e.bn.eps, Fin_pad

        def mha(m):
            return _f16(m.in_proj_weight), _f32(m.in_proj_bias), _f16(m.out_proj.weight), _f32(m.out_proj.bias)

        def ln(m):
            return _f32(m.weight), _f32(m.bias), m.eps

        P["enc"] = [dict(att=mha(l.self_attn.attention), w1=_f16(l.linear1.weight), b1=_f32(l.linear1.bias),
                         w2=_f16(l.linear2.weight), b2=_f32(l.linear2.bias), n1=ln(l.norm1), n2=ln(l.norm2))
                    for l in e.layers]
        P["convert.w1"] = _f16(self.dec.convert.weight[:, :D])
        P["dec"] = [dict(att=mha(l.temp_attn.attention), spk=mha(l.spk_attn), w1=_f16(l.linear1.weight),
                         b1=_f32(l.linear1.bias), w2=_f16(l.linear2.weight), b2=_f32(l.linear2.bias),
                         n1=ln(l.norm1), n2=ln(l.norm2), n3=ln(l.norm3)) for l in self.dec.layers]
        self._prep, self._prep_key, self._pc = P, key, {}
        return P

    def _convert_const(self, C):
        """pc[c] = convert.weight[:, D:] pe[c] + convert.bias (a (C, D) constant of the weights): eend_convert_const_f32."""
        if C not in self._pc:
            from . import lib as _lib
            D = self.n_units
            dev = self.dec.convert.weight.device
            W = self.dec.convert.weight.detach().to(F32).contiguous()
            b = self.dec.convert.bias.detach().to(F32).contiguous()
            pe = self.dec.pos_enc.pe[0, :C].to(device=dev, dtype=F32).contiguous()
            pc = torch.empty(C, D, dtype=F32, device=dev)
            L = _lib.load()
            _lib.check(L.eend_convert_const_f32(0, W.data_ptr(), b.data_ptr(), pe.data_ptr(), pc.data_ptr(), None, None, None, C,
                                                torch.cuda.current_stream().cuda_stream), "eend_convert_const_f32")
            self._pc[C] = pc
        return self._pc[C]

    def reset_streaming_state(self):
        """Forget the K/V history and the conv ring buffer (start of a new stream)."""
        self._enc_kv = self._dec_kv = None
        self.cnn.buffer.clear()
        self.cnn.t = 0

    def _scratch(self, N, dev, F):
        if self._sc is None or self._sc["N"] < N or self._sc["F"] < F:
            D = self.n_units
            e = lambda *s, dt=F16: torch.empty(*s, dtype=dt, device=dev)
            self._sc = dict(N=N, F=F, xin16=torch.zeros(N, self._prepare()["Fin_pad"], dtype=F16, device=dev),
                            h32=e(N, D, dt=F32), h16=e(N, D), qkv=e(N, 3 * D), o16=e(N, D), ff=e(N * F),
                            a32=e(N, D, dt=F32), a16=e(N, D))
        return self._sc

    # ------------------------------------------------------------------ one frame
    @torch.no_grad()
    def test(self, x_t, max_nspks: int = 6, dummy_conv_input=False):
        """x_t (1,1,in) -> y_t (1,1,S) or None during the first conv_delay frames (reference :31-60)."""
        P = self._prepare()
        dev = self.cnn.conv.weight.device
        D, H, C = self.n_units, self._H, max_nspks
        Fmax = max([l["w1"].shape[0] for l in P["enc"] + P["dec"]] + [1])
        sc = self._scratch(max(1, C), dev, Fmax)
        if dummy_conv_input:
            emb_t = torch.zeros(1, 1, D, device=dev)                         # reference :42-43
        else:
            if x_t.shape[0] != 1:
                raise NotImplementedError("the reference's streaming model is single-stream (B=1)")
            if self._enc_kv is None:
                self._enc_kv = [_KvCache(1, H, dev) for _ in P["enc"]]
            x = x_t.to(device=dev, dtype=F32).reshape(1, 1, -1).contiguous()
            xin16, h32, h16 = sc["xin16"][:1], sc["h32"][:1], sc["h16"][:1]
            qkv, o16 = sc["qkv"][:1], sc["o16"][:1]
            ops.bn_cast_pad(x, P["bn"], xin16, 1, 1, True, P["bn.eps"])
            ops.linear_res_ln(xin16, P["in.w"], P["in.b"], None, P["in.g"], P["in.beta"], h32, h16, P["in.eps"])
            for L, kv in zip(P["enc"], self._enc_kv):
                Fi = L["w1"].shape[0]
                ff = sc["ff"][:Fi].view(1, Fi)
                kv.ensure_room()
                ops.linear(h16, L["att"][0], L["att"][1], qkv)
                ops.attn_decode(qkv, kv.k, kv.v, o16, 1, H, kv.cap, kv.t)
                kv.t += 1
                ops.linear_res_ln(o16, L["att"][2], L["att"][3], h32, L["n1"][0], L["n1"][1], h32, h16, L["n1"][2])
                ops.linear(h16, L["w1"], L["b1"], ff, relu=True)
                ops.linear_res_ln(ff, L["w2"], L["b2"], h32, L["n2"][0], L["n2"][1], h32, h16, L["n2"][2])
            emb_t = h32.view(1, 1, D).clone()

        emb_t = self.cnn(emb_t.transpose(1, 2))                              # (1,D,1) or None
        if emb_t is None:
            return None
        e32 = emb_t.transpose(1, 2).reshape(1, D)
        e32 = (e32 / torch.linalg.vector_norm(e32, dim=-1, keepdim=True)).contiguous()   # reference :50

        if self._dec_kv is None or self._dec_kv[0].N != C:
            self._dec_kv = [_KvCache(C, H, dev) for _ in P["dec"]]
        a32, a16 = sc["a32"][:C], sc["a16"][:C]
        qkv, o16 = sc["qkv"][:C], sc["o16"][:C]
        ops.convert_fanout(e32.to(F16), P["convert.w1"], self._convert_const(C), a32, a16, 1, 1, C)
        for L, kv in zip(P["dec"], self._dec_kv):
            Fi = L["w1"].shape[0]
            ff = sc["ff"][:C * Fi].view(C, Fi)
            kv.ensure_room()
            ops.linear(a16, L["att"][0], L["att"][1], qkv)
            ops.attn_decode(qkv, kv.k, kv.v, o16, C, H, kv.cap, kv.t)
            kv.t += 1
            ops.linear_res_ln(o16, L["att"][2], L["att"][3], a32, L["n1"][0], L["n1"][1], a32, a16, L["n1"][2])
            ops.linear(a16, L["spk"][0], L["spk"][1], qkv)
            ops.spk_attn(qkv, o16, 1, C, 1, H)
            ops.linear_res_ln(o16, L["spk"][2], L["spk"][3], a32, L["n2"][0], L["n2"][1], a32, a16, L["n2"][2])
            ops.linear(a16, L["w1"], L["b1"], ff, relu=True)
            ops.linear_res_ln(ff, L["w2"], L["b2"], a32, L["n3"][0], L["n3"][1], a32, a16, L["n3"][2])
        attr = torch.empty(1, 1, C, D, dtype=F32, device=dev)
        y = torch.empty(1, 1, C, dtype=F32, device=dev)
        ops.head_l2dot(e32, a32, attr, y, 1, 1, 1, C, D)
        return y


class FsStreamSession:
    """Frame-by-frame FS-EEND with all streaming state in fixed HBM buffers and the per-frame work replayed from three
    captured hipGraphs (FS-EEND/streaming_infer_dia.py's per-frame loop is the procedure reproduced):

        G_enc  : BatchNorm + input projection + the incremental encoder layers (K/V caches appended in place)
        G_conv : push the frame into the 19-frame look-ahead window, Conv1d, L2 norm
        G_dec  : `convert` fan-out, incremental decoder layers (time-axis K/V caches, speaker-axis attention), head

    The history length lives in device memory (`eend_attn_decode_dev_f16` reads it, `eend_counter_add_i32` bumps it
    inside the graph), so one capture serves every frame until a K/V cache is full; the caches then double and the
    graphs are captured again (one capture per capacity bucket).  Outputs are bit-identical to the eager
    `StreamingTransformerEDADiarization.test` (tests/test_fs_streaming.py), which issues ~60 launches per frame from
    Python."""

    def __init__(self, model: "StreamingTransformerEDADiarization", max_nspks: int = 6, cap: int = 1024, use_graph: bool = True):
        self.m, self.C, self.use_graph = model, max_nspks, use_graph
        P = model._prepare()
        dev = model.cnn.conv.weight.device
        self.dev, self.D, self.H = dev, model.n_units, model._H
        D, H, C = self.D, self.H, max_nspks
        self.k = model.cnn.kernel_size
        self.center = model.cnn.center
        self.Fmax = max([l["w1"].shape[0] for l in P["enc"] + P["dec"]] + [1])
        e = lambda *s_, dt=F16: torch.zeros(*s_, dtype=dt, device=dev)
        N = max(1, C)
        self.x_in = e(1, 1, model._in_size, dt=F32)
        self.xin16 = e(1, P["Fin_pad"])
        self.h32, self.h16 = e(N, D, dt=F32), e(N, D)
        self.qkv, self.o16, self.ff = e(N, 3 * D), e(N, D), e(N * self.Fmax)
        self.enc_out = e(1, D, dt=F32)
        self.win16 = e(1, self.k * D)                                   # [tap*D + c]: the look-ahead window, oldest tap first
        self._shift = e(1, (self.k - 1) * D)
        self.conv32 = e(1, D, dt=F32)
        self.e32, self.e16 = e(1, D, dt=F32), e(1, D)
        self.attr = e(1, 1, C, D, dt=F32)
        self.logits = e(1, 1, C, dt=F32)
        self.t_enc = torch.zeros(1, dtype=torch.int32, device=dev)
        self.t_dec = torch.zeros(1, dtype=torch.int32, device=dev)
        self.cap = 0
        self._alloc_caches(cap, keep=False)
        self.reset()

    # ---- state
    def _alloc_caches(self, cap, keep):
        P = self.m._prepare()
        old = (getattr(self, "enc_kv", None), getattr(self, "dec_kv", None), self.cap)
        mk = lambda N: tuple(torch.zeros(N, self.H, cap, 64, dtype=F16, device=self.dev) for _ in range(2))
        self.enc_kv = [mk(1) for _ in P["enc"]]
        self.dec_kv = [mk(self.C) for _ in P["dec"]]
        if keep:
            for new, prev in zip(self.enc_kv + self.dec_kv, old[0] + old[1]):
                for a, b in zip(new, prev):
                    a[:, :, :old[2]] = b
        self.cap = cap
        self._graphs = None
        # long histories: key-split decode kernel (ops.attn_decode_split) and its partial-result scratch
        self._split = cap >= ops.SPLIT_DECODE_MIN_CAP
        self._dec_ws = torch.empty(ops.attn_decode_split_ws(max(1, self.C), self.H, cap), dtype=F32, device=self.dev) if self._split else None

    def reset(self):
        """Start of a new stream."""
        self.t_enc.zero_(); self.t_dec.zero_()
        self.win16.zero_(); self.enc_out.zero_()
        self.n_enc = self.n_dec = self.t = 0                             # host mirrors of the device counters / frames pushed

    # ---- the three stages (eager bodies; captured once per cache capacity)
    def _enc(self):
        P, H = self.m._prepare(), self.H
        h32, h16, qkv, o16 = self.h32[:1], self.h16[:1], self.qkv[:1], self.o16[:1]
        ops.bn_cast_pad(self.x_in, P["bn"], self.xin16, 1, 1, True, P["bn.eps"])
        ops.linear_res_ln(self.xin16, P["in.w"], P["in.b"], None, P["in.g"], P["in.beta"], h32, h16, P["in.eps"])
        for L, (kc, vc) in zip(P["enc"], self.enc_kv):
            Fi = L["w1"].shape[0]
            ff = self.ff[:Fi].view(1, Fi)
            ops.linear(h16, L["att"][0], L["att"][1], qkv)
            self._decode(qkv, kc, vc, o16, 1, self.t_enc)
            ops.linear_res_ln(o16, L["att"][2], L["att"][3], h32, L["n1"][0], L["n1"][1], h32, h16, L["n1"][2])
            ops.linear(h16, L["w1"], L["b1"], ff, relu=True)
            ops.linear_res_ln(ff, L["w2"], L["b2"], h32, L["n2"][0], L["n2"][1], h32, h16, L["n2"][2])
        ops.counter_add(self.t_enc, 1)
        self.enc_out.copy_(h32)

    def _decode(self, qkv, kc, vc, o16, N, t_dev):
        if self._split:
            ops.attn_decode_split(qkv, kc, vc, o16, self._dec_ws, N, self.H, self.cap, t_dev)
        else:
            ops.attn_decode_dev(qkv, kc, vc, o16, N, self.H, self.cap, t_dev)

    def _conv(self):
        D, k = self.D, self.k
        self._shift.copy_(self.win16[:, D:])
        self.win16[:, :(k - 1) * D].copy_(self._shift)
        self.win16[:, (k - 1) * D:].copy_(self.enc_out)                  # f32 -> f16, as StreamingConv1d casts its window
        wr, bias = self.m.cnn._weights()[:2]
        ops.linear_res_scale(self.win16, wr, bias, None, 1.0, self.conv32, None)
        torch.div(self.conv32, torch.linalg.vector_norm(self.conv32, dim=-1, keepdim=True), out=self.e32)   # reference :50
        self.e16.copy_(self.e32)

    def _dec(self):
        P, H, C, D = self.m._prepare(), self.H, self.C, self.D
        a32, a16, qkv, o16 = self.h32[:C], self.h16[:C], self.qkv[:C], self.o16[:C]
        ops.convert_fanout(self.e16, P["convert.w1"], self.m._convert_const(C), a32, a16, 1, 1, C)
        for L, (kc, vc) in zip(P["dec"], self.dec_kv):
            Fi = L["w1"].shape[0]
            ff = self.ff[:C * Fi].view(C, Fi)
            ops.linear(a16, L["att"][0], L["att"][1], qkv)
            self._decode(qkv, kc, vc, o16, C, self.t_dec)
            ops.linear_res_ln(o16, L["att"][2], L["att"][3], a32, L["n1"][0], L["n1"][1], a32, a16, L["n1"][2])
            ops.linear(a16, L["spk"][0], L["spk"][1], qkv)
            ops.spk_attn(qkv, o16, 1, C, 1, H)
            ops.linear_res_ln(o16, L["spk"][2], L["spk"][3], a32, L["n2"][0], L["n2"][1], a32, a16, L["n2"][2])
            ops.linear(a16, L["w1"], L["b1"], ff, relu=True)
            ops.linear_res_ln(ff, L["w2"], L["b2"], a32, L["n3"][0], L["n3"][1], a32, a16, L["n3"][2])
        ops.counter_add(self.t_dec, 1)
        ops.head_l2dot(self.e32, a32, self.attr, self.logits, 1, 1, 1, C, D)

    def _capture(self):
        keep = [t_.clone() for t_ in (self.t_enc, self.t_dec, self.win16, self.enc_out, self.x_in)]
        snap = [[c.clone() for c in kv] for kv in self.enc_kv + self.dec_kv] if (self.n_enc or self.n_dec) else None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                                        # warm-up: workspaces, operand caches
            self._enc(); self._conv(); self._dec()
        torch.cuda.current_stream().wait_stream(s)
        gs = []
        for fn in (self._enc, self._conv, self._dec):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            gs.append(g)
        self._graphs = gs
        for dst, src in zip((self.t_enc, self.t_dec, self.win16, self.enc_out, self.x_in), keep):    # undo the warm-up's effects
            dst.copy_(src)
        if snap is not None:
            for kv, sv in zip(self.enc_kv + self.dec_kv, snap):
                for a, b in zip(kv, sv):
                    a.copy_(b)

    def _run(self, i):
        if not self.use_graph:
            return (self._enc, self._conv, self._dec)[i]()
        if self._graphs is None:
            self._capture()
        self._graphs[i].replay()

    def seek(self, t: int):
        """Benchmarking aid: continue as if `t` frames had already been pushed -- the K/V caches keep whatever they hold
        (zeros after construction), only the history counters move, so the per-frame cost at a given stream position
        (it grows with t: every decode step reads the whole K/V history) can be measured without streaming up to it."""
        while t + 2 >= self.cap:
            self._alloc_caches(2 * self.cap, keep=True)
        self.n_enc, self.n_dec, self.t = t, max(0, t - self.center), t
        self.t_enc.fill_(self.n_enc)
        self.t_dec.fill_(self.n_dec)

    def _room(self):
        if max(self.n_enc, self.n_dec) + 1 >= self.cap:                   # next capacity bucket: bigger caches, new graphs
            self._alloc_caches(2 * self.cap, keep=True)

    def _check_weights(self):
        """The captured graphs hold raw pointers into the model's operand copies (model._prepare()): if the weights were
        refreshed since the capture (load_state_dict, .to(), an optimiser step) those copies are stale or freed -- capture
        again on the new ones instead of replaying over dead memory."""
        P = self.m._prep
        if P is None or (self.t & 255) == 0:
            P = self.m._prepare()
        if P is not getattr(self, "_P_captured", None):
            self._P_captured = P
            self._graphs = None

    @torch.no_grad()
    def push(self, x_t):
        """x_t: features of the next frame ((1,1,in) / (1,in) / (in,)) -> logits (1,1,C) of frame t - conv_delay, or None
        during the first conv_delay frames."""
        self._check_weights()
        self._room()
        self.x_in.copy_(x_t.reshape(1, 1, -1))
        self._run(0)
        self.n_enc += 1
        return self._emit()

    def _emit(self):
        self._run(1)
        self.t += 1
        if self.t < self.center + 1:
            return None
        self._run(2)
        self.n_dec += 1
        return self.logits.clone()

    @torch.no_grad()
    def flush(self):
        """the last conv_delay frames: zero embeddings through the look-ahead window (the reference driver's
        `dummy_conv_input=True` calls)"""
        out = []
        for _ in range(self.center):
            self._room()
            self.enc_out.zero_()
            y = self._emit()
            if y is not None:
                out.append(y)
        return out


# ---------------------------------------------------------------------------------------------
# parameter transfer (reference FS-EEND/nnet/utils/copy_params.py:7-62)
# ---------------------------------------------------------------------------------------------
def _copy_mha(dst: nn.MultiheadAttention, src: nn.MultiheadAttention):
    with torch.no_grad():              # in-place copy_ on the parameters themselves: bumps _version (the weight-cache key)
        dst.in_proj_weight.copy_(src.in_proj_weight)
        dst.in_proj_bias.copy_(src.in_proj_bias)
        dst.out_proj.weight.copy_(src.out_proj.weight)
        dst.out_proj.bias.copy_(src.out_proj.bias)


def _copy_mod(dst: nn.Module, src: nn.Module):
    dst.load_state_dict(src.state_dict())


def copy_params_with_masked_emb_encoder(masked_enc, streaming_enc):
    _copy_mod(streaming_enc.bn, masked_enc.bn)
    _copy_mod(streaming_enc.proj, masked_enc.encoder)
    _copy_mod(streaming_enc.proj_norm, masked_enc.encoder_norm)
    for std, inc in zip(masked_enc.transformer_encoder.layers, streaming_enc.layers):
        _copy_mha(inc.self_attn.attention, std.self_attn)
        for name in ("linear1", "linear2", "norm1", "norm2"):
            _copy_mod(getattr(inc, name), getattr(std, name))


def copy_params_with_conv1d(standard_conv1d, streaming_conv1d):
    _copy_mod(streaming_conv1d.conv, standard_conv1d)


def copy_params_with_masked_decoder(masked_dec, streaming_dec):
    _copy_mod(streaming_dec.pos_enc, masked_dec.pos_enc)
    _copy_mod(streaming_dec.convert, masked_dec.convert)
    for msk, st in zip(masked_dec.attractor_decoder.layers, streaming_dec.layers):
        _copy_mha(st.temp_attn.attention, msk.self_attn1)
        _copy_mha(st.spk_attn, msk.self_attn2)
        _copy_mod(st.linear1, msk.linear1)
        _copy_mod(st.linear2, msk.linear2)
        _copy_mod(st.norm1, msk.norm11)
        _copy_mod(st.norm2, msk.norm21)
        _copy_mod(st.norm3, msk.norm22)


def copy_params_from_masked_to_streaming(masked_fs_eend, streaming_fs_eend):
    copy_params_with_masked_emb_encoder(masked_fs_eend.enc, streaming_fs_eend.enc)
    copy_params_with_conv1d(masked_fs_eend.cnn, streaming_fs_eend.cnn)
    copy_params_with_masked_decoder(masked_fs_eend.dec, streaming_fs_eend.dec)
