"""Host-side mirror of the reference's LS-EEND model (Conformer-with-retention encoder,
retention x speaker-attention attractor decoder).

Drop-in for ``nnet.model.onl_conformer_retention_enc_1dcnn_tfm_retention_enc_linear_non_autoreg_
pos_enc_l2norm_emb_loss_mask`` (reference LS-EEND/nnet/model/...emb_loss_mask.py): same class
names, constructor arguments, attribute tree (``enc.encoder.layers[i].sequential[j].module...``,
``dec.layers[i]``, ``cnn``; what LS-EEND/streaming_infer_dia.py:30-44 reads) and state_dict keys,
so reference checkpoints load and seeded default init is bit-identical.

The torch.nn modules below are parameter containers only (no forward); the arithmetic runs
in libeend_hip.so through ops.py.
"""
import math
import weakref
from typing import Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .fs_model import PositionalEncoding, WorkspaceCache, _f16, _f32

# One fast form + one general fallback per operator, selected by SHAPE (round 6: the A/B environment switches of rounds 2 - 5 are gone;
# their questions are answered in profiles/OPTIMISATION_LOG.md):
#   retention + projections   ret_stream.hip (H = 4, chunk <= 512, ...)   | retention_proj + retention_chunk (retention.hip / _full.hip)
#   decoder layer head        spk_stream.hip, f32 residual (C <= 12: the limit of every speaker-axis kernel)
#   decoder layer tail        ffn.hip PRE form, out-projection weight as a hi / lo f16 pair, hi / lo f16 copies of the f32 output rows
#   Conformer half-step FFNs  ffn.hip (the packed-stream form measured slower on this model's f32 residual stream: round 4)
#   decoder input             convert_f32.hip (f32 MFMA; the batch forward hands over f32 embeddings) | f16 form (chunked long-form path)
#   look-ahead conv           conv_stream.hip (256 channels)              | implicit-GEMM epilogue (gemm.hip)
from .lib import EendHipError
from . import ls_stream
from .ls_stream import StreamingConv1d  # noqa: F401  (re-exported: the reference defines it next to the model)


# --------------------------------------------------------------------------- parameter containers
class _Tag(nn.Module):
    """Parameter-less placeholder keeping nn.Sequential indices aligned with the reference
    (Swish / GLU / Dropout / Transpose slots)."""

    def __init__(self, what: str):
        super().__init__()
        self.what = what


class Linear(nn.Module):
    """nn.Linear wrapper with xavier weight / zero bias (reference conformer/modules.py:36-49)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        nn.init.xavier_uniform_(self.linear.weight)
        if bias:
            nn.init.zeros_(self.linear.bias)


class ResidualConnectionModule(nn.Module):
    def __init__(self, module, module_factor=1.0, input_factor=1.0):
        super().__init__()
        self.module = module
        self.module_factor = module_factor
        self.input_factor = input_factor


class FeedForwardModule(nn.Module):
    def __init__(self, encoder_dim, expansion_factor, dropout_p):
        super().__init__()
        self.sequential = nn.Sequential(
            nn.LayerNorm(encoder_dim),
            Linear(encoder_dim, encoder_dim * expansion_factor, bias=True),
            _Tag("swish"), nn.Dropout(p=dropout_p),
            Linear(encoder_dim * expansion_factor, encoder_dim, bias=True),
            nn.Dropout(p=dropout_p))


class RetNetRelPos(nn.Module):
    """Buffers only; decay == log(1) (reference modules/retention.py:15-23)."""

    def __init__(self, embed_dim, num_heads, recurrent_chunk_size):
        super().__init__()
        angle = 1.0 / (10000 ** torch.linspace(0, 1, embed_dim // num_heads // 2))
        angle = angle.unsqueeze(-1).repeat(1, 2).flatten()
        decay = torch.log(torch.tensor([1] * num_heads, dtype=torch.float))
        self.register_buffer("angle", angle)
        self.register_buffer("decay", decay)
        self.recurrent_chunk_size = recurrent_chunk_size


class MultiScaleRetention(nn.Module):
    def __init__(self, embed_dim, num_heads, value_factor=1):
        super().__init__()
        self.factor, self.embed_dim, self.num_heads = value_factor, embed_dim, num_heads
        self.head_dim = embed_dim * value_factor // num_heads
        self.key_dim = embed_dim // num_heads
        self.scaling = self.key_dim ** -0.5
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.k_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.v_proj = nn.Linear(embed_dim, embed_dim * value_factor, bias=True)
        self.g_proj = nn.Linear(embed_dim, embed_dim * value_factor, bias=True)
        self.out_proj = nn.Linear(embed_dim * value_factor, embed_dim, bias=True)
        self.group_norm = nn.LayerNorm(self.head_dim, eps=1e-6, elementwise_affine=False)
        for p in (self.q_proj, self.k_proj, self.v_proj, self.g_proj):
            nn.init.xavier_uniform_(p.weight, gain=2 ** -2.5)
        nn.init.xavier_uniform_(self.out_proj.weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class MultiHeadedSelfRetentionModule(nn.Module):
    def __init__(self, d_model, num_heads, recurrent_chunk_size=500, dropout_p=0.1):
        super().__init__()
        self.layer_norm = nn.LayerNorm(d_model)
        self.ret_pos = RetNetRelPos(d_model, num_heads, recurrent_chunk_size)
        self.self_attn = MultiScaleRetention(d_model, num_heads, value_factor=1)
        self.dropout = nn.Dropout(p=dropout_p)


class PointwiseConv1d(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=True)


class DepthwiseConv1d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, padding):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, groups=in_channels, stride=1,
                              padding=padding, bias=False)


class ConformerConvModule(nn.Module):
    def __init__(self, in_channels, kernel_size=31, expansion_factor=2, dropout_p=0.1):
        super().__init__()
        assert expansion_factor == 2
        self.sequential = nn.Sequential(
            nn.LayerNorm(in_channels), _Tag("transpose"),
            PointwiseConv1d(in_channels, in_channels * expansion_factor), _Tag("glu"),
            DepthwiseConv1d(in_channels, in_channels, kernel_size, padding=kernel_size - 1),
            nn.BatchNorm1d(in_channels), _Tag("swish"),
            PointwiseConv1d(in_channels, in_channels), nn.Dropout(p=dropout_p))


class ConformerEncoderBlock(nn.Module):
    def __init__(self, encoder_dim, num_attention_heads, feed_forward_expansion_factor, conv_expansion_factor,
                 feed_forward_dropout_p, attention_dropout_p, conv_dropout_p, conv_kernel_size, half_step_residual,
                 recurrent_chunk_size):
        super().__init__()
        self.feed_forward_residual_factor = 0.5 if half_step_residual else 1
        f = self.feed_forward_residual_factor
        self.sequential = nn.Sequential(
            ResidualConnectionModule(FeedForwardModule(encoder_dim, feed_forward_expansion_factor, feed_forward_dropout_p), f),
            ResidualConnectionModule(MultiHeadedSelfRetentionModule(encoder_dim, num_attention_heads, recurrent_chunk_size,
                                                                    attention_dropout_p)),
            ResidualConnectionModule(ConformerConvModule(encoder_dim, conv_kernel_size, conv_expansion_factor, conv_dropout_p)),
            ResidualConnectionModule(FeedForwardModule(encoder_dim, feed_forward_expansion_factor, feed_forward_dropout_p), f),
            nn.LayerNorm(encoder_dim))


class ConformerEncoder(nn.Module):
    def __init__(self, input_dim, encoder_dim, num_layers, num_attention_heads, feed_forward_expansion_factor,
                 conv_expansion_factor, feed_forward_dropout_p, attention_dropout_p, conv_dropout_p, conv_kernel_size,
                 half_step_residual, recurrent_chunk_size):
        super().__init__()
        self._conv_kernel_size = conv_kernel_size
        self.input_projection = Linear(input_dim, encoder_dim)
        self.layer_norm = nn.LayerNorm(encoder_dim)
        self.layers = nn.ModuleList([ConformerEncoderBlock(
            encoder_dim, num_attention_heads, feed_forward_expansion_factor, conv_expansion_factor,
            feed_forward_dropout_p, attention_dropout_p, conv_dropout_p, conv_kernel_size, half_step_residual,
            recurrent_chunk_size) for _ in range(num_layers)])


class EmbeddingEncoderModule(nn.Module):
    def __init__(self, in_size, n_units, n_heads, n_layers, recurrent_chunk_size, feed_forward_expansion_factor=8,
                 conv_expansion_factor=2, dropout=0.1, conv_kernel_size=16, half_step_residual=True, max_seqlen=500):
        super().__init__()
        self.max_seqlen = max_seqlen
        self.recurrent_chunk_size = recurrent_chunk_size
        self.encoder = ConformerEncoder(in_size, n_units, n_layers, n_heads, feed_forward_expansion_factor,
                                        conv_expansion_factor, dropout, dropout, dropout, conv_kernel_size,
                                        half_step_residual, recurrent_chunk_size)

    def forward_one_step(self, x_t: Tensor, t: int, ret_states: list, conv_caches: list) -> Tensor:
        """x_t (B,1,in) -> (B,1,D); states updated in place (reference LS model :291-293)."""
        return ls_stream.enc_step(self._owner(), x_t, t, ret_states, conv_caches)


class TransformerEncoderFusionLayer(nn.Module):
    """Parameters of the LS decoder layer (reference modules/merge_retnet_layer.py:71-110)."""

    def __init__(self, d_model, nhead, recurrent_chunk_size=500, dim_feedforward=2048, dropout=0.1,
                 layer_norm_eps=1e-5, batch_first=True):
        super().__init__()
        self.ret_pos1 = RetNetRelPos(d_model, nhead, recurrent_chunk_size)
        self.self_attn1 = MultiScaleRetention(d_model, nhead, value_factor=1)
        self.self_attn2 = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=batch_first)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm11 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm12 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm21 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm22 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.dropout11, self.dropout21, self.dropout2 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)


class MaskedTransformerDecoderModel(nn.Module):
    def __init__(self, in_size, n_heads, n_units, n_layers, recurrent_chunk_size, dim_feedforward, dropout=0.5,
                 max_seqlen=500, has_pos=False, mask_delay=0):
        super().__init__()
        self.in_size, self.n_heads, self.n_units, self.n_layers = in_size, n_heads, n_units, n_layers
        self.has_pos, self.max_seqlen, self.mask_delay = has_pos, max_seqlen, mask_delay
        self.encoder = nn.Linear(in_size, n_units)            # dead, kept for checkpoints
        self.encoder_norm = nn.LayerNorm(n_units)             # dead
        self.pos_enc = PositionalEncoding(n_units, dropout)
        self.convert = nn.Linear(n_units * 2, n_units)
        self.layers = nn.ModuleList([TransformerEncoderFusionLayer(n_units, n_heads, recurrent_chunk_size, dim_feedforward,
                                                                   dropout, batch_first=True) for _ in range(n_layers)])

    def forward_one_step(self, emb_t: Tensor, t: int, max_nspks: int, ret_states: list) -> Tensor:
        """emb_t (B,1,D) -> (B,1,C,D); retention states updated in place (reference LS model :235-243)."""
        return ls_stream.dec_step(self._owner(), emb_t, t, max_nspks, ret_states)


def _ret_pack(msr: MultiScaleRetention):
    """[q; k * dk^-0.5; v; g] rows (4D, D) f16 + bias (4D,) f32 (retention.py:200-205)."""
    s = msr.scaling
    w = torch.cat([msr.q_proj.weight, msr.k_proj.weight * s, msr.v_proj.weight, msr.g_proj.weight], dim=0)
    b = torch.cat([msr.q_proj.bias, msr.k_proj.bias * s, msr.v_proj.bias, msr.g_proj.bias], dim=0)
    return _f16(w), _f32(b)


def _ret_pack32(msr: MultiScaleRetention):
    """The same packed projection in f32: operand of the frame-by-frame sessions (eend_retention_proj_step_f32)."""
    s = msr.scaling
    w = torch.cat([msr.q_proj.weight, msr.k_proj.weight * s, msr.v_proj.weight, msr.g_proj.weight], dim=0)
    return _f32(w)


class _Workspace:
    def __init__(self, dev, B, Tp, C, D, F_enc, F_dec, Fin_pad, H, nc, with_lo=True):
        f16, f32 = torch.float16, torch.float32
        Me, Md = B * Tp, B * C * Tp
        Mx = max(Me, Md)
        e = lambda *s, dt: torch.empty(*s, dtype=dt, device=dev)
        self.xin16 = torch.zeros(Me, Fin_pad, dtype=f16, device=dev)
        self.h32, self.h16, self.x16 = e(Me, D, dt=f32), e(Me, D, dt=f16), e(Me, D, dt=f16)
        self.q, self.k, self.kt, self.vt = (e(Mx * D, dt=f16) for _ in range(4))
        self.g = e(Mx, D, dt=f16)
        self.o16 = torch.zeros(Mx, D, dtype=f16, device=dev)     # rows of skipped padding chunks are never written: keep them finite
        self.glu16, self.dw16 = e(Me, D, dt=f16), e(Me, D, dt=f16)
        self.emb16 = e(Me, D, dt=f16)
        self.a32, self.a16 = e(Md, D, dt=f32), e(Md, D, dt=f16)
        # f16 remainder of a32's rows: the second operand of the fused retention's query path (ret_stream.hip); only allocated where
        # that operator runs (~3 GB at B = 512, C = 12, Tp = 1024 that nothing read on the fallback path)
        self.a16lo = e(Md if with_lo else 0, D, dt=f16)
        nseq = max(B, B * C)
        self.st = e(nseq * H * nc * 2 * 4096, dt=f16)
        self.cscale, self.sexp = e(nseq * H * nc, dt=f32), e(nseq * H * nc, dt=f32)


class OnlineConformerRetentionDADiarization(nn.Module):
    """LS-EEND on MI355X (reference LS model :14-147)."""

    def __init__(self, n_speakers, in_size, n_units, n_heads, enc_n_layers, dec_n_layers, dropout, max_seqlen,
                 recurrent_chunk_size: int = 500, feed_forward_expansion_factor: int = 8, dec_dim_feedforward: int = 2048,
                 conv_expansion_factor: int = 2, conv_kernel_size: int = 16, half_step_residual: bool = True,
                 conv_delay=9, mask_delay=0):
        super().__init__()
        if n_units != 256 or n_heads != 4:
            raise NotImplementedError("HIP kernels are specialised for n_units=256, n_heads=4 (all reference configs)")
        self.n_speakers, self.n_units, self.delay = n_speakers, n_units, conv_delay
        self.max_seqlen, self.recurrent_chunk_size = max_seqlen, recurrent_chunk_size
        self.enc = EmbeddingEncoderModule(in_size=in_size, n_units=n_units, n_heads=n_heads, n_layers=enc_n_layers,
                                          recurrent_chunk_size=recurrent_chunk_size,
                                          feed_forward_expansion_factor=feed_forward_expansion_factor,
                                          conv_expansion_factor=conv_expansion_factor, dropout=dropout,
                                          conv_kernel_size=conv_kernel_size, half_step_residual=half_step_residual,
                                          max_seqlen=max_seqlen)
        self.dec = MaskedTransformerDecoderModel(in_size, n_heads=n_heads, n_units=n_units, n_layers=dec_n_layers,
                                                 recurrent_chunk_size=recurrent_chunk_size,
                                                 dim_feedforward=dec_dim_feedforward, dropout=dropout,
                                                 max_seqlen=max_seqlen, mask_delay=mask_delay)
        self.cnn = nn.Conv1d(n_units, n_units, kernel_size=2 * conv_delay + 1, padding=conv_delay)
        self._in_size, self._n_heads = in_size, n_heads
        self._prep = self._prep_key = None
        self._ws, self._pc, self._step_scratch = WorkspaceCache(), {}, {}
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.refresh_weights())
        # back-references for the one-step API (bypass nn.Module registration: no module cycle)
        object.__setattr__(self.enc, "_owner", weakref.ref(self))
        object.__setattr__(self.dec, "_owner", weakref.ref(self))

    # ------------------------------------------------------------------ weight preparation
    def refresh_weights(self):
        """Drop the cached operand copies (needed after writes through `.data`, which do not bump _version)."""
        self._prep = None

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._prep = None
        return out

    def _fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _prepare(self):
        key = self._fingerprint()
        if self._prep is not None and key == self._prep_key:
            return self._prep
        dev = self.cnn.weight.device
        if dev.type != "cuda":
            raise EendHipError("model parameters must live on the GPU: the HIP path has no CPU fallback")
        for name, buf in self.named_buffers():
            # the retention kernels hard-code decay == 1 (log-decay 0) and no rotary shift, as the reference ships
            # (modules/retention.py:20, :209-213); a checkpoint with another decay must fail loudly, not silently
            if name.endswith(".decay") and bool((buf != 0).any()):
                raise NotImplementedError(f"{name} is not all-zero: the HIP retention kernels implement decay == 1 only")
        D = self.n_units
        P = {}
        e = self.enc.encoder
        Fin = self._in_size
        Fin_pad = (Fin + 63) // 64 * 64
        w = torch.zeros(D, Fin_pad, dtype=torch.float16, device=dev)
        w[:, :Fin] = e.input_projection.linear.weight.detach().to(torch.float16)
        P["in.w"], P["in.b"] = w, _f32(e.input_projection.linear.bias)
        w32 = torch.zeros(D, Fin_pad, dtype=torch.float32, device=dev)            # f32 frame step (ls_stream.enc_step)
        w32[:, :Fin] = e.input_projection.linear.weight.detach()
        P["in.w32"] = w32
        P["in.g"], P["in.beta"], P["in.eps"] = _f32(e.layer_norm.weight), _f32(e.layer_norm.bias), e.layer_norm.eps
        P["Fin_pad"] = Fin_pad
        blocks = []
        for blk in e.layers:
            s = blk.sequential
            ffa, ret, cm, ffb, ln_e = s[0].module.sequential, s[1].module, s[2].module.sequential, s[3].module.sequential, s[4]
            pw1 = cm[2].conv.weight.detach()[:, :, 0]                                   # (2D, D)
            pb1 = cm[2].conv.bias.detach()
            inter = torch.stack([pw1[:D], pw1[D:]], dim=1).reshape(2 * D, D)            # rows (value_n, gate_n) interleaved
            interb = torch.stack([pb1[:D], pb1[D:]], dim=1).reshape(2 * D)
            wq, bq = _ret_pack(ret.self_attn)
            bn = cm[5]
            blocks.append(dict(
                fa=s[0].module_factor, fb=s[3].module_factor,
                lna=(_f32(ffa[0].weight), _f32(ffa[0].bias), ffa[0].eps),
                w1a=_f16(ffa[1].linear.weight), b1a=_f32(ffa[1].linear.bias),
                w2a=_f16(ffa[4].linear.weight), b2a=_f32(ffa[4].linear.bias),
                lnb=(_f32(ret.layer_norm.weight), _f32(ret.layer_norm.bias), ret.layer_norm.eps),
                wqkvg=wq, bqkvg=bq, wqkvg32=_ret_pack32(ret.self_attn), gn_eps=ret.self_attn.group_norm.eps,
                wo=_f16(ret.self_attn.out_proj.weight), bo=_f32(ret.self_attn.out_proj.bias),
                lnc=(_f32(cm[0].weight), _f32(cm[0].bias), cm[0].eps),
                pw1=_f16(inter), pb1=_f32(interb),
                dw=_f32(cm[4].conv.weight[:, 0, :]),
                bn=tuple(_f32(t) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)), bn_eps=bn.eps,
                pw2=_f16(cm[7].conv.weight[:, :, 0]), pb2=_f32(cm[7].conv.bias),
                lnd=(_f32(ffb[0].weight), _f32(ffb[0].bias), ffb[0].eps),
                w1b=_f16(ffb[1].linear.weight), b1b=_f32(ffb[1].linear.bias),
                w2b=_f16(ffb[4].linear.weight), b2b=_f32(ffb[4].linear.bias),
                # f32 half-step FFN weights of the frame step (ls_stream.enc_step)
                w1a32=_f32(ffa[1].linear.weight), w2a32=_f32(ffa[4].linear.weight),
                w1b32=_f32(ffb[1].linear.weight), w2b32=_f32(ffb[4].linear.weight),
                lne=(_f32(ln_e.weight), _f32(ln_e.bias), ln_e.eps)))
        for Bk in blocks:
            Bk["wrs"] = ops.retention_stream_pack(Bk["wqkvg32"])
        P["blocks"] = blocks
        cw = self.cnn.weight.detach()
        P["cnn.w"] = cw.permute(0, 2, 1).reshape(cw.shape[0], -1).to(torch.float16).contiguous()
        P["cnn.b"], P["cnn.k"], P["cnn.pad"] = _f32(self.cnn.bias), cw.shape[2], self.cnn.padding[0]
        if cw.shape[0] == 256 and ops.conv_stream_ok(cw.shape[1], cw.shape[2], self.cnn.padding[0]):
            P["cnn.ws"] = ops.conv_stream_pack(P["cnn.w"], cw.shape[2])      # the look-ahead conv on the packed weight stream (conv_stream.hip)
        P["cnn.w32"] = cw.permute(0, 2, 1).reshape(cw.shape[0], -1).to(torch.float32).contiguous()     # f32 frame steps
        P["convert.w1"] = _f16(self.dec.convert.weight[:, :D])
        P["convert.w32"] = self.dec.convert.weight.detach().to(torch.float32).contiguous()     # frame steps (ls_stream.dec_step)
        dl = []
        for l in self.dec.layers:
            wq, bq = _ret_pack(l.self_attn1)
            dl.append(dict(
                wqkvg=wq, bqkvg=bq, wqkvg32=_ret_pack32(l.self_attn1), gn_eps=l.self_attn1.group_norm.eps,
                out1_w=_f16(l.self_attn1.out_proj.weight), out1_b=_f32(l.self_attn1.out_proj.bias),
                in2_w=_f16(l.self_attn2.in_proj_weight), in2_b=_f32(l.self_attn2.in_proj_bias),
                out2_w=_f16(l.self_attn2.out_proj.weight), out2_b=_f32(l.self_attn2.out_proj.bias),
                out2_wlo=_f16(l.self_attn2.out_proj.weight.detach().float() - l.self_attn2.out_proj.weight.detach().to(torch.float16).float()),
                w1=_f16(l.linear1.weight), b1=_f32(l.linear1.bias), w2=_f16(l.linear2.weight), b2=_f32(l.linear2.bias),
                g11=_f32(l.norm11.weight), be11=_f32(l.norm11.bias), eps11=l.norm11.eps,
                g21=_f32(l.norm21.weight), be21=_f32(l.norm21.bias), eps21=l.norm21.eps,
                g22=_f32(l.norm22.weight), be22=_f32(l.norm22.bias), eps22=l.norm22.eps,
                # f32 weights of the all-f32 frame step (ls_stream.dec_step: <= 16 rows per frame)
                out1_w32=_f32(l.self_attn1.out_proj.weight), in2_w32=_f32(l.self_attn2.in_proj_weight),
                out2_w32=_f32(l.self_attn2.out_proj.weight), w1_32=_f32(l.linear1.weight), w2_32=_f32(l.linear2.weight)))
        for Ld in dl:
            Ld["wrs"] = ops.retention_stream_pack(Ld["wqkvg32"])
            Ld["ws1"] = ops.spk_stream_pack(Ld["out1_w"], Ld["in2_w"])
            Fd = Ld["w1"].shape[0]      # layer tail on the packed stream (LO form) where its shape allows, else the un-packed ffn.hip launch
            Ld["ws2"] = ops.ffn_stream_pack_lo(Ld["out2_w"], Ld["out2_wlo"], Ld["w1"], Ld["w2"]) if (Fd % 64 == 0 and 64 <= Fd <= 2048) else None
        P["dec.layers"] = dl
        self._prep, self._prep_key, self._pc = P, key, {}
        return P

    def _convert_const(self, C):
        """pc[c] = convert.weight[:, D:] pe[c] + convert.bias  (a (C, D) constant of the weights): eend_convert_const_f32."""
        if C not in self._pc:
            from . import lib as _lib
            D = self.enc.n_units if hasattr(self.enc, "n_units") else self.n_units
            dev = self.dec.convert.weight.device
            W = self.dec.convert.weight.detach().to(torch.float32).contiguous()
            b = self.dec.convert.bias.detach().to(torch.float32).contiguous()
            pe = self.dec.pos_enc.pe[0, :C].to(device=dev, dtype=torch.float32).contiguous()
            pc = torch.empty(C, D, dtype=torch.float32, device=dev)
            L = _lib.load()
            _lib.check(L.eend_convert_const_f32(0, W.data_ptr(), b.data_ptr(), pe.data_ptr(), pc.data_ptr(), None, None, None, C,
                                                torch.cuda.current_stream().cuda_stream), "eend_convert_const_f32")
            self._pc[C] = pc
        return self._pc[C]

    def _workspace(self, dev, B, Tp, C, nc):
        key = (str(dev), B, Tp, C, nc)
        ws = self._ws.get(key)
        if ws is None:
            P = self._prep
            F_enc = P["blocks"][0]["w1a"].shape[0] if P["blocks"] else 0
            F_dec = P["dec.layers"][0]["w1"].shape[0] if P["dec.layers"] else 0
            with_lo = self._n_heads == 4 and ops.retention_stream_ok(self.recurrent_chunk_size, Tp)
            ws = _Workspace(dev, B, Tp, C, self.n_units, F_enc, F_dec, P["Fin_pad"], self._n_heads, nc, with_lo=with_lo)
            self._ws.put(key, ws)
        return ws

    # ------------------------------------------------------------------ the hot path
    def _encode_span(self, P, ws, part, B, Tc, Tp, states=None, halos=None, carry_in=False):
        """Conformer-retention encoder (conformer/encoder.py:194-201, :76-113) over one span of Tc frames per sequence
        (slab rows Tp): `part` = the span's feature tensors.  Result: ws.h16 (f16 encoder output rows).  `states` /
        `halos` (lists per block, long-form walk): the retention state after the span is written to states[i] (and read
        from it when carry_in), the conv module's k-1 frames of left context are read from / written to halos[i].
        The ONE place the block's kernel sequence lives: `test` and `test_chunked` both walk through here."""
        D, H, L = self.n_units, self._n_heads, self.recurrent_chunk_size
        Me = B * Tp
        ops.gather_bn_cast_pad(part, None, ws.xin16, Tc, Tp, 0.0, False)        # pad_sequence(0) (LS model :280) + cast
        ops.linear_res_ln(ws.xin16, P["in.w"], P["in.b"], None, P["in.g"], P["in.beta"], ws.h32, ws.h16, P["in.eps"])
        q, k, kt, vt = ws.q[:Me * D], ws.k[:Me * D], ws.kt[:Me * D], ws.vt[:Me * D]
        g, o16 = ws.g[:Me], ws.o16[:Me]
        nb = len(P["blocks"])
        K1 = self.enc.encoder._conv_kernel_size - 1
        for i, Bk in enumerate(P["blocks"]):
            if i == 0:
                ops.layernorm_f16(ws.h32, Bk["lna"][0], Bk["lna"][1], ws.x16, Bk["lna"][2])
            # x += fa * FFN(LN_a x)                      -> x16 = LN_b(x)      (hidden activations stay on chip: ffn.hip)
            ops.ffn_fused(ws.x16, Bk["w1a"], Bk["b1a"], Bk["w2a"], Bk["b2a"], ws.h32, Bk["lnb"][0], Bk["lnb"][1],
                          ws.h32, ws.x16, ops.ACT_SWISH, Bk["fa"], Bk["lnb"][2], residual_unnormalised=True)
            # x += Retention(LN_b x)                     -> x16 = LN_c(x)
            if H == 4 and ops.retention_stream_ok(L, Tp):      # projections + retention in one operator (ret_stream.hip)
                ops.retention_stream(ws.x16, None, Bk["wrs"], Bk["bqkvg"], o16, ws.st, ws.cscale, ws.sexp, B, Tp, L, Bk["gn_eps"], t_valid=Tc,
                                     state_in=states[i] if (states is not None and carry_in) else None,
                                     state_out=states[i] if states is not None else None)
            else:
                ops.retention_proj(ws.x16, Bk["wqkvg"], Bk["bqkvg"], q, k, kt, vt, g, B, Tp, H)
                ops.retention_chunk(q, k, kt, vt, g, o16, ws.st, ws.cscale, ws.sexp, B, H, Tp, L, Bk["gn_eps"], t_valid=Tc,
                                    state_in=states[i] if (states is not None and carry_in) else None,
                                    state_out=states[i] if states is not None else None)
            ops.linear_res_scale_ln16(o16, Bk["wo"], Bk["bo"], ws.h32, 1.0, Bk["lnc"][0], Bk["lnc"][1],
                                      ws.h32, ws.x16, Bk["lnc"][2])
            # x += ConvModule(x): 1x1 + GLU, causal depthwise + BN + swish, 1x1   -> x16 = LN_d(x)
            ops.linear_glu(ws.x16, Bk["pw1"], Bk["pb1"], ws.glu16)
            ops.dwconv_bn_swish(ws.glu16, Bk["dw"], Bk["bn"], ws.dw16, B, Tp, Bk["bn_eps"], halo16=None if halos is None else halos[i])
            if halos is not None:
                halos[i] = ws.glu16.view(B, Tp, D)[:, Tc - K1:Tc].clone()       # next span's left context (a copy: glu16 is reused)
            ops.linear_res_scale_ln16(ws.dw16, Bk["pw2"], Bk["pb2"], ws.h32, 1.0, Bk["lnd"][0], Bk["lnd"][1],
                                      ws.h32, ws.x16, Bk["lnd"][2])
            # x = LN_e(x + fb * FFN(LN_d x))
            ops.ffn_fused(ws.x16, Bk["w1b"], Bk["b1b"], Bk["w2b"], Bk["b2b"], ws.h32, Bk["lne"][0], Bk["lne"][1],
                          ws.h32, ws.h16, ops.ACT_SWISH, Bk["fb"], Bk["lne"][2])
            if i + 1 < nb:
                nx = P["blocks"][i + 1]["lna"]
                ops.layernorm_f16(ws.h32, nx[0], nx[1], ws.x16, nx[2])

    def _decode_span(self, P, ws, emb16, pc, B, Tc, Tp, C, states=None, carry_in=False, emb32=None):
        """Attractor decoder (LS model :215-220; merge_retnet_layer.py:233-253) over one span: emb16 (B*Tp, D) f16 unit
        embeddings -> ws.a32 (f32 attractor rows (b, c, t)).  `states`: as in _encode_span, per decoder layer."""
        D, H, L = self.n_units, self._n_heads, self.recurrent_chunk_size
        Md = B * C * Tp
        # the fused retention's query path reads the decoder rows as a hi / lo f16 pair (ret_stream.hip): the producers of the
        # layer inputs -- the f32 decoder-input linear and the layer tail -- write the remainder next to the f16 copy
        ret_fused = H == 4 and ops.retention_stream_ok(L, Tp)
        xlo = ret_fused and emb32 is not None and ws.a16lo.numel() > 0
        if emb32 is not None:
            # decoder input in f32 (exact-f32 MFMA): the retention's per-head LayerNorm (eps 1e-6) amplifies the f16 operand rounding
            # of this linear; at 12 speaker slots the f16 form left the 1e-3 bar (golden ls_c12_T1000: 1.3e-3)
            ops.convert_fanout_f32(emb32, P["convert.w32"], pc, ws.a32, ws.a16, B, Tp, C, out16lo=ws.a16lo if xlo else None)
        else:
            ops.convert_fanout(emb16, P["convert.w1"], pc, ws.a32, ws.a16, B, Tp, C)
        q, k, kt, vt = ws.q[:Md * D], ws.k[:Md * D], ws.kt[:Md * D], ws.vt[:Md * D]
        g, o16 = ws.g[:Md], ws.o16[:Md]
        nd = len(P["dec.layers"])
        for j, Ld in enumerate(P["dec.layers"]):
            if ret_fused:
                ops.retention_stream(ws.a16, ws.a16lo if xlo else None, Ld["wrs"], Ld["bqkvg"], o16, ws.st, ws.cscale, ws.sexp, B * C, Tp, L,
                                     Ld["gn_eps"], t_valid=Tc, state_in=states[j] if (states is not None and carry_in) else None,
                                     state_out=states[j] if states is not None else None)
            else:
                ops.retention_proj(ws.a16, Ld["wqkvg"], Ld["bqkvg"], q, k, kt, vt, g, B * C, Tp, H)
                ops.retention_chunk(q, k, kt, vt, g, o16, ws.st, ws.cscale, ws.sexp, B * C, H, Tp, L, Ld["gn_eps"], t_valid=Tc,
                                    state_in=states[j] if (states is not None and carry_in) else None,
                                    state_out=states[j] if states is not None else None)
            # out-projection + residual + norm11 + the speaker-axis attention in one launch, in place on the f32 stream (q, k, v in
            # registers; spk_stream.hip, any slot count the speaker-axis kernels take, C <= 12)
            ops.attnout_spk_stream_res32(o16, Ld["ws1"], Ld["out1_b"], ws.a32, Ld["g11"], Ld["be11"], Ld["eps11"], ws.a32, Ld["in2_b"],
                                         o16, B, C, Tp)
            # out-projection (hi / lo f16 weight pair) + norm21 + FFN + norm22 in one launch; f32 rows out, plus their hi / lo f16 copies
            # (ffn_stream.hip LO form: one wave owns 32 / 48 rows end to end; [327680, 2048]: 797 us un-packed ffn.hip -> 674 us before the lo product)
            lo_out = ws.a16lo if (xlo and j + 1 < nd) else None
            if Ld["ws2"] is not None:
                ops.attnout_ffn_stream_lo(o16, Ld["ws2"], Ld["out2_b"], ws.a32, Ld["g21"], Ld["be21"], Ld["eps21"], Ld["b1"], Ld["b2"],
                                          Ld["g22"], Ld["be22"], Ld["eps22"], ws.a32, ws.a16, lo_out)
            else:
                ops.attnout_ffn_fused(o16, Ld["out2_w"], Ld["out2_b"], ws.a32, Ld["g21"], Ld["be21"], Ld["eps21"], Ld["w1"], Ld["b1"],
                                      Ld["w2"], Ld["b2"], Ld["g22"], Ld["be22"], Ld["eps22"], ws.a32, ws.a16, out16lo=lo_out, wo_lo=Ld["out2_wlo"])

    def _run(self, src: Sequence[Tensor], ilens: Sequence[int], C: int):
        P = self._prepare()
        dev = self.cnn.weight.device
        D, L = self.n_units, self.recurrent_chunk_size
        srcs = [s.to(device=dev, dtype=torch.float32).contiguous() for s in src]
        B, T = len(srcs), max(int(s.shape[0]) for s in srcs)
        Tpad = math.ceil(T / L) * L                        # reference pads to a chunk multiple (:281-283)
        Tp = ops.frames_pad(Tpad)
        nc = (Tp + L - 1) // L
        ws = self._workspace(dev, B, Tp, C, nc)
        il_key = tuple(min(int(l), T) for l in ilens)
        if getattr(ws, "il_key", None) != il_key:
            ws.il = torch.tensor(il_key, dtype=torch.int32, device=dev)
            ws.il_key = il_key
        Me = B * Tp
        # chunks that are pure slab padding (rows Tpad..Tp) are skipped: the span's valid length is Tpad
        self._encode_span(P, ws, srcs, B, Tpad, Tp)

        # ---- truncate / zero re-pad, look-ahead conv, L2 (LS model :80-87)
        emb32 = torch.empty(Me, D, dtype=torch.float32, device=dev)
        if "cnn.ws" in P:
            ops.conv1d_l2norm_stream(ws.h16, P["cnn.ws"], P["cnn.b"], ws.il, emb32, ws.emb16, B, Tp, P["cnn.k"], P["cnn.pad"])
        else:
            ops.conv1d_l2norm(ws.h16, P["cnn.w"], P["cnn.b"], ws.il, emb32, ws.emb16, B, Tp, D, P["cnn.k"], P["cnn.pad"])
        self._decode_span(P, ws, ws.emb16, self._convert_const(C), B, Tpad, Tp, C, emb32=emb32)
        attr = torch.empty(B, T, C, D, dtype=torch.float32, device=dev)
        logits = torch.empty(B, T, C, dtype=torch.float32, device=dev)
        ops.head_l2dot(emb32, ws.a32, attr, logits, B, T, Tp, C, D)
        return logits, emb32.view(B, Tp, D), attr, T, Tp

    # ------------------------------------------------------------------ long-form: a few chunks at a time (BASELINE config 5)
    @torch.no_grad()
    def test_chunked(self, src, ilens, max_nspks=6, chunk_frames: Optional[int] = None, return_attractors: bool = True):
        """`test()` for long recordings with O(chunk) activation memory: the recording is walked in super-chunks of
        `chunk_frames` frames; what the reference's batch form carries implicitly is carried explicitly between calls
        -- per retention layer the f32 chunk state (retention.py:176-180), per Conformer conv module the k-1 input frames
        in front of the super-chunk (convolution.py:65-68).  The encoder output (0.5 KB/frame) is kept for the whole
        recording because the look-ahead Conv1d (LS model :86) needs 9 future frames at every super-chunk edge; it and the
        embeddings are the only O(T) buffers besides the outputs.

        chunk_frames must be a multiple of lcm(recurrent_chunk_size, 64) (8000 for the shipped configs): then every
        kernel sees each frame at the same position modulo its tile sizes as in the monolithic call, and the results are
        BIT-IDENTICAL to `test()` (tests/test_ls_longform.py).  Both share _encode_span / _decode_span."""
        P = self._prepare()
        dev = self.cnn.weight.device
        D, H, L = self.n_units, self._n_heads, self.recurrent_chunk_size
        unit = L * 64 // math.gcd(L, 64)
        if chunk_frames is None:
            chunk_frames = unit
        if chunk_frames % unit:
            raise ValueError(f"chunk_frames must be a multiple of lcm(recurrent_chunk_size, 64) = {unit}")
        srcs = [s.to(device=dev, dtype=torch.float32).contiguous() for s in src]
        B, T = len(srcs), max(int(s.shape[0]) for s in srcs)
        C = max_nspks
        Tpad = math.ceil(T / L) * L
        Tp_full = ops.frames_pad(Tpad)
        f16, f32 = torch.float16, torch.float32
        il = torch.tensor([min(int(l), T) for l in ilens], dtype=torch.int32, device=dev)
        enc16 = torch.zeros(B * Tp_full, D, dtype=f16, device=dev)            # encoder output, whole recording
        nb, nd = len(P["blocks"]), len(P["dec.layers"])
        enc_state = [torch.zeros(B, H, 64, 64, dtype=f32, device=dev) for _ in range(nb)]
        enc_halo = [None] * nb
        spans = [(s0, min(s0 + chunk_frames, Tpad)) for s0 in range(0, Tpad, chunk_frames)]

        # ---- pass 1: Conformer-retention encoder, super-chunk by super-chunk
        for s0, s1 in spans:
            Tc = s1 - s0
            Tp = ops.frames_pad(Tc)
            ws = self._workspace(dev, B, Tp, C, (Tp + L - 1) // L)
            part = [x[s0:min(s1, x.shape[0])] if x.shape[0] > s0 else x[:0] for x in srcs]     # may be empty: all pad (0)
            self._encode_span(P, ws, part, B, Tc, Tp, states=enc_state, halos=enc_halo, carry_in=s0 > 0)
            enc16.view(B, Tp_full, D)[:, s0:s1] = ws.h16.view(B, Tp, D)[:, :Tc]

        # ---- look-ahead conv + L2 norm over the whole recording (one launch; 1.5 KB/frame)
        emb32 = torch.empty(B * Tp_full, D, dtype=f32, device=dev)
        emb16 = torch.empty(B * Tp_full, D, dtype=f16, device=dev)
        if "cnn.ws" in P:
            ops.conv1d_l2norm_stream(enc16, P["cnn.ws"], P["cnn.b"], il, emb32, emb16, B, Tp_full, P["cnn.k"], P["cnn.pad"])
        else:
            ops.conv1d_l2norm(enc16, P["cnn.w"], P["cnn.b"], il, emb32, emb16, B, Tp_full, D, P["cnn.k"], P["cnn.pad"])
        del enc16

        # ---- pass 2: attractor decoder + head, super-chunk by super-chunk
        logits = torch.empty(B, T, C, dtype=f32, device=dev)
        attr = torch.empty(B, T, C, D, dtype=f32, device=dev) if return_attractors else None
        dec_state = [torch.zeros(B * C, H, 64, 64, dtype=f32, device=dev) for _ in range(nd)]
        pc = self._convert_const(C)
        for s0, s1 in spans:
            Tc = s1 - s0
            Tv = min(s1, T) - s0                                   # real frames of this super-chunk
            Tp = ops.frames_pad(Tc)
            ws = self._workspace(dev, B, Tp, C, (Tp + L - 1) // L)
            e16 = torch.zeros(B, Tp, D, dtype=f16, device=dev)
            e32 = torch.zeros(B, Tp, D, dtype=f32, device=dev)
            e16[:, :Tc] = emb16.view(B, Tp_full, D)[:, s0:s1]
            e32[:, :Tc] = emb32.view(B, Tp_full, D)[:, s0:s1]
            self._decode_span(P, ws, e16.view(-1, D), pc, B, Tc, Tp, C, states=dec_state, carry_in=s0 > 0, emb32=e32.view(-1, D))
            if Tv > 0:
                direct = B == 1 and return_attractors              # the output slices are contiguous: no staging copy
                lg = logits[:, s0:s0 + Tv] if direct else torch.empty(B, Tv, C, dtype=f32, device=dev)
                at = attr[:, s0:s0 + Tv] if direct else torch.empty(B, Tv, C, D, dtype=f32, device=dev)
                ops.head_l2dot(e32.view(-1, D), ws.a32, at, lg, B, Tv, Tp, C, D)
                if not direct:
                    logits[:, s0:s0 + Tv] = lg
                    if return_attractors:
                        attr[:, s0:s0 + Tv] = at
        emb = emb32.view(B, Tp_full, D)
        return ([logits[b, :l] for b, l in enumerate(ilens)], [emb[b, :l] for b, l in enumerate(ilens)],
                [attr[b, :l] for b, l in enumerate(ilens)] if return_attractors else None)

    @torch.no_grad()
    def test(self, src, ilens, max_nspks=6):
        """reference LS model :125-147."""
        logits, emb, attr, T, Tp = self._run(src, ilens, max_nspks)
        return ([logits[b, :l] for b, l in enumerate(ilens)], [emb[b, :l] for b, l in enumerate(ilens)],
                [attr[b, :l] for b, l in enumerate(ilens)])

    def forward(self, src, tgt, ilens):
        """reference LS model :74-122.  Under torch.no_grad(): values only (eval-mode numerics).  With gradients enabled
        (the reference's training_step, LS-EEND/train/oln_tfm_enc_dec_on_the_fly.py:78): the HIP training forward runs inside a
        torch.autograd.Function whose backward is the hand-written HIP backward (autograd.py, train_ls.LsTrainStep)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd import ls_forward_with_grad
            return ls_forward_with_grad(self, src, tgt, ilens)
        n_speakers = [t.shape[1] for t in tgt]
        C = max(n_speakers)
        logits, emb, attr, T, Tp = self._run(src, ilens, C)
        dev = logits.device
        seq_len = max(int(l) for l in ilens)
        tgt_pad = [nn.functional.pad(t.to(dev, torch.float32), (0, C - t.shape[1])) for t in tgt]
        tgt_pad = nn.utils.rnn.pad_sequence(tgt_pad, padding_value=0.0, batch_first=True).contiguous()
        lens = torch.tensor([int(l) for l in ilens], dtype=torch.int32, device=dev)
        loss = ops.emb_consistency(emb, tgt_pad, seq_len, lens=lens,              # length-masked embeddings (:100), sum / sum(len^2) (:113)
                                   inv_count=1.0 / sum(int(l) * int(l) for l in ilens))
        output = [logits[b, :l, :n] for b, (l, n) in enumerate(zip(ilens, n_speakers))]
        embs = [emb[b, :l].clone() for b, l in enumerate(ilens)]
        attractors = [attr[b, :l, 1:n] for b, (l, n) in enumerate(zip(ilens, n_speakers))]
        return output, loss, embs, attractors
