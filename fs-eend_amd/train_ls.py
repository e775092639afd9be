"""LS-EEND training step on MI355X: forward with saved activations, hand-written backward, Adam -- all HIP.

Reference behaviour reproduced (paths relative to the reference root, LS-EEND/):
  * nnet/model/onl_conformer_retention_enc_..._emb_loss_mask.py:74-122   model.forward in train mode: zero padding to the
    retention chunk multiple (:281-283), the Conformer conv modules' BatchNorm1d on batch statistics
    (conformer/convolution.py:143), the length-masked embedding-consistency loss (:92-113)
  * nnet/conformer/encoder.py:76-113   pre-norm residual blocks: x += 0.5 FFN(LN x); x += Ret(LN x); x += Conv(LN x);
    x += 0.5 FFN(LN x); LN
  * nnet/modules/retention.py:146-228  chunk-recurrent retention; `inner_scale` / `kv_scale` are detached (:163,:180), so
    the backward is that of a linear attention with a constant per-row factor (csrc/attn_bwd.hip RET, retention_bwd.hip)
  * nnet/modules/merge_retnet_layer.py:233-253  decoder layer: retention over time, MHA over speaker slots, ReLU FFN
  * train/oln_tfm_enc_dec_on_the_fly.py:52-92 training_step; train/utils/loss.py:136-142 standard_loss
  * train_dia_simu.py:97-117 Adam(betas (0.9, 0.98), eps 1e-9) x Noam; :159-173 gradient_clip_val, `sync_batchnorm`

SyncBatchNorm (train_dia_simu.py:167): every conv-module BatchNorm exchanges one (mean, M2, n) triple per rank in the
forward (all-gather of 513 floats, merged exactly on the device) and its two per-channel gradient sums in the backward
(all-reduce of 512 floats) over the training process group; with one rank nothing is exchanged.

Dropout sites (hash masks, see train.py): encoder block i -> 16*i + {0 FFN-a hidden, 1 FFN-a out, 2 retention out,
3 conv out, 4 FFN-b hidden, 5 FFN-b out}; decoder layer j -> 4096 + 16*j + {1 dropout11, 2 speaker-attention
probabilities, 3 dropout21, 4 FFN hidden, 5 dropout2}.

There is no autograd / eager fallback: everything below is a call into libeend_hip.so.
"""
import math
import os
from typing import Optional, Sequence

import torch
from torch import Tensor

from . import ops
from .lib import EendHipError
from .shard import all_reduce_bn_sums, gather_bn_stats
from .train import (BF16, D, F16, F32, H, I32, WS_FLOATS, TrainStepBase, _Site, _call, drop_step_seed)

# The Macaron half-step FFNs of the Conformer blocks as one training-forward launch each (eend_ffn_swish_train_f16): bit 0 = FFN_a,
# bit 1 = FFN_b.  Built in round 5 and kept off there because it moved the gradient norms of two ill-conditioned tensors of golden
# ls_train_clip (decoder layer 1's retention q / k projections) over a fixed 1e-2 bar; with the bar conditioned on the reference's own
# fp32-vs-fp64 gap of each tensor (round 6, tests/test_train_step_ls.py) every golden is green with both fused: on.
MACARON_FUSED = 3

ENC_FFA_HID, ENC_FFA_OUT, ENC_RET, ENC_CONV, ENC_FFB_HID, ENC_FFB_OUT = 0, 1, 2, 3, 4, 5
SITE_OUT1, SITE_SPK, SITE_OUT2, SITE_FF, SITE_FFOUT = 1, 2, 3, 4, 5


def never_graded(name: str) -> bool:
    """Parameters of the LS model that are not on the forward path: the decoder's dead input projection and the fusion
    layers' norm12 (merge_retnet_layer.py:93)."""
    return name.startswith("dec.encoder.") or name.startswith("dec.encoder_norm.") or ".norm12." in name


class _RetSave:
    """Saved tensors of one retention module: bf16 head layouts for the backward products, gate pre-activation,
    normalised rows, per-(row, head) 1/sigma * detached scale, and the gated output (out_proj input)."""

    def __init__(self, dev, nseq, Tp, L):
        n = nseq * Tp * D
        self.q, self.k, self.v = (torch.empty(n, dtype=BF16, device=dev) for _ in range(3))
        # [d][t] copies: only the two-kernel backward of chunk lengths beyond 512 reads them (attn_bwd.hip, retention_bwd.hip)
        self.qt, self.kt, self.vt = ((torch.empty(n, dtype=BF16, device=dev) for _ in range(3)) if L > 512 else (None, None, None))
        self.g = torch.empty(nseq * Tp, D, dtype=F16, device=dev)
        self.rhat = torch.zeros(nseq * Tp, D, dtype=F16, device=dev)        # rows of skipped padding chunks stay zero
        self.rc = torch.zeros(nseq * Tp, H, dtype=F32, device=dev)
        self.ctx = torch.zeros(nseq * Tp, D, dtype=F16, device=dev)


class _LsBuffers:
    """Device buffers of one (B, Tp, C) training shape."""

    def __init__(self, dev, B, Tp, C, n_enc, n_dec, F_enc, F_dec, Fin_pad, L):
        e = lambda *s, dt: torch.empty(*s, dtype=dt, device=dev)
        Me, Md = B * Tp, B * C * Tp
        Mx = max(Me, Md)
        nseq_max = max(B, B * C)
        nc = max(1, Tp // L)
        self.xin16 = torch.zeros(Me, Fin_pad, dtype=F16, device=dev)
        self.h32 = e(Me, D, dt=F32)
        self.site0 = _Site(dev, Me)
        self.enc = []
        for _ in range(n_enc):
            self.enc.append(dict(
                lnA=_Site(dev, Me), za=e(Me, F_enc, dt=F16), aa=e(Me, F_enc, dt=F16),
                lnB=_Site(dev, Me), ret=_RetSave(dev, B, Tp, L),
                lnC=_Site(dev, Me), P=e(Me, 2 * D, dt=F16), c16=e(Me, D, dt=F16), s16=e(Me, D, dt=F16),
                bn_mean=e(D, dt=F32), bn_var=e(D, dt=F32), bn_n=e(1, dt=F32),
                lnD=_Site(dev, Me), zb=e(Me, F_enc, dt=F16), ab=e(Me, F_enc, dt=F16),
                lnE=_Site(dev, Me)))
        self.emb32, self.emb16, self.inv_norm = e(Me, D, dt=F32), e(Me, D, dt=F16), e(Me, dt=F32)
        self.a32, self.a16 = e(Md, D, dt=F32), e(Md, D, dt=F16)
        self.dec = [dict(ret=_RetSave(dev, B * C, Tp, L), s11=_Site(dev, Md), qkv=e(Md, 3 * D, dt=F16), o2=e(Md, D, dt=F16),
                         s21=_Site(dev, Md), hid=e(Md, F_dec, dt=F16), s22=_Site(dev, Md)) for _ in range(n_dec)]
        # forward transients (f16 operands of the retention forward kernels)
        self.fq, self.fk, self.fkt, self.fvt = (e(Mx * D, dt=F16) for _ in range(4))
        self.st = e(nseq_max * H * nc * 2 * 4096, dt=F16)
        self.cscale, self.sexp = e(nseq_max * H * nc, dt=F32), e(nseq_max * H * nc, dt=F32)
        self.kv_ws = e(nseq_max * H * nc * 4096, dt=F32)
        self.bn_stats = e(2 * D + 1, dt=F32)
        # backward temporaries
        self.g32 = e(Md, D, dt=F32)
        self.ge32 = e(Me, D, dt=F32)
        self.de32 = e(Me, D, dt=F32)
        self.ds16 = e(Mx, D, dt=BF16)
        self.dctx16 = e(Mx, D, dt=BF16)
        self.dctx32 = e(Mx, D, dt=F32)
        self.dy16 = e(Mx, D, dt=BF16)
        self.dh16 = e(max(Me * max(F_enc, 2 * D), Md * F_dec), dt=BF16)
        self.dqkvg = e(Mx, 4 * D, dt=BF16)
        self.dqkv16 = e(Md, 3 * D, dt=BF16)
        self.ot = e(Mx * D, dt=BF16)
        self.ott = e(Mx * D, dt=BF16) if L > 512 else None      # head-transposed o~: the two-kernel backward only
        self.g_ws = e(nseq_max * H * nc * 4096, dt=F32)
        self.st_bwd = e(nseq_max * H * nc * 6 * 4096, dt=BF16)
        self.bn_sums = e(2 * D, dt=F32)
        self.gsum16 = e(Me, D, dt=BF16)
        self.demb16 = e(Me, D, dt=BF16)
        self.conv_tmp = e(D * 19 * D, dt=F32)
        self.dpc = e(C, D, dt=F32)
        self.pc = e(C, D, dt=F32)
        self.logits = e(B, Tp, C, dt=F32)
        self.loss = torch.zeros(4, dtype=F32, device=dev)         # [bce, emb, -, -]


class LsTrainStep(TrainStepBase):
    """One training step of LS-EEND (`OnlineConformerRetentionDADiarization` mirror) entirely in HIP."""

    def __init__(self, model, warmup: Optional[int] = 100000, lr: float = 1.0, schedule_scale: float = 1.0, grad_clip: float = 5.0,
                 betas=(0.9, 0.98), eps: float = 1e-9, bn_momentum: float = 0.1, process_group=None, drop_seed: int = 0,
                 sync_batchnorm: bool = True):
        from .ls_model import OnlineConformerRetentionDADiarization
        if not isinstance(model, OnlineConformerRetentionDADiarization):
            raise TypeError("LsTrainStep drives fs_eend_amd.ls_model.OnlineConformerRetentionDADiarization")
        self.L = int(model.recurrent_chunk_size)
        if self.L > 512 or self.L % 4:
            raise NotImplementedError("the training retention kernels keep a chunk on chip: recurrent_chunk_size <= 512, % 4 == 0")
        for name, buf in model.named_buffers():
            if name.endswith(".decay") and bool((buf != 0).any()):
                raise NotImplementedError(f"{name} is not all-zero: the HIP retention kernels implement decay == 1 only")
        self.sync_bn = bool(sync_batchnorm)
        self._init_common(model, warmup, lr, schedule_scale, grad_clip, betas, eps, bn_momentum, process_group, drop_seed)
        self._build_weight_table()

    # ------------------------------------------------------------------ weight operand copies
    def _ret_tables(self, key, pfx):
        """Packed retention projections of `pfx` (q_proj, k_proj, v_proj, g_proj: four (256, 256) + (256,) parameters laid
        out back to back in the flat buffer): forward [q; k * dk^-1/2; v; g] f16 (1024, 256) + bias f32 (1024), and the
        un-scaled transposed bf16 (256, 1024) of the data-gradient GEMM (the backward delivers sk * dk)."""
        fl = self.flat
        names = [pfx + n for n in ("q_proj", "k_proj", "v_proj", "g_proj")]
        stride = fl.offsets[names[1] + ".weight"] - fl.offsets[names[0] + ".weight"]
        for a, b in zip(names[:-1], names[1:]):
            if fl.offsets[b + ".weight"] - fl.offsets[a + ".weight"] != stride or fl.offsets[b + ".bias"] - fl.offsets[a + ".bias"] != stride:
                raise EendHipError("unexpected flat layout of the retention projections")
        s = 64 ** -0.5
        w = torch.zeros(4 * D, D, dtype=F16, device=self.dev)
        b = torch.zeros(4 * D, dtype=F32, device=self.dev)
        wT = torch.zeros(D, 4 * D, dtype=BF16, device=self.dev)
        qw, qb = names[0] + ".weight", names[0] + ".bias"
        self._add_entry(w, 0, qw, (1, D, D), (0, D, 1), 0)
        self._add_entry(w, D * D, qw, (1, D, D), (0, D, 1), 0, off=stride, nscale=1, scale=s)
        self._add_entry(w, 2 * D * D, qw, (2, D, D), (stride, D, 1), 0, off=2 * stride)
        self._add_entry(b, 0, qb, (1, 1, D), (0, 0, 1), 2)
        self._add_entry(b, D, qb, (1, 1, D), (0, 0, 1), 2, off=stride, nscale=1, scale=s)
        self._add_entry(b, 2 * D, qb, (2, 1, D), (stride, 0, 1), 2, off=2 * stride)
        self._add_entry(wT, 0, qw, (D, 4, D), (1, stride, D), 1)           # wT[k][j*256 + n] = W_j[n][k]
        self.W[key + ".wqkvg"], self.W[key + ".bqkvg"], self.W[key + ".wqkvgT"] = w, b, wT

    def _build_weight_table(self):
        m, dev = self.model, self.dev
        self._table_begin()
        plain, transposed, add = self._plain, self._transposed, self._add
        e = m.enc.encoder
        Fin = m._in_size
        self.Fin, self.Fin_pad = Fin, (Fin + 127) // 128 * 128
        plain("in.w", "enc.encoder.input_projection.linear.weight", D, Fin, kpad=self.Fin_pad)
        self.F_enc = e.layers[0].sequential[0].module.sequential[1].linear.out_features if len(e.layers) else D
        self.kdw = e._conv_kernel_size
        for i in range(len(e.layers)):
            s_ = f"enc.encoder.layers.{i}.sequential."
            for tag, j in (("a", 0), ("b", 3)):
                plain(f"e{i}.w1{tag}", s_ + f"{j}.module.sequential.1.linear.weight", self.F_enc, D)
                transposed(f"e{i}.w1{tag}T", s_ + f"{j}.module.sequential.1.linear.weight", self.F_enc, D)
                plain(f"e{i}.w2{tag}", s_ + f"{j}.module.sequential.4.linear.weight", D, self.F_enc)
                transposed(f"e{i}.w2{tag}T", s_ + f"{j}.module.sequential.4.linear.weight", D, self.F_enc)
            self._ret_tables(f"e{i}", s_ + "1.module.self_attn.")
            plain(f"e{i}.wo", s_ + "1.module.self_attn.out_proj.weight", D, D)
            transposed(f"e{i}.woT", s_ + "1.module.self_attn.out_proj.weight", D, D)
            plain(f"e{i}.pw1", s_ + "2.module.sequential.2.conv.weight", 2 * D, D)           # (512, 256, 1): value rows, gate rows
            transposed(f"e{i}.pw1T", s_ + "2.module.sequential.2.conv.weight", 2 * D, D)
            plain(f"e{i}.pw2", s_ + "2.module.sequential.7.conv.weight", D, D)
            transposed(f"e{i}.pw2T", s_ + "2.module.sequential.7.conv.weight", D, D)
        k = m.cnn.kernel_size[0]
        self.ktaps, self.cpad = k, m.cnn.padding[0]
        add("cnn.w", "cnn.weight", (D, k, D), (D * k, 1, k), 0, alloc=(D, k * D))
        add("cnn.wd", "cnn.weight", (D, k, D), (k, -1, D * k), 1, off=k - 1, alloc=(D, k * D))
        add("convert.w1", "dec.convert.weight", (D, 1, D), (2 * D, 0, 1), 0)
        add("convert.w1T", "dec.convert.weight", (D, 1, D), (1, 0, 2 * D), 1)
        self.F_dec = m.dec.layers[0].linear1.out_features if len(m.dec.layers) else D
        for i, l in enumerate(m.dec.layers):
            p_ = f"dec.layers.{i}."
            Fh = l.linear1.out_features
            self._ret_tables(f"d{i}", p_ + "self_attn1.")
            plain(f"d{i}.out1_w", p_ + "self_attn1.out_proj.weight", D, D)
            transposed(f"d{i}.out1_wT", p_ + "self_attn1.out_proj.weight", D, D)
            plain(f"d{i}.in2_w", p_ + "self_attn2.in_proj_weight", 3 * D, D)
            transposed(f"d{i}.in2_wT", p_ + "self_attn2.in_proj_weight", 3 * D, D)
            plain(f"d{i}.out2_w", p_ + "self_attn2.out_proj.weight", D, D)
            transposed(f"d{i}.out2_wT", p_ + "self_attn2.out_proj.weight", D, D)
            plain(f"d{i}.w1", p_ + "linear1.weight", Fh, D)
            transposed(f"d{i}.w1T", p_ + "linear1.weight", Fh, D)
            plain(f"d{i}.w2", p_ + "linear2.weight", D, Fh)
            transposed(f"d{i}.w2T", p_ + "linear2.weight", D, Fh)
        self._table_end()
        self.pe = m.dec.pos_enc.pe[0].to(device=dev, dtype=F32).contiguous()

    # ------------------------------------------------------------------ helpers
    def _buffers(self, B, Tp, C) -> _LsBuffers:
        key = (B, Tp, C)
        b = self._bufs.get(key)
        if b is None:
            if len(self._bufs) >= 2:
                self._bufs.clear()
            m = self.model
            b = _LsBuffers(self.dev, B, Tp, C, len(m.enc.encoder.layers), len(m.dec.layers), self.F_enc, self.F_dec, self.Fin_pad,
                           self.L)
            self._bufs[key] = b
        return b

    def _pit_assign(self, P, ys, ts, n_spk):
        return P.pit_loss_multispk(ys, ts, n_spk)           # train/oln_tfm_enc_dec_spk_pit_on_the_fly.py:92 (Hungarian)

    def _prenorm_out(self, a16, K, w, bias, alpha, ln, site: _Site, h32, M, drop):
        """h32 <- dropout(a16 W^T + bias) * alpha + h32; site <- LayerNorm(h32) (the next sub-layer's pre-norm)."""
        _call("eend_linear_res_scale_ln_train_f16", a16, a16.stride(0), w, w.stride(0), bias, h32, alpha, self._P(ln + ".weight"),
              self._P(ln + ".bias"), 1e-5, h32, site.out16, site.xhat, site.rstd, M, K, drop)

    proj_stream_min_rows = 49152        # rows from which the packed-stream projection beats the two launches (tests set 0 to force it)

    def _ret_fwd(self, bf, x16, wkey, sv: _RetSave, nseq, Tp, Tv):
        W = self.W
        n = nseq * Tp * D
        q, k, kt, vt = bf.fq[:n], bf.fk[:n], bf.fkt[:n], bf.fvt[:n]
        ws = self._proj_streams.get(wkey + ".wqkvg")
        groups = [dict(rows=q, kind=2, rows2=sv.q), dict(rows=k, kind=2, rows2=sv.k, heads_t=kt), dict(rows2=sv.v, heads_t=vt),
                  dict(rows=sv.g, kind=1, ld=D)]
        if ws is not None and sv.qt is None and nseq * Tp >= self.proj_stream_min_rows and ops.proj_stream_ok(x16.stride(0), nseq * Tp, 4 * D, Tp, H, groups):
            # one pass over the rows for the f16 operands of the forward kernel AND the bf16 head rows the backward keeps (proj_stream.hip:
            # [393216, 1024] 526 us against 803 us for the two projections below, [65536, 1024] 101 against 149 us; below about 48 k rows
            # the 256-row tiles leave CUs idle and the two launches win)
            ops.proj_stream(x16, ws, W[wkey + ".bqkvg"], nseq * Tp, 4 * D, Tp, H, groups)
        else:
            ops.retention_proj(x16, W[wkey + ".wqkvg"], W[wkey + ".bqkvg"], q, k, kt, vt, sv.g, nseq, Tp, H)
            _call("eend_inproj_heads_train_bf16", x16, x16.stride(0), W[wkey + ".wqkvg"], W[wkey + ".bqkvg"], sv.q, sv.qt, sv.k, sv.kt,
                  sv.v, sv.vt, nseq, Tp, H)
        _call("eend_retention_chunk_train_f16", q, k, kt, vt, sv.g, sv.ctx, sv.rhat, sv.rc, bf.st, bf.kv_ws, bf.cscale, bf.sexp, nseq, H,
              Tp, self.L, D, D, 1e-6, Tv)

    # ------------------------------------------------------------------ forward (saves activations)
    def forward(self, src: Sequence[Tensor], labels: Sequence[Tensor], ilens: Sequence[int], pit: bool = False, dropout: bool = True,
                fused_loss: bool = True):
        """Train-mode model.forward with every activation the backward needs saved.  labels: prepared (T_i, nspk_i+2)
        tensors (oln_tfm_enc_dec_on_the_fly.py:53-75).
        fused_loss=True : also standard_loss + masked emb-consistency loss and their gradients w.r.t. attractors /
                          embeddings (bf.loss = [bce, emb_loss]); `backward(bf)` then completes the step.
        fused_loss=False: stop at the head (bf.logits_full (B,T,C), bf.attr_n (B,T,C,D), bf.loss[1] = emb_loss): the
                          caller computes its own loss on the logits and passes d loss / d logits to `backward`."""
        m, W, dev, L = self.model, self.W, self.dev, self.L
        srcs = [s.to(device=dev, dtype=F32).contiguous() for s in src]
        B, T = len(srcs), max(int(s.shape[0]) for s in srcs)
        ncols = [int(l.shape[1]) for l in labels]
        C = max(ncols)
        Tv = math.ceil(T / L) * L                       # the reference's padded length (LS model :281-283)
        Tp = ops.frames_pad(Tv)
        bf = self._buffers(B, Tp, C)
        Me, Md = B * Tp, B * C * Tp
        il = [min(int(l), T) for l in ilens]
        if any(int(l) != int(s.shape[0]) for l, s in zip(ilens, srcs)):
            # the reference pads the decoder and windows the emb-consistency loss by max(ilens) AFTER truncating the embeddings
            # (LS model :80-84, :100); this step derives both from the feature lengths, which is the same thing only when they agree
            # -- as they always do in the reference's training_step
            raise EendHipError("LsTrainStep.forward: ilens must equal the feature lengths (ilens shorter than the features are not supported)")
        key = (tuple(il), tuple(ncols), T)
        if getattr(bf, "len_key", None) != key:              # cached: no H2D copies in the steady state
            bf.il = torch.tensor(il, dtype=I32, device=dev)
            bf.tl = torch.full((B,), Tv, dtype=I32, device=dev)
            bf.nc = torch.tensor(ncols, dtype=I32, device=dev)
            bf.inv_sq = 1.0 / float(sum(l * l for l in il))
            bf.len_key = key
        bf.shape = (B, T, Tp, C)
        bf.Tv = Tv
        self._fwd_count += 1
        bf.drop_base = drop_step_seed(self.drop_seed, self._fwd_count) if (dropout and self.drop_p > 0.0) else None
        bf.drop_specs = {}
        dr = lambda site: self._drop(bf, site)
        n_frames = sum(int(l.shape[0]) for l in labels)
        if all(tuple(l.shape) == (T, C) for l in labels):
            lab = torch.stack([l.to(device=dev, dtype=F32) for l in labels]).contiguous()
        else:
            lab = torch.zeros(B, T, C, dtype=F32, device=dev)
            for b_, l in enumerate(labels):
                lab[b_, :l.shape[0], :l.shape[1]] = l.to(device=dev, dtype=F32)
        bf.labels = lab

        # ---- input projection + LayerNorm (conformer/encoder.py:194-196); pad_sequence(0) + cast in one launch
        ptrs, lens = self._table_for(srcs, T)
        _call("eend_gather_bn_cast_pad_f16", ptrs, lens, 0.0, None, None, None, None, 0.0, bf.xin16, B, T, Tp, self.Fin, self.Fin_pad, 0)
        self._linear_ln(bf.xin16, W["in.w"], self._P("enc.encoder.input_projection.linear.bias"), None, "enc.encoder.layer_norm",
                        bf.site0, bf.h32, Me, self.Fin_pad)
        h32 = bf.h32
        # the Macaron half-step FFNs as one launch each (eend_ffn_swish_train_f16; MACARON_FUSED above) where the shape allows it
        _mac = MACARON_FUSED
        fused_ffn = self.F_enc % 64 == 0 and (Me + 128) * self.F_enc * 2 < (1 << 32)      # the Macaron FFNs as one launch each
        fused_a, fused_b = fused_ffn and bool(_mac & 1), fused_ffn and bool(_mac & 2)
        for i, sv in enumerate(bf.enc):
            s_ = f"enc.encoder.layers.{i}.sequential."
            so = 16 * i
            ffa, ffb, ret, cm = s_ + "0.module.sequential.", s_ + "3.module.sequential.", s_ + "1.module.", s_ + "2.module.sequential."
            # x += 0.5 * FFN_a(LN x)                                              (feed_forward.py:47-57)
            _call("eend_layernorm_train_f16", h32, self._P(ffa + "0.weight"), self._P(ffa + "0.bias"), 1e-5, sv["lnA"].out16,
                  sv["lnA"].xhat, sv["lnA"].rstd, Me)
            if fused_a:          # z, swish + dropout, second linear, dropout, half-step residual, the next LayerNorm: one launch (ffn.hip MODE 3)
                _call("eend_ffn_swish_train_f16", sv["lnA"].out16, D, W[f"e{i}.w1a"], self._P(ffa + "1.linear.bias"), W[f"e{i}.w2a"],
                      self._P(ffa + "4.linear.bias"), h32, 0.5, self._P(ret + "layer_norm.weight"), self._P(ret + "layer_norm.bias"), 1e-5, h32,
                      sv["lnB"].out16, sv["za"], sv["aa"], sv["lnB"].xhat, sv["lnB"].rstd, Me, self.F_enc, 1, dr(so + ENC_FFA_HID),
                      dr(so + ENC_FFA_OUT))
            else:
                ops.linear(sv["lnA"].out16, W[f"e{i}.w1a"], self._P(ffa + "1.linear.bias"), sv["za"])
                _call("eend_swish_dropout_f16", sv["za"], sv["aa"], Me, self.F_enc, dr(so + ENC_FFA_HID))
                self._prenorm_out(sv["aa"], self.F_enc, W[f"e{i}.w2a"], self._P(ffa + "4.linear.bias"), 0.5, ret + "layer_norm", sv["lnB"],
                                  h32, Me, dr(so + ENC_FFA_OUT))
            # x += Retention(LN x)                                                (conformer/attention.py:99-112)
            self._ret_fwd(bf, sv["lnB"].out16, f"e{i}", sv["ret"], B, Tp, Tv)
            self._prenorm_out(sv["ret"].ctx, D, W[f"e{i}.wo"], self._P(ret + "self_attn.out_proj.bias"), 1.0, cm + "0", sv["lnC"], h32,
                              Me, dr(so + ENC_RET))
            # x += ConvModule(LN x): 1x1 + GLU, causal depthwise, BatchNorm (batch statistics), swish, 1x1  (convolution.py:138-149)
            ops.linear(sv["lnC"].out16, W[f"e{i}.pw1"], self._P(cm + "2.conv.bias"), sv["P"])
            _call("eend_glu_dwconv_f16", sv["P"], self._P(cm + "4.conv.weight"), sv["c16"], B, Tp, Tv, self.kdw)
            _call("eend_bn_batch_stats_f16", sv["c16"], self.ws, WS_FLOATS, bf.bn_stats, B, Tp, Tv)
            # SyncBatchNorm: one (mean, M2, n) triple per rank, merged exactly (shard.gather_bn_stats; R = 1 without peers)
            stats, R = (bf.bn_stats.view(1, -1), 1) if not self.sync_bn else gather_bn_stats(bf.bn_stats, self.group)
            bnm = m.enc.encoder.layers[i].sequential[2].module.sequential[5]
            _call("eend_bn_merge_f32", stats, R, sv["bn_mean"], sv["bn_var"], sv["bn_n"], bnm.running_mean, bnm.running_var,
                  self.bn_momentum)
            bnm.num_batches_tracked += 1
            _call("eend_bn_swish_f16", sv["c16"], sv["bn_mean"], sv["bn_var"], bnm.eps, self._P(cm + "5.weight"), self._P(cm + "5.bias"),
                  sv["s16"], Me)
            self._prenorm_out(sv["s16"], D, W[f"e{i}.pw2"], self._P(cm + "7.conv.bias"), 1.0, ffb + "0", sv["lnD"], h32, Me,
                              dr(so + ENC_CONV))
            # x = LN(x + 0.5 * FFN_b(LN x))
            if fused_b:
                _call("eend_ffn_swish_train_f16", sv["lnD"].out16, D, W[f"e{i}.w1b"], self._P(ffb + "1.linear.bias"), W[f"e{i}.w2b"],
                      self._P(ffb + "4.linear.bias"), h32, 0.5, self._P(s_ + "4.weight"), self._P(s_ + "4.bias"), 1e-5, h32,
                      sv["lnE"].out16, sv["zb"], sv["ab"], sv["lnE"].xhat, sv["lnE"].rstd, Me, self.F_enc, 0, dr(so + ENC_FFB_HID),
                      dr(so + ENC_FFB_OUT))
            else:
                ops.linear(sv["lnD"].out16, W[f"e{i}.w1b"], self._P(ffb + "1.linear.bias"), sv["zb"])
                _call("eend_swish_dropout_f16", sv["zb"], sv["ab"], Me, self.F_enc, dr(so + ENC_FFB_HID))
                self._linear_ln(sv["ab"], W[f"e{i}.w2b"], self._P(ffb + "4.linear.bias"), h32, s_ + "4", sv["lnE"], h32, Me, self.F_enc,
                                dr(so + ENC_FFB_OUT), alpha=0.5)
        enc_out16 = bf.enc[-1]["lnE"].out16 if bf.enc else bf.site0.out16
        bf.enc_out16 = enc_out16

        # ---- truncate / zero re-pad, look-ahead conv, L2 norm (LS model :80-87)
        _call("eend_conv1d_l2norm_train_f16", enc_out16, W["cnn.w"], self._P("cnn.bias"), bf.il, bf.emb32, bf.emb16, bf.inv_norm, B, Tp, D,
              self.ktaps, self.cpad)

        # ---- attractor decoder (LS model :215-220; merge_retnet_layer.py:233-253)
        _call("eend_convert_const_f32", 0, self._P("dec.convert.weight"), self._P("dec.convert.bias"), self.pe, bf.pc, None, None,
              None, C)
        ops.convert_fanout(bf.emb16, W["convert.w1"], bf.pc, bf.a32, bf.a16, B, Tp, C)
        x16 = bf.a16
        for i, sv in enumerate(bf.dec):
            p_ = f"dec.layers.{i}."
            so = 4096 + 16 * i
            self._ret_fwd(bf, x16, f"d{i}", sv["ret"], B * C, Tp, Tv)
            self._linear_ln(sv["ret"].ctx, W[f"d{i}.out1_w"], self._P(p_ + "self_attn1.out_proj.bias"), bf.a32, p_ + "norm11", sv["s11"],
                            bf.a32, Md, D, dr(so + SITE_OUT1))
            ops.linear(sv["s11"].out16, W[f"d{i}.in2_w"], self._P(p_ + "self_attn2.in_proj_bias"), sv["qkv"])
            _call("eend_spk_attn_train_f16", sv["qkv"], sv["o2"], B, C, Tp, H, 0.125, dr(so + SITE_SPK))
            self._linear_ln(sv["o2"], W[f"d{i}.out2_w"], self._P(p_ + "self_attn2.out_proj.bias"), bf.a32, p_ + "norm21", sv["s21"],
                            bf.a32, Md, D, dr(so + SITE_OUT2))
            self._ffn(sv["s21"].out16, W[f"d{i}.w1"], self._P(p_ + "linear1.bias"), sv["hid"], W[f"d{i}.w2"], self._P(p_ + "linear2.bias"),
                      bf.a32, p_ + "norm22", sv["s22"], bf.a32, Md, dr(so + SITE_FF), dr(so + SITE_FFOUT))
            x16 = sv["s22"].out16

        # ---- head + BCE (+ PIT label choice) + masked emb-consistency loss, and their gradients (LS model :89-117)
        if not fused_loss:
            bf.logits_full = torch.empty(B, T, C, dtype=F32, device=dev)
            bf.attr_n = torch.empty(B, T, C, D, dtype=F32, device=dev)
            ops.head_l2dot(bf.emb32, bf.a32, bf.attr_n, bf.logits_full, B, T, Tp, C, D)
            bf.loss[1] = ops.emb_consistency(bf.emb32.view(B, Tp, D), lab, T, lens=bf.il, inv_count=bf.inv_sq)
            return bf
        if pit:
            lab = self._pit_labels(bf, lab, il, ncols)
            bf.labels = lab
        _call("eend_head_bce_f32", bf.emb32, bf.a32, lab, bf.il, bf.nc, 1.0 / float(n_frames), None, bf.logits, bf.g32, bf.de32,
              self.ws, WS_FLOATS, bf.loss[0:1], B, T, Tp, C)
        bf.loss[1] = ops.emb_consistency(bf.emb32.view(B, Tp, D), lab, T, lens=bf.il, inv_count=bf.inv_sq)
        _call("eend_emb_consistency_bwd_f16", bf.emb16, lab, bf.il, bf.inv_sq, bf.de32, B, T, Tp, D, C)
        return bf

    # ------------------------------------------------------------------ backward
    def _ln_bwd2(self, g, g_is_bf16, site: _Site, ln, ds32, accumulate, M, ds16=None, alpha16=1.0, drop=None, bias=None):
        _call("eend_layernorm_bwd2_f32", g, 1 if g_is_bf16 else 0, site.xhat, site.rstd, self._P(ln + ".weight"), ds32,
              1 if accumulate else 0, ds16, alpha16, self.ws, WS_FLOATS, self._G(ln + ".weight"), self._G(ln + ".bias"),
              None if bias is None else self._G(bias), M, drop)

    def _resgrad(self, g32, ds16, alpha, bias, M, drop):
        _call("eend_resgrad_cast_bf16", g32, ds16, alpha, self.ws, WS_FLOATS, self._G(bias), M, drop)

    def _ret_bwd(self, bf, g32, ds16, sv: _RetSave, x_in16, nseq, M, wkey, pfx, prenorm_site=None, prenorm_ln=None, next_ln=None):
        """backward of x -> x + out_proj(retention(x')) given ds16 = gradient w.r.t. the branch output (bf16); x' = x
        (decoder, post-norm: the input gradient joins g32 directly) or x' = LN(x) (encoder: through the LayerNorm)."""
        W = self.W
        B, T, Tp, C = bf.shape
        dctx, dq = bf.dctx32[:M], bf.dqkvg[:M]
        self._wgrad(ds16, sv.ctx, M, D, D, pfx + "out_proj.weight")
        # the out-projection's data gradient stays f32: the per-head LayerNorm backward that consumes it cancels its two
        # largest components (mean and the component along rhat), which amplifies any rounding applied before it
        _call("eend_gemm_acc_bf16", ds16, D, W[wkey + ".woT" if wkey[0] == "e" else wkey + ".out1_wT"], D, None, 1.0, dctx, None, M, D)
        _call("eend_retention_bwd_bf16", sv.q, sv.qt, sv.k, sv.kt, sv.v, sv.vt, dctx, sv.g, D, sv.rhat, sv.rc, bf.ot, bf.ott, bf.kv_ws,
              bf.g_ws, bf.st_bwd, dq, 4 * D, nseq, H, Tp, self.L, bf.Tv, 0.125)
        projs = ("q_proj", "k_proj", "v_proj", "g_proj")
        offw = [self.flat.offsets[pfx + nm + ".weight"] for nm in projs]
        offb = [self.flat.offsets[pfx + nm + ".bias"] for nm in projs]
        stride = offw[1] - offw[0]
        if all(offw[j] - offw[0] == j * stride and offb[j] - offb[0] == j * stride for j in range(4)) and stride >= D * D:
            # the four projections' (weight, bias) gradients are equally spaced in the flat buffer: ONE weight-gradient launch over dq's 1024
            # columns (x read once instead of four times, two reductions instead of eight)
            _call("eend_wgrad_bias_grouped_bf16", dq, 4 * D, x_in16, x_in16.stride(0), 1, M, 4 * D, D, self.ws, WS_FLOATS,
                  self._G(pfx + "q_proj.weight"), self._G(pfx + "q_proj.bias"), D, stride, 1.0)
        else:
            for j, nm in enumerate(projs):
                blk = dq[:, j * D:(j + 1) * D]
                _call("eend_wgrad_bias_bf16", blk, 4 * D, x_in16, x_in16.stride(0), 1, M, D, D, self.ws, WS_FLOATS,
                      self._G(pfx + nm + ".weight"), D, D, self._G(pfx + nm + ".bias"), 1.0, 0)
        if prenorm_site is None and next_ln is not None:
            site, ln, ndrop, nbias = next_ln
            self._gemm_acc_ln_bwd(dq, 4 * D, W[wkey + ".wqkvgT"], g32, site, ln, ds16, M, ndrop, nbias)
        elif prenorm_site is None:
            self._gemm_acc(dq, 4 * D, W[wkey + ".wqkvgT"], g32, M)
        else:
            dy = bf.dy16[:M]
            _call("eend_gemm_bf16", dq, 4 * D, W[wkey + ".wqkvgT"], 4 * D, None, dy, D, M, D, 4 * D)
            self._ln_bwd2(dy, True, prenorm_site, prenorm_ln, g32, True, M)

    def _ffn_swish_bwd(self, bf, g32, ds16, z16, a16, site: _Site, ln, M, wkey, tag, pfx, hid_drop):
        """backward of the branch W2 dropout(swish(W1 LN(x) + b1)) + b2 given ds16 = gradient w.r.t. the branch output."""
        W = self.W
        F_ = self.F_enc
        dh = bf.dh16[:M * F_].view(M, F_)
        self._wgrad(ds16, a16, M, D, F_, pfx + "4.linear.weight")
        _call("eend_gemm_bf16", ds16, D, W[f"{wkey}.w2{tag}T"], D, None, dh, F_, M, F_, D)
        _call("eend_swish_bwd_bf16", dh, z16, M, F_, hid_drop)
        self._wgrad_bias(dh, site.out16, M, F_, D, pfx + "1.linear.weight", pfx + "1.linear.bias")
        dy = bf.dy16[:M]
        _call("eend_gemm_bf16", dh, F_, W[f"{wkey}.w1{tag}T"], F_, None, dy, D, M, D, F_)
        self._ln_bwd2(dy, True, site, ln, g32, True, M)

    def backward(self, bf: _LsBuffers, dlogits: Optional[Tensor] = None, emb_loss_grad: float = 1.0):
        """Gradients w.r.t. every parameter -> self.flat.grads.  After forward(fused_loss=True): of bce + emb_loss.  After
        forward(fused_loss=False): of the caller's loss, given dlogits = d loss / d logits (B, T, C) f32 and
        emb_loss_grad = d loss / d emb_loss."""
        W = self.W
        B, T, Tp, C = bf.shape
        Tv = bf.Tv
        if dlogits is not None:
            dl = dlogits.to(device=self.dev, dtype=F32).contiguous()
            if tuple(dl.shape) != (B, T, C):
                raise EendHipError(f"backward: dlogits must be ({B}, {T}, {C})")
            _call("eend_head_bce_f32", bf.emb32, bf.a32, None, None, None, 0.0, dl, None, bf.g32, bf.de32, self.ws, WS_FLOATS,
                  bf.loss[2:3], B, T, Tp, C)
            if emb_loss_grad != 0.0:
                _call("eend_emb_consistency_bwd_f16", bf.emb16, bf.labels, bf.il, float(emb_loss_grad) * bf.inv_sq, bf.de32, B, T, Tp, D, C)
        Me, Md = B * Tp, B * C * Tp
        m = self.model
        ds16, dctx16, dqkv16 = bf.ds16, bf.dctx16, bf.dqkv16
        dr = lambda site: self._drop(bf, site)
        ff_scale = 1.0 / (1.0 - self.drop_p) if bf.drop_base is not None else 1.0

        # ---- decoder layers, last to first; bf.g32 = gradient w.r.t. the layer output
        g32 = bf.g32
        for i in reversed(range(len(bf.dec))):
            sv = bf.dec[i]
            p_ = f"dec.layers.{i}."
            x_in16 = bf.dec[i - 1]["s22"].out16 if i > 0 else bf.a16
            dsd = ds16[:Md]
            so = 4096 + 16 * i
            if i == len(bf.dec) - 1:      # (the other layers' norm22 backward ran in the epilogue of the layer above's last data-gradient GEMM)
                self._ln_bwd(g32, sv["s22"], p_ + "norm22", dsd, Md, dr(so + SITE_FFOUT), p_ + "linear2.bias")
            self._ffn_bwd(g32, dsd, bf.dh16, sv["hid"], sv["s21"].out16, Md, f"d{i}", p_, "norm22", ff_scale)
            self._ln_bwd(g32, sv["s21"], p_ + "norm21", dsd, Md, dr(so + SITE_OUT2), p_ + "self_attn2.out_proj.bias")
            self._wgrad(dsd, sv["o2"], Md, D, D, p_ + "self_attn2.out_proj.weight")
            _call("eend_gemm_bf16", dsd, D, W[f"d{i}.out2_wT"], D, None, dctx16[:Md], D, Md, D, D)
            _call("eend_spk_attn_bwd_bf16", sv["qkv"], dctx16[:Md], dqkv16[:Md], B, C, Tp, H, 0.125, dr(so + SITE_SPK))
            self._wgrad_bias(dqkv16[:Md], sv["s11"].out16, Md, 3 * D, D, p_ + "self_attn2.in_proj_weight", p_ + "self_attn2.in_proj_bias")
            # the in-projection's data gradient joins the stream; norm11's backward in the same launch (train.TrainStepBase._gemm_acc_ln_bwd)
            self._gemm_acc_ln_bwd(dqkv16[:Md], 3 * D, W[f"d{i}.in2_wT"], g32, sv["s11"], p_ + "norm11", dsd, Md, dr(so + SITE_OUT1),
                                  p_ + "self_attn1.out_proj.bias")
            below = None
            if i > 0:                     # the layer below ends in norm22: its backward rides on this block's last GEMM
                pb = f"dec.layers.{i - 1}."
                below = (bf.dec[i - 1]["s22"], pb + "norm22", dr(so - 16 + SITE_FFOUT), pb + "linear2.bias")
            self._ret_bwd(bf, g32, dsd, sv["ret"], x_in16, B * C, Md, f"d{i}", p_ + "self_attn1.", next_ln=below)

        # ---- convert fan-out (LS model :216-217, factored): g32 = gradient w.r.t. attr0
        _call("eend_convert_fanout_bwd_f32", g32, bf.gsum16, self.ws, WS_FLOATS, bf.dpc, B, Tp, C)
        _call("eend_convert_const_f32", 1, None, None, self.pe, None, bf.dpc, self._G("dec.convert.weight"), self._G("dec.convert.bias"), C)
        self._wgrad(bf.gsum16, bf.emb16, Me, D, D, "dec.convert.weight", ld_out=2 * D, k_out=D)
        _call("eend_gemm_acc_bf16", bf.gsum16, D, W["convert.w1T"], D, bf.de32, 1.0, bf.de32, None, Me, D)

        # ---- L2 norm + look-ahead conv (LS model :86-87)
        _call("eend_l2norm_bwd_bf16", bf.emb32, bf.de32, bf.inv_norm, bf.demb16, B, Tv, Tp)
        self._bias_grad(bf.demb16, Me, D, "cnn.bias")
        _call("eend_conv1d_wgrad_bf16", bf.demb16, bf.enc_out16, bf.il, B, Tp, D, self.ktaps, self.cpad, self.ws, WS_FLOATS, bf.conv_tmp,
              self._G("cnn.weight"))
        _call("eend_conv1d_dgrad_bf16", bf.demb16, W["cnn.wd"], bf.tl, bf.il, bf.ge32, B, Tp, D, self.ktaps, self.ktaps - 1 - self.cpad)

        # ---- Conformer blocks, last to first; g32 = gradient w.r.t. the block output
        g32 = bf.ge32
        dse = ds16[:Me]
        for i in reversed(range(len(bf.enc))):
            sv = bf.enc[i]
            s_ = f"enc.encoder.layers.{i}.sequential."
            so = 16 * i
            ffa, ffb, ret, cm = s_ + "0.module.sequential.", s_ + "3.module.sequential.", s_ + "1.module.", s_ + "2.module.sequential."
            # block-final LayerNorm over x + 0.5 FFN_b(LN_d x): g32 <- gradient w.r.t. that sum; dse <- 0.5 * dropout-masked copy
            self._ln_bwd2(g32, False, sv["lnE"], s_ + "4", g32, False, Me, ds16=dse, alpha16=0.5, drop=dr(so + ENC_FFB_OUT),
                          bias=ffb + "4.linear.bias")
            self._ffn_swish_bwd(bf, g32, dse, sv["zb"], sv["ab"], sv["lnD"], ffb + "0", Me, f"e{i}", "b", ffb, dr(so + ENC_FFB_HID))
            # conv module
            self._resgrad(g32, dse, 1.0, cm + "7.conv.bias", Me, dr(so + ENC_CONV))
            self._wgrad(dse, sv["s16"], Me, D, D, cm + "7.conv.weight")
            dsw = dctx16[:Me]
            _call("eend_gemm_bf16", dse, D, W[f"e{i}.pw2T"], D, None, dsw, D, Me, D, D)
            bnm = m.enc.encoder.layers[i].sequential[2].module.sequential[5]
            bn_args = (sv["c16"], sv["bn_mean"], sv["bn_var"], bnm.eps, self._P(cm + "5.weight"), self._P(cm + "5.bias"))
            _call("eend_bn_swish_bwd_stats_bf16", dsw, *bn_args, self.ws, WS_FLOATS, bf.bn_sums, self._G(cm + "5.weight"),
                  self._G(cm + "5.bias"), B, Tp, Tv)
            if self.sync_bn:                             # SyncBatchNorm backward: the two per-channel sums become global
                all_reduce_bn_sums(bf.bn_sums, self.group)
            _call("eend_bn_swish_bwd_apply_bf16", dsw, *bn_args, bf.bn_sums, sv["bn_n"], B, Tp, Tv)
            dP = bf.dh16[:Me * 2 * D].view(Me, 2 * D)
            _call("eend_dwconv_glu_bwd_bf16", dsw, sv["P"], self._P(cm + "4.conv.weight"), dP, self.ws, WS_FLOATS,
                  self._G(cm + "4.conv.weight"), B, Tp, Tv, self.kdw)
            self._wgrad_bias(dP, sv["lnC"].out16, Me, 2 * D, D, cm + "2.conv.weight", cm + "2.conv.bias")
            dy = bf.dy16[:Me]
            _call("eend_gemm_bf16", dP, 2 * D, W[f"e{i}.pw1T"], 2 * D, None, dy, D, Me, D, 2 * D)
            self._ln_bwd2(dy, True, sv["lnC"], cm + "0", g32, True, Me)
            # retention
            self._resgrad(g32, dse, 1.0, ret + "self_attn.out_proj.bias", Me, dr(so + ENC_RET))
            self._ret_bwd(bf, g32, dse, sv["ret"], sv["lnB"].out16, B, Me, f"e{i}", ret + "self_attn.", prenorm_site=sv["lnB"],
                          prenorm_ln=ret + "layer_norm")
            # FFN_a
            self._resgrad(g32, dse, 0.5, ffa + "4.linear.bias", Me, dr(so + ENC_FFA_OUT))
            self._ffn_swish_bwd(bf, g32, dse, sv["za"], sv["aa"], sv["lnA"], ffa + "0", Me, f"e{i}", "a", ffa, dr(so + ENC_FFA_HID))

        # ---- input projection + LayerNorm (conformer/encoder.py:195-196)
        self._ln_bwd(g32, bf.site0, "enc.encoder.layer_norm", dse, Me, None, "enc.encoder.input_projection.linear.bias")
        _call("eend_wgrad_bf16", dse, D, bf.xin16, self.Fin_pad, 1, Me, D, self.Fin_pad, self.ws, WS_FLOATS,
              self._G("enc.encoder.input_projection.linear.weight"), self.Fin, self.Fin, 1.0, 0)
