"""FS-EEND training step on MI355X: forward with saved activations, hand-written backward, Adam -- all HIP.

Reference behaviour reproduced (paths relative to the reference root):
  * FS-EEND/nnet/model/onl_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm.py:32-65   model.forward (train mode:
    BatchNorm1d batch statistics over the -1-padded input, :165-166)
  * FS-EEND/train/utils/loss.py:119-125   standard_loss (label_delay = 0)
  * FS-EEND/train_dia.py:83-100,153       Adam(betas (0.9, 0.98), eps 1e-9) x NoamScheduler, gradient_clip_val
  * torch autograd                         replaced by the backward kernels of csrc/{wgrad,attn_bwd,train_rows,
                                           embloss_bwd}.hip and the bf16 gradient GEMMs of gemm.hip

Design:
  * parameters, gradients and the Adam moments are four FLAT f32 buffers (`FlatState`); every nn.Parameter of the
    mirror model is a view into the parameter buffer, so checkpoints / state_dict keep working, the optimiser is one
    elementwise launch, and data parallelism is ONE all-reduce of the gradient buffer (RCCL over xGMI);
  * the 8 tensors the reference never back-propagates into keep an all-zero gradient slice (Adam's update for a
    zero gradient with zero moments is exactly zero -- the reference's Adam skips them because .grad is None);
  * MFMA operand copies of the weights (f16 forward layouts, bf16 transposed backward layouts) are rebuilt from the
    updated f32 parameters by one table-driven launch per step;
  * dropout (all ten nn.Dropout / attention-dropout sites of the two layer types) is applied inside the producing
    kernels' epilogues from a counter-based hash of (seed, element index) -- include/eend_hip.h `eend_dropout` -- and
    recomputed, not stored, by the backward.  The reference's masks come from torch's Philox stream and cannot be
    matched by any other implementation, so parity with dropout is pinned against the oracle driven with the SAME
    hash masks (tests/test_train_step.py), and the p = 0 path against the reference's own gradients.

There is no autograd / eager fallback: everything below is a call into libeend_hip.so.
"""
import ctypes
import math
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

from . import lib as _lib
from . import ops
from .lib import EendHipError

F16, BF16, F32, I32 = torch.float16, torch.bfloat16, torch.float32, torch.int32
D = 256
H = 4
WS_FLOATS = 32 * 1024 * 1024          # f32 scratch for split-sum partials (128 MB)
# (the FFN of a post-norm block -- linear1 + ReLU + dropout + linear2 + dropout + residual + LayerNorm -- is one launch forward and one for
#  the data gradients, ffn.hip MODE 3 / 4, wherever its shape allows: hidden width a multiple of 64, 32-bit row offsets; else GEMM launches)


def _call(name: str, *args):
    """One C-ABI call: tensors -> device pointers, current stream appended, return code checked."""
    L = _lib.load()
    a = []
    for x in args:
        if isinstance(x, Tensor):
            if not x.is_cuda:
                raise EendHipError(f"{name}: expected GPU tensors (the HIP path has no CPU fallback)")
            a.append(x.data_ptr())
        else:
            a.append(x)
    a.append(torch.cuda.current_stream().cuda_stream)
    _lib.check(getattr(L, name)(*a), name)


def _fmix32(h: int) -> int:
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def drop_step_seed(seed: int, fwd_count: int) -> int:
    """Per-forward base seed of the dropout hash."""
    return _fmix32((seed & 0xFFFFFFFF) ^ _fmix32(fwd_count + 0x9E3779B9))


def drop_site_seed(base: int, site: int) -> int:
    """Seed of one dropout site (`eend_dropout.seed`)."""
    return _fmix32(base ^ _fmix32(site * 0x9E3779B1 + 0x7F4A7C15))


def never_graded(name: str) -> bool:
    """Parameters of the FS model that are not on the forward path: the decoder's dead input projection
    and the fusion layers' norm12."""
    return name.startswith("dec.encoder.") or name.startswith("dec.encoder_norm.") or ".norm12." in name


class FlatState:
    """Flat f32 parameter / gradient / Adam-moment buffers; the model's parameters become views of `params`."""

    def __init__(self, model: torch.nn.Module):
        named = list(model.named_parameters())
        dev = named[0][1].device
        if dev.type != "cuda":
            raise EendHipError("training needs the model on the GPU: the HIP path has no CPU fallback")
        from .shard import flat_layout
        self.names = [n for n, _ in named]
        self.offsets, off = flat_layout([(n, tuple(p.shape)) for n, p in named])      # 16-byte aligned slices
        self.numel = off
        self.params = torch.zeros(off, dtype=F32, device=dev)
        self.grads = torch.zeros(off, dtype=F32, device=dev)
        self.m = torch.zeros(off, dtype=F32, device=dev)
        self.v = torch.zeros(off, dtype=F32, device=dev)
        with torch.no_grad():
            for n, p in named:
                o = self.offsets[n]
                view = self.params[o:o + p.numel()].view(p.shape)
                view.copy_(p.detach().to(F32))
                p.data = view
        self.shapes = {n: tuple(p.shape) for n, p in named}

    def p(self, name: str) -> Tensor:
        o = self.offsets[name]
        return self.params[o:o + math.prod(self.shapes[name])].view(self.shapes[name])

    def g(self, name: str) -> Tensor:
        o = self.offsets[name]
        return self.grads[o:o + math.prod(self.shapes[name])].view(self.shapes[name])


def noam_lr(opt_step: int, d_model: int, warmup: int, scale: float = 1.0, base_lr: float = 1.0) -> float:
    """Learning rate of optimiser step `opt_step` (1-based): utlis/scheduler.py:3-28 stepped once per optimiser step
    (oln_tfm_enc_dec.py:274); the scheduler's construction-time step makes step k run at last_epoch = k - 1 (>= 1)."""
    e = max(1, opt_step - 1)
    return base_lr * scale * d_model ** (-0.5) * min(e ** (-0.5), e * warmup ** (-1.5))


class _Site:
    """Saved tensors of one `linear + residual + LayerNorm` site: f16 output, normalised rows, 1/sigma."""

    def __init__(self, dev, M):
        self.out16 = torch.empty(M, D, dtype=F16, device=dev)
        self.xhat = torch.empty(M, D, dtype=F16, device=dev)
        self.rstd = torch.empty(M, dtype=F32, device=dev)


class _AttnSave:
    def __init__(self, dev, nseq, Tp):
        n = nseq * Tp * D
        self.q, self.k, self.v, self.vt = (torch.empty(n, dtype=BF16, device=dev) for _ in range(4))
        # [d][t] copies of Q, K: read only by the two-kernel backward of windows beyond 512 frames (attn_bwd.hip)
        self.qt, self.kt = ((torch.empty(n, dtype=BF16, device=dev) for _ in range(2)) if Tp > 512 else (None, None))
        self.lse = torch.empty(nseq * H * Tp, dtype=F32, device=dev)
        self.ctx = torch.empty(nseq * Tp, D, dtype=F16, device=dev)


class _Buffers:
    """Device buffers of one (B, Tp, C) training shape."""

    def __init__(self, dev, B, Tp, C, n_enc, n_dec, F_enc, F_dec, Fin, Fin_pad):
        e = lambda *s, dt: torch.empty(*s, dtype=dt, device=dev)
        Me, Md = B * Tp, B * C * Tp
        Mx = max(Me, Md)
        self.xin16 = torch.zeros(Me, Fin_pad, dtype=F16, device=dev)
        self.bn_mean, self.bn_var = e(Fin, dt=F32), e(Fin, dt=F32)
        self.h32 = e(Me, D, dt=F32)
        self.site0 = _Site(dev, Me)
        self.enc = [dict(att=_AttnSave(dev, B, Tp), s1=_Site(dev, Me), hid=e(Me, F_enc, dt=F16), s2=_Site(dev, Me))
                    for _ in range(n_enc)]
        self.emb32, self.emb16, self.inv_norm = e(Me, D, dt=F32), e(Me, D, dt=F16), e(Me, dt=F32)
        self.a32, self.a16 = e(Md, D, dt=F32), e(Md, D, dt=F16)
        self.dec = [dict(att=_AttnSave(dev, B * C, Tp), s11=_Site(dev, Md), qkv=e(Md, 3 * D, dt=F16), o2=e(Md, D, dt=F16),
                         s21=_Site(dev, Md), hid=e(Md, F_dec, dt=F16), s22=_Site(dev, Md)) for _ in range(n_dec)]
        # backward temporaries
        self.g32 = e(Md, D, dt=F32)
        self.ge32 = e(Me, D, dt=F32)
        self.de32 = e(Me, D, dt=F32)
        self.ds16 = e(Mx, D, dt=BF16)
        self.dctx16 = e(Mx, D, dt=BF16)
        self.dh16 = e(Mx * max(F_enc, F_dec), dt=BF16)
        self.dqkv16 = e(Mx, 3 * D, dt=BF16)
        self.dot_ws = e(Mx * D, dt=BF16)
        self.dh_ws = e(max(B, B * C) * H * Tp, dt=F32)
        self.gsum16 = e(Me, D, dt=BF16)
        self.demb16 = e(Me, D, dt=BF16)
        self.dy_in = e(Me, Fin_pad, dt=BF16)
        self.conv_tmp = e(D * 19 * D, dt=F32)
        self.dpc = e(C, D, dt=F32)
        self.pc = e(C, D, dt=F32)
        self.logits = e(B, Tp, C, dt=F32)
        self.loss = torch.zeros(4, dtype=F32, device=dev)         # [bce, emb, -, -]


class TrainStepBase:
    """What the FS-EEND and LS-EEND training steps share: the flat parameter / gradient / Adam buffers, the weight
    re-layout table, dropout site bookkeeping, the generic backward helpers (LayerNorm, ReLU FFN, bias / weight
    gradients), the data-parallel gradient exchange and the optimiser."""

    def _init_common(self, model, warmup, lr, schedule_scale, grad_clip, betas, eps, bn_momentum, process_group, drop_seed):
        ps = set()
        for name, mod in model.named_modules():
            if name.endswith("pos_enc.dropout") or name.endswith("pos_encoder.dropout"):
                continue                     # PositionalEncoding.forward returns the table un-dropped (model :218-224)
            if isinstance(mod, torch.nn.Dropout):
                ps.add(float(mod.p))
            if isinstance(mod, torch.nn.MultiheadAttention):
                ps.add(float(mod.dropout))
        if len(ps) > 1:
            raise NotImplementedError(f"one dropout ratio for the whole model (the reference's constructor), got {sorted(ps)}")
        self.drop_p = ps.pop() if ps else 0.0
        if not 0.0 <= self.drop_p < 1.0:
            raise ValueError("dropout ratio must be in [0, 1)")
        self.drop_seed = int(drop_seed) & 0xFFFFFFFF
        self._fwd_count = 0
        self.model = model
        # warmup None: no scheduler -- the optimiser runs at the constant configured lr (train_dia.py:95-100, scheduler = None)
        self.warmup, self.base_lr, self.sched_scale, self.clip = warmup, lr, schedule_scale, grad_clip
        self.b1, self.b2, self.eps, self.bn_momentum = betas[0], betas[1], eps, bn_momentum
        self.opt_step = 0
        self.group = process_group
        self.flat = FlatState(model)
        dev = self.flat.params.device
        self.dev = dev
        self.ws = torch.empty(WS_FLOATS, dtype=F32, device=dev)
        self.hp = torch.zeros(4, dtype=F32, device=dev)
        self.gsumsq = torch.zeros(1, dtype=F32, device=dev)
        # hyper-parameters travel through a RING of pinned host buffers: a slot is rewritten only after the async copy
        # that read it has completed (event), so queuing several steps ahead of the GPU never races the H2D copy
        self._hp_ring = [torch.zeros(4, dtype=F32).pin_memory() for _ in range(8)]
        self._hp_events = [None] * 8
        self._bufs = {}
        self._ptr_tables = {}
        self.last = {}

    # ------------------------------------------------------------------ weight re-layout table
    def _table_begin(self):
        self.W, self._entries = {}, []

    def _add_entry(self, dst: Tensor, dst_off_elems: int, pname: str, dims, strides, dtype, off=0, cpad=None, nscale=0, scale=1.0):
        """dst[a][b][c] (contiguous A x B x Cpad at dst + dst_off_elems) = convert(params[off(pname) + off + a*sa + b*sb + c*sc])."""
        A, B_, C_ = dims
        e = _lib.PrepEntry()
        e.src = self.flat.params.data_ptr()
        e.off = self.flat.offsets[pname] + off
        e.dst = dst.data_ptr() + dst_off_elems * dst.element_size()
        e.A, e.B, e.C, e.Cpad = A, B_, C_, C_ if cpad is None else cpad
        e.sa, e.sb, e.sc = strides
        e.dtype, e.nscale, e.scale, e.reserved = dtype, nscale, float(scale), 0
        self._entries.append(e)

    def _add(self, key, pname, dims, strides, dtype, off=0, cpad=None, nscale=0, scale=1.0, alloc=None):
        A, B_, C_ = dims
        cpad = C_ if cpad is None else cpad
        dt = {0: F16, 1: BF16, 2: F32}[dtype]
        shape = alloc if alloc is not None else (A * B_, cpad)
        t = torch.zeros(*shape, dtype=dt, device=self.dev)
        self.W[key] = t
        self._add_entry(t, 0, pname, dims, strides, dtype, off=off, cpad=cpad, nscale=nscale, scale=scale)
        return t

    def _plain(self, key, pname, N, K, dtype=0, kpad=None, nscale=0, scale=1.0):        # [N][K] row-major copy
        return self._add(key, pname, (N, 1, K), (K, 0, 1), dtype, cpad=kpad, nscale=nscale, scale=scale)

    def _transposed(self, key, pname, N, K, ld=None, rows_alloc=None):                 # bf16 [K][N]: dst[k][n] = W[n][k]
        ld = K if ld is None else ld
        return self._add(key, pname, (K, 1, N), (1, 0, ld), 1, alloc=(rows_alloc or K, N))

    def _vec(self, key, pname, n, nscale=0, scale=1.0):
        return self._add(key, pname, (n, 1, 1), (1, 0, 0), 2, nscale=nscale, scale=scale, alloc=(n,))

    def _table_end(self):
        arr = (_lib.PrepEntry * len(self._entries))(*self._entries)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self._table = raw.to(self.dev)
        self._n_entries = len(self._entries)
        # post-norm ReLU FFN blocks (keys X.w1 / X.w2 / X.w1T / X.w2T): packed weight streams of ffn_train_stream.hip, re-packed from the
        # operand copies after every parameter update -- (W1, W2) f16 for the forward, (W2^T, W1^T) bf16 for the data gradient
        self._ffn_streams = {}
        for key in list(self.W):
            if key.endswith(".w1") and all(key[:-3] + sfx in self.W for sfx in (".w2", ".w1T", ".w2T")):
                pre = key[:-3]
                Fh = self.W[key].shape[0]
                n = _lib.load().eend_ffn_train_stream_elems(Fh)
                if n > 0 and self.W[key].shape[1] == D:
                    self._ffn_streams[pre] = (torch.empty(n, dtype=F16, device=self.dev), torch.empty(n, dtype=BF16, device=self.dev), Fh)
        self._stream_of_w1 = {self.W[pre + ".w1"].data_ptr(): pre for pre in self._ffn_streams}
        # K = 256 input projections with a packed-stream form (proj_stream.hip): the retention q / k / v / g operand copies (X.wqkvg)
        self._proj_streams = {}
        for key in list(self.W):
            w = self.W[key]
            if key.endswith(".wqkvg") and w.dtype == F16 and w.dim() == 2 and w.shape[1] == D:
                n = _lib.load().eend_proj_stream_elems(int(w.shape[0]))
                if n > 0:
                    self._proj_streams[key] = torch.empty(n, dtype=F16, device=self.dev)
        # data gradients g += dY W of the K -> 256 projections with a large K (attention in-projections X.in_wT / X.in2_wT: K = 768, retention
        # projections X.wqkvgT: K = 1024): packed streams of gemm_acc_stream.hip from the transposed bf16 operand copies [256][K]
        self._gacc_streams = {}
        for key in list(self.W):
            w = self.W[key]
            if (key.endswith(".in_wT") or key.endswith(".in2_wT") or key.endswith(".wqkvgT")) and w.dtype == BF16 and w.dim() == 2 and w.shape[0] == D:
                n = _lib.load().eend_gemm_acc_stream_elems(int(w.shape[1]))
                if n > 0:
                    self._gacc_streams[key] = torch.empty(n, dtype=BF16, device=self.dev)
        self._gacc_of_w = {self.W[k].data_ptr(): k for k in self._gacc_streams}
        # time-axis attention in-projections (X.in_w / X.in1_w, f16 [768][256]): packed for the one-launch training forward (attn_stream.hip)
        self._attn_packed = {}
        for key in list(self.W):
            w = self.W[key]
            if (key.endswith(".in_w") or key.endswith(".in1_w")) and w.dtype == F16 and tuple(w.shape) == (3 * D, D):
                self._attn_packed[key] = torch.empty(_lib.load().eend_inproj_attn_packed_elems(), dtype=F16, device=self.dev)
        self._attn_packed_of_w = {self.W[k].data_ptr(): k for k in self._attn_packed}

    def prep_weights(self):
        """f32 parameters -> MFMA operand copies (one launch), then the packed FFN streams."""
        _call("eend_prep_weights", self._table, self._n_entries)
        for pre, (fw, bw, Fh) in self._ffn_streams.items():
            _call("eend_ffn_train_stream_pack", self.W[pre + ".w1"], self.W[pre + ".w2"], fw, Fh)
            _call("eend_ffn_train_stream_pack", self.W[pre + ".w2T"], self.W[pre + ".w1T"], bw, Fh)
        for key, buf in self._proj_streams.items():
            _call("eend_proj_stream_pack_f16", self.W[key], buf, int(self.W[key].shape[0]))
        for key, buf in self._attn_packed.items():
            _call("eend_inproj_attn_pack_f16", self.W[key], buf)
        for key, buf in self._gacc_streams.items():
            w = self.W[key]
            _call("eend_gemm_acc_stream_pack_bf16", w, w.stride(0), buf, int(w.shape[1]))

    def _ffn_stream_for(self, w1, M, F):
        """Key of the packed streams serving this FFN at M rows, or None (the un-packed, row-major entries then)."""
        pre = self._stream_of_w1.get(w1.data_ptr())
        if pre is None or M % 16 != 0 or not _lib.load().eend_ffn_train_stream_ok(M, F, D):
            return None
        return pre

    # ------------------------------------------------------------------ helpers
    def _table_for(self, srcs, T):
        key = tuple((s.data_ptr(), s.shape[0]) for s in srcs)
        tab = self._ptr_tables.get(key)
        if tab is None:
            if len(self._ptr_tables) > 64:
                self._ptr_tables.clear()
            tab = (torch.tensor([k[0] for k in key], dtype=torch.int64, device=self.dev),
                   torch.tensor([min(k[1], T) for k in key], dtype=I32, device=self.dev))
            self._ptr_tables[key] = tab
        return tab

    def _P(self, name):          # f32 parameter view
        return self.flat.p(name)

    def _G(self, name):          # f32 gradient view
        return self.flat.g(name)

    def _drop(self, bf, site: int):
        """eend_dropout of one site for the forward that filled `bf` (None <=> no dropout): the seed is a hash of
        (drop_seed, forward count, site id), so the backward asks for the same spec and gets the same mask."""
        if bf.drop_base is None:
            return None
        spec = bf.drop_specs.get(site)
        if spec is None:
            spec = _lib.Dropout(drop_site_seed(bf.drop_base, site), int(round(self.drop_p * (1 << 24))), 1.0 / (1.0 - self.drop_p))
            bf.drop_specs[site] = spec
        return ctypes.byref(spec)

    def _linear_ln(self, a16, w, bias, res, ln, site, out32, M, K, drop=None, alpha=1.0):
        _call("eend_linear_res_ln_train_f16", a16, a16.stride(0), w, w.stride(0), bias, res, alpha, self._P(ln + ".weight"),
              self._P(ln + ".bias"), 1e-5, out32, site.out16, site.xhat, site.rstd, M, K, drop)

    def _ffn(self, x16, w1, b1, hid, w2, b2, res, ln, site, out32, M, drop_hidden=None, drop_out=None):
        """linear1 + ReLU + dropout + linear2 + dropout + residual + LayerNorm of a post-norm block.  One launch (ffn.hip MODE 3: the hidden
        activations are written once for the backward and never re-read) where the shape allows, else the two GEMM launches."""
        F = hid.shape[1]
        pre = self._ffn_stream_for(w1, M, F) if x16.shape[1] == 256 and x16.stride(0) == 256 else None
        if pre is not None:
            # packed weight stream (ffn_train_stream.hip): hid is written in the BLOCKED layout its two consumers read (_ffn_bwd)
            _call("eend_ffn_train_stream_f16", x16, 256, self._ffn_streams[pre][0], b1, b2, res, 1.0, self._P(ln + ".weight"), self._P(ln + ".bias"),
                  1e-5, out32, site.out16, hid, site.xhat, site.rstd, M, F, drop_hidden, drop_out)
            return
        if F % 64 == 0 and (M + 128) * F * 2 < (1 << 32) and x16.shape[1] == 256:
            _call("eend_ffn_train_f16", x16, x16.stride(0), w1, b1, w2, b2, res, 1.0, self._P(ln + ".weight"), self._P(ln + ".bias"), 1e-5,
                  out32, site.out16, hid, site.xhat, site.rstd, M, F, drop_hidden, drop_out)
            return
        self._linear_relu(x16, w1, b1, hid, drop_hidden)
        self._linear_ln(hid, w2, b2, res, ln, site, out32, M, F, drop_out)

    def _linear_relu(self, a16, w, bias, out16, drop=None):
        M, K = a16.shape
        N = out16.shape[1]
        _call("eend_linear_relu_train_f16", a16, a16.stride(0), w, w.stride(0), bias, out16, out16.stride(0), M, N, K, drop)

    def _bias_grad(self, dy16, M, N, gname, scale=1.0):
        _call("eend_colsum_f32", dy16, dy16.stride(0), M, N, 1, self.ws, WS_FLOATS, self._G(gname), scale, 0)

    def _wgrad(self, dy16, x, M, N, K, gname, x_is_f16=True, ld_out=None, k_out=None, goff=0, scale=1.0):
        g = self._G(gname).view(-1)[goff:]
        _call("eend_wgrad_bf16", dy16, dy16.stride(0), x, x.stride(0), 1 if x_is_f16 else 0, M, N, K, self.ws, WS_FLOATS, g,
              K if ld_out is None else ld_out, K if k_out is None else k_out, scale, 0)

    def _wgrad_bias(self, dy16, x, M, N, K, wname, bname, x_is_f16=True):
        """Weight gradient AND bias gradient of one linear layer from one pass over dy16 (eend_wgrad_bias_bf16)."""
        _call("eend_wgrad_bias_bf16", dy16, dy16.stride(0), x, x.stride(0), 1 if x_is_f16 else 0, M, N, K, self.ws, WS_FLOATS,
              self._G(wname), K, K, self._G(bname), 1.0, 0)

    def _ln_bwd(self, g32, site, ln, ds16, M, drop=None, bias=None):
        """`drop`: the spec of the sub-layer output dropout in front of this LayerNorm's residual sum -- the bf16 branch
        gradient ds16 gets the mask, the f32 residual-stream gradient g32 does not.  `bias`: the bias parameter of the
        linear layer in front of the LayerNorm; its gradient (the column sums of ds16) comes out of the same pass."""
        _call("eend_layernorm_bwd_f32", g32, site.xhat, site.rstd, self._P(ln + ".weight"), g32, ds16, self.ws, WS_FLOATS,
              self._G(ln + ".weight"), self._G(ln + ".bias"), None if bias is None else self._G(bias), M, drop)

    def _gemm_acc(self, a16, K, wT, g32, M):
        """g32 += a16 wT^T in place: the packed-stream kernel where it pays (see _gemm_acc_ln_bwd), else gemm.hip's tiles."""
        key = self._gacc_of_w.get(wT.data_ptr())
        if key is not None and M >= self.gacc_stream_min_rows and _lib.load().eend_gemm_acc_stream_ok(M, K, K):
            _call("eend_gemm_acc_stream_bf16", a16, K, self._gacc_streams[key], g32, M, K)
        else:
            _call("eend_gemm_acc_bf16", a16, K, wT, K, g32, 1.0, g32, None, M, K)

    attn_train_fused = True             # (tests compare against the two-launch forward by clearing it)
    gacc_stream_min_rows = 98304        # rows from which gemm_acc_stream.hip + eend_layernorm_bwd_f32 beat the fused tiled launch (tests set 0)

    def _gemm_acc_ln_bwd(self, a16, K, wT, g32, site, ln, ds16, M, drop=None, bias=None):
        """g32 += a16 wT^T (the data gradient of a K -> 256 linear joining the residual-gradient stream), then the LayerNorm backward of the
        post-norm site `site` that stream now stands in front of -- one launch (gemm.hip EPI_RES_LNBWD): `_ln_bwd`'s outputs, the f32 stream
        read and written once instead of twice."""
        key = self._gacc_of_w.get(wT.data_ptr())
        if key is not None and M >= self.gacc_stream_min_rows and _lib.load().eend_gemm_acc_stream_ok(M, K, K):
            # from about 100 k rows the packed-stream GEMM (wave-owned rows, no k-tile barriers) + the LayerNorm backward as its own pass beat
            # the tiled GEMM with the LayerNorm backward in its epilogue: [196608, 256, 768] 172 + 95 us against 314 us, [393216, 256, 1024]
            # 363 + 192 against 665 us (same box)
            _call("eend_gemm_acc_stream_bf16", a16, K, self._gacc_streams[key], g32, M, K)
            self._ln_bwd(g32, site, ln, ds16, M, drop, bias)
            return
        _call("eend_gemm_acc_lnbwd_bf16", a16, K, wT, K, g32, site.xhat, site.rstd, self._P(ln + ".weight"), g32, ds16, self.ws, WS_FLOATS,
              self._G(ln + ".weight"), self._G(ln + ".bias"), None if bias is None else self._G(bias), M, K, drop)

    def _ffn_bwd(self, g32, ds16, dh16, hid, x_in16, M, wkey, p_, norm, drop_scale=1.0):
        """backward of x -> LN(x + W2 relu(W1 x + b1) + b2); g32 in/out (gradient w.r.t. output -> w.r.t. x)."""
        W = self.W
        Fh = hid.shape[1]
        dh = dh16[:M * Fh].view(M, Fh)
        pre = self._ffn_stream_for(W[wkey + ".w1"], M, Fh) if x_in16.stride(0) == 256 else None      # the same predicate as the forward's
        if pre is not None:
            # hid and dH in the blocked layout: flags 2 (X blocked) / 4 (dY blocked) of the weight-gradient entries
            g2 = self._G(p_ + "linear2.weight").view(-1)
            _call("eend_wgrad_bf16", ds16, D, hid, Fh, 1 | 2, M, D, Fh, self.ws, WS_FLOATS, g2, Fh, Fh, 1.0, 0)
            _call("eend_ffn_bwd_data_stream_bf16", ds16, D, self._ffn_streams[pre][1], hid, drop_scale, dh, g32, M, Fh)
            _call("eend_wgrad_bias_bf16", dh, Fh, x_in16, D, 1 | 4, M, Fh, D, self.ws, WS_FLOATS, self._G(p_ + "linear1.weight"), D, D,
                  self._G(p_ + "linear1.bias"), 1.0, 0)
            return
        self._wgrad(ds16, hid, M, D, Fh, p_ + "linear2.weight")
        if Fh % 64 == 0 and (M + 128) * Fh * 2 < (1 << 32):
            # dH = scale * (dY W2) under the saved mask, and g += dH W1, in one launch (ffn.hip MODE 4): dH is written once for the weight
            # gradient below and not re-read by the data path
            _call("eend_ffn_bwd_data_bf16", ds16, D, W[wkey + ".w2T"], hid, W[wkey + ".w1T"], drop_scale, dh, g32, M, Fh)
            self._wgrad_bias(dh, x_in16, M, Fh, D, p_ + "linear1.weight", p_ + "linear1.bias")
            return
        _call("eend_gemm_relu_bwd_bf16", ds16, D, W[wkey + ".w2T"], D, hid, Fh, dh, Fh, M, Fh, D, drop_scale)
        self._wgrad_bias(dh, x_in16, M, Fh, D, p_ + "linear1.weight", p_ + "linear1.bias")
        _call("eend_gemm_acc_bf16", dh, Fh, W[wkey + ".w1T"], Fh, g32, 1.0, g32, None, M, Fh)

    def _pit_labels(self, bf, lab, il, ncols):
        """train/oln_tfm_enc_dec_spk_pit.py:78-87: re-order the speaker columns (1..ncols-2) of the labels by
        batch_pit_n_speaker_loss on the same columns of the logits (device PIT kernels, pit.py)."""
        from . import pit as P
        B, T, Tp, C = bf.shape
        logits = torch.empty(B, T, C, dtype=F32, device=self.dev)
        attr = torch.empty(B, T, C, D, dtype=F32, device=self.dev)
        ops.head_l2dot(bf.emb32, bf.a32, attr, logits, B, T, Tp, C, D)
        n_spk = [n - 2 for n in ncols]
        S = max(n_spk)
        ys = [torch.nn.functional.pad(logits[b, :il[b], 1:1 + n_spk[b]], (0, S - n_spk[b])) for b in range(B)]
        ts = [torch.nn.functional.pad(lab[b, :il[b], 1:1 + n_spk[b]], (0, S - n_spk[b])) for b in range(B)]
        perm = self._pit_assign(P, ys, ts, n_spk)
        out = lab.clone()
        for b in range(B):
            out[b, :il[b], 1:1 + n_spk[b]] = perm[b]
        return out

    def _pit_assign(self, P, ys, ts, n_spk):
        return P.batch_pit_n_speaker_loss(ys, ts, n_spk)[1]

    # ------------------------------------------------------------------ optimiser
    def accumulate_grads(self, index: int, count: int):
        """Micro-batch `index` of `count` (accumulate_grad_batches, train_dia.py:151): the backward kernels overwrite
        flat.grads, so the running mean over micro-batches lives in a second flat buffer; after the last one it is
        moved back and the usual all-reduce / optimiser step follow."""
        if count <= 1:
            return
        if getattr(self, "_gacc", None) is None:
            self._gacc = torch.zeros_like(self.flat.grads)
        _call("eend_grad_accumulate_f32", self._gacc, self.flat.grads, 1.0 / count, 1 if index == 0 else 0, self.flat.numel)
        if index == count - 1:
            _call("eend_grad_accumulate_f32", self.flat.grads, self._gacc, 1.0, 1, self.flat.numel)

    def all_reduce_grads(self):
        """Data parallelism: ONE all-reduce (mean) of the flat gradient buffer over RCCL (torch.distributed 'nccl')."""
        from .shard import all_reduce_mean
        all_reduce_mean(self.flat.grads, self.group)

    def current_lr(self, opt_step: int) -> float:
        if self.warmup is None:
            return float(self.base_lr)
        return noam_lr(opt_step, D, self.warmup, self.sched_scale, self.base_lr)

    def optimizer_step(self):
        self.opt_step += 1
        lr = self.current_lr(self.opt_step)
        t = self.opt_step
        slot = t % len(self._hp_ring)
        if self._hp_events[slot] is not None:
            self._hp_events[slot].synchronize()         # the copy that last read this slot has completed
        host = self._hp_ring[slot]
        host[0], host[1], host[2], host[3] = lr, 1 - self.b1 ** t, 1 - self.b2 ** t, self.clip
        self.hp.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hp_events[slot] = ev
        fl = self.flat
        _call("eend_grad_sumsq_f32", fl.grads, fl.numel, self.ws, WS_FLOATS, self.gsumsq)
        _call("eend_adam_step_f32", fl.params, fl.grads, fl.m, fl.v, fl.numel, self.hp, self.gsumsq, self.b1, self.b2, self.eps)
        self.last_lr = lr
        self.prep_weights()
        self.model._prep = None            # the inference-path operand cache is stale now
        return lr

    def step(self, src, labels, ilens, pit: bool = False):
        """forward + backward + (all-reduce) + Adam.  Returns dict(loss, bce, emb, lr, gradnorm) of device scalars /
        floats; nothing here synchronises with the host."""
        if self.opt_step == 0 and not getattr(self, "_prepped", False):
            self.prep_weights()
            self._prepped = True
        bf = self.forward(src, labels, ilens, pit=pit)
        self.backward(bf)
        self.all_reduce_grads()
        lr = self.optimizer_step()
        return dict(bce=bf.loss[0], emb=bf.loss[1], loss=bf.loss[0] + bf.loss[1], lr=lr, gradnorm=self.gsumsq.sqrt())

    # ------------------------------------------------------------------ checkpoint state (optimiser side)
    def optimizer_state(self) -> dict:
        """Everything beyond the model's state_dict that a resumed run needs: Adam moments, step counters."""
        return dict(m=self.flat.m.detach().clone(), v=self.flat.v.detach().clone(), opt_step=self.opt_step,
                    fwd_count=self._fwd_count, names=list(self.flat.names), numel=self.flat.numel)

    def load_optimizer_state(self, st: dict):
        if list(st["names"]) != list(self.flat.names) or int(st["numel"]) != self.flat.numel:
            raise EendHipError("optimizer state does not match this model's flat layout")
        self.flat.m.copy_(st["m"].to(self.dev))
        self.flat.v.copy_(st["v"].to(self.dev))
        self.opt_step, self._fwd_count = int(st["opt_step"]), int(st["fwd_count"])


class FsTrainStep(TrainStepBase):
    """One training step of FS-EEND (`OnlineTransformerDADiarization` mirror) entirely in HIP."""

    def __init__(self, model, warmup: Optional[int] = 100000, lr: float = 1.0, schedule_scale: float = 1.0, grad_clip: float = 5.0,
                 betas=(0.9, 0.98), eps: float = 1e-9, bn_momentum: float = 0.1, process_group=None, drop_seed: int = 0):
        from .fs_model import OnlineTransformerDADiarization
        if not isinstance(model, OnlineTransformerDADiarization):
            raise TypeError("FsTrainStep drives fs_eend_amd.fs_model.OnlineTransformerDADiarization")
        if model.enc.mask_delay != model.dec.mask_delay:
            raise NotImplementedError
        self._init_common(model, warmup, lr, schedule_scale, grad_clip, betas, eps, bn_momentum, process_group, drop_seed)
        self._build_weight_table()

    # ------------------------------------------------------------------ weight operand copies
    def _build_weight_table(self):
        m, fl, dev = self.model, self.flat, self.dev
        self._table_begin()
        W = self.W
        add, plain, transposed, vec = self._add, self._plain, self._transposed, self._vec

        Fin = m.enc.in_size
        self.Fin, self.Fin_pad = Fin, (Fin + 127) // 128 * 128
        qs = ops.QSCALE_LOG2
        plain("enc.in.w", "enc.encoder.weight", D, Fin, kpad=self.Fin_pad)
        transposed("enc.in.wT", "enc.encoder.weight", D, Fin, rows_alloc=self.Fin_pad)
        for i, l in enumerate(m.enc.transformer_encoder.layers):
            p_ = f"enc.transformer_encoder.layers.{i}."
            Fh = l.linear1.out_features
            plain(f"e{i}.in_w", p_ + "self_attn.in_proj_weight", 3 * D, D, nscale=D, scale=qs)
            vec(f"e{i}.in_b", p_ + "self_attn.in_proj_bias", 3 * D, nscale=D, scale=qs)
            transposed(f"e{i}.in_wT", p_ + "self_attn.in_proj_weight", 3 * D, D)
            plain(f"e{i}.out_w", p_ + "self_attn.out_proj.weight", D, D)
            transposed(f"e{i}.out_wT", p_ + "self_attn.out_proj.weight", D, D)
            plain(f"e{i}.w1", p_ + "linear1.weight", Fh, D)
            transposed(f"e{i}.w1T", p_ + "linear1.weight", Fh, D)
            plain(f"e{i}.w2", p_ + "linear2.weight", D, Fh)
            transposed(f"e{i}.w2T", p_ + "linear2.weight", D, Fh)
        k = m.cnn.kernel_size[0]
        self.ktaps, self.cpad = k, m.cnn.padding[0]
        # forward: [co][tap*256 + ci]; data gradient: [ci][tap'*256 + co] = W[co][ci][k-1-tap']
        add("cnn.w", "cnn.weight", (D, k, D), (D * k, 1, k), 0, alloc=(D, k * D))
        add("cnn.wd", "cnn.weight", (D, k, D), (k, -1, D * k), 1, off=k - 1, alloc=(D, k * D))
        add("convert.w1", "dec.convert.weight", (D, 1, D), (2 * D, 0, 1), 0)
        add("convert.w1T", "dec.convert.weight", (D, 1, D), (1, 0, 2 * D), 1)
        for i, l in enumerate(m.dec.attractor_decoder.layers):
            p_ = f"dec.attractor_decoder.layers.{i}."
            Fh = l.linear1.out_features
            plain(f"d{i}.in1_w", p_ + "self_attn1.in_proj_weight", 3 * D, D, nscale=D, scale=qs)
            vec(f"d{i}.in1_b", p_ + "self_attn1.in_proj_bias", 3 * D, nscale=D, scale=qs)
            transposed(f"d{i}.in1_wT", p_ + "self_attn1.in_proj_weight", 3 * D, D)
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
        self.pe = m.dec.pos_enc.pe[0].to(device=dev, dtype=F32).contiguous()       # (5000, 256) sinusoid rows


    def _buffers(self, B, Tp, C) -> _Buffers:
        key = (B, Tp, C)
        b = self._bufs.get(key)
        if b is None:
            if len(self._bufs) >= 2:
                self._bufs.clear()
            m = self.model
            Fe = m.enc.transformer_encoder.layers[0].linear1.out_features if len(m.enc.transformer_encoder.layers) else D
            Fd = m.dec.attractor_decoder.layers[0].linear1.out_features if len(m.dec.attractor_decoder.layers) else D
            b = _Buffers(self.dev, B, Tp, C, len(m.enc.transformer_encoder.layers), len(m.dec.attractor_decoder.layers),
                         Fe, Fd, self.Fin, self.Fin_pad)
            self._bufs[key] = b
        return b


    # ------------------------------------------------------------------ dropout sites
    # site ids: encoder layer i -> 16*i + k, decoder layer i -> 4096 + 16*i + k, with k:
    SITE_ATT, SITE_OUT1, SITE_SPK, SITE_OUT2, SITE_FF, SITE_FFOUT = 0, 1, 2, 3, 4, 5

    def _attn_fwd(self, x16, w, bias, sv: _AttnSave, nseq, Tp, mask_delay, kv_len, drop=None):
        key = self._attn_packed_of_w.get(w.data_ptr())
        if key is not None and Tp <= 512 and H == 4 and self.attn_train_fused:
            # in-projection + attention in one launch, Q / K / V^T on chip in between; bf16 head rows + lse saved for the backward
            _call("eend_inproj_attn_train_bf16", x16, x16.stride(0), self._attn_packed[key], bias, sv.ctx, D, sv.q, sv.k, sv.v, sv.lse, nseq, H, Tp,
                  mask_delay, kv_len, drop)
            return
        _call("eend_inproj_heads_train_bf16", x16, x16.stride(0), w, bias, sv.q, sv.qt, sv.k, sv.kt, sv.v, sv.vt, nseq, Tp, H)
        _call("eend_attn_causal_lse_bf16", sv.q, sv.k, sv.vt, sv.ctx, sv.lse, nseq, H, Tp, D, mask_delay, kv_len, ops.LN2, drop)


    # ------------------------------------------------------------------ forward (saves activations)
    def forward(self, src: Sequence[Tensor], labels: Sequence[Tensor], ilens: Sequence[int], pit: bool = False,
                fused_loss: bool = True, dropout: bool = True):
        """Train-mode model.forward with every activation the backward needs saved.  labels: prepared (T_i, nspk_i+2)
        tensors (oln_tfm_enc_dec.py:53-75).
        fused_loss=True : also standard_loss + emb-consistency loss and their gradients w.r.t. attractors / embeddings
                          (bf.loss = [bce, emb_loss]); `backward(bf)` then completes the step.
        fused_loss=False: stop at the head (bf.logits_full (B,T,C), bf.attr_n (B,T,C,D), bf.loss[1] = emb_loss): the
                          caller computes its own loss on the logits and passes d loss / d logits to `backward`."""
        m, W, dev = self.model, self.W, self.dev
        srcs = [s.to(device=dev, dtype=F32).contiguous() for s in src]
        B, T = len(srcs), max(int(s.shape[0]) for s in srcs)
        ncols = [int(l.shape[1]) for l in labels]
        C = max(ncols)
        Tp = ops.frames_pad(T)
        bf = self._buffers(B, Tp, C)
        Me, Md = B * Tp, B * C * Tp
        il = [min(int(l), T) for l in ilens]
        key = (tuple(il), tuple(ncols), T)
        if getattr(bf, "len_key", None) != key:              # cached: no H2D copies in the steady state
            bf.il = torch.tensor(il, dtype=I32, device=dev)
            bf.tl = torch.full((B,), T, dtype=I32, device=dev)
            bf.nc = torch.tensor(ncols, dtype=I32, device=dev)
            bf.len_key = key
        bf.shape = (B, T, Tp, C)
        bf.srcs = srcs
        self._fwd_count += 1
        bf.drop_base = drop_step_seed(self.drop_seed, self._fwd_count) if (dropout and self.drop_p > 0.0) else None
        bf.drop_specs = {}
        dr = lambda site: self._drop(bf, site)
        n_frames = sum(int(l.shape[0]) for l in labels)
        if all(tuple(l.shape) == (T, C) for l in labels):    # equal-length chunks (the training set-up): one stack
            lab = torch.stack([l.to(device=dev, dtype=F32) for l in labels]).contiguous()
        else:
            lab = torch.zeros(B, T, C, dtype=F32, device=dev)
            for b_, l in enumerate(labels):
                lab[b_, :l.shape[0], :l.shape[1]] = l.to(device=dev, dtype=F32)
        bf.labels = lab
        enc = m.enc
        delay_e = enc.mask_delay if enc.has_mask else Tp
        kv_e = T
        bf.delay_e, bf.kv_e = delay_e, kv_e

        # ---- BatchNorm1d, train mode: batch statistics over all B*T padded frames, running-stat update (model :165-166)
        ptrs, lens = self._table_for(srcs, T)
        bf.ptrs, bf.lens = ptrs, lens
        _call("eend_bn_train_stats_f32", ptrs, lens, -1.0, self.ws, WS_FLOATS, bf.bn_mean, bf.bn_var, enc.bn.running_mean,
              enc.bn.running_var, self.bn_momentum, B, T, self.Fin)
        enc.bn.num_batches_tracked += 1
        _call("eend_gather_bn_cast_pad_f16", ptrs, lens, -1.0, self._P("enc.bn.weight"), self._P("enc.bn.bias"), bf.bn_mean,
              bf.bn_var, enc.bn.eps, bf.xin16, B, T, Tp, self.Fin, self.Fin_pad, 1)
        self._linear_ln(bf.xin16, W["enc.in.w"], self._P("enc.encoder.bias"), None, "enc.encoder_norm", bf.site0, bf.h32, Me,
                        self.Fin_pad)
        x16 = bf.site0.out16
        for i, sv in enumerate(bf.enc):
            p_ = f"enc.transformer_encoder.layers.{i}."
            so = 16 * i
            self._attn_fwd(x16, W[f"e{i}.in_w"], W[f"e{i}.in_b"], sv["att"], B, Tp, delay_e, kv_e, dr(so + self.SITE_ATT))
            self._linear_ln(sv["att"].ctx, W[f"e{i}.out_w"], self._P(p_ + "self_attn.out_proj.bias"), bf.h32, p_ + "norm1", sv["s1"],
                            bf.h32, Me, D, dr(so + self.SITE_OUT1))
            self._ffn(sv["s1"].out16, W[f"e{i}.w1"], self._P(p_ + "linear1.bias"), sv["hid"], W[f"e{i}.w2"], self._P(p_ + "linear2.bias"),
                      bf.h32, p_ + "norm2", sv["s2"], bf.h32, Me, dr(so + self.SITE_FF), dr(so + self.SITE_FFOUT))
            x16 = sv["s2"].out16
        bf.enc_out16 = x16

        # ---- truncate / look-ahead conv / L2 norm (model :38-41)
        _call("eend_conv1d_l2norm_train_f16", x16, W["cnn.w"], self._P("cnn.bias"), bf.il, bf.emb32, bf.emb16, bf.inv_norm, B, Tp, D,
              self.ktaps, self.cpad)

        # ---- attractor decoder (model :112-118)
        _call("eend_convert_const_f32", 0, self._P("dec.convert.weight"), self._P("dec.convert.bias"), self.pe, bf.pc, None, None,
              None, C)
        ops.convert_fanout(bf.emb16, W["convert.w1"], bf.pc, bf.a32, bf.a16, B, Tp, C)
        x16 = bf.a16
        dm = m.dec.mask_delay
        for i, sv in enumerate(bf.dec):
            p_ = f"dec.attractor_decoder.layers.{i}."
            so = 4096 + 16 * i
            self._attn_fwd(x16, W[f"d{i}.in1_w"], W[f"d{i}.in1_b"], sv["att"], B * C, Tp, dm, T, dr(so + self.SITE_ATT))
            self._linear_ln(sv["att"].ctx, W[f"d{i}.out1_w"], self._P(p_ + "self_attn1.out_proj.bias"), bf.a32, p_ + "norm11",
                            sv["s11"], bf.a32, Md, D, dr(so + self.SITE_OUT1))
            ops.linear(sv["s11"].out16, W[f"d{i}.in2_w"], self._P(p_ + "self_attn2.in_proj_bias"), sv["qkv"])
            _call("eend_spk_attn_train_f16", sv["qkv"], sv["o2"], B, C, Tp, H, 0.125, dr(so + self.SITE_SPK))
            self._linear_ln(sv["o2"], W[f"d{i}.out2_w"], self._P(p_ + "self_attn2.out_proj.bias"), bf.a32, p_ + "norm21", sv["s21"],
                            bf.a32, Md, D, dr(so + self.SITE_OUT2))
            self._ffn(sv["s21"].out16, W[f"d{i}.w1"], self._P(p_ + "linear1.bias"), sv["hid"], W[f"d{i}.w2"], self._P(p_ + "linear2.bias"),
                      bf.a32, p_ + "norm22", sv["s22"], bf.a32, Md, dr(so + self.SITE_FF), dr(so + self.SITE_FFOUT))
            x16 = sv["s22"].out16

        # ---- head + BCE (+ PIT label choice) + emb-consistency loss, and their gradients w.r.t. attractors / embeddings
        if not fused_loss:
            bf.logits_full = torch.empty(B, T, C, dtype=F32, device=dev)
            bf.attr_n = torch.empty(B, T, C, D, dtype=F32, device=dev)
            ops.head_l2dot(bf.emb32, bf.a32, bf.attr_n, bf.logits_full, B, T, Tp, C, D)
            bf.loss[1] = ops.emb_consistency(bf.emb32.view(B, Tp, D), lab, T)
            return bf
        if pit:
            lab = self._pit_labels(bf, lab, il, ncols)
            bf.labels = lab
        _call("eend_head_bce_f32", bf.emb32, bf.a32, lab, bf.il, bf.nc, 1.0 / float(n_frames), None, bf.logits, bf.g32, bf.de32,
              self.ws, WS_FLOATS, bf.loss[0:1], B, T, Tp, C)
        emb_loss = ops.emb_consistency(bf.emb32.view(B, Tp, D), lab, T)
        bf.loss[1] = emb_loss
        _call("eend_emb_consistency_bwd_f16", bf.emb16, lab, None, 0.0, bf.de32, B, T, Tp, D, C)
        return bf

    # ------------------------------------------------------------------ backward
    def _attn_bwd(self, g32, ds16, dctx16, dqkv16, sv: _AttnSave, x_in16, nseq, Tp, M, w_outT, w_inT, p_out, p_in, bf, delay,
                  kv_len, T, drop=None, next_ln=None):
        """backward of x -> x + out_proj(causal_mha(in_proj x)) given ds16 = gradient w.r.t. that sum (bf16) and
        g32 = the same in f32 (residual path); g32 += gradient through the attention branch."""
        self._wgrad(ds16, sv.ctx, M, D, D, p_out + ".weight")
        _call("eend_gemm_bf16", ds16, D, w_outT, D, None, dctx16, D, M, D, D)
        _call("eend_attn_causal_bwd_bf16", sv.q, sv.qt, sv.k, sv.kt, sv.v, dctx16, D, sv.ctx, D, sv.lse, bf.dot_ws, bf.dh_ws, dqkv16,
              3 * D, nseq, H, Tp, delay, kv_len, T, 1.0, 0.125, ops.LN2, drop)
        self._wgrad_bias(dqkv16, x_in16, M, 3 * D, D, p_in + "_weight", p_in + "_bias")
        if next_ln is not None:           # (site, LayerNorm name, dropout spec, bias name) of the LayerNorm whose output this block's input is
            site, ln, ndrop, nbias = next_ln
            self._gemm_acc_ln_bwd(dqkv16, 3 * D, w_inT, g32, site, ln, ds16, M, ndrop, nbias)
        else:
            self._gemm_acc(dqkv16, 3 * D, w_inT, g32, M)

    def backward(self, bf: _Buffers, dlogits: Optional[Tensor] = None, emb_loss_grad: float = 1.0):
        """Gradients w.r.t. every parameter -> self.flat.grads.  After forward(fused_loss=True): of bce + emb_loss.
        After forward(fused_loss=False): of the caller's loss, given dlogits = d loss / d logits (B, T, C) f32 and
        emb_loss_grad = d loss / d emb_loss."""
        W = self.W
        B, T, Tp, C = bf.shape
        if dlogits is not None:
            dl = dlogits.to(device=self.dev, dtype=F32).contiguous()
            if tuple(dl.shape) != (B, T, C):
                raise EendHipError(f"backward: dlogits must be ({B}, {T}, {C})")
            _call("eend_head_bce_f32", bf.emb32, bf.a32, None, None, None, 0.0, dl, None, bf.g32, bf.de32, self.ws, WS_FLOATS,
                  bf.loss[2:3], B, T, Tp, C)
            if emb_loss_grad != 0.0:
                _call("eend_emb_consistency_bwd_f16", bf.emb16, bf.labels, None, float(emb_loss_grad) / (float(B) * T * T), bf.de32,
                      B, T, Tp, D, C)
        Me, Md = B * Tp, B * C * Tp
        m = self.model
        dm = m.dec.mask_delay
        ds16, dctx16, dqkv16 = bf.ds16, bf.dctx16, bf.dqkv16
        dr = lambda site: self._drop(bf, site)
        ff_scale = 1.0 / (1.0 - self.drop_p) if bf.drop_base is not None else 1.0

        # ---- decoder layers, last to first; bf.g32 = gradient w.r.t. the layer output
        g32 = bf.g32
        for i in reversed(range(len(bf.dec))):
            sv = bf.dec[i]
            p_ = f"dec.attractor_decoder.layers.{i}."
            x_in16 = bf.dec[i - 1]["s22"].out16 if i > 0 else bf.a16
            dsd = ds16[:Md]
            so = 4096 + 16 * i
            if i == len(bf.dec) - 1:      # (the other layers' norm22 backward ran in the epilogue of the layer above's last data-gradient GEMM)
                self._ln_bwd(g32, sv["s22"], p_ + "norm22", dsd, Md, dr(so + self.SITE_FFOUT), p_ + "linear2.bias")
            self._ffn_bwd(g32, dsd, bf.dh16, sv["hid"], sv["s21"].out16, Md, f"d{i}", p_, "norm22", ff_scale)
            # speaker-axis attention block (merge_tfm_encoder.py:373, :388-394)
            self._ln_bwd(g32, sv["s21"], p_ + "norm21", dsd, Md, dr(so + self.SITE_OUT2), p_ + "self_attn2.out_proj.bias")
            self._wgrad(dsd, sv["o2"], Md, D, D, p_ + "self_attn2.out_proj.weight")
            _call("eend_gemm_bf16", dsd, D, W[f"d{i}.out2_wT"], D, None, dctx16[:Md], D, Md, D, D)
            _call("eend_spk_attn_bwd_bf16", sv["qkv"], dctx16[:Md], dqkv16[:Md], B, C, Tp, H, 0.125, dr(so + self.SITE_SPK))
            self._wgrad_bias(dqkv16[:Md], sv["s11"].out16, Md, 3 * D, D, p_ + "self_attn2.in_proj_weight", p_ + "self_attn2.in_proj_bias")
            # ... its in-projection's data gradient joins the stream, and norm11's backward runs in the same launch
            # time-axis attention block (:364, :379-385)
            self._gemm_acc_ln_bwd(dqkv16[:Md], 3 * D, W[f"d{i}.in2_wT"], g32, sv["s11"], p_ + "norm11", dsd, Md, dr(so + self.SITE_OUT1),
                                  p_ + "self_attn1.out_proj.bias")
            below = None
            if i > 0:                     # the layer below ends in norm22: its backward rides on this block's last GEMM
                pb = f"dec.attractor_decoder.layers.{i - 1}."
                below = (bf.dec[i - 1]["s22"], pb + "norm22", dr(so - 16 + self.SITE_FFOUT), pb + "linear2.bias")
            self._attn_bwd(g32, dsd, dctx16[:Md], dqkv16[:Md], sv["att"], x_in16, B * C, Tp, Md, W[f"d{i}.out1_wT"],
                           W[f"d{i}.in1_wT"], p_ + "self_attn1.out_proj", p_ + "self_attn1.in_proj", bf, dm, T, T,
                           dr(so + self.SITE_ATT), next_ln=below)

        # ---- convert fan-out (model :113-114, factored): g32 = gradient w.r.t. attr0
        _call("eend_convert_fanout_bwd_f32", g32, bf.gsum16, self.ws, WS_FLOATS, bf.dpc, B, Tp, C)
        _call("eend_convert_const_f32", 1, None, None, self.pe, None, bf.dpc, self._G("dec.convert.weight"), self._G("dec.convert.bias"), C)
        self._wgrad(bf.gsum16, bf.emb16, Me, D, D, "dec.convert.weight", ld_out=2 * D, k_out=D)
        _call("eend_gemm_acc_bf16", bf.gsum16, D, W["convert.w1T"], D, bf.de32, 1.0, bf.de32, None, Me, D)

        # ---- L2 norm + look-ahead conv (model :40-41); de32 = gradient w.r.t. the unit embeddings (head + emb loss + convert)
        _call("eend_l2norm_bwd_bf16", bf.emb32, bf.de32, bf.inv_norm, bf.demb16, B, T, Tp)
        self._bias_grad(bf.demb16, Me, D, "cnn.bias")
        _call("eend_conv1d_wgrad_bf16", bf.demb16, bf.enc_out16, bf.il, B, Tp, D, self.ktaps, self.cpad, self.ws, WS_FLOATS, bf.conv_tmp,
              self._G("cnn.weight"))
        _call("eend_conv1d_dgrad_bf16", bf.demb16, W["cnn.wd"], bf.tl, bf.il, bf.ge32, B, Tp, D, self.ktaps, self.ktaps - 1 - self.cpad)

        # ---- encoder layers, last to first
        g32 = bf.ge32
        dse = ds16[:Me]
        for i in reversed(range(len(bf.enc))):
            sv = bf.enc[i]
            p_ = f"enc.transformer_encoder.layers.{i}."
            x_in16 = bf.enc[i - 1]["s2"].out16 if i > 0 else bf.site0.out16
            so = 16 * i
            if i == len(bf.enc) - 1:      # (the other layers' norm2 backward ran in the epilogue of the layer above's last data-gradient GEMM)
                self._ln_bwd(g32, sv["s2"], p_ + "norm2", dse, Me, dr(so + self.SITE_FFOUT), p_ + "linear2.bias")
            self._ffn_bwd(g32, dse, bf.dh16, sv["hid"], sv["s1"].out16, Me, f"e{i}", p_, "norm2", ff_scale)
            self._ln_bwd(g32, sv["s1"], p_ + "norm1", dse, Me, dr(so + self.SITE_OUT1), p_ + "self_attn.out_proj.bias")
            if i > 0:
                pb = f"enc.transformer_encoder.layers.{i - 1}."
                below = (bf.enc[i - 1]["s2"], pb + "norm2", dr(so - 16 + self.SITE_FFOUT), pb + "linear2.bias")
            else:                         # the encoder's input LayerNorm (model :166)
                below = (bf.site0, "enc.encoder_norm", None, "enc.encoder.bias")
            self._attn_bwd(g32, dse, dctx16[:Me], dqkv16[:Me], sv["att"], x_in16, B, Tp, Me, W[f"e{i}.out_wT"], W[f"e{i}.in_wT"],
                           p_ + "self_attn.out_proj", p_ + "self_attn.in_proj", bf, bf.delay_e, bf.kv_e, T, dr(so + self.SITE_ATT), next_ln=below)

        # ---- input projection + LayerNorm + BatchNorm (model :166,:173-174)
        if len(bf.enc) == 0:
            self._ln_bwd(g32, bf.site0, "enc.encoder_norm", dse, Me, None, "enc.encoder.bias")
        _call("eend_wgrad_bf16", dse, D, bf.xin16, self.Fin_pad, 1, Me, D, self.Fin_pad, self.ws, WS_FLOATS, self._G("enc.encoder.weight"),
              self.Fin, self.Fin, 1.0, 0)
        _call("eend_gemm_bf16", dse, D, W["enc.in.wT"], D, None, bf.dy_in, self.Fin_pad, Me, self.Fin_pad, D)
        _call("eend_bn_bwd_f32", bf.ptrs, bf.lens, -1.0, bf.bn_mean, bf.bn_var, m.enc.bn.eps, bf.dy_in, self.Fin_pad, self.ws, WS_FLOATS,
              self._G("enc.bn.weight"), self._G("enc.bn.bias"), B, T, Tp, self.Fin)

