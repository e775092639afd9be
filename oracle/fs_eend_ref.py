"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the
FS-EEND frame-wise diarization forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; it is the checker, never the thing shipped or measured as the
product.  The product path (fs-eend_amd/) never imports anything under oracle/.

Pinned against the reference by tests/golden/*.npz, generated in the build
container by oracle/gen_golden.py, which imports /root/reference/FS-EEND/nnet
(torch 2.10 CPU fp32) and stores weights + inputs + per-stage outputs.

Everything is written as explicit tensor algebra on a flat ``state_dict`` (the
reference's own parameter names), no torch.nn modules, so that the arithmetic
the HIP kernels must reproduce is spelled out.  Citations are file:line under
/root/reference/FS-EEND/.

``dtype`` may be torch.float32 (the parity oracle) or torch.float64 (an
"exact" arbiter used to judge which of two fp32-class results is closer).

``q`` is an optional operand-quantiser hook ``q(tensor, role) -> tensor`` used
only by precision-study tests (emulating bf16 MFMA operands); None = exact.
"""
import math
from typing import Callable, Dict, List, Optional, Sequence

import torch

Tensor = torch.Tensor
LN_EPS = 1e-5          # torch.nn.LayerNorm default, used everywhere in FS-EEND
BN_EPS = 1e-5          # torch.nn.BatchNorm1d default (nnet/model/onl_tfm...l2norm.py:142)


def _id(x, role=None):
    return x


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def linear(x: Tensor, w: Tensor, b: Optional[Tensor], q=_id, role="lin") -> Tensor:
    """y = x @ w.T + b   (torch.nn.Linear)."""
    y = q(x, role + ".a") @ q(w, role + ".w").t()
    return y if b is None else y + b


def layer_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float = LN_EPS) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)          # biased, as torch
    y = (x - mu) / torch.sqrt(var + eps)
    if w is not None:
        y = y * w + b
    return y


def causal_allowed(T: int, mask_delay: int = 0, device=None) -> Tensor:
    """Boolean (T,T): allowed[i,j] <=> j - i <= mask_delay.

    Restates _generate_square_subsequent_mask (model :107-110, :152-155):
    ``triu(ones, diagonal=-mask_delay).T`` then 0 / -inf fill.  Bit-exact
    integer predicate; the HIP kernel evaluates it per element and never
    materialises the (T,T) tensor.
    """
    i = torch.arange(T, device=device)[:, None]
    j = torch.arange(T, device=device)[None, :]
    return (j - i) <= mask_delay


def mha(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
        n_heads: int, allowed: Optional[Tensor], q=_id, role="mha",
        kv: Optional[Tensor] = None, pdrop=None) -> Tensor:
    """torch.nn.MultiheadAttention(x,x,x) on (N, L, D) batch-first data.

    packed in-proj (3D x D), per-head scaled dot product with scale 1/sqrt(dh),
    additive {0,-inf} mask, softmax over keys, AV, out-proj.
    Call sites: nn.TransformerEncoderLayer (model :147), _sa_block1/_sa_block2
    (modules/merge_tfm_encoder.py:379-394).  ``kv`` (N, S, D) lets the
    streaming restatement attend over a longer key/value sequence
    (modules/streaming_tfm.py:29-35).
    """
    N, L, D = x.shape
    dh = D // n_heads
    src = x if kv is None else kv
    S = src.shape[1]
    qkv_q = linear(x, in_w[:D], in_b[:D], q, role + ".inq")
    qkv_k = linear(src, in_w[D:2 * D], in_b[D:2 * D], q, role + ".ink")
    qkv_v = linear(src, in_w[2 * D:], in_b[2 * D:], q, role + ".inv")
    qh = qkv_q.reshape(N, L, n_heads, dh).transpose(1, 2)      # (N,H,L,dh)
    kh = qkv_k.reshape(N, S, n_heads, dh).transpose(1, 2)
    vh = qkv_v.reshape(N, S, n_heads, dh).transpose(1, 2)
    s = (q(qh, role + ".q") @ q(kh, role + ".k").transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if allowed is not None:
        s = s.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(s, dim=-1)
    if pdrop is not None:                                       # nn.MultiheadAttention(dropout=p): dropout of the probabilities
        p = pdrop(p)
    o = q(p, role + ".p") @ q(vh, role + ".v")                 # (N,H,L,dh)
    o = o.transpose(1, 2).reshape(N, L, D)
    return linear(o, out_w, out_b, q, role + ".out")


def sinusoid_pe(n_rows: int, d_model: int, dtype=torch.float32) -> Tensor:
    """PositionalEncoding table rows 0..n_rows-1 (model :206-214).  The
    reference builds it in fp32; we do the same and then cast."""
    pe = torch.zeros(n_rows, d_model)
    position = torch.arange(0, n_rows, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.to(dtype)


# ----------------------------------------------------------------------------
# the batch (masked) model
# ----------------------------------------------------------------------------
def encoder_layer(x: Tensor, sd: Dict[str, Tensor], pfx: str, n_heads: int,
                  allowed: Optional[Tensor], q=_id, drop=None, site0: int = 0) -> Tensor:
    """Post-norm nn.TransformerEncoderLayer, ReLU (model :147).
    x = LN1(x + drop1(MHA(x))); x = LN2(x + drop2(W2 drop(relu(W1 x)))); the attention probabilities are dropped
    too.  ``drop`` (oracle/dropout_ref.HashDropout or None) supplies the masks; x: (B, T, D)."""
    N, T, _ = x.shape
    a = mha(x, sd[pfx + "self_attn.in_proj_weight"], sd[pfx + "self_attn.in_proj_bias"],
            sd[pfx + "self_attn.out_proj.weight"], sd[pfx + "self_attn.out_proj.bias"],
            n_heads, allowed, q, "enc.mha",
            pdrop=None if drop is None else (lambda p: drop.attn(p, site0 + drop.SITE_ATT)))
    if drop is not None:
        a = drop.rows(a, site0 + drop.SITE_OUT1, drop.seq_rows(N, T, x.device))
    x = layer_norm(x + a, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"])
    h = torch.relu(linear(x, sd[pfx + "linear1.weight"], sd[pfx + "linear1.bias"], q, "enc.ff1"))
    if drop is not None:
        h = drop.rows(h, site0 + drop.SITE_FF, drop.seq_rows(N, T, x.device))
    f = linear(h, sd[pfx + "linear2.weight"], sd[pfx + "linear2.bias"], q, "enc.ff2")
    if drop is not None:
        f = drop.rows(f, site0 + drop.SITE_FFOUT, drop.seq_rows(N, T, x.device))
    return layer_norm(x + f, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])


def encoder(src: Sequence[Tensor], sd: Dict[str, Tensor], n_heads: int, n_layers: int,
            has_mask: bool = True, mask_delay: int = 0, q=_id, dtype=torch.float32,
            taps: Optional[dict] = None, bn_batch_stats: Optional[dict] = None, drop=None) -> Tensor:
    """MaskedTransformerEncoderModel.forward (model :162-188).

    pad(-1) -> BatchNorm1d -> Linear -> LN -> n_layers causal post-norm Transformer layers.
    Returns (B, Tmax, D).  BatchNorm uses the running statistics (eval mode) unless
    ``bn_batch_stats`` is a dict: then it is train mode -- statistics over all B*Tmax padded
    frames (the -1 padding included, exactly as ``self.bn(src.transpose(1, 2))`` sees it),
    biased variance for the normalisation; the dict receives ``mean``, ``var_biased`` and
    ``count`` for the caller's running-statistics update.
    """
    x = torch.nn.utils.rnn.pad_sequence([s.to(dtype) for s in src], padding_value=-1.0,
                                        batch_first=True)                      # :165
    if bn_batch_stats is None:
        x = (x - sd["enc.bn.running_mean"]) / torch.sqrt(sd["enc.bn.running_var"] + BN_EPS)
    else:
        flat = x.reshape(-1, x.shape[-1])
        mu = flat.mean(dim=0)
        var = ((flat - mu) ** 2).mean(dim=0)
        bn_batch_stats.update(mean=mu.detach(), var_biased=var.detach(), count=flat.shape[0])
        x = (x - mu) / torch.sqrt(var + BN_EPS)
    x = x * sd["enc.bn.weight"] + sd["enc.bn.bias"]                            # :166
    x = linear(x, sd["enc.encoder.weight"], sd["enc.encoder.bias"], q, "enc.in")   # :173
    x = layer_norm(x, sd["enc.encoder_norm.weight"], sd["enc.encoder_norm.bias"])  # :174
    if taps is not None:
        taps["enc_in"] = x
    allowed = causal_allowed(x.shape[1], mask_delay, x.device) if has_mask else None
    for i in range(n_layers):
        x = encoder_layer(x, sd, f"enc.transformer_encoder.layers.{i}.", n_heads, allowed, q, drop, 16 * i)
        if taps is not None:
            taps[f"enc_l{i}"] = x
    return x


def lookahead_conv_l2(enc_out: Tensor, ilens: Sequence[int], sd: Dict[str, Tensor],
                      q=_id) -> Tensor:
    """Truncate to ilen, zero re-pad (model :38-39), Conv1d(D,D,k,padding=9)
    (model :30,:40), then x/||x||_2 with no eps (model :41)."""
    emb = [e[:l] for e, l in zip(enc_out, ilens)]
    emb = torch.nn.utils.rnn.pad_sequence(emb, padding_value=0.0, batch_first=True)
    w = sd["cnn.weight"]                                   # (Dout, Din, k)
    k = w.shape[-1]
    pad = 9                                                # hard-coded in FS (model :30)
    xp = torch.nn.functional.pad(emb, (0, 0, pad, k - 1 - pad))
    B, T, D = emb.shape
    # explicit tap sum == cross-correlation of nn.Conv1d
    win = xp.unfold(1, k, 1)                               # (B, T, Din, k)
    y = q(win.reshape(B, T, D * k), "cnn.a") @ q(w.reshape(w.shape[0], D * k), "cnn.w").t()
    y = y + sd["cnn.bias"]
    return y / torch.linalg.vector_norm(y, dim=-1, keepdim=True)


def fusion_layer(x: Tensor, sd: Dict[str, Tensor], pfx: str, n_heads: int,
                 allowed: Optional[Tensor], q=_id, drop=None, site0: int = 0) -> Tensor:
    """TransformerEncoderFusionLayer.forward (modules/merge_tfm_encoder.py:356-376),
    post-norm branch.  x: (B, T, C, D).  ``drop``: the six dropout sites of the layer (:385 dropout11 + self_attn1's
    probabilities, :394 dropout21 + self_attn2's probabilities, :398 dropout, :399 dropout2)."""
    B, T, C, D = x.shape
    y = x.transpose(1, 2).reshape(B * C, T, D)
    a = mha(y, sd[pfx + "self_attn1.in_proj_weight"], sd[pfx + "self_attn1.in_proj_bias"],
            sd[pfx + "self_attn1.out_proj.weight"], sd[pfx + "self_attn1.out_proj.bias"],
            n_heads, allowed, q, "dec.mha_t",
            pdrop=None if drop is None else (lambda p: drop.attn(p, site0 + drop.SITE_ATT)))
    if drop is not None:
        a = drop.rows(a, site0 + drop.SITE_OUT1, drop.seq_rows(B * C, T, x.device))
    y = layer_norm(y + a, sd[pfx + "norm11.weight"], sd[pfx + "norm11.bias"])      # :364
    y = y.reshape(B, C, T, D).transpose(1, 2).reshape(B * T, C, D)
    a = mha(y, sd[pfx + "self_attn2.in_proj_weight"], sd[pfx + "self_attn2.in_proj_bias"],
            sd[pfx + "self_attn2.out_proj.weight"], sd[pfx + "self_attn2.out_proj.bias"],
            n_heads, None, q, "dec.mha_s",
            pdrop=None if drop is None else (lambda p: drop.spk(p, site0 + drop.SITE_SPK, B, T)))
    rows = None if drop is None else drop.slot_rows(B, T, C, x.device)              # (B*T, C) slab rows (b*C + c)*Tp + t
    if drop is not None:
        a = drop.rows(a, site0 + drop.SITE_OUT2, rows)
    y = layer_norm(y + a, sd[pfx + "norm21.weight"], sd[pfx + "norm21.bias"])      # :373
    h = torch.relu(linear(y, sd[pfx + "linear1.weight"], sd[pfx + "linear1.bias"], q, "dec.ff1"))
    if drop is not None:
        h = drop.rows(h, site0 + drop.SITE_FF, rows)
    f = linear(h, sd[pfx + "linear2.weight"], sd[pfx + "linear2.bias"], q, "dec.ff2")
    if drop is not None:
        f = drop.rows(f, site0 + drop.SITE_FFOUT, rows)
    y = layer_norm(y + f, sd[pfx + "norm22.weight"], sd[pfx + "norm22.bias"])      # :374
    return y.reshape(B, T, C, D)


def decoder(emb: Tensor, max_nspks: int, sd: Dict[str, Tensor], n_heads: int, n_layers: int,
            mask_delay: int = 0, q=_id, taps: Optional[dict] = None, drop=None) -> Tensor:
    """MaskedTransformerDecoderModel.forward (model :112-118).
    attr0[b,t,c] = convert([emb[b,t]; pe[c]]), then n_layers fusion layers.
    NB the decoder always applies the causal mask (:116), whatever has_mask."""
    B, T, D = emb.shape
    pe = sd["dec.pos_enc.pe"][0, :max_nspks].to(emb.dtype)               # (C, D) rows = speaker slot
    cat = torch.cat([emb[:, :, None, :].expand(B, T, max_nspks, D),
                     pe[None, None].expand(B, T, max_nspks, D)], dim=-1)
    x = linear(cat, sd["dec.convert.weight"], sd["dec.convert.bias"], q, "dec.convert")
    if taps is not None:
        taps["attr0"] = x
    allowed = causal_allowed(T, mask_delay, emb.device)
    for i in range(n_layers):
        x = fusion_layer(x, sd, f"dec.attractor_decoder.layers.{i}.", n_heads, allowed, q, drop, 4096 + 16 * i)
        if taps is not None:
            taps[f"dec_l{i}"] = x
    return x


def fs_test(src: Sequence[Tensor], ilens: Sequence[int], sd: Dict[str, Tensor], *, n_heads: int,
            enc_n_layers: int, dec_n_layers: int, max_nspks: int = 6, has_mask: bool = True,
            mask_delay: int = 0, q=_id, dtype=torch.float32, taps: Optional[dict] = None):
    """OnlineTransformerDADiarization.test (model :67-84).

    Returns (logits list[(T_i,C)], emb list[(T_i,D)], attractors list[(T_i,C,D)]).
    """
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    enc_out = encoder(src, sd, n_heads, enc_n_layers, has_mask, mask_delay, q, dtype, taps)
    emb = lookahead_conv_l2(enc_out, ilens, sd, q)
    if taps is not None:
        taps["emb"] = emb
    attr = decoder(emb, max_nspks, sd, n_heads, dec_n_layers, mask_delay, q, taps)
    attr = attr / torch.linalg.vector_norm(attr, dim=-1, keepdim=True)       # :76
    out = (q(emb, "head.e")[:, :, None, :] * q(attr, "head.a")).sum(-1)     # :79  <emb, attr_c>
    logits = [o[:l] for o, l in zip(out, ilens)]
    embs = [e[:l] for e, l in zip(emb, ilens)]
    attrs = [a[:l] for a, l in zip(attr, ilens)]
    return logits, embs, attrs


def emb_consistency_loss(emb: Tensor, tgt_pad: Tensor) -> Tensor:
    """model :46-57: MSE between cos-sim(emb) and cos-sim(labels), (B,T,T) maps,
    eps 1e-6 added to the norm products."""
    attn_map = emb @ emb.transpose(-1, -2)
    n = torch.linalg.vector_norm(emb, dim=-1, keepdim=True)
    attn_map = attn_map / (n @ n.transpose(-1, -2) + 1e-6)
    label_map = tgt_pad @ tgt_pad.transpose(-1, -2)
    tn = torch.linalg.vector_norm(tgt_pad, dim=-1, keepdim=True)
    label_map = label_map / (tn @ tn.transpose(-1, -2) + 1e-6)
    return ((attn_map - label_map) ** 2).mean()


def fs_forward(src: Sequence[Tensor], tgt: Sequence[Tensor], ilens: Sequence[int],
               sd: Dict[str, Tensor], *, n_heads: int, enc_n_layers: int, dec_n_layers: int,
               has_mask: bool = True, mask_delay: int = 0, q=_id, dtype=torch.float32,
               bn_batch_stats: Optional[dict] = None, drop=None):
    """OnlineTransformerDADiarization.forward (model :32-65); BN running stats (eval) or, with
    ``bn_batch_stats`` a dict, batch statistics (train mode, see encoder()).  Dropout is off unless ``drop`` is
    an oracle/dropout_ref.HashDropout (the masks are then the HIP path's counter-hash masks, NOT torch's)."""
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    n_speakers = [t.shape[1] for t in tgt]
    C = max(n_speakers)
    enc_out = encoder(src, sd, n_heads, enc_n_layers, has_mask, mask_delay, q, dtype,
                      bn_batch_stats=bn_batch_stats, drop=drop)
    emb = lookahead_conv_l2(enc_out, ilens, sd, q)
    attr = decoder(emb, C, sd, n_heads, dec_n_layers, mask_delay, q, drop=drop)
    attr = attr / torch.linalg.vector_norm(attr, dim=-1, keepdim=True)
    tgt_pad = [torch.nn.functional.pad(t.to(dtype), (0, C - t.shape[1])) for t in tgt]
    tgt_pad = torch.nn.utils.rnn.pad_sequence(tgt_pad, padding_value=0.0, batch_first=True)
    loss = emb_consistency_loss(emb, tgt_pad)
    out = (emb[:, :, None, :] * attr).sum(-1)
    logits = [o[:l, :n] for o, l, n in zip(out, ilens, n_speakers)]
    embs = [e[:l] for e, l in zip(emb, ilens)]
    attrs = [a[:l, 1:n] for a, l, n in zip(attr, ilens, n_speakers)]
    return logits, loss, embs, attrs


# ----------------------------------------------------------------------------
# frame-by-frame streaming model (modules/streaming_tfm.py,
# model/streaming_tfm_enc_1dcnn...l2norm.py), parameters addressed through the
# *masked* model's names (utils/copy_params.py:7-57 defines the mapping).
# ----------------------------------------------------------------------------
class FsStreamingRef:
    """StreamingTransformerEDADiarization.test restated over the masked
    model's state_dict.  One frame in -> one (conv_delay-delayed) frame out."""

    def __init__(self, sd: Dict[str, Tensor], *, n_heads: int, enc_n_layers: int,
                 dec_n_layers: int, conv_delay: int = 9, dtype=torch.float32):
        self.sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        self.H, self.Le, self.Ld = n_heads, enc_n_layers, dec_n_layers
        self.k = 2 * conv_delay + 1
        self.center = self.k // 2
        self.dtype = dtype
        self.enc_cache: List[Optional[Tensor]] = [None] * enc_n_layers      # x history per layer
        self.dec_cache: List[Optional[Tensor]] = [None] * dec_n_layers
        self.buf: List[Tensor] = []
        self.t = 0

    def _enc_step(self, x_t: Tensor) -> Tensor:
        sd = self.sd
        x = (x_t - sd["enc.bn.running_mean"]) / torch.sqrt(sd["enc.bn.running_var"] + BN_EPS)
        x = x * sd["enc.bn.weight"] + sd["enc.bn.bias"]                    # streaming_tfm.py:115
        x = layer_norm(linear(x, sd["enc.encoder.weight"], sd["enc.encoder.bias"]),
                       sd["enc.encoder_norm.weight"], sd["enc.encoder_norm.bias"])
        for i in range(self.Le):
            p = f"enc.transformer_encoder.layers.{i}."
            # K/V cache holds the layer *inputs*; keys/values are re-projected
            # (IncrementalSelfAttention caches key=value=x, :28-35)
            hist = x if self.enc_cache[i] is None else torch.cat([self.enc_cache[i], x], dim=1)
            self.enc_cache[i] = hist
            a = mha(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
                    sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"],
                    self.H, None, kv=hist)
            x = layer_norm(a + x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
            f = linear(torch.relu(linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                       sd[p + "linear2.weight"], sd[p + "linear2.bias"])
            x = layer_norm(f + x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        return x

    def _conv_step(self, e_t: Tensor) -> Optional[Tensor]:
        """StreamingConv1d.forward (streaming_tfm.py:141-167)."""
        self.t += 1
        self.buf.append(e_t)                       # (B,1,D)
        self.buf = self.buf[-self.k:]
        left = self.k - len(self.buf)
        win = [torch.zeros_like(e_t)] * left + self.buf
        win = torch.cat(win, dim=1)                # (B,k,D)
        w = self.sd["cnn.weight"]                  # (Dout,Din,k)
        y = torch.einsum("bkd,odk->bo", win, w) + self.sd["cnn.bias"]
        return y[:, None, :] if self.t >= self.center + 1 else None

    def _dec_step(self, emb_t: Tensor, C: int) -> Tensor:
        sd = self.sd
        B, _, D = emb_t.shape
        pe = sd["dec.pos_enc.pe"][0, :C].to(self.dtype)
        cat = torch.cat([emb_t[:, :, None, :].expand(B, 1, C, D), pe[None, None].expand(B, 1, C, D)], -1)
        x = linear(cat, sd["dec.convert.weight"], sd["dec.convert.bias"])   # (B,1,C,D)
        for i in range(self.Ld):
            p = f"dec.attractor_decoder.layers.{i}."
            xt = x.transpose(1, 2).reshape(B * C, 1, D)
            hist = xt if self.dec_cache[i] is None else torch.cat([self.dec_cache[i], xt], dim=1)
            self.dec_cache[i] = hist
            a = mha(xt, sd[p + "self_attn1.in_proj_weight"], sd[p + "self_attn1.in_proj_bias"],
                    sd[p + "self_attn1.out_proj.weight"], sd[p + "self_attn1.out_proj.bias"],
                    self.H, None, kv=hist)
            y = layer_norm(a + xt, sd[p + "norm11.weight"], sd[p + "norm11.bias"])
            y = y.reshape(B, C, D)
            a = mha(y, sd[p + "self_attn2.in_proj_weight"], sd[p + "self_attn2.in_proj_bias"],
                    sd[p + "self_attn2.out_proj.weight"], sd[p + "self_attn2.out_proj.bias"],
                    self.H, None)
            y = layer_norm(a + y, sd[p + "norm21.weight"], sd[p + "norm21.bias"])
            f = linear(torch.relu(linear(y, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                       sd[p + "linear2.weight"], sd[p + "linear2.bias"])
            y = layer_norm(f + y, sd[p + "norm22.weight"], sd[p + "norm22.bias"])
            x = y[:, None]
        return x

    def test(self, x_t: Tensor, max_nspks: int = 6, dummy_conv_input: bool = False):
        """streaming model :31-60.  x_t (B,1,in)."""
        x_t = x_t.to(self.dtype)
        if dummy_conv_input:
            e = torch.zeros(1, 1, self.sd["cnn.weight"].shape[1], dtype=self.dtype)
        else:
            e = self._enc_step(x_t)
        e = self._conv_step(e)
        if e is None:
            return None
        e = e / torch.linalg.vector_norm(e, dim=-1, keepdim=True)
        a = self._dec_step(e, max_nspks)
        a = a / torch.linalg.vector_norm(a, dim=-1, keepdim=True)
        return (e[:, :, None, :] * a).sum(-1)                              # (B,1,C)


# ----------------------------------------------------------------------------
# callers' arithmetic that the bench/tests need (train/utils/loss.py)
# ----------------------------------------------------------------------------
def standard_loss(ys: Sequence[Tensor], ts: Sequence[Tensor]) -> Tensor:
    """train/utils/loss.py:119-125 with label_delay=0."""
    losses = [torch.nn.functional.binary_cross_entropy_with_logits(y, t) * len(y)
              for y, t in zip(ys, ts)]
    return torch.stack(losses).sum() / sum(t.shape[0] for t in ts)


def calc_diarization_error(pred: Tensor, label: Tensor) -> Dict[str, float]:
    """train/utils/loss.py:198-236 with label_delay=0 (frame-level DER counters)."""
    decisions = torch.sigmoid(pred) > 0.5
    n_ref = label.sum(dim=-1).long()
    n_sys = decisions.sum(dim=-1).long()
    res = {}
    res["speech_scored"] = int((n_ref > 0).sum())
    res["speech_miss"] = int(((n_ref > 0) & (n_sys == 0)).sum())
    res["speech_falarm"] = int(((n_ref == 0) & (n_sys > 0)).sum())
    res["speaker_scored"] = int(n_ref.sum())
    res["speaker_miss"] = int(torch.clamp(n_ref - n_sys, min=0).sum())
    res["speaker_falarm"] = int(torch.clamp(n_sys - n_ref, min=0).sum())
    n_map = ((label == 1) & (decisions == 1)).sum(dim=-1)
    res["speaker_error"] = int((torch.min(n_ref, n_sys) - n_map).sum())
    res["correct"] = float((label == decisions).sum()) / label.shape[1]
    res["diarization_error"] = res["speaker_miss"] + res["speaker_falarm"] + res["speaker_error"]
    res["frames"] = len(label)
    return res
