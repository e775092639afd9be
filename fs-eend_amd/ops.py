"""Tensor-level wrappers over the C-ABI (include/eend_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every
arithmetic op of the hot path is one call into libeend_hip.so.  All wrappers
require CUDA(ROCm) tensors and raise on anything else -- there is no eager or
CPU fallback.
"""
import math

import torch

from . import lib as _lib

F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.EendHipError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.EendHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.EendHipError(f"{name}: expected a contiguous tensor")


def frames_pad(T: int) -> int:
    """Frames per sequence slab: T rounded up to a multiple of 64."""
    return (T + 63) // 64 * 64


def bn_cast_pad(x, bn, out16, T, Tp, apply_bn=True, eps=1e-5):
    """x f32 (B,T,Fin) -> out16 f16 (B*Tp, Fpad).  bn = (weight, bias, mean, var) or None."""
    L = _lib.load()
    _chk(x, F32, "x"); _chk(out16, F16, "out16")
    B, Fin = x.shape[0], x.shape[2]
    Fpad = out16.shape[-1]
    w = b = m = v = None
    if apply_bn:
        w, b, m, v = bn
        for t, n in ((w, "bn.weight"), (b, "bn.bias"), (m, "bn.mean"), (v, "bn.var")):
            _chk(t, F32, n)
    _lib.check(L.eend_bn_cast_pad_f16(_p(x), _p(w), _p(b), _p(m), _p(v), eps, _p(out16), B, T, Tp, Fin, Fpad,
                                      1 if apply_bn else 0, _stream()), "eend_bn_cast_pad_f16")
    return out16


ACT_NONE, ACT_RELU, ACT_SWISH = 0, 1, 2


def linear(a16, w16, bias, out16, relu=False, act=None):
    """out16 = act(a16 @ w16.T + bias); a16 (M,K) f16, w16 (N,K) f16."""
    if act is None:
        act = ACT_RELU if relu else ACT_NONE
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(w16, F16, "w16"); _chk(bias, F32, "bias"); _chk(out16, F16, "out16")
    M, K = a16.shape
    N = w16.shape[0]
    _lib.check(L.eend_linear_f16(_p(a16), a16.stride(0), _p(w16), w16.stride(0), _p(bias), _p(out16),
                                 out16.stride(0), M, N, K, act, _stream()), "eend_linear_f16")
    return out16


def inproj_heads(a16, w16, bias, q, k, vt, nseq, Tp, H):
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(w16, F16, "w16"); _chk(bias, F32, "bias")
    _chk(q, BF16, "q"); _chk(k, BF16, "k"); _chk(vt, BF16, "vt")
    K = a16.shape[1]
    _lib.check(L.eend_inproj_heads_bf16(_p(a16), a16.stride(0), _p(w16), w16.stride(0), _p(bias), _p(q), _p(k),
                                        _p(vt), nseq, Tp, H, 64, K, _stream()), "eend_inproj_heads_bf16")


def linear_glu(a16, wi16, bias_i, out16):
    """GLU(a16 @ W.T + b) with value/gate rows interleaved in wi16 (2N, K); out16 (M, N)."""
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(wi16, F16, "wi16"); _chk(bias_i, F32, "bias_i"); _chk(out16, F16, "out16")
    M, K = a16.shape
    _lib.check(L.eend_linear_glu_f16(_p(a16), a16.stride(0), _p(wi16), wi16.stride(0), _p(bias_i), _p(out16),
                                     out16.stride(0), M, wi16.shape[0], K, _stream()), "eend_linear_glu_f16")


def linear_res_scale_ln16(a16, w16, bias, res, alpha, gamma, beta, out32, out16, eps=1e-5):
    """out32 = (a16 @ w16.T + bias) * alpha + res ; out16 = f16(LN(out32) * gamma + beta)."""
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(w16, F16, "w16"); _chk(bias, F32, "bias"); _chk(res, F32, "res")
    _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta"); _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    M, K = a16.shape
    _lib.check(L.eend_linear_res_scale_ln16_f16(_p(a16), a16.stride(0), _p(w16), w16.stride(0), _p(bias), _p(res),
                                                float(alpha), _p(gamma), _p(beta), eps, _p(out32), _p(out16), M, K,
                                                _stream()), "eend_linear_res_scale_ln16_f16")


def retention_proj(a16, wqkvg16, bias, q, k, kt, vt, g, nseq, Tp, H):
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(wqkvg16, F16, "wqkvg"); _chk(bias, F32, "bias")
    for t, n in ((q, "q"), (k, "k"), (kt, "kt"), (vt, "vt"), (g, "g")):
        _chk(t, F16, n)
    _lib.check(L.eend_retention_proj_f16(_p(a16), a16.stride(0), _p(wqkvg16), wqkvg16.stride(0), _p(bias), _p(q),
                                         _p(k), _p(kt), _p(vt), _p(g), nseq, Tp, H, 64, a16.shape[1], _stream()),
               "eend_retention_proj_f16")


_KV_WS = {}


def retention_chunk(q, k, kt, vt, g, o16, st_ws, cscale_ws, sexp_ws, nseq, H, Tp, chunk, gn_eps=1e-6, t_valid=0,
                    state_in=None, state_out=None):
    L = _lib.load()
    for t, n in ((q, "q"), (k, "k"), (kt, "kt"), (vt, "vt"), (g, "g"), (o16, "o16"), (st_ws, "st_ws")):
        _chk(t, F16, n)
    _chk(cscale_ws, F32, "cscale_ws"); _chk(sexp_ws, F32, "sexp_ws"); _chk(state_in, F32, "state_in"); _chk(state_out, F32, "state_out")
    for st_ in (state_in, state_out):
        if st_ is not None and st_.numel() < nseq * H * 4096:
            raise _lib.EendHipError("retention_chunk: carried state must be (nseq, H, 64, 64) f32")
    nc = (Tp + chunk - 1) // chunk
    if st_ws.numel() < nseq * H * nc * 2 * 4096 or cscale_ws.numel() < nseq * H * nc or sexp_ws.numel() < nseq * H * nc:
        raise _lib.EendHipError("retention_chunk: workspace too small")
    need = nseq * H * nc * 4096
    kv_ws = _KV_WS.get(str(q.device))
    if kv_ws is None or kv_ws.numel() < need:                  # f32 per-chunk K^T V workspace, grown on demand
        kv_ws = torch.empty(need, dtype=F32, device=q.device)
        _KV_WS[str(q.device)] = kv_ws
    _lib.check(L.eend_retention_chunk_f16(_p(q), _p(k), _p(kt), _p(vt), _p(g), _p(o16), _p(st_ws), _p(kv_ws),
                                          _p(cscale_ws), _p(sexp_ws), nseq, H, Tp, chunk, o16.stride(0), g.stride(0),
                                          gn_eps, int(t_valid), _p(state_in), _p(state_out), _stream()), "eend_retention_chunk_f16")


def retention_stream_ok(L, Tp, ldx=256, ldo=256):
    return bool(_lib.load().eend_retention_stream_ok(int(L), int(Tp), int(ldx), int(ldo)))


def retention_stream_pack(wqkvg32):
    """Pack the f32 [q; k * dk^-0.5; v; g] rows (1024, 256) into the weight stream of retention_stream (hi / lo parts of the q rows)."""
    L = _lib.load()
    _chk(wqkvg32, F32, "wqkvg32")
    if wqkvg32.shape != (1024, 256):
        raise _lib.EendHipError("retention_stream_pack: expected the packed projection [1024][256]")
    out = torch.empty(L.eend_retention_stream_elems(), dtype=F16, device=wqkvg32.device)
    _lib.check(L.eend_retention_stream_pack_f16(_p(wqkvg32), _p(out), _stream()), "eend_retention_stream_pack_f16")
    return out


def retention_stream(x16, xlo16, wstream, bias, o16, st_ws, cscale_ws, sexp_ws, nseq, Tp, chunk, gn_eps=1e-6, t_valid=0,
                     state_in=None, state_out=None):
    """o16 = swish(g) * LN_head(retention(q, k, v)) with q / k / v / g projected on chip from the rows x16 (ret_stream.hip); xlo16
    (optional): the f16 remainder of the f32 stream behind x16 (query path precision).  H = 4."""
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(xlo16, F16, "xlo16"); _chk(wstream, F16, "wstream"); _chk(bias, F32, "bias"); _chk(o16, F16, "o16")
    _chk(st_ws, F16, "st_ws"); _chk(cscale_ws, F32, "cscale_ws"); _chk(sexp_ws, F32, "sexp_ws")
    _chk(state_in, F32, "state_in"); _chk(state_out, F32, "state_out")
    H = 4
    if x16.shape != (nseq * Tp, 256) or o16.shape[0] < nseq * Tp or bias.numel() != 1024 or wstream.numel() != L.eend_retention_stream_elems() or \
            (xlo16 is not None and (xlo16.shape != x16.shape or xlo16.stride(0) != x16.stride(0))):
        raise _lib.EendHipError("retention_stream: shape mismatch")
    for st_ in (state_in, state_out):
        if st_ is not None and st_.numel() < nseq * H * 4096:
            raise _lib.EendHipError("retention_stream: carried state must be (nseq, H, 64, 64) f32")
    nc = (Tp + chunk - 1) // chunk
    if st_ws.numel() < nseq * H * nc * 2 * 4096 or cscale_ws.numel() < nseq * H * nc or sexp_ws.numel() < nseq * H * nc:
        raise _lib.EendHipError("retention_stream: workspace too small")
    need = nseq * H * nc * 4096
    kv_ws = _KV_WS.get(str(x16.device))
    if kv_ws is None or kv_ws.numel() < need:                  # f32 per-chunk K^T V workspace, grown on demand
        kv_ws = torch.empty(need, dtype=F32, device=x16.device)
        _KV_WS[str(x16.device)] = kv_ws
    _lib.check(L.eend_retention_stream_f16(_p(x16), x16.stride(0), _p(xlo16), _p(wstream), _p(bias), _p(o16), o16.stride(0), _p(st_ws),
                                           _p(kv_ws), _p(cscale_ws), _p(sexp_ws), nseq, Tp, chunk, gn_eps, int(t_valid), _p(state_in),
                                           _p(state_out), _stream()), "eend_retention_stream_f16")


def layernorm_f16(x32, gamma, beta, out16, eps=1e-5):
    L = _lib.load()
    _chk(x32, F32, "x32"); _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta"); _chk(out16, F16, "out16")
    M, D = x32.shape
    _lib.check(L.eend_layernorm_f16(_p(x32), _p(gamma), _p(beta), eps, _p(out16), M, D, _stream()), "eend_layernorm_f16")


def dwconv_bn_swish(x16, w, bn, out16, nseq, Tp, eps=1e-5, halo16=None):
    """x16/out16 f16 (nseq*Tp, D); w f32 (D, k); bn = (weight, bias, mean, var); halo16 f16 (nseq, k-1, D) or None."""
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(w, F32, "w"); _chk(out16, F16, "out16")
    for t in bn:
        _chk(t, F32, "bn")
    D, k = w.shape
    _chk(halo16, F16, "halo16")
    if halo16 is not None and tuple(halo16.shape) != (nseq, k - 1, D):
        raise _lib.EendHipError("dwconv_bn_swish: halo must be (nseq, k-1, D)")
    _lib.check(L.eend_dwconv_bn_swish_f16(_p(x16), _p(w), _p(bn[0]), _p(bn[1]), _p(bn[2]), _p(bn[3]), eps, _p(out16),
                                          nseq, Tp, D, k, _p(halo16), _stream()), "eend_dwconv_bn_swish_f16")


def linear_res_ln(a16, w16, bias, res, gamma, beta, out32, out16, eps=1e-5, alpha=1.0):
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(w16, F16, "w16"); _chk(bias, F32, "bias"); _chk(res, F32, "res")
    _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta"); _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    M, K = a16.shape
    if w16.shape[0] != 256:
        raise _lib.EendHipError("linear_res_ln: N must be 256")
    _lib.check(L.eend_linear_res_ln_f16(_p(a16), a16.stride(0), _p(w16), w16.stride(0), _p(bias), _p(res),
                                        float(alpha), _p(gamma), _p(beta), eps, _p(out32), _p(out16), M, K, _stream()),
               "eend_linear_res_ln_f16")


def linear_res_scale(a16, w16, bias, res, alpha, out32, out16):
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(w16, F16, "w16"); _chk(bias, F32, "bias"); _chk(res, F32, "res")
    _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    M, K = a16.shape
    if w16.shape[0] != 256:
        raise _lib.EendHipError("linear_res_scale: N must be 256")
    _lib.check(L.eend_linear_res_scale_f16(_p(a16), a16.stride(0), _p(w16), w16.stride(0), _p(bias), _p(res),
                                           float(alpha), _p(out32), _p(out16), M, K, _stream()),
               "eend_linear_res_scale_f16")


def conv1d_l2norm(x16, wr16, bias, ilens_i32, out32, out16, nseq, Tp, cin, ktaps, pad):
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(wr16, F16, "wr16"); _chk(bias, F32, "bias"); _chk(ilens_i32, torch.int32, "ilens")
    _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    _lib.check(L.eend_conv1d_l2norm_f16(_p(x16), _p(wr16), _p(bias), _p(ilens_i32), _p(out32), _p(out16), nseq,
                                        Tp, cin, ktaps, pad, _stream()), "eend_conv1d_l2norm_f16")


def conv_stream_ok(cin, ktaps, pad):
    return bool(_lib.load().eend_conv_stream_ok(int(cin), int(ktaps), int(pad)))


def conv_stream_pack(wr16, ktaps):
    """Pack the tap-major conv weight Wr f16 [256][ktaps*256] into the weight stream of conv1d_l2norm_stream."""
    L = _lib.load()
    _chk(wr16, F16, "wr16")
    if wr16.shape != (256, ktaps * 256):
        raise _lib.EendHipError("conv_stream_pack: expected Wr [256][ktaps*256]")
    out = torch.empty(L.eend_conv_stream_elems(int(ktaps)), dtype=F16, device=wr16.device)
    _lib.check(L.eend_conv_stream_pack_f16(_p(wr16), _p(out), int(ktaps), _stream()), "eend_conv_stream_pack_f16")
    return out


def conv1d_l2norm_stream(x16, wstream, bias, ilens_i32, out32, out16, nseq, Tp, ktaps, pad):
    """conv1d_l2norm on the packed weight stream (conv_stream.hip; cin = 256)."""
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(wstream, F16, "wstream"); _chk(bias, F32, "bias"); _chk(ilens_i32, torch.int32, "ilens")
    _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    if x16.shape != (nseq * Tp, 256) or out32.shape != (nseq * Tp, 256) or out16.shape != (nseq * Tp, 256) or \
            wstream.numel() != L.eend_conv_stream_elems(int(ktaps)):
        raise _lib.EendHipError("conv1d_l2norm_stream: shape mismatch")
    _lib.check(L.eend_conv1d_l2norm_stream_f16(_p(x16), _p(wstream), _p(bias), _p(ilens_i32), _p(out32), _p(out16), nseq, Tp, ktaps, pad,
                                               _stream()), "eend_conv1d_l2norm_stream_f16")


def convert_fanout(e16, w1_16, pc, out32, out16, B, Tp, C):
    L = _lib.load()
    _chk(e16, F16, "e16"); _chk(w1_16, F16, "w1"); _chk(pc, F32, "pc"); _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    _lib.check(L.eend_convert_fanout_f16(_p(e16), _p(w1_16), _p(pc), _p(out32), _p(out16), B, Tp, C, _stream()),
               "eend_convert_fanout_f16")


def convert_fanout_f32(e32, w32, pc, out32, out16, B, Tp, C, out16lo=None):
    """convert_fanout with f32 operands (exact-f32 MFMA): e32 (B*Tp, 256) f32, w32 the (256, >= 256) f32 convert.weight; out16lo
    (optional): the f16 remainder of the f32 rows."""
    L = _lib.load()
    _chk(e32, F32, "e32"); _chk(w32, F32, "w32"); _chk(pc, F32, "pc"); _chk(out32, F32, "out32"); _chk(out16, F16, "out16"); _chk(out16lo, F16, "out16lo")
    _lib.check(L.eend_convert_fanout_f32(_p(e32), _p(w32), w32.stride(0), _p(pc), _p(out32), _p(out16), _p(out16lo), B, Tp, C, _stream()),
               "eend_convert_fanout_f32")


LN2 = math.log(2.0)
# Multiply the q rows of an in-projection (weight and bias) by QSCALE_LOG2 and call attn_causal with
# scale=LN2: the scores then leave the QK^T MFMA already scaled and in the log2 domain, which is what the
# whole-sequence attention kernel's cheap softmax path needs (attn_full.hip, LAZY).
QSCALE_LOG2 = (1.0 / math.sqrt(64.0)) * math.log2(math.e)


def attn_causal(q, k, vt, o16, nseq, H, Tp, mask_delay=0, kv_len=None, scale=1.0 / math.sqrt(64.0)):
    L = _lib.load()
    _chk(q, BF16, "q"); _chk(k, BF16, "k"); _chk(vt, BF16, "vt"); _chk(o16, F16, "o16")
    _lib.check(L.eend_attn_causal_bf16(_p(q), _p(k), _p(vt), _p(o16), nseq, H, Tp, o16.stride(0), mask_delay,
                                       Tp if kv_len is None else kv_len, scale, _stream()), "eend_attn_causal_bf16")


def inproj_attn_pack(w_in):
    """Pack in_proj_weight f16 [768][256] (q rows pre-scaled by QSCALE_LOG2) for inproj_attn_causal_packed."""
    L = _lib.load()
    _chk(w_in, F16, "w_in")
    if w_in.shape != (768, 256):
        raise _lib.EendHipError("inproj_attn_pack: expected in_proj_weight [768][256]")
    out = torch.empty(L.eend_inproj_attn_packed_elems(), dtype=F16, device=w_in.device)
    _lib.check(L.eend_inproj_attn_pack_f16(_p(w_in), _p(out), _stream()), "eend_inproj_attn_pack_f16")
    return out


def inproj_attn_causal_packed(x16, w_packed, b_in, o16, nseq, H, Tp, mask_delay=0, kv_len=None):
    """Packed in-projection + causal attention in one launch (attn_stream.hip; Tp = 64 m <= 512, H = 4): Q, K, V never reach HBM."""
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(w_packed, F16, "w_packed"); _chk(b_in, F32, "b_in"); _chk(o16, F16, "o16")
    if w_packed.numel() != L.eend_inproj_attn_packed_elems() or b_in.numel() != 768 or x16.shape[0] < nseq * Tp or o16.shape[0] < nseq * Tp:
        raise _lib.EendHipError("inproj_attn_causal_packed: shape mismatch")
    _lib.check(L.eend_inproj_attn_causal_packed_f16(_p(x16), x16.stride(0), _p(w_packed), _p(b_in), _p(o16), nseq, H, Tp,
                                                    o16.stride(0), mask_delay, Tp if kv_len is None else kv_len, _stream()),
               "eend_inproj_attn_causal_packed_f16")


def inproj_attn_long_scratch(nseq, Tp, mask_delay=0, kv_len=None):
    """(partial-row f16 elements, lse f32 elements) eend_inproj_attn_causal_long_f16 needs for this shape, or None if it does not cover it."""
    import ctypes
    L = _lib.load()
    a, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
    rc = L.eend_inproj_attn_long_scratch_elems(nseq, Tp, mask_delay, Tp if kv_len is None else kv_len, ctypes.byref(a), ctypes.byref(b))
    return None if rc != 0 else (a.value, b.value)


def inproj_attn_causal_long(x16, w_packed, b_in, o16, part16, lse32, nseq, H, Tp, mask_delay=0, kv_len=None):
    """Packed in-projection + causal attention for windows of more than 512 frames (attn_stream.hip, groups of 512 frames + combine)."""
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(w_packed, F16, "w_packed"); _chk(b_in, F32, "b_in"); _chk(o16, F16, "o16"); _chk(part16, F16, "part16"); _chk(lse32, F32, "lse32")
    kv = Tp if kv_len is None else kv_len
    need = inproj_attn_long_scratch(nseq, Tp, mask_delay, kv)
    if need is None:
        raise _lib.EendHipError("inproj_attn_causal_long: shape not covered")
    if (w_packed.numel() != L.eend_inproj_attn_packed_elems() or b_in.numel() != 768 or x16.shape[0] < nseq * Tp or o16.shape[0] < nseq * Tp
            or part16.numel() < need[0] or lse32.numel() < need[1]):
        raise _lib.EendHipError("inproj_attn_causal_long: shape mismatch")
    _lib.check(L.eend_inproj_attn_causal_long_f16(_p(x16), x16.stride(0), _p(w_packed), _p(b_in), _p(o16), _p(part16), _p(lse32), nseq, H, Tp,
                                                  o16.stride(0), mask_delay, kv, _stream()),
               "eend_inproj_attn_causal_long_f16")


def proj_stream_pack(w16, out=None):
    """Re-order W f16 [N][256] (N = 256 n <= 1024) into the packed weight stream of proj_stream; None if the shape has no stream form."""
    L = _lib.load()
    _chk(w16, F16, "w16")
    n = L.eend_proj_stream_elems(int(w16.shape[0])) if (w16.dim() == 2 and w16.shape[1] == 256) else 0
    if n <= 0:
        return None
    if out is None:
        out = torch.empty(n, dtype=F16, device=w16.device)
    _lib.check(L.eend_proj_stream_pack_f16(_p(w16), _p(out), int(w16.shape[0]), _stream()), "eend_proj_stream_pack_f16")
    return out


def _proj_groups(groups):
    arr = (_lib.ProjGroup * len(groups))()
    for a, gd in zip(arr, groups):
        rows, kind, t = gd.get("rows"), gd.get("kind", 0), gd.get("heads_t")
        a.rows = _p(rows); a.rows_kind = kind if rows is not None else 0
        a.rows_bf16 = 1 if (rows is not None and rows.dtype == BF16) else 0
        a.rows_ld = int(gd.get("ld", 0))
        a.rows2_bf16_heads = _p(gd.get("rows2"))
        a.heads_t = _p(t); a.heads_t_bf16 = 1 if (t is not None and t.dtype == BF16) else 0
    return arr


def proj_stream_ok(ldx, M, N, Tp, H, groups):
    return bool(_lib.load().eend_proj_stream_ok(int(ldx), int(M), int(N), int(Tp), int(H), _proj_groups(groups)))


def proj_stream(x16, wstream, bias, M, N, Tp, H, groups):
    """Y = x16 W^T + bias on the packed stream; groups: one dict per 256 output features with rows (+ kind 1 row-major / 2 head rows, ld),
    rows2 (bf16 head rows), heads_t (transposed head rows) -- see eend_proj_stream_f16."""
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(wstream, F16, "wstream"); _chk(bias, F32, "bias")
    if wstream.numel() != L.eend_proj_stream_elems(N) or bias.numel() != N or len(groups) != N // 256 or x16.shape[0] < M:
        raise _lib.EendHipError("proj_stream: shape mismatch")
    for gd in groups:
        for k in ("rows2",):
            if gd.get(k) is not None:
                _chk(gd[k], BF16, k)
    _lib.check(L.eend_proj_stream_f16(_p(x16), x16.stride(0), _p(wstream), _p(bias), M, N, Tp, H, _proj_groups(groups), _stream()),
               "eend_proj_stream_f16")


def spk_stream_pack(wo16, win16):
    """Pack Wo1 [256][256] + in_proj_weight [768][256] (f16) into the weight stream of attnout_spk_stream."""
    L = _lib.load()
    _chk(wo16, F16, "wo16"); _chk(win16, F16, "win16")
    if wo16.shape != (256, 256) or win16.shape != (768, 256):
        raise _lib.EendHipError("spk_stream_pack: expected Wo [256][256] and W_in [768][256]")
    out = torch.empty(L.eend_spk_stream_elems(), dtype=F16, device=wo16.device)
    _lib.check(L.eend_spk_stream_pack_f16(_p(wo16), _p(win16), _p(out), _stream()), "eend_spk_stream_pack_f16")
    return out


def spk_stream_ok(C, Tp):
    return bool(_lib.load().eend_spk_stream_ok(int(C), int(Tp)))


def attnout_spk_stream(a16, wstream, bo, res16, g1, be1, eps1, x16, b_in, out16, B, C, Tp):
    """x16 = LN11(a16 @ Wo1.T + bo + res16); out16 = MHA over the C slots of (x16 @ W_in.T + b_in), one launch.
    x16 may be res16 and out16 may be a16."""
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(wstream, F16, "wstream"); _chk(res16, F16, "res16"); _chk(x16, F16, "x16"); _chk(out16, F16, "out16")
    for n, t in (("bo", bo), ("g1", g1), ("be1", be1), ("b_in", b_in)):
        _chk(t, F32, n)
    M = B * C * Tp
    if a16.shape != (M, 256) or res16.shape != (M, 256) or x16.shape != (M, 256) or out16.shape != (M, 256) or b_in.numel() != 768:
        raise _lib.EendHipError("attnout_spk_stream: shape mismatch")
    if wstream.numel() != L.eend_spk_stream_elems():
        raise _lib.EendHipError("attnout_spk_stream: weight stream has the wrong size")
    _lib.check(L.eend_attnout_spk_stream_f16(_p(a16), a16.stride(0), _p(wstream), _p(bo), _p(res16), _p(g1), _p(be1), eps1, _p(x16),
                                             _p(b_in), _p(out16), B, C, Tp, 0.125, _stream()), "eend_attnout_spk_stream_f16")


def attnout_spk_stream_res32(a16, wstream, bo, res32, g1, be1, eps1, x32, b_in, out16, B, C, Tp):
    """attnout_spk_stream on an f32 residual stream: x32 = LN11(a16 @ Wo1.T + bo + res32) (f32 rows; x32 may be res32), out16 as above."""
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(wstream, F16, "wstream"); _chk(res32, F32, "res32"); _chk(x32, F32, "x32"); _chk(out16, F16, "out16")
    for n, t in (("bo", bo), ("g1", g1), ("be1", be1), ("b_in", b_in)):
        _chk(t, F32, n)
    M = B * C * Tp
    if a16.shape != (M, 256) or res32.shape != (M, 256) or x32.shape != (M, 256) or out16.shape != (M, 256) or b_in.numel() != 768:
        raise _lib.EendHipError("attnout_spk_stream_res32: shape mismatch")
    if wstream.numel() != L.eend_spk_stream_elems():
        raise _lib.EendHipError("attnout_spk_stream_res32: weight stream has the wrong size")
    _lib.check(L.eend_attnout_spk_stream_res32_f16(_p(a16), a16.stride(0), _p(wstream), _p(bo), _p(res32), _p(g1), _p(be1), eps1, _p(x32),
                                                   _p(b_in), _p(out16), B, C, Tp, 0.125, _stream()), "eend_attnout_spk_stream_res32_f16")


def spk_attn(qkv16, o16, B, C, Tp, H):
    L = _lib.load()
    _chk(qkv16, F16, "qkv16"); _chk(o16, F16, "o16")
    _lib.check(L.eend_spk_attn_f16(_p(qkv16), _p(o16), B, C, Tp, H, 1.0 / math.sqrt(64.0), _stream()),
               "eend_spk_attn_f16")


def emb_consistency(emb32, labels, T, lens=None, inv_count=0.0):
    """MSE between the cosine-similarity maps of the embeddings and of the labels (FS model :46-57).
    emb32 f32 (B, Tp, D) (frames >= T ignored), labels f32 (B, T, C) zero-padded -> 0-dim f32 tensor.
    lens (int32 device tensor, B) zeroes the embeddings beyond each length and inv_count overrides the
    1/(B*T*T) normalisation (LS model :92-113)."""
    L = _lib.load()
    _chk(emb32, F32, "emb32"); _chk(labels, F32, "labels"); _chk(lens, torch.int32, "lens")
    B, Tp, D = emb32.shape
    if labels.shape[0] != B or labels.shape[1] != T or Tp < T:
        raise _lib.EendHipError("emb_consistency: shape mismatch")
    nt = (T + 63) // 64
    ws = torch.empty(B * nt * nt, dtype=F32, device=emb32.device)
    out = torch.empty(1, dtype=F32, device=emb32.device)
    _lib.check(L.eend_emb_consistency_f32(_p(emb32), _p(labels), _p(lens) if lens is not None else None, float(inv_count),
                                          _p(ws), _p(out), B, T, Tp, D, labels.shape[2], _stream()),
               "eend_emb_consistency_f32")
    return out[0]


def head_l2dot(emb32, attr, attr_out, logits, B, T, Tp, C, D):
    """attr: the un-normalised attractor rows, f32 or f16 (B*C*Tp, D)."""
    L = _lib.load()
    _chk(emb32, F32, "emb32"); _chk(attr_out, F32, "attr_out"); _chk(logits, F32, "logits")
    if attr.dtype == F16:
        _chk(attr, F16, "attr")
        _lib.check(L.eend_head_l2dot_a16_f32(_p(emb32), _p(attr), _p(attr_out), _p(logits), B, T, Tp, C, D, _stream()),
                   "eend_head_l2dot_a16_f32")
        return
    _chk(attr, F32, "attr")
    _lib.check(L.eend_head_l2dot_f32(_p(emb32), _p(attr), _p(attr_out), _p(logits), B, T, Tp, C, D, _stream()),
               "eend_head_l2dot_f32")


def retention_step(qkvg16, kv_state, scale_in, scale_out, out16, N, H, gn_eps=1e-6):
    L = _lib.load()
    _chk(qkvg16, F16, "qkvg16"); _chk(kv_state, F32, "kv_state"); _chk(scale_in, F32, "scale_in")
    _chk(scale_out, F32, "scale_out"); _chk(out16, F16, "out16")
    _lib.check(L.eend_retention_step_f16(_p(qkvg16), _p(kv_state), _p(scale_in), _p(scale_out), _p(out16), N, H, gn_eps,
                                         _stream()), "eend_retention_step_f16")


def retention_proj_step(x32, ln, w32, b32, qkvg32, N):
    """qkvg32 (N, 1024) f32 = LayerNorm(x32 (N, 256)) @ w32.T + b32, all f32 (ln = (gamma, beta, eps) or None)."""
    L = _lib.load()
    _chk(x32, F32, "x32"); _chk(w32, F32, "w32"); _chk(b32, F32, "b32"); _chk(qkvg32, F32, "qkvg32")
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    _lib.check(L.eend_retention_proj_step_f32(_p(x32), _p(g), _p(b), float(eps), _p(w32), _p(b32), _p(qkvg32), N, _stream()),
               "eend_retention_proj_step_f32")


def retention_step_f32(qkvg32, kv_state, scale_in, scale_out, out16, N, H, gn_eps=1e-6, out32=None):
    L = _lib.load()
    _chk(qkvg32, F32, "qkvg32"); _chk(kv_state, F32, "kv_state"); _chk(scale_in, F32, "scale_in")
    _chk(scale_out, F32, "scale_out"); _chk(out16, F16, "out16"); _chk(out32, F32, "out32")
    _lib.check(L.eend_retention_step_f32(_p(qkvg32), _p(kv_state), _p(scale_in), _p(scale_out), _p(out16), _p(out32), N, H, gn_eps,
                                         _stream()), "eend_retention_step_f32")


# The f32 frame-step linears (skinny.hip) serve any number of rows in groups of 16 (round 4: multi-stream sessions take the
# all-f32 step too -- their f16 step drifted past the 1e-3 bar within an hour): no row limit.
STEP_F32_MAX_ROWS = 1 << 30


def linear_step_f32(a32, w32, bias, out32, act=ACT_NONE):
    """out32 = act(a32 w32^T + bias), everything f32 (the all-f32 LS decoder frame step)."""
    L = _lib.load()
    _chk(a32, F32, "a32"); _chk(w32, F32, "w32"); _chk(bias, F32, "bias"); _chk(out32, F32, "out32")
    M, K = a32.shape
    _lib.check(L.eend_linear_step_f32(_p(a32), a32.stride(0), _p(w32), w32.stride(0), _p(bias), _p(out32), out32.stride(0), M,
                                      w32.shape[0], K, act, _stream()), "eend_linear_step_f32")


def linear_res_ln_step_f32(a32, w32, bias, res, gamma, beta, out32, eps=1e-5, alpha=1.0, out16=None):
    """out32 = LayerNorm((a32 w32^T + bias) * alpha + res), f32 operands; out32 may alias res."""
    L = _lib.load()
    _chk(a32, F32, "a32"); _chk(w32, F32, "w32"); _chk(bias, F32, "bias"); _chk(res, F32, "res"); _chk(gamma, F32, "gamma")
    _chk(beta, F32, "beta"); _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    M, K = a32.shape
    if w32.shape[0] != 256:
        raise _lib.EendHipError("linear_res_ln_step_f32: N must be 256")
    _lib.check(L.eend_linear_res_ln_step_f32(_p(a32), a32.stride(0), _p(w32), w32.stride(0), _p(bias), _p(res), float(alpha), _p(gamma),
                                             _p(beta), eps, _p(out32), _p(out16), M, K, _stream()), "eend_linear_res_ln_step_f32")


def linear_res_scale_ln_step_f32(a32, w32, bias, res, alpha, gamma, beta, out32, ln_out32=None, ln_out16=None, eps=1e-5):
    """out32 = (a32 w32^T + bias) * alpha + res (un-normalised stream, may alias res); ln_out32 / ln_out16 = LayerNorm(out32)."""
    L = _lib.load()
    for n, t in (("a32", a32), ("w32", w32), ("bias", bias), ("res", res), ("gamma", gamma), ("beta", beta), ("out32", out32), ("ln_out32", ln_out32)):
        _chk(t, F32, n)
    _chk(ln_out16, F16, "ln_out16")
    M, K = a32.shape
    if w32.shape[0] != 256:
        raise _lib.EendHipError("linear_res_scale_ln_step_f32: N must be 256")
    _lib.check(L.eend_linear_res_scale_ln_step_f32(_p(a32), a32.stride(0), _p(w32), w32.stride(0), _p(bias), _p(res), float(alpha),
                                                   _p(gamma), _p(beta), eps, _p(out32), _p(ln_out32), _p(ln_out16), M, K, _stream()),
               "eend_linear_res_scale_ln_step_f32")


def layernorm_rows_f32(x32, gamma, beta, out32, eps=1e-5):
    L = _lib.load()
    _chk(x32, F32, "x32"); _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta"); _chk(out32, F32, "out32")
    if x32.shape[-1] != 256:
        raise _lib.EendHipError("layernorm_rows_f32: rows of 256 features")
    _lib.check(L.eend_layernorm_rows_f32(_p(x32), _p(gamma), _p(beta), eps, _p(out32), x32.numel() // 256, _stream()),
               "eend_layernorm_rows_f32")


def l2norm_rows_f32(x32, y32):
    L = _lib.load()
    _chk(x32, F32, "x32"); _chk(y32, F32, "y32")
    if x32.shape[-1] != 256:
        raise _lib.EendHipError("l2norm_rows_f32: rows of 256 features")
    _lib.check(L.eend_l2norm_rows_f32(_p(x32), _p(y32), x32.numel() // 256, _stream()), "eend_l2norm_rows_f32")


def spk_attn_step_f32(qkv32, out32, B, C):
    L = _lib.load()
    _chk(qkv32, F32, "qkv32"); _chk(out32, F32, "out32")
    _lib.check(L.eend_spk_attn_step_f32(_p(qkv32), _p(out32), B, C, 1.0 / math.sqrt(64.0), _stream()), "eend_spk_attn_step_f32")


def convert_fanout_step_f32(emb32, w32, pc, out32, out16, B, C):
    """One frame of `convert(cat(emb, pe_c))` in f32: emb32 (B, 256), w32 the (256, 512) convert.weight, pc (C, 256)."""
    L = _lib.load()
    _chk(emb32, F32, "emb32"); _chk(w32, F32, "w32"); _chk(pc, F32, "pc"); _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    _lib.check(L.eend_convert_fanout_step_f32(_p(emb32), _p(w32), w32.stride(0), _p(pc), _p(out32), _p(out16), B, C, _stream()),
               "eend_convert_fanout_step_f32")


def dwconv_step(x16, cache, w, bn, out16, eps=1e-5):
    """x16/out16 f16 (B, D); cache f32 (B, D, k-1) shifted in place; w f32 (D, k)."""
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(cache, F32, "cache"); _chk(w, F32, "w"); _chk(out16, F16, "out16")
    for t in bn:
        _chk(t, F32, "bn")
    B, D = x16.shape
    _lib.check(L.eend_dwconv_step_f16(_p(x16), _p(cache), _p(w), _p(bn[0]), _p(bn[1]), _p(bn[2]), _p(bn[3]), eps,
                                      _p(out16), B, D, w.shape[1], _stream()), "eend_dwconv_step_f16")


def attn_decode(qkv16, k_cache, v_cache, out16, N, H, cap, t):
    """Append the new token's k/v (row t) to the caches (N,H,cap,64) f16 and attend over t+1 tokens."""
    L = _lib.load()
    _chk(qkv16, F16, "qkv16"); _chk(k_cache, F16, "k_cache"); _chk(v_cache, F16, "v_cache"); _chk(out16, F16, "out16")
    _lib.check(L.eend_attn_decode_f16(_p(qkv16), _p(k_cache), _p(v_cache), _p(out16), N, H, cap, t,
                                      1.0 / math.sqrt(64.0), _stream()), "eend_attn_decode_f16")


def attn_decode_dev(qkv16, k_cache, v_cache, out16, N, H, cap, t_dev):
    """attn_decode with the token count in device memory (int32 tensor of one element): graph-capturable."""
    L = _lib.load()
    _chk(qkv16, F16, "qkv16"); _chk(k_cache, F16, "k_cache"); _chk(v_cache, F16, "v_cache"); _chk(out16, F16, "out16")
    _chk(t_dev, torch.int32, "t_dev")
    _lib.check(L.eend_attn_decode_dev_f16(_p(qkv16), _p(k_cache), _p(v_cache), _p(out16), N, H, cap, _p(t_dev),
                                          1.0 / math.sqrt(64.0), _stream()), "eend_attn_decode_dev_f16")


SPLIT_DECODE_MIN_CAP = 4096      # cache capacities from here on take the key-split decode kernel


def attn_decode_split_ws(N, H, cap):
    return N * H * ((cap + 511) // 512) * 66


def attn_decode_split(qkv16, k_cache, v_cache, out16, ws, N, H, cap, t_dev):
    """attn_decode_dev with the history split over cap/512 workgroups per (n, h) + a merge kernel (long streams)."""
    L = _lib.load()
    _chk(qkv16, F16, "qkv16"); _chk(k_cache, F16, "k_cache"); _chk(v_cache, F16, "v_cache"); _chk(out16, F16, "out16")
    _chk(t_dev, torch.int32, "t_dev"); _chk(ws, F32, "ws")
    _lib.check(L.eend_attn_decode_split_f16(_p(qkv16), _p(k_cache), _p(v_cache), _p(out16), _p(ws), ws.numel(), N, H, cap, _p(t_dev),
                                            1.0 / math.sqrt(64.0), _stream()), "eend_attn_decode_split_f16")


def counter_add(counter_i32, inc=1):
    L = _lib.load()
    _chk(counter_i32, torch.int32, "counter")
    _lib.check(L.eend_counter_add_i32(_p(counter_i32), inc, _stream()), "eend_counter_add_i32")


_PTR_TABLES = {}


def gather_bn_cast_pad(src, bn, out16, T, Tp, pad_value, apply_bn=True, eps=1e-5):
    """src: list of B f32 GPU tensors (T_i, Fin) -> out16 f16 (B*Tp, Fpad) in one launch (pointer table)."""
    L = _lib.load()
    _chk(out16, F16, "out16")
    for s_ in src:
        _chk(s_, F32, "src[i]")
    B, Fin, Fpad = len(src), src[0].shape[1], out16.shape[-1]
    if Fpad > 512:
        raise _lib.EendHipError(f"gather_bn_cast_pad: padded feature width {Fpad} > 512 (in_size {Fin}): the gather kernel holds "
                                "four feature pairs per lane; the shipped configs use in_size 345")
    tab = _ptr_table(src, T, out16.device)
    w = b = m = v = None
    if apply_bn:
        w, b, m, v = bn
    _lib.check(L.eend_gather_bn_cast_pad_f16(_p(tab[0]), _p(tab[1]), float(pad_value), _p(w), _p(b), _p(m), _p(v), eps,
                                             _p(out16), B, T, Tp, Fin, Fpad, 1 if apply_bn else 0, _stream()),
               "eend_gather_bn_cast_pad_f16")
    return out16


def _ptr_table(src, T, dev):
    key = tuple((s_.data_ptr(), s_.shape[0]) for s_ in src)
    tab = _PTR_TABLES.get(key)
    if tab is None:                                            # tiny H2D upload, cached per (pointers, lengths)
        if len(_PTR_TABLES) > 64:
            _PTR_TABLES.clear()
        tab = (torch.tensor([k[0] for k in key], dtype=torch.int64, device=dev),
               torch.tensor([min(k[1], T) for k in key], dtype=torch.int32, device=dev))
        _PTR_TABLES[key] = tab
    return tab


def encoder_input_ok(src, Tp, w16):
    """True where encoder_input covers the shapes (else the caller runs gather_bn_cast_pad + linear_res_ln)."""
    Fin = src[0].shape[1]
    return (bool(_lib.load().eend_encoder_input_ok(int(Fin), int(Tp), int(w16.stride(0)))) and w16.shape[0] == 256
            and all(s_.is_contiguous() and s_.data_ptr() % 16 == 0 and s_.shape[1] == Fin for s_ in src))


def encoder_input(src, bn, w16, bias, gamma, beta, out32, out16, T, Tp, pad_value, bn_eps=1e-5, eps=1e-5):
    """out = LayerNorm(BN(pad_sequence(src, pad_value)) @ w16.T + bias) in one launch (encin.hip): src list of B f32 GPU tensors
    (T_i, Fin); w16 f16 (256, >= ceil32(Fin)) zero-padded; out16 f16 (B*Tp, 256), out32 optional."""
    L = _lib.load()
    _chk(w16, F16, "w16"); _chk(out16, F16, "out16"); _chk(out32, F32, "out32")
    for n, t in (("bias", bias), ("gamma", gamma), ("beta", beta)) + tuple(("bn", t) for t in bn):
        _chk(t, F32, n)
    for s_ in src:
        _chk(s_, F32, "src[i]")
    B, Fin = len(src), src[0].shape[1]
    if out16.shape != (B * Tp, 256) or not encoder_input_ok(src, Tp, w16):
        raise _lib.EendHipError("encoder_input: unsupported shapes (see eend_encoder_input_ok)")
    tab = _ptr_table(src, T, out16.device)
    w, b, m, v = bn
    _lib.check(L.eend_encoder_input_f16(_p(tab[0]), _p(tab[1]), float(pad_value), _p(w), _p(b), _p(m), _p(v), bn_eps, _p(w16), w16.stride(0),
                                        _p(bias), _p(gamma), _p(beta), eps, _p(out32), _p(out16), B, T, Tp, Fin, _stream()),
               "eend_encoder_input_f16")


def linear_res16_ln(a16, w16, bias, res16, gamma, beta, out32, out16, eps=1e-5, alpha=1.0):
    """out = LayerNorm((a16 @ w16.T + bias) * alpha + res16) with the residual read from the f16 stream; out32 may be None."""
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(w16, F16, "w16"); _chk(bias, F32, "bias"); _chk(res16, F16, "res16")
    _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta"); _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    M, K = a16.shape
    if w16.shape[0] != 256:
        raise _lib.EendHipError("linear_res16_ln: N must be 256")
    _lib.check(L.eend_linear_res16_ln_f16(_p(a16), a16.stride(0), _p(w16), w16.stride(0), _p(bias), _p(res16), float(alpha), _p(gamma),
                                          _p(beta), eps, _p(out32), _p(out16), M, K, _stream()), "eend_linear_res16_ln_f16")


def attnout_ffn_fused_res16(a16, wo, bo, res16, g1, be1, eps1, w1, b1, w2, b2, g2, be2, eps2, out32, out16):
    """attnout_ffn_fused with the out-projection's residual read from the f16 stream; out32 may be None."""
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(wo, F16, "wo"); _chk(w1, F16, "w1"); _chk(w2, F16, "w2"); _chk(res16, F16, "res16")
    for n, t in (("bo", bo), ("g1", g1), ("be1", be1), ("b1", b1), ("b2", b2), ("g2", g2), ("be2", be2), ("out32", out32)):
        _chk(t, F32, n)
    _chk(out16, F16, "out16")
    M, K = a16.shape
    Fh = w1.shape[0]
    if K != 256 or wo.shape != (256, 256) or w1.shape[1] != 256 or w2.shape != (256, Fh):
        raise _lib.EendHipError("attnout_ffn_fused_res16: expected d_model 256")
    _lib.check(L.eend_attnout_ffn_fused_res16_f16(_p(a16), a16.stride(0), _p(wo), _p(bo), _p(res16), _p(g1), _p(be1), eps1, _p(w1),
                                                  _p(b1), _p(w2), _p(b2), _p(g2), _p(be2), eps2, _p(out32), _p(out16), M, Fh,
                                                  _stream()), "eend_attnout_ffn_fused_res16_f16")


def attnout_ffn_fused(a16, wo, bo, res, g1, be1, eps1, w1, b1, w2, b2, g2, be2, eps2, out32, out16, out16lo=None, wo_lo=None):
    """x = LN1(a16 @ wo.T + bo + res); out = LN2(relu(x @ w1.T + b1) @ w2.T + b2 + x): the attention
    out-projection, both residual adds, both LayerNorms and the FFN of a post-LN layer in one launch."""
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(wo, F16, "wo"); _chk(w1, F16, "w1"); _chk(w2, F16, "w2")
    for n, t in (("bo", bo), ("res", res), ("g1", g1), ("be1", be1), ("b1", b1), ("b2", b2), ("g2", g2), ("be2", be2), ("out32", out32)):
        _chk(t, F32, n)
    _chk(out16, F16, "out16"); _chk(out16lo, F16, "out16lo"); _chk(wo_lo, F16, "wo_lo")
    M, K = a16.shape
    Fh = w1.shape[0]
    if K != 256 or wo.shape != (256, 256) or w1.shape[1] != 256 or w2.shape != (256, Fh) or (wo_lo is not None and wo_lo.shape != (256, 256)):
        raise _lib.EendHipError("attnout_ffn_fused: expected d_model 256")
    _lib.check(L.eend_attnout_ffn_fused_f16(_p(a16), a16.stride(0), _p(wo), _p(bo), _p(res), _p(g1), _p(be1), eps1, _p(w1), _p(b1),
                                            _p(w2), _p(b2), _p(g2), _p(be2), eps2, _p(out32), _p(out16), _p(out16lo), _p(wo_lo), M, Fh, _stream()),
               "eend_attnout_ffn_fused_f16")


def ffn_stream_pack(wo, w1, w2):
    """Pack Wo (or None), W1 [F][256], W2 [256][F] into the MFMA-fragment stream of the stream kernels (once per
    parameter version).  With wo the stream serves attnout_ffn_stream, without it ffn_stream."""
    L = _lib.load()
    _chk(w1, F16, "w1"); _chk(w2, F16, "w2")
    Fh = w1.shape[0]
    if w1.shape[1] != 256 or w2.shape != (256, Fh) or Fh % 64 or Fh < 64 or not w1.is_contiguous() or not w2.is_contiguous():
        raise _lib.EendHipError("ffn_stream_pack: expected contiguous W1 [F][256], W2 [256][F], F a multiple of 64")
    if wo is not None:
        _chk(wo, F16, "wo")
        if wo.shape != (256, 256) or not wo.is_contiguous():
            raise _lib.EendHipError("ffn_stream_pack: expected contiguous Wo [256][256]")
    n = L.eend_ffn_stream_elems(Fh, 1 if wo is not None else 0)
    out = torch.empty(n, dtype=F16, device=w1.device)
    _lib.check(L.eend_ffn_stream_pack_f16(_p(wo), _p(w1), _p(w2), _p(out), Fh, _stream()), "eend_ffn_stream_pack_f16")
    return out


def ffn_stream_pack_lo(wo, wo_lo, w1, w2):
    """ffn_stream_pack with the out-projection weight as an f16 hi / lo pair: the stream of attnout_ffn_stream_lo."""
    L = _lib.load()
    for n, t in (("wo", wo), ("wo_lo", wo_lo), ("w1", w1), ("w2", w2)):
        _chk(t, F16, n)
    Fh = w1.shape[0]
    if (w1.shape[1] != 256 or w2.shape != (256, Fh) or Fh % 64 or Fh < 64 or wo.shape != (256, 256) or wo_lo.shape != (256, 256)
            or not all(t.is_contiguous() for t in (wo, wo_lo, w1, w2))):
        raise _lib.EendHipError("ffn_stream_pack_lo: expected contiguous Wo / Wo_lo [256][256], W1 [F][256], W2 [256][F], F a multiple of 64")
    out = torch.empty(L.eend_ffn_stream_elems(Fh, 2), dtype=F16, device=w1.device)
    _lib.check(L.eend_ffn_stream_pack_lo_f16(_p(wo), _p(wo_lo), _p(w1), _p(w2), _p(out), Fh, _stream()), "eend_ffn_stream_pack_lo_f16")
    return out


def attnout_ffn_stream_lo(a16, wstream, bo, res, g1, be1, eps1, b1, b2, g2, be2, eps2, out32, out16, out16lo):
    """attnout_ffn_fused(..., out16lo, wo_lo) on a packed weight stream (ffn_stream_pack_lo): f32 residual rows in, f32 + f16 hi / lo out."""
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(wstream, F16, "wstream"); _chk(out16, F16, "out16"); _chk(out16lo, F16, "out16lo")
    for n, t in (("bo", bo), ("res", res), ("g1", g1), ("be1", be1), ("b1", b1), ("b2", b2), ("g2", g2), ("be2", be2), ("out32", out32)):
        _chk(t, F32, n)
    M, K = a16.shape
    Fh = b1.shape[0]
    if K != 256 or wstream.numel() != L.eend_ffn_stream_elems(Fh, 2) or res is None:
        raise _lib.EendHipError("attnout_ffn_stream_lo: expected d_model 256 and a stream packed with Wo / Wo_lo for this F")
    _lib.check(L.eend_attnout_ffn_stream_lo_f16(_p(a16), a16.stride(0), _p(wstream), _p(bo), _p(res), _p(g1), _p(be1), eps1, _p(b1), _p(b2),
                                                _p(g2), _p(be2), eps2, _p(out32), _p(out16), _p(out16lo), M, Fh, _stream()),
               "eend_attnout_ffn_stream_lo_f16")


def ffn_stream_max_rows(lda=256):
    """Rows one launch of the packed-stream layer-tail kernels takes; larger M is served in several launches inside the C ABI."""
    return int(_lib.load().eend_ffn_stream_max_rows(int(lda)))


def debug_ffn_stream_set(tile_fragments=0, max_rows_per_launch=0):
    """Test hook: force the tile size (2 / 3 fragments) / cap the rows per launch of the packed-stream layer-tail kernels (0 = default)."""
    _lib.check(_lib.load().eend_debug_ffn_stream_set(int(tile_fragments), int(max_rows_per_launch)), "eend_debug_ffn_stream_set")


def stream_ok(Fh):
    """Whether the packed-stream kernels take this hidden width."""
    return Fh % 64 == 0 and 64 <= Fh <= 2048


def attnout_ffn_stream(a16, wstream, bo, res, res16, g1, be1, eps1, b1, b2, g2, be2, eps2, out32, out16):
    """attnout_ffn_fused[_res16] on a packed weight stream (ffn_stream_pack(wo, w1, w2)); exactly one of res (f32) / res16."""
    L = _lib.load()
    _chk(a16, F16, "a16"); _chk(wstream, F16, "wstream"); _chk(res, F32, "res"); _chk(res16, F16, "res16")
    for n, t in (("bo", bo), ("g1", g1), ("be1", be1), ("b1", b1), ("b2", b2), ("g2", g2), ("be2", be2), ("out32", out32)):
        _chk(t, F32, n)
    _chk(out16, F16, "out16")
    M, K = a16.shape
    Fh = b1.shape[0]
    if K != 256 or wstream.numel() != L.eend_ffn_stream_elems(Fh, 1):
        raise _lib.EendHipError("attnout_ffn_stream: expected d_model 256 and a stream packed with Wo for this F")
    _lib.check(L.eend_attnout_ffn_stream_f16(_p(a16), a16.stride(0), _p(wstream), _p(bo), _p(res), _p(res16), _p(g1), _p(be1), eps1,
                                             _p(b1), _p(b2), _p(g2), _p(be2), eps2, _p(out32), _p(out16), M, Fh, _stream()),
               "eend_attnout_ffn_stream_f16")


def ffn_stream(x16, wstream, b1, b2, res, gamma, beta, out32, out16, act=ACT_RELU, alpha=1.0, eps=1e-5,
               residual_unnormalised=False):
    """ffn_fused on a packed weight stream (ffn_stream_pack(None, w1, w2))."""
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(wstream, F16, "wstream"); _chk(b1, F32, "b1"); _chk(b2, F32, "b2")
    _chk(res, F32, "res"); _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta"); _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    M, K = x16.shape
    Fh = b1.shape[0]
    if K != 256 or wstream.numel() != L.eend_ffn_stream_elems(Fh, 0):
        raise _lib.EendHipError("ffn_stream: expected d_model 256 and a stream packed without Wo for this F")
    _lib.check(L.eend_ffn_stream_f16(_p(x16), x16.stride(0), _p(wstream), _p(b1), _p(b2), _p(res), float(alpha),
                                     _p(gamma), _p(beta), eps, _p(out32), _p(out16), M, Fh, act,
                                     1 if residual_unnormalised else 0, _stream()), "eend_ffn_stream_f16")


def ffn_fused(x16, w1, b1, w2, b2, res, gamma, beta, out32, out16, act=ACT_RELU, alpha=1.0, eps=1e-5,
              residual_unnormalised=False):
    """out = LN((act(x16 @ w1.T + b1) @ w2.T + b2) * alpha + res); hidden activations stay on chip."""
    L = _lib.load()
    _chk(x16, F16, "x16"); _chk(w1, F16, "w1"); _chk(w2, F16, "w2"); _chk(b1, F32, "b1"); _chk(b2, F32, "b2")
    _chk(res, F32, "res"); _chk(gamma, F32, "gamma"); _chk(beta, F32, "beta"); _chk(out32, F32, "out32"); _chk(out16, F16, "out16")
    M, K = x16.shape
    Fh = w1.shape[0]
    if K != 256 or w1.shape[1] != 256 or w2.shape != (256, Fh):
        raise _lib.EendHipError("ffn_fused: expected d_model 256")
    _lib.check(L.eend_ffn_fused_f16(_p(x16), x16.stride(0), _p(w1), _p(b1), _p(w2), _p(b2), _p(res), float(alpha),
                                    _p(gamma), _p(beta), eps, _p(out32), _p(out16), M, Fh, act,
                                    1 if residual_unnormalised else 0, _stream()), "eend_ffn_fused_f16")
