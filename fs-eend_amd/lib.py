"""ctypes binding of libeend_hip.so (C-ABI: include/eend_hip.h).

Fails loudly: if the shared object is missing or a symbol is absent the import
of the product path raises -- there is no CPU / eager fallback anywhere."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EEND_HIP_LIB: load another build of the same library (same-box A/B perf studies, tools/ab_variants.sh)
LIB_PATH = os.environ.get("EEND_HIP_LIB") or os.path.join(_HERE, "csrc", "libeend_hip.so")

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_l = ctypes.c_long

ABI_VERSION = 5          # EEND_ABI_VERSION of include/eend_hip.h; load() refuses a library that reports another

# name -> argtypes, exactly the prototypes of include/eend_hip.h
PROTOTYPES = {
    "eend_abi_version": [],
    "eend_bn_cast_pad_f16": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "eend_gather_bn_cast_pad_f16": [_vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "eend_linear_f16": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "eend_inproj_heads_bf16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "eend_linear_glu_f16": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "eend_linear_res_ln_f16": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_linear_res16_ln_f16": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_linear_res_scale_ln16_f16": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_ffn_fused_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp],
    "eend_attnout_ffn_fused_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "eend_attnout_ffn_fused_res16_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_convert_fanout_f32": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "eend_ffn_stream_elems": [_i, _i],
    "eend_ffn_stream_pack_f16": [_vp, _vp, _vp, _vp, _i, _vp],
    "eend_ffn_stream_pack_lo_f16": [_vp, _vp, _vp, _vp, _vp, _i, _vp],
    "eend_attnout_ffn_stream_lo_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _vp],
    "eend_ffn_stream_f16": [_vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp],
    "eend_conv_stream_elems": [_i],
    "eend_conv_stream_ok": [_i, _i, _i],
    "eend_conv_stream_pack_f16": [_vp, _vp, _i, _vp],
    "eend_conv1d_l2norm_stream_f16": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "eend_encoder_input_ok": [_i, _i, _i],
    "eend_encoder_input_f16": [_vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp],
    "eend_inproj_attn_packed_elems": [],
    "eend_inproj_attn_pack_f16": [_vp, _vp, _vp],
    "eend_inproj_attn_causal_packed_f16": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "eend_inproj_attn_train_bf16": [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "eend_gemm_acc_stream_elems": [_i],
    "eend_gemm_acc_stream_ok": [_i, _i, _i],
    "eend_gemm_acc_stream_pack_bf16": [_vp, _i, _vp, _i, _vp],
    "eend_gemm_acc_stream_bf16": [_vp, _i, _vp, _vp, _i, _i, _vp],
    "eend_proj_stream_elems": [_i],
    "eend_proj_stream_pack_f16": [_vp, _vp, _i, _vp],
    "eend_proj_stream_ok": [_i, _i, _i, _i, _i, _vp],
    "eend_proj_stream_f16": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "eend_inproj_attn_long_scratch_elems": [_i, _i, _i, _i, _vp, _vp],
    "eend_inproj_attn_causal_long_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "eend_spk_stream_elems": [],
    "eend_spk_stream_ok": [_i, _i],
    "eend_spk_stream_pack_f16": [_vp, _vp, _vp, _vp],
    "eend_attnout_spk_stream_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "eend_attnout_spk_stream_res32_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "eend_ffn_stream_max_rows": [_i],
    "eend_debug_ffn_stream_set": [_i, _l],
    "eend_attnout_ffn_stream_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_emb_consistency_f32": [_vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "eend_activity_median_u8": [_vp, _i, _i, _i, _f, _i, _vp, _vp],
    "eend_activity_segments_i32": [_vp, _i, _i, _vp, _vp, _i, _vp],
    "eend_der_counters_u64": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp],
    "eend_stft_logmel23_f32": [_vp, _l, _l, _i, _vp, _vp, _vp, _vp],
    "eend_feature_meannorm_f32": [_vp, _vp, _i, _i, _i, _vp],
    "eend_splice_subsample_f32": [_vp, _i, _i, _i, _i, _vp, _vp],
    "eend_pit_cost_f64": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "eend_pit_assign_i32": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "eend_retention_proj_f16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "eend_retention_chunk_f16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp],
    "eend_retention_stream_elems": [],
    "eend_retention_stream_ok": [_i, _i, _i, _i],
    "eend_retention_stream_pack_f16": [_vp, _vp, _vp],
    "eend_retention_stream_f16": [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp],
    "eend_attn_decode_f16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "eend_attn_decode_dev_f16": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _vp],
    "eend_counter_add_i32": [_vp, _i, _vp],
    "eend_attn_decode_split_f16": [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp, _f, _vp],
    "eend_retention_step_f16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "eend_retention_proj_step_f32": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp],
    "eend_retention_step_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "eend_linear_step_f32": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "eend_linear_res_ln_step_f32": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_spk_attn_step_f32": [_vp, _vp, _i, _i, _f, _vp],
    "eend_linear_res_scale_ln_step_f32": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _vp],
    "eend_layernorm_rows_f32": [_vp, _vp, _vp, _f, _vp, _i, _vp],
    "eend_l2norm_rows_f32": [_vp, _vp, _i, _vp],
    "eend_convert_fanout_step_f32": [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "eend_dwconv_step_f16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp],
    "eend_layernorm_f16": [_vp, _vp, _vp, _f, _vp, _i, _i, _vp],
    "eend_dwconv_bn_swish_f16": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp, _vp],
    "eend_linear_res_scale_f16": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_conv1d_l2norm_f16": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "eend_convert_fanout_f16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "eend_attn_causal_bf16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "eend_spk_attn_f16": [_vp, _vp, _i, _i, _i, _i, _f, _vp],
    "eend_head_l2dot_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "eend_head_l2dot_a16_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    # ---- training step
    "eend_linear_res_ln_train_f16": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp],
    "eend_linear_relu_train_f16": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "eend_ffn_swish_train_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "eend_ffn_bwd_data_bf16": [_vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_ffn_train_f16": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "eend_ffn_train_stream_elems": [_i],
    "eend_ffn_train_stream_ok": [_i, _i, _i],
    "eend_ffn_train_stream_pack": [_vp, _vp, _vp, _i, _vp],
    "eend_ffn_train_stream_f16": [_vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "eend_ffn_bwd_data_stream_bf16": [_vp, _i, _vp, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_spk_attn_train_f16": [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp],
    "eend_conv1d_l2norm_train_f16": [_vp] * 7 + [_i] * 5 + [_vp],
    "eend_inproj_heads_train_bf16": [_vp, _i, _vp, _vp] + [_vp] * 6 + [_i, _i, _i, _vp],
    "eend_attn_causal_lse_bf16": [_vp] * 5 + [_i] * 6 + [_f, _vp, _vp],
    "eend_attn_causal_bwd_bf16": [_vp] * 5 + [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i] + [_i] * 6 + [_f, _f, _f, _vp, _vp],
    "eend_gemm_bf16": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "eend_gemm_relu_bwd_bf16": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp],
    "eend_wgrad_bias_grouped_bf16": [_vp, _i, _vp, _i, _i, _l, _i, _i, _vp, _l, _vp, _vp, _i, _l, _f, _vp],
    "eend_gemm_acc_lnbwd_bf16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _i, _i, _vp, _vp],
    "eend_gemm_acc_bf16": [_vp, _i, _vp, _i, _vp, _f, _vp, _vp, _i, _i, _vp],
    "eend_conv1d_dgrad_bf16": [_vp] * 5 + [_i] * 5 + [_vp],
    "eend_wgrad_bf16": [_vp, _i, _vp, _i, _i, _l, _i, _i, _vp, _l, _vp, _i, _i, _f, _i, _vp],
    "eend_wgrad_bias_bf16": [_vp, _i, _vp, _i, _i, _l, _i, _i, _vp, _l, _vp, _i, _i, _vp, _f, _i, _vp],
    "eend_conv1d_wgrad_bf16": [_vp] * 3 + [_i] * 5 + [_vp, _l, _vp, _vp, _vp],
    "eend_colsum_f32": [_vp, _i, _l, _i, _i, _vp, _l, _vp, _f, _i, _vp],
    "eend_layernorm_bwd_f32": [_vp] * 6 + [_vp, _l, _vp, _vp, _vp, _l, _vp, _vp],
    "eend_head_bce_f32": [_vp] * 5 + [_f] + [_vp] * 4 + [_vp, _l, _vp] + [_i] * 4 + [_vp],
    "eend_l2norm_bwd_bf16": [_vp] * 4 + [_i] * 3 + [_vp],
    "eend_convert_fanout_bwd_f32": [_vp, _vp, _vp, _l, _vp, _i, _i, _i, _vp],
    "eend_convert_const_f32": [_i] + [_vp] * 7 + [_i, _vp],
    "eend_spk_attn_bwd_bf16": [_vp] * 3 + [_i] * 4 + [_f, _vp, _vp],
    "eend_bn_train_stats_f32": [_vp, _vp, _f, _vp, _l] + [_vp] * 4 + [_f, _i, _i, _i, _vp],
    "eend_bn_bwd_f32": [_vp, _vp, _f, _vp, _vp, _f, _vp, _i, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp],
    "eend_emb_consistency_bwd_f16": [_vp, _vp, _vp, _f, _vp] + [_i] * 5 + [_vp],
    "eend_grad_sumsq_f32": [_vp, _l, _vp, _l, _vp, _vp],
    "eend_adam_step_f32": [_vp] * 4 + [_l, _vp, _vp, _f, _f, _f, _vp],
    "eend_prep_weights": [_vp, _i, _vp],
    "eend_grad_accumulate_f32": [_vp, _vp, _f, _i, _l, _vp],
    # ---- LS-EEND training step
    "eend_swish_dropout_f16": [_vp, _vp, _l, _i, _vp, _vp],
    "eend_swish_bwd_bf16": [_vp, _vp, _l, _i, _vp, _vp],
    "eend_layernorm_train_f16": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _l, _vp],
    "eend_layernorm_bwd2_f32": [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, _l, _vp, _vp, _vp, _l, _vp, _vp],
    "eend_resgrad_cast_bf16": [_vp, _vp, _f, _vp, _l, _vp, _l, _vp, _vp],
    "eend_linear_res_scale_ln_train_f16": [_vp, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp],
    "eend_glu_dwconv_f16": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "eend_bn_batch_stats_f16": [_vp, _vp, _l, _vp, _i, _i, _i, _vp],
    "eend_bn_merge_f32": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp],
    "eend_bn_swish_f16": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _l, _vp],
    "eend_bn_swish_bwd_stats_bf16": [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _vp],
    "eend_bn_swish_bwd_apply_bf16": [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "eend_dwconv_glu_bwd_bf16": [_vp, _vp, _vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _vp],
    "eend_retention_chunk_train_f16": [_vp] * 12 + [_i] * 6 + [_f, _i, _vp],
    "eend_retention_bwd_bf16": [_vp] * 8 + [_i, _vp, _vp] + [_vp] * 6 + [_i] * 6 + [_f, _vp],
}


class PrepEntry(ctypes.Structure):
    """eend_prep_entry of include/eend_hip.h."""
    _fields_ = [("src", ctypes.c_void_p), ("off", ctypes.c_long), ("dst", ctypes.c_void_p),
                ("A", ctypes.c_int), ("B", ctypes.c_int), ("C", ctypes.c_int), ("Cpad", ctypes.c_int),
                ("sa", ctypes.c_long), ("sb", ctypes.c_long), ("sc", ctypes.c_long),
                ("dtype", ctypes.c_int), ("nscale", ctypes.c_int), ("scale", ctypes.c_float), ("reserved", ctypes.c_int)]


class ProjGroup(ctypes.Structure):
    """eend_proj_group of include/eend_hip.h."""
    _fields_ = [("rows", ctypes.c_void_p), ("rows_kind", ctypes.c_int), ("rows_bf16", ctypes.c_int), ("rows_ld", ctypes.c_int),
                ("rows2_bf16_heads", ctypes.c_void_p), ("heads_t", ctypes.c_void_p), ("heads_t_bf16", ctypes.c_int)]


class Dropout(ctypes.Structure):
    """eend_dropout of include/eend_hip.h."""
    _fields_ = [("seed", ctypes.c_uint), ("thresh24", ctypes.c_uint), ("scale", ctypes.c_float)]


_lib = None


class EendHipError(RuntimeError):
    pass


def load():
    """Load the shared object (once) and type every exported entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EendHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EendHipError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    got = lib.eend_abi_version()
    if got != ABI_VERSION and not os.environ.get("EEND_HIP_LIB"):     # (an A/B study may load an older build on purpose)
        raise EendHipError(f"{LIB_PATH} has ABI version {got}, this binding was written for {ABI_VERSION} (include/eend_hip.h)")
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        raise EendHipError(f"{name} failed with code {rc} "
                           f"({'invalid argument' if rc == -1 else 'HIP launch failure' if rc == -2 else 'unknown'})")
