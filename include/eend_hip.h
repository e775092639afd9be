/*
 * eend_hip.h -- C-ABI of libeend_hip.so: the MI355X (gfx950) kernels behind the
 * FS-EEND / LS-EEND frame-wise diarization forward.
 *
 * The reference (Audio-WestlakeU/FS-EEND) is pure Python on torch.nn and has no FFI of
 * its own; each entry point below replaces the stock ATen kernels that one group of
 * reference lines dispatches, and cites those lines (paths relative to the reference
 * root).  The Python host (fs-eend_amd/) binds them with ctypes; INTEGRATION.md shows
 * the stub.
 *
 * Conventions (every entry point):
 *   - returns 0 on success, EEND_EINVAL (-1) for a rejected argument, EEND_ELAUNCH (-2)
 *     when the HIP launch failed; never throws, never allocates, never synchronises;
 *   - all pointers are DEVICE pointers owned by the caller and must stay alive until the
 *     stream has executed the call; `stream` is a hipStream_t passed as void*;
 *   - kernels are stateless and stream ordered (safe under hipGraph capture);
 *   - "slab" layout: activations of `nseq` sequences are stored [nseq][Tp][D] with
 *     Tp % 64 == 0 >= T (frames t >= T are padding the caller never reads);
 *   - f16 = IEEE binary16, bf16 = bfloat16, f32 = IEEE binary32; accumulation is f32.
 */
#ifndef EEND_HIP_H
#define EEND_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define EEND_OK 0
#define EEND_EINVAL (-1)
#define EEND_ELAUNCH (-2)

/* ABI version of this header (bumped on any signature change).
 *   4 (round 6): eend_attnout_ffn_fused_f16 gained out_lo_f16 / Wo_lo and eend_convert_fanout_f32 gained out_lo_f16 (round 5, shipped
 *      under 3 by mistake); a caller built against the version-3 header must refuse this library. */
/*   5 (round 6): the dropout mask function of eend_dropout changed (two 24-bit-multiply rounds instead of the murmur3 finaliser): a caller
 *      that regenerates masks itself (as oracle/dropout_ref.py does) must follow; eend_inproj_heads_train_bf16 accepts NULL Qt / Kt;
 *      x_is_f16 of eend_wgrad[_bias]_bf16 became a flag word (bits 1, 2: blocked operands); new eend_ffn_train_stream_* entries. */
#define EEND_ABI_VERSION 5
int eend_abi_version(void);

/* Eval-mode BatchNorm1d over features + cast + zero pad to the frame slab.
 * Replaces FS-EEND/nnet/model/onl_tfm_enc_1dcnn_enc_linear_non_autoreg_pos_enc_l2norm.py:165-166
 * (pad_sequence(-1) is done by the host; BN uses running stats) and, with apply_bn = 0,
 * the plain cast in front of LS-EEND's input projection (LS-EEND/nnet/conformer/encoder.py:195).
 * x f32 [B][T][Fin] -> out f16 [B][Tp][Fpad] (Fpad % 64 == 0, zero filled beyond Fin / T). */
int eend_bn_cast_pad_f16(const float* x, const float* bn_weight, const float* bn_bias,
                         const float* bn_mean, const float* bn_var, float eps, void* out_f16,
                         int B, int T, int Tp, int Fin, int Fpad, int apply_bn, void* stream);

/* The same, reading the B utterances through a device table of pointers (x_ptrs[b] -> f32
 * [lens[b]][Fin]) and using `pad_value` for frames lens[b] <= t < T: pad_sequence (FS model :165 pads
 * with -1, LS model :280 with 0) + BatchNorm + cast + slab padding in ONE launch. */
int eend_gather_bn_cast_pad_f16(const void* const* x_ptrs, const int* lens, float pad_value,
                                const float* bn_weight, const float* bn_bias, const float* bn_mean,
                                const float* bn_var, float eps, void* out_f16, int B, int T, int Tp, int Fin,
                                int Fpad, int apply_bn, void* stream);

/* Encoder input in ONE launch (encin.hip): pad_sequence(pad_value) + BatchNorm(eval) + cast, the input projection
 * (in_size -> 256) and its LayerNorm, i.e. eend_gather_bn_cast_pad_f16 followed by eend_linear_res_ln_f16 without a residual
 * (FS model :162-170: pad_sequence(-1), self.bn, enc.encoder, enc.encoder_norm) -- the f32 features are read once, the f16 copy
 * never exists.  x_ptrs: device table of B pointers to (len_b, Fin) f32 rows (16-byte aligned), lens[b] frames valid (frames
 * len_b <= t < T take pad_value through the BatchNorm, frames T <= t < Tp are zero before the projection, exactly as the pair);
 * W f16 [256][ldw] with columns >= Fin zero; out_f16 [B*Tp][256], out_f32 optional.  Supported where
 * eend_encoder_input_ok(Fin, Tp, ldw) != 0 (320 < Fin <= 384, Tp a multiple of 32); EEND_EINVAL otherwise. */
int eend_encoder_input_ok(int Fin, int Tp, int ldw);
int eend_encoder_input_f16(const float* const* x_ptrs, const int* lens, float pad_value, const float* bn_w, const float* bn_b,
                           const float* bn_mean, const float* bn_var, float bn_eps, const void* W_f16, int ldw, const float* bias,
                           const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, int B, int T, int Tp,
                           int Fin, void* stream);

/* out = act(A W^T + bias), f16 in / f16 out, f32 accumulate; act: 0 none, 1 ReLU, 2 Swish.
 * torch.nn.Linear call sites: FFN linear1+ReLU of nn.TransformerEncoderLayer (FS model :147) and of
 * the fusion layer (FS-EEND/nnet/modules/merge_tfm_encoder.py:397-399), packed in-proj of the speaker
 * MHA (merge_tfm_encoder.py:388-394), Linear+Swish of LS-EEND's FeedForwardModule
 * (LS-EEND/nnet/conformer/feed_forward.py:49-50).  A [M][lda], W [N][ldw], N % 128 == 0, K % 64 == 0. */
int eend_linear_f16(const void* A, int lda, const void* W, int ldw, const float* bias, void* out_f16,
                    int ldo, int M, int N, int K, int act, void* stream);

/* Pointwise Conv1d(D -> 2D) + GLU over channels (LS-EEND/nnet/conformer/convolution.py:141-142,
 * activation.py:39-41).  Wi / bias_i have the value and gate rows interleaved
 * (row 2n = W[n], row 2n+1 = W[n + N2/2]); out[m][n] = v_n * sigmoid(g_n), f16 [M][ldo]. */
int eend_linear_glu_f16(const void* A, int lda, const void* Wi, int ldw, const float* bias_i, void* out_f16,
                        int ldo, int M, int N2, int K, void* stream);

/* Packed MHA in-projection (3D x D) whose epilogue scatters Q, K (bf16 [nseq][H][Tp][dh]) and
 * V transposed (bf16 [nseq][H][dh][Tp]) for eend_attn_causal_bf16.  Replaces the in-proj half of
 * nn.MultiheadAttention in nn.TransformerEncoderLayer (FS model :147) and _sa_block1
 * (merge_tfm_encoder.py:379-385).  A f16 [nseq*Tp][lda], W f16 [3*H*dh][ldw], bias f32 [3*H*dh]. */
int eend_inproj_heads_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* Q_bf16,
                           void* K_bf16, void* Vt_bf16, int nseq, int Tp, int H, int dh, int K, void* stream);

/* out = LayerNorm((A W^T + bias) * alpha + res) * gamma + beta, N = 256; writes f32 (residual
 * stream) and f16 (next MFMA operand) copies; res / out32 / out16 / gamma may be null.  Replaces
 * Linear -> LayerNorm (FS model :173-174), out-proj + residual + norm1 / norm11 / norm21 and
 * linear2 + residual + norm2 / norm22 (torch TransformerEncoderLayer; merge_tfm_encoder.py:364,373-374),
 * and with alpha = 0.5 the half-step FFN + block-final LayerNorm of the Conformer block
 * (LS-EEND/nnet/conformer/encoder.py:104-110). */
int eend_linear_res_ln_f16(const void* A, int lda, const void* W, int ldw, const float* bias,
                           const float* res, float alpha, const float* gamma, const float* beta, float eps,
                           float* out_f32, void* out_f16, int M, int K, void* stream);

/* y = (A W^T + bias) * alpha + res -> out_f32 (residual stream, not normalised) and
 * LayerNorm(y) * gamma + beta -> out_f16: the pre-norm input of the NEXT Conformer sub-module
 * (feed_forward.py:48, attention.py:100, convolution.py:139) fused into its producer.  N = 256. */
int eend_linear_res_scale_ln16_f16(const void* A, int lda, const void* W, int ldw, const float* bias,
                                   const float* res, float alpha, const float* gamma, const float* beta,
                                   float eps, float* out_f32, void* out_f16, int M, int K, void* stream);

/* out = (A W^T + bias) * alpha + res, N = 256, f32 + f16 copies (no normalisation).  Replaces the
 * LS-EEND residual wrappers: ResidualConnectionModule with module_factor 0.5 / 1
 * (LS-EEND/nnet/conformer/modules.py:32-33) around the second Linear of FeedForwardModule
 * (feed_forward.py:47-57), the retention out_proj (LS-EEND/nnet/modules/retention.py:226) and the
 * last pointwise conv of ConformerConvModule (convolution.py:138-149). */
int eend_linear_res_scale_f16(const void* A, int lda, const void* W, int ldw, const float* bias,
                              const float* res, float alpha, float* out_f32, void* out_f16, int M, int K,
                              void* stream);

/* Look-ahead Conv1d(cin -> 256, ktaps, padding = pad) as an implicit GEMM, + bias, then
 * x / ||x||_2 (no eps).  Frames >= ilens[seq] read as zero (the reference truncates to ilen and
 * zero re-pads before the conv).  Replaces FS model :38-41 / LS model :80-87.
 * X f16 [nseq][Tp][cin]; Wr f16 [256][ktaps*cin] with Wr[o][tap*cin + i] = conv.weight[o][i][tap]. */
int eend_conv1d_l2norm_f16(const void* X, const void* Wr, const float* bias, const int* ilens,
                           float* out_f32, void* out_f16, int nseq, int Tp, int cin, int ktaps, int pad,
                           void* stream);

/* The same operator on a packed weight stream (conv_stream.hip; cin = 256, ktaps <= 24): the tile's input rows are staged once in
 * LDS and every tap reads them shifted, the weights -- re-ordered once per parameter version by eend_conv_stream_pack_f16 from the
 * same Wr [256][ktaps*256] into eend_conv_stream_elems(ktaps) f16 elements -- flow through an LDS-DMA ring.  EEND_EINVAL where
 * eend_conv_stream_ok(cin, ktaps, pad) == 0 (the caller keeps eend_conv1d_l2norm_f16). */
int eend_conv_stream_elems(int ktaps);
int eend_conv_stream_ok(int cin, int ktaps, int pad);
int eend_conv_stream_pack_f16(const void* Wr, void* stream_out, int ktaps, void* stream);
int eend_conv1d_l2norm_stream_f16(const void* X, const void* wstream, const float* bias, const int* ilens, float* out_f32,
                                  void* out_f16, int nseq, int Tp, int ktaps, int pad, void* stream);

/* attr0[(b,c)][t][:] = W1 emb[b][t][:] + pc[c][:], the factored form of
 * convert(cat(emb, pe[c])) (FS model :113-114, LS model :216-217): W1 = convert.weight[:, :D],
 * pc[c] = convert.weight[:, D:] pe[c] + convert.bias.  E f16 [B][Tp][256] ->
 * out f32/f16 [B*C][Tp][256] (decoder slab, sequence index = b*C + c). */
int eend_convert_fanout_f16(const void* E, const void* W1, const float* pc, float* out_f32, void* out_f16,
                            int B, int Tp, int C, void* stream);
/* The same stage with f32 operands on the exact-f32 MFMA: E_f32 [B][Tp][256], W_f32 = convert.weight ([256][ldw], the first 256
 * columns are W1).  The LS-EEND batch forward takes it (the decoder retention's per-head LayerNorm, eps 1e-6, amplifies the f16
 * operand rounding of this linear; with 12 speaker slots the f16 form left the 1e-3 bar).  out_f32 may be NULL.  out_lo_f16 (optional,
 * same layout as out_f16): f16(v - f16(v)), the remainder the retention's query path reads next to out_f16 (eend_retention_stream_f16). */
int eend_convert_fanout_f32(const float* E_f32, const float* W_f32, int ldw, const float* pc, float* out_f32, void* out_f16,
                            void* out_lo_f16, int B, int Tp, int C, void* stream);

/* Fused causal MHA core: softmax(mask(Q K^T / sqrt(dh))) V with
 * allowed(i,j) <=> j - i <= mask_delay && j < kv_len evaluated on indices (the (T,T) {0,-inf} tensor
 * of FS model :107-110,:152-155 is never built; has_mask=False is mask_delay >= Tp, kv_len = T).
 * dh = 64.  Q,K bf16 [nseq][H][Tp][64], Vt bf16 [nseq][H][64][Tp] -> O f16 [nseq*Tp][ldo]. */
int eend_attn_causal_bf16(const void* Q, const void* K, const void* Vt, void* O_f16, int nseq, int H,
                          int Tp, int ldo, int mask_delay, int kv_len, float scale, void* stream);

/* Whole position-wise feed-forward block in one launch, hidden activations never leave the CU:
 *   y = (act(X W1^T + b1) W2^T + b2) * alpha + res ;  out_f16 = LayerNorm(y) * gamma + beta ;
 *   out_f32 = LayerNorm(y)... (residual_stream_unnormalised = 0) or y (= 1).
 * act: 1 ReLU (nn.TransformerEncoderLayer linear1/linear2 + norm2, FS model :147; _ff_block + norm22,
 * FS merge_tfm_encoder.py:374,397-399, LS merge_retnet_layer.py:252,309-311), 2 Swish (Conformer
 * FeedForwardModule + half-step residual, LS conformer/feed_forward.py:47-57, encoder.py:76-110).
 * X f16 [M][ldx] (256 features), W1 f16 [F][256], W2 f16 [256][F], F % 64 == 0. */
int eend_ffn_fused_f16(const void* X, int ldx, const void* W1, const float* b1, const void* W2, const float* b2,
                       const float* res, float alpha, const float* gamma, const float* beta, float eps,
                       float* out_f32, void* out_f16, int M, int F, int act, int residual_stream_unnormalised,
                       void* stream);

/* Attention out-projection + residual + norm1 AND the position-wise feed-forward block + residual +
 * norm2 of one post-LN layer in ONE launch (ReLU FFN):
 *   x = LayerNorm1(A Wo^T + bo + res) * g1 + be1 ;  out = LayerNorm2(relu(x W1^T + b1) W2^T + b2 + x) * g2 + be2
 * i.e. nn.TransformerEncoderLayer's out_proj/dropout1/norm1 + linear1/linear2/norm2 (FS model :147) and
 * the second half of the fusion layers (out_proj of self_attn2 + norm21, _ff_block + norm22: FS
 * merge_tfm_encoder.py:371-376,397-399; LS merge_retnet_layer.py:248-253,309-311).  x never leaves the CU.
 * A f16 [M][lda] (attention output), Wo f16 [256][256], res f32 [M][256] (stream before the attention
 * sub-layer; may alias out_f32), out_f32 f32 [M][256], out_f16 f16 [M][256] (may alias A); out_lo_f16 (optional, f16 [M][256]):
 * f16(out - f16(out)), the remainder next to out_f16 (the next layer's retention reads both: eend_retention_stream_f16).  Wo_lo (optional,
 * f16 [256][256]): f16(Wo_f32 - f16(Wo_f32)) -- the out-projection then runs as two products on the same A fragments (the weight's f16
 * rounding was the largest single term of the worst LS-EEND logit at 12 speaker slots; +3 % of the launch). */
int eend_attnout_ffn_fused_f16(const void* A, int lda, const void* Wo, const float* bo, const float* res,
                               const float* g1, const float* be1, float eps1, const void* W1, const float* b1,
                               const void* W2, const float* b2, const float* g2, const float* be2, float eps2,
                               float* out_f32, void* out_f16, void* out_lo_f16, const void* Wo_lo, int M, int F, void* stream);
/* The two post-norm joins above with the residual taken from the f16 stream: in a post-norm stack (nn.TransformerEncoderLayer,
 * merge_tfm_encoder.py:356-376) the residual IS the previous LayerNorm's output, whose f16 copy the next MFMA reads anyway, so
 * the f32 stream's write + read (1 KB per row per sub-layer, the dominant traffic of the HBM-bound out-projection GEMM) can be
 * dropped; out_f32 may be NULL (only the last layer's output is needed in f32, by the head).  Emulated on the oracle: max
 * |d logit| 2.4e-4 -> 2.6e-4. */
int eend_linear_res16_ln_f16(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res_f16,
                             float alpha, const float* gamma, const float* beta, float eps, float* out_f32,
                             void* out_f16, int M, int K, void* stream);
int eend_attnout_ffn_fused_res16_f16(const void* A, int lda, const void* Wo, const float* bo, const void* res_f16,
                                     const float* g1, const float* be1, float eps1, const void* W1, const float* b1,
                                     const void* W2, const float* b2, const float* g2, const float* be2, float eps2,
                                     float* out_f32, void* out_f16, int M, int F, void* stream);

/* Rows one launch of the packed-stream layer-tail kernels addresses (32-bit buffer offsets, one grid of tiles of prefetch ahead);
 * eend_ffn_stream_f16 / eend_attnout_ffn_stream_f16 serve larger M in several launches over row ranges of that size. */
int eend_ffn_stream_max_rows(int lda);
/* TEST HOOK (process-wide, not for production callers): force the row-tile size of the packed-stream layer-tail kernels
 * (tile_fragments = 2 / 3 for 128- / 192-row tiles, 0 = the launcher's cost model) and cap the rows of one launch
 * (max_rows_per_launch > 0; 0 = the 32-bit addressing limit), so that the tile-size agreement and the multi-launch path can be
 * exercised at small sizes (tests/test_hip_ffn_stream.py).  Replaces the EEND_FS_NJ / EEND_FFN_STREAM_MAX_ROWS environment
 * switches of rounds 4 - 5. */
int eend_debug_ffn_stream_set(int tile_fragments, long max_rows_per_launch);

/* Round 4: the same two operators (eend_ffn_fused_f16 / eend_attnout_ffn_fused[_res16]_f16; reference sites as above:
 * nn.TransformerEncoderLayer of FS model :147, merge_tfm_encoder.py:356-399, LS merge_retnet_layer.py:240-253,
 * conformer/feed_forward.py:47-57) on a PACKED WEIGHT STREAM.  eend_ffn_stream_pack_f16 re-orders Wo (optional, [256][256]),
 * W1 ([F][256]) and W2 ([256][F]) once per parameter version into the sequence of 1-KB MFMA fragments the kernel consumes
 * (eend_ffn_stream_elems(F, with_wo) f16 elements; F a multiple of 64, at most 2048).  With Wo the stream serves eend_attnout_ffn_stream_f16
 * (the contraction index of W1 follows the register layout LayerNorm1 leaves), without it eend_ffn_stream_f16 (natural order).
 * The kernel gives one wave 48 token rows end to end: the hidden activations and x stay in its registers, weights flow
 * through a 4-slot LDS-DMA ring.  Results equal the un-packed entries up to fp32 summation order. */
int eend_ffn_stream_elems(int F, int with_wo);
int eend_ffn_stream_pack_f16(const void* Wo, const void* W1, const void* W2, void* stream_out, int F, void* stream);
int eend_ffn_stream_f16(const void* X, int ldx, const void* wstream, const float* b1, const float* b2,
                        const float* res, float alpha, const float* gamma, const float* beta, float eps,
                        float* out_f32, void* out_f16, int M, int F, int act, int residual_stream_unnormalised,
                        void* stream);
/* The LO form (round 6; LS-EEND decoder layer tail, merge_retnet_layer.py:240-253 on the f32 attractor rows): the out-projection weight
 * as an f16 hi / lo pair (Wo_lo = f16(Wo - f16(Wo)): a second product on the same input fragments; eend_ffn_stream_elems(F, 2) elements)
 * and the rows leaving as f32 AND as an f16 hi / lo pair (out_lo_f16 = f16(y - f16(y)); may be NULL), as eend_attnout_ffn_fused_f16
 * with Wo_lo / out_lo_f16 does on un-packed weights. */
int eend_ffn_stream_pack_lo_f16(const void* Wo, const void* Wo_lo, const void* W1, const void* W2, void* stream_out, int F, void* stream);
int eend_attnout_ffn_stream_lo_f16(const void* A, int lda, const void* wstream, const float* bo, const float* res, const float* g1,
                                   const float* be1, float eps1, const float* b1, const float* b2, const float* g2, const float* be2,
                                   float eps2, float* out_f32, void* out_f16, void* out_lo_f16, int M, int F, void* stream);
/* res (f32) or res_f16 (f16), exactly one non-null; out_f32 may be NULL */
int eend_attnout_ffn_stream_f16(const void* A, int lda, const void* wstream, const float* bo, const float* res,
                                const void* res_f16, const float* g1, const float* be1, float eps1, const float* b1,
                                const float* b2, const float* g2, const float* be2, float eps2,
                                float* out_f32, void* out_f16, int M, int F, void* stream);

/* First half of a speaker-fusion decoder layer in one launch on a packed weight stream (spk_stream.hip):
 *   x1 = LayerNorm11(A Wo1^T + bo1 + res)    (out-projection of the time-axis attention, residual, norm11)
 *   O  = MHA over the C slots of every frame of (x1 W_in^T + b_in)      (self_attn2 of the fusion layers)
 * i.e. eend_linear_res16_ln_f16 followed by eend_spk_qkv_attn_f16 (FS merge_tfm_encoder.py:356-394; LS
 * merge_retnet_layer.py:301-306), with q, k, v kept in f32 registers.  Rows are (b*C + c)*Tp + t.  Supported where
 * eend_spk_stream_ok(C, Tp) != 0 (1 <= C <= 12, Tp a multiple of 64 / 32 / 16 for C <= 3 / 6 / 12); other shapes return EEND_EINVAL and
 * the caller uses the two-launch path.  eend_spk_stream_pack_f16 re-orders Wo1 [256][256] and W_in [768][256] (both f16) into the stream
 * (eend_spk_stream_elems() f16 elements).  x_f16 may be res_f16 and O_f16 may be A (lda == 256): rows are read before written. */
int eend_spk_stream_elems(void);
int eend_spk_stream_ok(int C, int Tp);
int eend_spk_stream_pack_f16(const void* Wo, const void* W_in, void* stream_out, void* stream);
int eend_attnout_spk_stream_f16(const void* A, int lda, const void* wstream, const float* bo, const void* res_f16,
                                const float* g1, const float* be1, float eps1, void* x_f16, const float* b_in, void* O_f16,
                                int B, int C, int Tp, float scale, void* stream);
/* ... with an f32 residual stream (round 5; LS-EEND's decoder, merge_retnet_layer.py:301-306 behind the retention's out-projection: replaces
 * eend_linear_res_ln_f16 + eend_spk_qkv_attn_f16 there): res_f32 rows in, x1 rows out as f32 (x_f32 may be res_f32), no f16 copy of x1.
 * 16-byte aligned f32 buffers; same shapes and weight stream as above. */
int eend_attnout_spk_stream_res32_f16(const void* A, int lda, const void* wstream, const float* bo, const float* res_f32,
                                      const float* g1, const float* be1, float eps1, float* x_f32, const float* b_in, void* O_f16,
                                      int B, int C, int Tp, float scale, void* stream);


/* Embedding-consistency loss (FS model :46-57; LS model :92-113): mean over (b,i,j) of
 * (cos(emb_i, emb_j) - cos(label_i, label_j))^2 with the reference's "+1e-6" denominators, without
 * materialising the (B,T,T) maps.  emb f32 [B][Tp][D] (rows >= T ignored), labels f32 [B][T][C] zero-padded
 * (C <= 16), partial_ws f32 [B * ceil(T/64)^2] scratch, out f32 [1].  lens (device int [B]) may be null; when
 * given, embeddings of frames >= lens[b] count as zero (the LS model's length mask, :100).  inv_count > 0
 * replaces the default 1/(B*T*T) normalisation (LS: 1/sum(len^2)).  Exact-fp32 MFMA; deterministic. */
int eend_emb_consistency_f32(const float* emb, const float* labels, const int* lens, float inv_count,
                             float* partial_ws, float* out, int B, int T, int Tp, int D, int C, void* stream);

/* ---- output-side post-processing (FS-EEND/train/utils/make_rttm.py:10-28, train/utils/loss.py:198-236;
 * the LS-EEND copies are identical).  Bit exact against the reference.  ---- */

/* pred f32 [T][ld] (S <= ld columns used; probabilities) -> act u8 [T][S]: (pred > threshold), then the
 * zero-padded median of `median` (odd) frames along time, i.e. scipy.signal.medfilt(x, (median, 1)). */
int eend_activity_median_u8(const float* pred, int ld, int T, int S, float threshold, int median,
                            unsigned char* act, void* stream);

/* act u8 [T][S] -> per speaker the increasing frame indices (0..T) at which the zero-padded track changes:
 * changes i32 [S][cap] (even entries = segment starts, odd = ends; entries beyond cap are dropped),
 * counts i32 [S] (number found, may exceed cap). */
int eend_activity_segments_i32(const unsigned char* act, int T, int S, int* changes, int* counts, int cap,
                               void* stream);

/* Frame-level DER counters of pre-activations pred f32 [T][ldp] against 0/1 labels f32 [T][ldl], C columns,
 * label_delay as in the reference: counters u64 [8] = speech_scored, speech_miss, speech_falarm,
 * speaker_scored, speaker_miss, speaker_falarm, speaker_error, #(label == decision).  Zeroed by the call. */
int eend_der_counters_u64(const float* pred, int ldp, const float* label, int ldl, int T, int C, int label_delay,
                          unsigned long long* counters, void* stream);

/* ---- feature front-end ({LS,FS}-EEND/datasets/feature.py: stft :166-191, transform :43-131, splice :141-163,
 * subsample :133-138, extract_fbank :324-336); fp32 throughout.  ---- */

/* STFT (Hann window of 200 samples zero-padded to n_fft = 256, hop 80) -> power -> 23 mel bands -> log10(max(.,1e-10)).
 * Frame t uses samples y[first + 80 t + k], k = 0..199 (the non-zero window taps), zero outside [0, len):
 * first = -100 reproduces librosa.stft(center=True, pad_mode="constant"); a host that pre-pads the signal
 * itself (e.g. reflect) passes the padded buffer and first = 28.  dft f32 [200][288]: windowed DFT table,
 * row k = w[k] * cos(2 pi (k+28) n / 256) in column n (n <= 128) and -w[k] * sin(..) in column 144 + n, zeros
 * elsewhere; melT f32 [132][32]: mel filterbank transposed (rows = bins, 129 used), zero padded.
 * out f32 [n_frames][23]. */
int eend_stft_logmel23_f32(const float* y, long len, long first, int n_frames, const float* dft, const float* melT,
                           float* out, void* stream);

/* Column-mean normalisation of Y f32 [T][F] into out: mode 1 = subtract the mean over all frames
 * (logmel23_mn), mode 2 = subtract the running mean of frames 0..t (logmel23_cummn).  fp64 running sums. */
int eend_feature_meannorm_f32(const float* Y, float* out, int T, int F, int mode, void* stream);

/* out f32 [ceil(T/sub)][F (2 ctx + 1)]: row j = frames j sub - ctx .. j sub + ctx of Y f32 [T][F] side by side,
 * zero outside [0, T) (splice + subsample). */
int eend_splice_subsample_f32(const float* Y, int T, int F, int ctx, int sub, float* out, void* stream);

/* ---- permutation-invariant label assignment (FS-EEND/train/utils/loss.py:257-327 batch_pit_n_speaker_loss;
 * LS-EEND/train/utils/loss.py:350-379 pit_loss_multispk) ---- */

/* cost f64 [B][C][C]: cost[b][i][j] = sum_t BCEwithLogits(y[b][t][i], labels[b][t][j]) over all T frames of
 * the padded batch (the reference pads both logits and labels with -1, pad_sequence); y, labels f32 [B][T][C],
 * C <= 16.  batch_pit's losses[b][i][s] is cost[b][i][(i+s) % C]; pit_loss_multispk's cost_mxs is cost itself. */
int eend_pit_cost_f64(const float* y, const float* labels, int B, int T, int C, double* cost, void* stream);

/* Optimal assignment on the leading nspk[b] x nspk[b] block of every cost matrix (shortest augmenting paths,
 * fp64; the remaining slots keep their place): perm i32 [B][C] = label column assigned to prediction i,
 * loss f64 [B] = mean over the C slots of the assigned costs (batch_pit's per-utterance minimum). */
int eend_pit_assign_i32(const double* cost, const int* nspk, int B, int C, int* perm, double* loss, void* stream);

/* q/k/v/g projections of MultiScaleRetention (LS-EEND/nnet/modules/retention.py:200-207) in the
 * layouts eend_retention_chunk_f16 consumes.  Wqkvg f16 [4*H*dh][ldw] = rows of q_proj, k_proj * dk^-0.5,
 * v_proj, g_proj (bias likewise); Q,K f16 [nseq][H][Tp][dh]; Kt,Vt f16 [nseq][H][dh][Tp]; G f16 [M][H*dh]. */
int eend_retention_proj_f16(const void* A, int lda, const void* Wqkvg, int ldw, const float* bias, void* Q,
                            void* K, void* Kt, void* Vt, void* G, int nseq, int Tp, int H, int dh, int Kdim,
                            void* stream);

/* Chunk-recurrent retention with decay 1 (retention.py:146-194 + RetNetRelPos :30-47), the per-head
 * LayerNorm (eps gn_eps, no affine, :222) and the swish gate (:224) in one pass:
 * O = swish(G) * LN_head(retention(Q,K,V)).  L = recurrent_chunk_size; workspaces: St_ws f16
 * [nseq][H][nc][2][64][64], kv_ws f32 [nseq][H][nc][64][64], cscale_ws / sexp_ws f32 [nseq][H][nc],
 * nc = ceil(Tp / L).  Three launches: chunk-parallel K_c^T V_c, per-(seq,head) prefix scan, core.
 * T_valid (0 = Tp): frames at or beyond it are slab padding (Tp rounds the chunk-padded length up to 64);
 * chunks that start there are skipped and their rows of O are left untouched.
 * state_in / state_out (optional, f32 [nseq][H][64][64], the reference's `prev_key_value` before its scaling,
 * retention.py:176-180): the chunk state before the first / after the last chunk of this call, so a long recording
 * can be processed a few chunks at a time with the state carried across calls (BASELINE config 5). */
int eend_retention_chunk_f16(const void* Q, const void* K, const void* Kt, const void* Vt, const void* G,
                             void* O_f16, void* St_ws, float* kv_ws, float* cscale_ws, float* sexp_ws, int nseq,
                             int H, int Tp, int L, int ldo, int ldg, float gn_eps, int T_valid,
                             const float* state_in, float* state_out, void* stream);

/* The same operator with its four projections fused on chip (ret_stream.hip, round 5): replaces eend_retention_proj_f16 +
 * eend_retention_chunk_f16 (`MultiScaleRetention.forward`, LS-EEND/nnet/modules/retention.py:196-228, called from
 * conformer/attention.py and merge_retnet_layer.py:233-253): q / k / k^T / v^T / g never exist in HBM.  X f16 [nseq*Tp][ldx]: the
 * retention's input rows; Xlo (optional, same layout): f16(x - f16(x)) of the f32 stream the rows were rounded from -- with it the
 * QUERY projection carries ~22 significand bits (three f16 MFMA products); the query is a hi/lo f16 pair in the score and cross-chunk
 * products either way.  W_packed: eend_retention_stream_pack_f16 of the f32 [q; k * dk^-0.5; v; g] rows ([1024][256]) into
 * eend_retention_stream_elems() f16 elements (hi and lo parts of the q rows), once per parameter version; bias f32 [1024] likewise.
 * H = 4, dh = 64, L <= 512 (eend_retention_stream_ok; the caller keeps the two-call form otherwise).  Workspaces, T_valid,
 * state_in / state_out: as eend_retention_chunk_f16.  Three launches: chunk K^T V products (projecting K, V on the fly), prefix scan,
 * rows. */
int eend_retention_stream_elems(void);
int eend_retention_stream_ok(int L, int Tp, int ldx, int ldo);
int eend_retention_stream_pack_f16(const float* Wqkvg_f32, void* packed_out, void* stream);
int eend_retention_stream_f16(const void* X_f16, int ldx, const void* Xlo_f16, const void* W_packed, const float* bias, void* O_f16, int ldo,
                              void* St_ws, float* kv_ws, float* cscale_ws, float* sexp_ws, int nseq, int Tp, int L, float gn_eps,
                              int T_valid, const float* state_in, float* state_out, void* stream);

/* Stand-alone LayerNorm f32 [M][D] -> f16 (D <= 1024): second of two back-to-back LayerNorms
 * (conformer/encoder.py:110 then feed_forward.py:48; encoder.py:196 then feed_forward.py:48). */
int eend_layernorm_f16(const float* x, const float* gamma, const float* beta, float eps, void* out_f16, int M,
                       int D, void* stream);

/* Causal depthwise Conv1d (k taps, left context k-1, no bias) -> BatchNorm1d(eval) -> Swish
 * (conformer/convolution.py:65-68,143-147).  x,out f16 [nseq][Tp][D]; w f32 [D][k].  halo_f16 (optional,
 * [nseq][k-1][D]): the k-1 input frames preceding the slab (carried from the previous call); null = zeros. */
int eend_dwconv_bn_swish_f16(const void* x_f16, const float* w, const float* bn_weight, const float* bn_bias,
                             const float* bn_mean, const float* bn_var, float eps, void* out_f16, int nseq,
                             int Tp, int D, int k, const void* halo_f16, void* stream);

/* Incremental self-attention of FS-EEND streaming (FS-EEND/nnet/modules/streaming_tfm.py:15-37,
 * used by StreamingTransformerEncoderLayer :61-66 and StreamingAttractorDecoderLayer :197-201): the
 * packed in-proj of ONE new token per sequence (qkv f16 [N][3*H*64]) is appended to the projected
 * K/V caches (f16 [N][H][cap][64], `t` tokens already present, t < cap) and attends over all t+1
 * tokens.  out f16 [N][H*64]. */
int eend_attn_decode_f16(const void* qkv, void* K_cache, void* V_cache, void* out_f16, int N, int H, int cap,
                         int t, float scale, void* stream);
/* eend_attn_decode_dev_f16 for long histories (BASELINE config 5: t up to 36 000): the key axis is split over
 * cap/512 workgroups per (n, h) whose (max, sum, o) partials a second kernel merges with the new token -- the per-frame
 * cost follows HBM bandwidth instead of one wave's latency chain.  ws: f32 scratch of N*H*ceil(cap/512)*66 floats.
 * Same results as the single-wave kernel up to fp32 summation order. */
int eend_attn_decode_split_f16(const void* qkv, void* K_cache, void* V_cache, void* out_f16, float* ws, long ws_floats, int N,
                               int H, int cap, const int* t_dev, float scale, void* stream);

/* The same with the token count read from device memory (*t_dev), so that a captured hipGraph of the frame step
 * (FS-EEND/streaming_infer_dia.py's per-frame loop) stays valid while the history grows; *t_dev >= cap makes the
 * launch a no-op.  eend_counter_add_i32: *counter += inc on the stream (the graph's own "t += 1"). */
int eend_attn_decode_dev_f16(const void* qkv, void* K_cache, void* V_cache, void* out_f16, int N, int H, int cap,
                             const int* t_dev, float scale, void* stream);
int eend_counter_add_i32(int* counter, int inc, void* stream);

/* One frame of MultiScaleRetention.recurrent_forward (retention.py:126-144, decay 1) + per-head
 * LayerNorm + swish gate, state updated in place.  qkvg f16 [N][4*H*64] = [q | k*dk^-0.5 | v | g];
 * kv_state f32 [N][H][64][64] in the reference's incremental_state["prev_key_value"] layout;
 * scale_in / scale_out f32 [H] (incremental_state["scale"]; zero state + scale 0 before frame 0). */
int eend_retention_step_f16(const void* qkvg, float* kv_state, const float* scale_in, float* scale_out,
                            void* out_f16, int N, int H, float gn_eps, void* stream);
/* Frame-by-frame sessions: the retention projections of one frame in full f32 -- qkvg_f32 [N][1024] =
 * LayerNorm(x)[N][256] . Wqkvg^T + bias with the packed f32 weight [q; k*dk^-0.5; v; g] (ln_gamma == NULL: no
 * LayerNorm, the decoder's post-norm stream) -- and the recurrent step on those f32 projections.  The
 * recurrence amplifies operand rounding with the stream position (retention.py:126-144 normalises a state that grows like
 * sqrt(t)); f16 projections cost > 1e-3 on the logits late in a one-hour stream, f32 ones keep the reference's fp32
 * streaming within the 1e-3 bar (tests/test_long_horizon.py). */
int eend_retention_proj_step_f32(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* Wqkvg,
                                 const float* bias, float* qkvg_f32, int N, void* stream);
int eend_retention_step_f32(const float* qkvg, float* kv_state, const float* scale_in, float* scale_out, void* out_f16, float* out_f32,
                            int N, int H, float gn_eps, void* stream);      /* out_f16 and / or out_f32 [N][256] */
/* The remaining pieces of an all-f32 LS decoder frame step (merge_retnet_layer.py:255-276 with <= 16 rows = one frame x
 * max_nspks slots): y = act(A W^T + bias) with f32 activations AND f32 weights in torch's nn.Linear layout (act 0 none /
 * 1 relu / 2 swish; K % 8 == 0, M <= 16); the post-norm join out = LayerNorm((A W^T + bias) * alpha + res) (N = 256; out_f16
 * optional copy); the speaker-axis attention of the frame on f32 [B*C][768] = [q | k | v] rows (C <= 16).  Why f32: DESIGN
 * 9a -- the decoder retention's per-head LayerNorm amplifies f16 operand rounding ~30x on isolated frames. */
int eend_linear_step_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* out_f32, int ldo, int M, int N,
                         int K, int act, void* stream);
int eend_linear_res_ln_step_f32(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, float alpha,
                                const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, int M, int K,
                                void* stream);
int eend_spk_attn_step_f32(const float* qkv, float* out_f32, int B, int C, float scale, void* stream);
/* ... and of the Conformer encoder's f32 half-step FFNs (feed_forward.py:47-57 inside the pre-norm blocks of
 * conformer/encoder.py:76-113): the pre-norm join out_f32 = (A W^T + bias) * alpha + res (the un-normalised stream) with the NEXT
 * sub-layer's LayerNorm of it as ln_out_f32 and / or ln_out_f16; and a plain row LayerNorm f32 -> f32 (256 features). */
int eend_linear_res_scale_ln_step_f32(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, float alpha,
                                      const float* gamma, const float* beta, float eps, float* out_f32, float* ln_out_f32,
                                      void* ln_out_f16, int M, int K, void* stream);
int eend_layernorm_rows_f32(const float* x, const float* gamma, const float* beta, float eps, float* out_f32, int M, void* stream);
/* y[r] = x[r] / ||x[r]||_2, f32 rows of 256 features: the embedding normalisation (LS model :87, no eps) behind the f32
 * look-ahead conv of a frame step (the conv itself is eend_linear_step_f32 over the flattened 19-frame window). */
int eend_l2norm_rows_f32(const float* x, float* y, int rows, void* stream);
/* Frame-by-frame decoder input in f32: out[b*C + c] = W[:, :256] emb[b] + pc[c] (`convert(cat(emb, pe))`, LS model
 * :229-233; pc from eend_convert_const_f32).  W_f32 is the convert.weight parameter itself ([256][ldw], ldw = 512).
 * f32 for the same reason as the projections above: the decoder retention amplifies the f16 rounding of this linear ~30x
 * at some frames of a long stream.  out_f32 / out_f16 [B*C][256]; C <= 64. */
int eend_convert_fanout_step_f32(const float* emb_f32, const float* W_f32, int ldw, const float* pc, float* out_f32, void* out_f16,
                                 int B, int C, void* stream);

/* One frame of the causal depthwise conv + BatchNorm(eval) + Swish of ConformerConvModule.
 * forward_one_step (conformer/convolution.py:157-163); cache f32 [B][D][k-1] (the driver's
 * conv_caches layout, streaming_infer_dia.py:42-45) shifted in place.  x,out f16 [B][D]. */
int eend_dwconv_step_f16(const void* x_f16, float* cache, const float* w, const float* bn_weight,
                         const float* bn_bias, const float* bn_mean, const float* bn_var, float eps,
                         void* out_f16, int B, int D, int k, void* stream);

/* Unmasked MHA core over the C (<= 12) attractor slots of each frame (_sa_block2,
 * merge_tfm_encoder.py:388-394; LS-EEND/nnet/modules/merge_retnet_layer.py:301-306).
 * qkv f16 [B*C*Tp][768] (row = (b*C + c)*Tp + t) -> O f16 [B*C*Tp][256].  H = 4, dh = 64. */
int eend_spk_attn_f16(const void* qkv, void* O_f16, int B, int C, int Tp, int H, float scale,
                      void* stream);


/* Packed in-projection + causal multi-head attention in one launch (nn.MultiheadAttention(x, x, x) on the time axis:
 * nn.TransformerEncoderLayer.self_attn, FS model :147; self_attn1 of the fusion layers, merge_tfm_encoder.py:379-385),
 * Tp = 64 m <= 512 (round 6; rounds 4 - 5: 512 only), H = 4, d_model = 256 (attn_stream.hip): the in-projection weights pre-packed in MFMA fragment order, a wave keeps
 * the X rows of the 64 tokens whose queries it runs in registers, the head's 96 KB of weights arrive by LDS-DMA, Q, K and V never
 * reach HBM.  eend_inproj_attn_pack_f16 re-orders W_in f16 [768][256] (= in_proj_weight with the q rows pre-multiplied by
 * 1/sqrt(64) * log2(e)) into eend_inproj_attn_packed_elems() f16 elements, once per parameter version.  mask: key j visible to
 * query i iff j - i <= mask_delay and j < kv_len.  Equivalent to eend_inproj_heads_bf16 followed by
 * eend_attn_causal_bf16(scale = ln 2), which is what the caller uses for other chunk lengths (EEND_EINVAL here).  The key bias
 * is not applied (it cancels in the softmax); b_in is the [768] in_proj_bias, q part pre-scaled. */
int eend_inproj_attn_packed_elems(void);
int eend_inproj_attn_pack_f16(const void* W_in, void* packed_out, void* stream);
int eend_inproj_attn_causal_packed_f16(const void* X_f16, int ldx, const void* W_packed, const float* b_in, void* O_f16,
                                       int nseq, int H, int Tp, int ldo, int mask_delay, int kv_len, void* stream);

/* The same operator for windows of more than 512 frames (round 6; Tp = 64 m, 1 <= kv_len <= Tp, mask_delay >= 0; whole recordings at
 * test time, FS model :147 on an un-chunked sequence): the frames are cut into groups of 512 and the launch runs one item of the kernel
 * above per (sequence, head, query group, key group the mask reaches).  The item of a group with itself writes its rows of O; an item
 * of two different groups projects Q from the query group's rows and K / V from the key group's, and leaves normalised partial rows in
 * part_f16; every item leaves the log2 of its softmax denominators in lse_f32, and a second launch weighs the partial rows of a query
 * group with them (O = sum_i 2^(lse_i - lse) O_i).  Scratch sizes from eend_inproj_attn_long_scratch_elems (EEND_EINVAL for a shape
 * this form does not cover: Tp <= 512 -- use the entry above --, more than 64 items or 9 key groups per query group; the caller
 * keeps eend_inproj_heads_bf16 + eend_attn_causal_bf16 there).  Same weights, bias and mask convention as the entry above. */
int eend_inproj_attn_long_scratch_elems(int nseq, int Tp, int mask_delay, int kv_len, long long* part_f16_elems, long long* lse_f32_elems);
int eend_inproj_attn_causal_long_f16(const void* X_f16, int ldx, const void* W_packed, const float* b_in, void* O_f16, void* part_f16,
                                     float* lse_f32, int nseq, int H, int Tp, int ldo, int mask_delay, int kv_len, void* stream);


/* attractors / ||attractors||_2 and logits[b,t,c] = <emb[b,t], attractors[b,t,c]>
 * (FS model :43,:60 / :76,:79; LS model :89,:117).  emb f32 [B][Tp][D], attr f32 [B*C][Tp][D]
 * -> attr_out f32 [B][T][C][D], logits f32 [B][T][C]. */
int eend_head_l2dot_f32(const float* emb, const float* attr, float* attr_out, float* logits, int B, int T,
                        int Tp, int C, int D, void* stream);
/* The same reading the un-normalised attractors from the f16 stream (FS-EEND f16 residual mode: the last decoder LayerNorm's
 * output is then never written in f32). */
int eend_head_l2dot_a16_f32(const float* emb, const void* attr_f16, float* attr_out, float* logits, int B, int T,
                            int Tp, int C, int D, void* stream);


/* ======================================================================================================
 * TRAINING STEP (BASELINE config 4).  The reference trains through torch autograd + torch.optim.Adam under
 * PyTorch-Lightning (FS-EEND/train/oln_tfm_enc_dec.py:51-91 training_step, train/utils/loss.py:119-125
 * standard_loss, FS-EEND/train_dia.py:77-100 optimiser, :145-156 Trainer); the entry points below are the
 * hand-written backward of every forward op above, the loss, and the optimiser.  Gradient tensors that feed
 * an MFMA are bf16 (exponent range), saved forward activations f16, accumulation and the residual-gradient
 * stream f32.  `ws` arguments are caller-owned f32 scratch (`ws_floats` = its capacity in floats); parameter
 * gradients are written in fixed summation order (no atomics), so a step is bit-reproducible.
 * ====================================================================================================== */

/* Dropout (torch.nn.Dropout in train mode: FS model :147 / merge_tfm_encoder.py:209-219,385,394,398-399,609-614 and
 * nn.MultiheadAttention(dropout=p)).  The reference draws its masks from torch's Philox stream, which no other
 * implementation can reproduce; here a mask is a pure function of (seed, element index), so the backward recomputes it
 * instead of storing it and a step stays bit-reproducible:
 *   h = mix((a * 0x9E3779B1 + b) ^ seed),   keep <=> (h >> 8) >= thresh24,
 *   mix(x): x ^= x >> 16; x = (x & 0xFFFFFF) * 0x6B2F4D; x ^= x >> 13; x = (x & 0xFFFFFF) * 0x9E3779   (32-bit wrap-around; ABI 5 --
 *   versions up to 4 used the murmur3 finaliser, whose 32-bit multiplies are quarter rate on CDNA4)
 * with (a, b) = (row, column) of the element (attention: a = (seq*H + head)*Tp + query, b = key; speaker attention:
 * a = ((b*Tp + t)*4 + head)*16 + query slot, b = key slot).  thresh24 = round(p * 2^24), scale = 1/(1-p).
 * A null pointer or thresh24 == 0 means no dropout.  Host struct, read at call time. */
typedef struct eend_dropout {
    unsigned seed;
    unsigned thresh24;
    float scale;
} eend_dropout;

/* Training forward of nn.MultiheadAttention(x, x, x) on the time axis in one launch (attn_stream.hip, round 6; windows Tp = 64 m <= 512,
 * H = 4): eend_inproj_heads_train_bf16 + eend_attn_causal_lse_bf16 with Q / K / V^T staying on chip between projection and attention.
 * W_packed = eend_inproj_attn_pack_f16 of the f16 in-projection operand copy (q rows pre-scaled as for the two entries it replaces), b_in
 * [768] (the key bias IS applied here: the saved K is the reference's).  Out: O_f16 [nseq*Tp][ldo] (the heads' context rows), Q / K / V
 * bf16 head rows [nseq][H][Tp][64] and lse [nseq][H][Tp] (log2 domain) for eend_attn_causal_bwd_bf16; `drop` acts on the attention
 * probabilities, element ((seq*H + head)*Tp + query, key), as in eend_attn_causal_lse_bf16.  EEND_EINVAL for longer windows: the caller
 * keeps the two entries (and the [d][t] copies their backward reads). */
int eend_inproj_attn_train_bf16(const void* X_f16, int ldx, const void* W_packed, const float* b_in, void* O_f16, int ldo, void* Q_bf16,
                                void* K_bf16, void* V_bf16, float* lse, int nseq, int H, int Tp, int mask_delay, int kv_len,
                                const eend_dropout* drop, void* stream);

/* g_f32[M][256] += A[M][K] Wt^T in place on a packed weight stream (gemm_acc_stream.hip, round 6): eend_gemm_acc_bf16(res = out = g,
 * alpha = 1) for the data gradients whose K is large -- autograd of nn.MultiheadAttention's in_proj (K = 768; FS model :147,
 * merge_tfm_encoder.py:379-385) and of MultiScaleRetention's projections (K = 1024; LS retention.py:146-160).  A bf16 [M][lda],
 * Wt bf16 [256][ldw] (the transposed weight, as eend_gemm_acc_bf16 takes it), K a multiple of 128 in 256 .. 2048, re-ordered once per
 * parameter version by eend_gemm_acc_stream_pack_bf16 into eend_gemm_acc_stream_elems(K) 16-bit elements.  eend_gemm_acc_stream_ok is
 * the shape predicate; EEND_EINVAL otherwise (the caller keeps eend_gemm_acc_bf16 / eend_gemm_acc_lnbwd_bf16). */
int eend_gemm_acc_stream_elems(int K);
int eend_gemm_acc_stream_ok(int M, int K, int lda);
int eend_gemm_acc_stream_pack_bf16(const void* Wt, int ldw, void* stream_out, int K, void* stream);
int eend_gemm_acc_stream_bf16(const void* A, int lda, const void* wstream, float* g_f32, int M, int K, void* stream);

/* K = 256 input projections on a packed weight stream (proj_stream.hip, round 6): nn.MultiheadAttention's in_proj (FS model :147,
 * merge_tfm_encoder.py:379-385) and MultiScaleRetention's q / k / v / g projections (LS retention.py:146-160) in the training forward,
 * where the f16 operands of the forward kernel and the bf16 Q / K / V the hand-written backward keeps used to be two projections of the
 * same rows.  Y = X W^T + bias, X f16 [M][ldx] (256 features), W f16 [N][256], N = 256 n <= 1024, re-ordered once per parameter version
 * by eend_proj_stream_pack_f16 into eend_proj_stream_elems(N) f16 elements.  Each 256-feature output group g names up to three
 * destinations, all written from one pass over the rows:
 *   rows            rows_kind 1: [M][rows_ld] row-major (pointer at the group's first column); 2: head rows [seq][H][Tp][64];
 *                   f16, or bf16 with rows_bf16
 *   rows2_bf16_heads  a second copy as bf16 head rows
 *   heads_t         transposed head rows [seq][H][64][Tp] (f16, or bf16 with heads_t_bf16)
 * Head destinations need H = 4, Tp a multiple of 64 and M a multiple of Tp.  eend_proj_stream_ok is the shape predicate (pointers of
 * `groups` only tested for null / alignment); EEND_EINVAL on anything else -- the caller keeps eend_inproj_heads_train_bf16 /
 * eend_retention_proj_f16 / eend_gemm_f16 there. */
typedef struct eend_proj_group {
    void* rows;
    int rows_kind, rows_bf16, rows_ld;
    void* rows2_bf16_heads;
    void* heads_t;
    int heads_t_bf16;
} eend_proj_group;
int eend_proj_stream_elems(int N);
int eend_proj_stream_pack_f16(const void* W, void* stream_out, int N, void* stream);
int eend_proj_stream_ok(int ldx, int M, int N, int Tp, int H, const eend_proj_group* groups);
int eend_proj_stream_f16(const void* X, int ldx, const void* wstream, const float* bias, int M, int N, int Tp, int H,
                         const eend_proj_group* groups, void* stream);

/* eend_linear_res_ln_f16 that also saves what LayerNorm backward needs: xhat_f16 [M][256] = the normalised
 * row before the affine, rstd [M] = 1/sigma.  (torch.nn.LayerNorm inside nn.TransformerEncoderLayer, FS model
 * :147,:174; merge_tfm_encoder.py:364,373-374.)  `drop` acts on (A W^T + bias) before the residual (dropout1 /
 * dropout2 / dropout11 / dropout21), element (row m, column n). */
int eend_linear_res_ln_train_f16(const void* A, int lda, const void* W, int ldw, const float* bias, const float* res,
                                 float alpha, const float* gamma, const float* beta, float eps, float* out_f32,
                                 void* out_f16, void* xhat_f16, float* rstd, int M, int K, const eend_dropout* drop,
                                 void* stream);
/* relu(A W^T + bias) with dropout after the activation (the FFN's inner `self.dropout`, merge_tfm_encoder.py:398,613);
 * the stored activation is the dropped one, so its zeros are the ReLU-and-dropout mask of the backward. */
int eend_linear_relu_train_f16(const void* A, int lda, const void* W, int ldw, const float* bias, void* out_f16, int ldo,
                               int M, int N, int K, const eend_dropout* drop, void* stream);
/* The two entries above as ONE launch for a post-norm ReLU block (round 5; FS nn.TransformerEncoderLayer._ff_block + norm2,
 * merge_tfm_encoder.py:397-399 / :612-614; LS merge_retnet_layer.py:250-253):
 *   hid = drop_hidden(relu(X W1^T + b1))            f16 [M][F], written once and never re-read by this launch
 *   y   = drop_out(hid W2^T + b2) * alpha + res     out_f32 / out_f16 = LayerNorm(y); xhat_f16, rstd as eend_linear_res_ln_train_f16
 * Same dropout indexing (row, column) per site as the two-launch form, so a given eend_dropout produces the same masks.
 * K = N = 256, F a multiple of 64, M * F * 2 < 2^32 bytes, 16-byte aligned hid / xhat; out_f32 may be res (in place).
 * EEND_EINVAL outside that: the caller keeps the two launches. */
int eend_ffn_train_f16(const void* X, int ldx, const void* W1, const float* b1, const void* W2, const float* b2, const float* res,
                       float alpha, const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, void* hid_f16,
                       void* xhat_f16, float* rstd, int M, int F, const eend_dropout* drop_hidden, const eend_dropout* drop_out,
                       void* stream);
/* The Macaron half-step FFN of a Conformer block (LS-EEND conformer/feed_forward.py:47-57 inside modules.py:32-33's residual) as one
 * launch: z = X W1^T + b1 (saved, f16), a = drop_hidden(swish(z)) (saved, f16), y = drop_out(a W2^T + b2) * alpha + res; then either
 * (residual_stream_unnormalised = 1, the pre-norm join: eend_linear_res_scale_ln_train_f16) out_f32 = y and out_f16 / xhat / rstd = the NEXT
 * sub-layer's LayerNorm of y, or (0: the block-final LayerNorm, eend_linear_res_ln_train_f16) out_f32 = out_f16 = LayerNorm(y).
 * Replaces eend_linear_f16 + eend_swish_dropout_f16 + that GEMM entry.  Shapes as eend_ffn_train_f16. */
int eend_ffn_swish_train_f16(const void* X, int ldx, const void* W1, const float* b1, const void* W2, const float* b2, const float* res,
                             float alpha, const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, void* z_f16,
                             void* a_f16, void* xhat_f16, float* rstd, int M, int F, int residual_stream_unnormalised,
                             const eend_dropout* drop_hidden, const eend_dropout* drop_out, void* stream);
/* Data-gradient backward of the same block in one launch (round 5; replaces eend_gemm_relu_bwd_bf16 + eend_gemm_acc_bf16 of its FFN):
 *   dH = drop_scale * (dY W2) where hid_f16 != 0, else 0     bf16 [M][F], written once (the weight gradient of linear1 reads it)
 *   g  += dH W1                                              the f32 residual-gradient stream [M][256], in place
 * dY bf16 [M][ldy], W2T = W2 transposed, bf16 [F][256]; W1T = W1 transposed, bf16 [256][F] (the copies the two-launch form uses).
 * F a multiple of 64, M * F * 2 < 2^32 bytes, 16-byte aligned buffers; EEND_EINVAL outside that. */
int eend_ffn_bwd_data_bf16(const void* dY, int ldy, const void* W2T, const void* hid_f16, const void* W1T, float drop_scale,
                           void* dH_bf16, float* g_f32, int M, int F, void* stream);
/* Round 6: eend_ffn_train_f16 / eend_ffn_bwd_data_bf16 on a PACKED WEIGHT STREAM (ffn_train_stream.hip; same reference sites): one
 * wave owns 32 / 48 token rows end to end, the hidden units stay in its registers and leave (hid, dH) / arrive (the mask) as 16-byte
 * accesses per lane.  eend_ffn_train_stream_pack re-orders two 16-bit matrices A [F][256], B [256][F] into the fragment sequence the
 * kernel consumes (eend_ffn_train_stream_elems(F) elements; F a multiple of 64, at most 2048): (W1, W2) in f16 for the forward,
 * (W2^T, W1^T) in bf16 for the data gradient.  Pack once per parameter version.  Results, saved tensors and dropout masks as the
 * un-packed entries (up to fp32 summation order), EXCEPT the layout of hid_f16 and dH_bf16: BLOCKED [ceil(M/16)][F/32][16 rows][32 units]
 * (element (m, u) at ((m>>4) * (F/32) + (u>>5)) * 512 + (m&15) * 32 + (u&31); allocate ceil(M/16) * 16 rows) -- a wave then writes its 16
 * rows as one sequential stream of whole cache lines, where row-major rows made the launch write-pattern-bound (594 -> 488 us forward,
 * 812 -> 493 us data gradient at [196608, 2048]).  The only consumers are eend_ffn_bwd_data_stream_bf16 (the mask) and the weight
 * gradients (eend_wgrad[_bias]_bf16 with x_is_f16 & 2 / & 4).  eend_ffn_train_stream_ok: the row count one launch can address (32-bit
 * offsets); outside it -> EEND_EINVAL, the caller keeps the un-packed, row-major entries. */
int eend_ffn_train_stream_elems(int F);
int eend_ffn_train_stream_ok(int M, int F, int ldx);
int eend_ffn_train_stream_pack(const void* A, const void* B, void* stream_out, int F, void* stream);
int eend_ffn_train_stream_f16(const void* X, int ldx, const void* wstream, const float* b1, const float* b2, const float* res, float alpha,
                              const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, void* hid_f16, void* xhat_f16,
                              float* rstd, int M, int F, const eend_dropout* drop_hidden, const eend_dropout* drop_out, void* stream);
int eend_ffn_bwd_data_stream_bf16(const void* dY, int ldy, const void* wstream, const void* hid_f16, float drop_scale, void* dH_bf16,
                                  float* g_f32, int M, int F, void* stream);
/* eend_spk_attn_f16 with dropout of the attention probabilities. */
int eend_spk_attn_train_f16(const void* qkv, void* O_f16, int B, int C, int Tp, int H, float scale,
                            const eend_dropout* drop, void* stream);

/* eend_conv1d_l2norm_f16 that also saves inv_norm [nseq*Tp] = 1/||conv output|| (FS model :40-41). */
int eend_conv1d_l2norm_train_f16(const void* X, const void* Wr, const float* bias, const int* ilens, float* out_f32,
                                 void* out_f16, float* inv_norm, int nseq, int Tp, int cin, int ktaps, int pad,
                                 void* stream);

/* Packed MHA in-projection for training: Q, K, V in BOTH head layouts (bf16 [nseq][H][Tp][64] and
 * [nseq][H][64][Tp]) -- the forward attention reads Q, K, Vt, its backward Q, K, V and, for windows beyond 512 frames
 * (the two-kernel form), Qt, Kt.  Qt / Kt may be NULL (not written) when Tp <= 512; Vt may be NULL when no consumer reads it (the
 * retention backward at chunk lengths <= 512).  K = 256. */
int eend_inproj_heads_train_bf16(const void* A, int lda, const void* W, const float* bias, void* Q, void* Qt,
                                 void* K, void* Kt, void* V, void* Vt, int nseq, int Tp, int H, void* stream);

/* eend_attn_causal_bf16 that also writes lse [nseq][H][Tp]: the log2-domain log-sum-exp of every query row.
 * `drop`: dropout of the softmax probabilities (the normaliser stays un-dropped, as in torch). */
int eend_attn_causal_lse_bf16(const void* Q, const void* K, const void* Vt, void* O_f16, float* lse, int nseq, int H,
                              int Tp, int ldo, int mask_delay, int kv_len, float scale, const eend_dropout* drop,
                              void* stream);

/* Backward of the causal time-axis attention (torch autograd through nn.MultiheadAttention's core: FS model :147,
 * merge_tfm_encoder.py:379-385).  dO bf16 [nseq*Tp][ldo] (gradient w.r.t. the concatenated head outputs), O f16
 * the forward output; dQKV bf16 [nseq*Tp][ldg] receives dQ | dK | dV at columns 0 / 256 / 512 (+ h*64).
 * dOt_ws (bf16, nseq*Tp*256) and dh_ws (f32, nseq*H*Tp) are scratch.  scale_log2 as given to the forward
 * (its `scale` * log2 e); sq / sk: factors applied to dQ / dK (for the pre-scaled-q convention of
 * eend_attn_causal_bf16: scale_log2 = 1, sq = 1/sqrt(dh), sk = ln 2).  `drop`: the forward's spec; O_f16 is the
 * forward output (computed from the dropped probabilities), which keeps D_i = <dO_i, O_i> valid. */
int eend_attn_causal_bwd_bf16(const void* Q, const void* Qt, const void* K, const void* Kt, const void* V,
                              const void* dO, int ldo, const void* O_f16, int ldout, const float* lse, void* dOt_ws,
                              float* dh_ws, void* dQKV, int ldg, int nseq, int H, int Tp, int mask_delay, int kv_len,
                              int q_len, float scale_log2, float sq, float sk, const eend_dropout* drop, void* stream);

/* Gradient GEMMs, bf16 operands, f32 accumulate (autograd of every torch.nn.Linear on the path):
 *   out = A W^T (+ bias)            -> bf16 [M][ldo]                 N % 128 == 0, K % 64 == 0 */
int eend_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* out_bf16, int ldo,
                   int M, int N, int K, void* stream);
/*   out = drop_scale * (A W^T) where act != 0    (ReLU [+ dropout] backward; act = the saved forward activation, any
 *   2-byte float; drop_scale = 1/(1-p) of eend_linear_relu_train_f16, or 1) */
int eend_gemm_relu_bwd_bf16(const void* A, int lda, const void* W, int ldw, const void* act, int ldact,
                            void* out_bf16, int ldo, int M, int N, int K, float drop_scale, void* stream);
/*   out_f32 = (A W^T) * alpha + res_f32 (N = 256; the residual-gradient stream), optional bf16 copy */
int eend_gemm_acc_bf16(const void* A, int lda, const void* W, int ldw, const float* res_f32, float alpha,
                       float* out_f32, void* out_bf16, int M, int K, void* stream);

/* eend_gemm_acc_bf16 FOLLOWED by eend_layernorm_bwd_f32 of the post-norm site in front of the branch, in one launch (round 6): the
 * f32 gradient stream is read once and written once instead of twice each.
 *   g   = A W^T + g_f32                  (the data gradient of a K -> 256 linear joining the gradient w.r.t. a LayerNorm OUTPUT)
 *   dz  = rstd * (g gamma - mean(g gamma) - x_hat mean(g gamma x_hat))  -> ds_f32 (may alias g_f32), ds_bf16 = bf16 of dz under `drop`
 *   dgamma = sum_rows g x_hat, dbeta = sum_rows g, dbias (optional) = sum_rows of the masked dz      (overwritten, fixed summation order)
 * A bf16 [M][lda], W bf16 [256][ldw] (the transposed copy), x_hat f16 [M][256], ws >= ceil(M/64) * 768 floats. */
int eend_gemm_acc_lnbwd_bf16(const void* A, int lda, const void* W, int ldw, const float* g_f32, const void* xhat_f16, const float* rstd,
                             const float* gamma, float* ds_f32, void* ds_bf16, float* ws, long ws_floats, float* dgamma, float* dbeta,
                             float* dbias, int M, int K, const eend_dropout* drop, void* stream);

/* Conv1d(256,256,k) data gradient as an implicit GEMM over (tap, c_out) (autograd of FS model :40): dY bf16 slab,
 * Wd bf16 [c_in][k*c_out] with Wd[ci][tap'*c_out + co] = W[co][ci][k-1-tap']; frames >= src_lens[seq] of dY count as
 * zero; out_f32 rows t >= mask_lens[seq] are written as zero (the input was truncated to ilen, FS model :38-39). */
int eend_conv1d_dgrad_bf16(const void* dY, const void* Wd, const int* src_lens, const int* mask_lens, float* out_f32,
                           int nseq, int Tp, int cout, int ktaps, int pad, void* stream);

/* Weight gradient out[n][k] (row stride ld_out, k < K_out) (+)= scale * sum_m dY[m][n] X[m][k]; dY bf16 [M][lda],
 * X f16 (x_is_f16 & 1) or bf16 [M][ldb]; N, K % 128 == 0.  x_is_f16 & 2: X is stored in the blocked layout of
 * eend_ffn_train_stream_f16's hid (ldb = its full width F); x_is_f16 & 4: dY is (eend_ffn_bwd_data_stream_bf16's dH, lda = F). */
int eend_wgrad_bf16(const void* dY, int lda, const void* X, int ldb, int x_is_f16, long M, int N, int K, float* ws,
                    long ws_floats, float* out, int ld_out, int K_out, float scale, int accumulate, void* stream);
/* eend_wgrad_bias_bf16 for N / group_rows linear layers that share the input X and whose (weight, bias) gradients sit equally spaced in
 * one buffer (the q / k / v / g projections of a retention module in the flat gradient buffer): rows g * group_rows .. of dY's N columns go
 * to out + g * group_stride (row stride K), their bias gradient to bias_out + g * group_stride.  One pass over X instead of N / group_rows,
 * one reduction launch each for weights and biases.  Overwrites (no accumulate). */
int eend_wgrad_bias_grouped_bf16(const void* dY, int lda, const void* X, int ldb, int x_is_f16, long M, int N, int K, float* ws,
                                 long ws_floats, float* out, float* bias_out, int group_rows, long group_stride, float scale, void* stream);
/* The same with the bias gradient on the side: bias_out[n] = scale * sum_m dY[m][n] (+ bias_out if accumulate), accumulated
 * from the dY tiles the kernel stages anyway instead of by a separate pass over dY (eend_colsum_f32).  ws needs
 * nsplit * (N*K + N) floats. */
int eend_wgrad_bias_bf16(const void* dY, int lda, const void* X, int ldb, int x_is_f16, long M, int N, int K, float* ws,
                         long ws_floats, float* out, int ld_out, int K_out, float* bias_out, float scale, int accumulate, void* stream);
/* Conv1d weight gradient into the parameter's own (c_out, c_in, k) layout; tmp: f32 [c_out][k*c_in] scratch. */
int eend_conv1d_wgrad_bf16(const void* dY, const void* X_f16, const int* ilens, int nseq, int Tp, int cin, int ktaps,
                           int pad, float* ws, long ws_floats, float* tmp, float* out, void* stream);
/* out[n] (+)= scale * sum_m Y[m][n]   (bias gradients); Y bf16 or f16 [M][ld]. */
int eend_colsum_f32(const void* Y, int ld, long M, int N, int is_bf16, float* ws, long ws_floats, float* out,
                    float scale, int accumulate, void* stream);

/* LayerNorm backward (autograd of torch.nn.LayerNorm): g = gradient w.r.t. the output (f32 [M][256]); writes the
 * gradient w.r.t. the input as f32 (ds_f32, may alias g) and bf16, and dgamma / dbeta [256].  `drop`: the spec of the
 * producing eend_linear_res_ln_train_f16 -- applied to the bf16 copy only (the branch gradient), not to ds_f32 (the
 * residual stream).  dbias (optional, [256]): column sums of that branch gradient = the bias gradient of the linear layer in
 * front of the LayerNorm (saves a separate pass over ds_bf16). */
int eend_layernorm_bwd_f32(const float* g, const void* xhat_f16, const float* rstd, const float* gamma, float* ds_f32,
                           void* ds_bf16, float* ws, long ws_floats, float* dgamma, float* dbeta, float* dbias, long M,
                           const eend_dropout* drop, void* stream);

/* Head forward + standard_loss + their gradient in one pass (FS model :43,:60; train/utils/loss.py:119-125 with
 * label_delay = 0): labels f32 [B][T][C] (prepared: silence / speakers / none columns, zero padded), ilens / ncols
 * [B]; loss_out[0] = BCE loss; da f32 slab [(b*C+c)*Tp+t][256] = gradient w.r.t. the un-normalised attractors,
 * de f32 [B*Tp][256] = gradient w.r.t. the unit embeddings (overwritten); logits (optional) f32 [B][T][C].
 * With dlogits_in (f32 [B][T][C]: the caller's own d loss / d logits, e.g. torch autograd over the reference's
 * standard_loss) the loss part is skipped and that gradient is propagated (labels / ilens / ncols may be null). */
int eend_head_bce_f32(const float* emb, const float* attr, const float* labels, const int* ilens, const int* ncols,
                      float inv_frames, const float* dlogits_in, float* logits, float* da, float* de, float* ws,
                      long ws_floats, float* loss_out, int B, int T, int Tp, int C, void* stream);

/* x / ||x|| backward (FS model :41): y unit rows f32, dy f32, inv_norm from the forward -> dx bf16, rows t >= T zero. */
int eend_l2norm_bwd_bf16(const float* y, const float* dy, const float* inv_norm, void* dx_bf16, int B, int T, int Tp,
                         void* stream);

/* `convert` fan-out backward (FS model :113-114): gsum bf16 [B*Tp][256] = sum over speaker slots of g0,
 * dpc f32 [C][256] = sum over (b, t) of g0 per slot. */
int eend_convert_fanout_bwd_f32(const float* g0, void* gsum_bf16, float* ws, long ws_floats, float* dpc, int B, int Tp,
                                int C, void* stream);
/* mode 0: pc[c] = W[:, 256:] pe[c] + bias (the forward's per-slot constant); mode 1: dW[:, 256:] and dbias from dpc.
 * W / dW: the (256, 512) convert weight. */
int eend_convert_const_f32(int mode, const float* W, const float* bias, const float* pe, float* pc, const float* dpc,
                           float* dW, float* dbias, int C, void* stream);

/* Speaker-axis attention backward (merge_tfm_encoder.py:388-394): qkv f16 [rows][768] from the forward in-projection,
 * dO bf16 [rows][256] -> dqkv bf16 [rows][768]. */
int eend_spk_attn_bwd_bf16(const void* qkv_f16, const void* dO_bf16, void* dqkv_bf16, int B, int C, int Tp, int H,
                           float scale, const eend_dropout* drop, void* stream);

/* Train-mode BatchNorm1d statistics over the padded input (FS model :165-166): mean / biased var [F] of all B*T
 * frames (pad_value for frames beyond each length) and the running-statistics update (momentum, unbiased var). */
int eend_bn_train_stats_f32(const void* const* x_ptrs, const int* lens, float pad_value, float* ws, long ws_floats,
                            float* mean, float* var, float* run_mean, float* run_var, float momentum, int B, int T,
                            int F, void* stream);
/* BatchNorm weight / bias gradients from dy bf16 [B*Tp][ld] (gradient w.r.t. the BN output). */
int eend_bn_bwd_f32(const void* const* x_ptrs, const int* lens, float pad_value, const float* mean, const float* var,
                    float eps, const void* dy_bf16, int ld, float* ws, long ws_floats, float* dgamma, float* dbeta,
                    int B, int T, int Tp, int F, void* stream);

/* Embedding-consistency loss gradient (FS model :46-57): de f32 [B*Tp][256] += d loss / d emb. */
int eend_emb_consistency_bwd_f16(const void* emb_f16, const float* labels, const int* lens, float inv_count, float* de,
                                 int B, int T, int Tp, int D, int C, void* stream);

/* Optimiser on flat f32 buffers (FS-EEND/train_dia.py:83-100,153): sum of squares of the gradient; Adam step with
 * clip_grad_norm_ folded in (hp = {lr, 1-beta1^t, 1-beta2^t, max_norm} on the device). */
int eend_grad_sumsq_f32(const float* g, long n, float* ws, long ws_floats, float* out, void* stream);
int eend_adam_step_f32(float* p, const float* g, float* m, float* v, long n, const float* hp, const float* gsumsq,
                       float beta1, float beta2, float eps, void* stream);

/* Gradient accumulation over micro-batches (Lightning accumulate_grad_batches, FS-EEND/train_dia.py:151):
 * acc = (first ? 0 : acc) + scale * g on the flat buffers. */
int eend_grad_accumulate_f32(float* acc, const float* g, float scale, int first, long n, void* stream);

/* One entry of the weight re-layout table of eend_prep_weights: dst[a][b][c] (dims A x B x Cpad, zero for
 * c >= C) = convert(src[off + a*sa + b*sb + c*sc] * (a < nscale ? scale : 1)); dtype 0 f16, 1 bf16, 2 f32. */
typedef struct eend_prep_entry {
    const float* src;
    long off;
    void* dst;
    int A, B, C, Cpad;
    long sa, sb, sc;
    int dtype, nscale;
    float scale;
    int reserved;
} eend_prep_entry;
/* MFMA-operand copies of all parameters (f16 forward layouts, bf16 transposed backward layouts) in one launch;
 * `table` is a device array of n entries. */
int eend_prep_weights(const eend_prep_entry* table, int n, void* stream);

/* ======================================================================================================
 * LS-EEND TRAINING STEP (the LS half of BASELINE config 4: LS-EEND/train/oln_tfm_enc_dec_on_the_fly.py:52-92
 * training_step under train_dia_simu.py:97-117,159-173).  Same conventions as the FS-EEND training entries above.
 * Sequence slabs have Tp rows of which the first Tv (the reference's chunk-padded length, LS model :281-283) exist
 * in the reference; rows t >= Tv never enter a statistic and receive zero gradients.
 * ====================================================================================================== */

/* a = dropout(swish(z)) (conformer/feed_forward.py:51-52: Swish, nn.Dropout); z f16 [M][F] is the saved
 * pre-activation.  Backward, in place on the bf16 gradient: dz = da * keep * scale * swish'(z). */
int eend_swish_dropout_f16(const void* z_f16, void* a_f16, long M, int F, const eend_dropout* drop, void* stream);
int eend_swish_bwd_bf16(void* dz_bf16, const void* z_f16, long M, int F, const eend_dropout* drop, void* stream);

/* torch.nn.LayerNorm(256) forward keeping x_hat / 1/sigma for its backward (the pre-norm of the first sub-layer of a
 * Conformer block, feed_forward.py:48, applied to the previous block's output). */
int eend_layernorm_train_f16(const float* x, const float* gamma, const float* beta, float eps, void* y_f16,
                             void* xhat_f16, float* rstd, long M, void* stream);

/* LayerNorm backward, general form.  g: gradient w.r.t. the LayerNorm output, f32 or (g_is_bf16) bf16 [M][256].
 * ds_f32 (optional): the input gradient, overwritten or (accumulate) added to -- the latter is the pre-norm residual
 * block of conformer/modules.py:32-33, where the LayerNorm sits on the branch.  ds_bf16 (optional) = alpha16 *
 * dropout(input gradient), dbias (optional, needs ds_bf16) its column sums.  dgamma / dbeta [256] are overwritten. */
int eend_layernorm_bwd2_f32(const void* g, int g_is_bf16, const void* xhat_f16, const float* rstd, const float* gamma,
                            float* ds_f32, int accumulate, void* ds_bf16, float alpha16, float* ws, long ws_floats,
                            float* dgamma, float* dbeta, float* dbias, long M, const eend_dropout* drop, void* stream);

/* Residual-stream gradient g f32 [M][256] -> gradient of a pre-norm branch output: ds_bf16 = alpha * dropout(g)
 * (ResidualConnectionModule module_factor, the branch's trailing nn.Dropout) and dbias [256] = its column sums. */
int eend_resgrad_cast_bf16(const float* g, void* ds_bf16, float alpha, float* ws, long ws_floats, float* dbias, long M,
                           const eend_dropout* drop, void* stream);

/* eend_linear_res_scale_ln16_f16 for training: out_f32 = dropout(A W^T + bias) * alpha + res (un-normalised
 * residual stream), out_f16 = LayerNorm(out_f32) with x_hat / 1/sigma saved (the next sub-layer's pre-norm). */
int eend_linear_res_scale_ln_train_f16(const void* A, int lda, const void* W, int ldw, const float* bias,
                                       const float* res, float alpha, const float* gamma, const float* beta, float eps,
                                       float* out_f32, void* out_f16, void* xhat_f16, float* rstd, int M, int K,
                                       const eend_dropout* drop, void* stream);

/* ConformerConvModule, train mode (conformer/convolution.py:138-149).  P f16 [nseq*Tp][512] = pointwise-conv-1
 * output (value | gate); c = causal depthwise conv (k taps, zero left context) of value * sigmoid(gate), f16
 * [nseq*Tp][256], zero for t >= Tv. */
int eend_glu_dwconv_f16(const void* P_f16, const float* w, void* c_f16, int nseq, int Tp, int Tv, int k, void* stream);
/* BatchNorm1d batch statistics of c over the nseq*Tv valid frames, two-pass: stats f32 [513] = mean[256], M2[256]
 * (sum of squared deviations), n.  The triples of several ranks merge exactly (SyncBatchNorm, train_dia_simu.py:167). */
int eend_bn_batch_stats_f16(const void* c_f16, float* ws, long ws_floats, float* stats, int nseq, int Tp, int Tv,
                            void* stream);
/* stats [R][513] of R ranks -> mean, biased var [256], n_out[1] = total frame count; running statistics updated
 * like torch BatchNorm1d / SyncBatchNorm in train mode (momentum, unbiased variance) when run_mean is given. */
int eend_bn_merge_f32(const float* stats, int R, float* mean, float* var, float* n_out, float* run_mean, float* run_var,
                      float momentum, void* stream);
/* s = swish(BatchNorm(c)) with the given batch statistics, f16 [M][256]. */
int eend_bn_swish_f16(const void* c_f16, const float* mean, const float* var, float eps, const float* gamma,
                      const float* beta, void* s_f16, long M, void* stream);
/* Backward of the two, pass 1: sums f32 [512] = per-channel sum of d_y and of d_y * c_hat over this rank's valid
 * frames (d_y = ds * swish'(BN(c))); also written as dbeta / dgamma.  Pass 2 (sums / n may be the all-reduced global
 * ones): ds <- d_c = gamma * rstd * (d_y - S1/n - c_hat * S2/n), zero for t >= Tv, in place (bf16). */
int eend_bn_swish_bwd_stats_bf16(const void* ds_bf16, const void* c_f16, const float* mean, const float* var, float eps,
                                 const float* gamma, const float* beta, float* ws, long ws_floats, float* sums,
                                 float* dgamma, float* dbeta, int nseq, int Tp, int Tv, void* stream);
int eend_bn_swish_bwd_apply_bf16(void* ds_bf16, const void* c_f16, const float* mean, const float* var, float eps,
                                 const float* gamma, const float* beta, const float* sums, const float* n_dev, int nseq,
                                 int Tp, int Tv, void* stream);
/* Depthwise conv + GLU backward: dP bf16 [nseq*Tp][512] (gradient of the pointwise-conv-1 output), dw f32 [256][k]. */
int eend_dwconv_glu_bwd_bf16(const void* dc_bf16, const void* P_f16, const float* w, void* dP_bf16, float* ws,
                             long ws_floats, float* dw, int nseq, int Tp, int Tv, int k, void* stream);

/* eend_retention_chunk_f16 (chunk-resident kernel, L <= 512) that also saves rhat_f16 [nseq*Tp][ldo] = the per-head
 * normalised retention rows and rc f32 [nseq*Tp][H] = 1/sigma times the detached row scale (retention.py:163,180,185:
 * inner_scale / kv_scale carry no gradient). */
int eend_retention_chunk_train_f16(const void* Q, const void* K, const void* Kt, const void* Vt, const void* G,
                                   void* O_f16, void* rhat_f16, float* rc, void* St_ws, float* kv_ws, float* cscale_ws,
                                   float* sexp_ws, int nseq, int H, int Tp, int L, int ldo, int ldg, float gn_eps,
                                   int T_valid, void* stream);
/* MultiScaleRetention backward from the gradient of its out_proj input (f32 [nseq*Tp][256]; retention.py:196-228, chunk-recurrent form
 * :146-194): swish gate and per-head LayerNorm backward, then the linear-attention backward with the detached scales,
 *   dq_t = sum_{s<=t} (o~_t.v_s) k_s,  dk_s = sum_{t>=s} (o~_t.v_s) q_t,  dv_s = sum_{t>=s} (q_t.k_s) o~_t,
 * intra-chunk on bf16 MFMA tiles, across chunks through 64x64 prefix / suffix states.  Q..Vt: bf16 head layouts of
 * eend_inproj_heads_train_bf16 (k rows pre-scaled by dk^-1/2); dqkvg bf16 [nseq*Tp][ldq]: dq | sk*dk | dv | dg at
 * columns 0 / 256 / 512 / 768.  ot_ws, ott_ws: bf16 scratch [nseq*Tp*256]; kv_ws, g_ws: f32 [nseq*H*nc*4096];
 * St_ws: bf16 [nseq*H*nc*6*4096].  Chunk lengths L <= 512 read only the row-major head layouts (Q, K, V; the transposed
 * fragments come out of the LDS reads): Qt, Kt, Vt and ott_ws may then be NULL, and eend_inproj_heads_train_bf16 need not write them. */
int eend_retention_bwd_bf16(const void* Q, const void* Qt, const void* K, const void* Kt, const void* V, const void* Vt,
                            const float* dctx_f32, const void* g_f16, int ldg, const void* rhat_f16, const float* rc,
                            void* ot_ws, void* ott_ws, float* kv_ws, float* g_ws, void* St_ws, void* dqkvg_bf16, int ldq,
                            int nseq, int H, int Tp, int L, int T_valid, float sk, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EEND_HIP_H */
