// Internal (C++) launch interface shared by the .hip translation units and the
// C-ABI shim (api.hip).  Nothing here is exported; the exported surface is
// include/eend_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/eend_hip.h"
#include "common.h"

enum GemmEpilogue {
    EPI_PLAIN_F16 = 0,       // out16[m][n] = f16(acc + bias)
    EPI_PLAIN_RELU_F16 = 1,  // out16[m][n] = f16(relu(acc + bias))
    EPI_QK_HEADS = 2,        // bf16 Q -> out16, K -> out16b, layout [seq][H][Tp][dh]
    EPI_VT_HEADS = 3,        // bf16 V^T -> out16, layout [seq][H][dh][Tp]
    EPI_RES_LN = 4,          // LN(acc + bias + res) -> out32 (f32) and out16 (f16); N == 256
    EPI_L2NORM = 5,          // implicit-GEMM Conv1d + bias, then x/||x||_2 -> out32, out16; N == 256
    EPI_CONVERT = 6,         // attr0 fan-out over C speaker slots (+pc[c]) -> out32, out16; N == 256
    EPI_RES_SCALE = 7,       // (acc + bias)*alpha + res -> out32, out16 (no norm); N == 256
    EPI_RES_SCALE_LN16 = 8,  // y = (acc + bias)*alpha + res -> out32 ; LN(y)*gamma+beta -> out16; N == 256
    EPI_PLAIN_SWISH_F16 = 9, // out16[m][n] = f16(swish(acc + bias))
    EPI_GLU_F16 = 10,        // rows of W interleaved (value, gate): out16[m][n/2] = v * sigmoid(g)
    EPI_QK_HEADS_F16 = 11,   // as EPI_QK_HEADS, f16 outputs (retention)
    EPI_KTVT_HEADS_F16 = 12, // K^T -> out16, V^T -> out16b, f16 [seq][H][dh][Tp] (retention)
    // ---- training step (gradient GEMMs run on bf16 operands: GemmParams::bf16 = 1)
    EPI_PLAIN_BF16 = 13,     // out16[m][n] = bf16(acc + bias)
    EPI_MASK_BF16 = 14,      // out16[m][n] = bf16(acc) where mask[m][n] != 0 else 0   (ReLU backward)
    EPI_F32_ROWMASK = 15,    // out32[m][n] = acc for frames t < mask_lens[seq], else 0 (ALOAD_CONV: Conv1d data gradient)
    EPI_RES_LN_TRAIN = 16,   // EPI_RES_LN that also saves x_hat (xhat16) and 1/sigma (rstat) of every row
    EPI_L2NORM_TRAIN = 17,   // EPI_L2NORM that also saves 1/||x|| (rstat)
    EPI_RES_SCALE_LN16_TRAIN = 18,  // EPI_RES_SCALE_LN16 (pre-norm residual blocks, LS-EEND Conformer) with dropout of
                                    // (acc + bias) and x_hat / 1/sigma of the LayerNorm saved
    EPI_RES_LNBWD = 19,      // bf16 data-gradient GEMM into the f32 gradient stream (EPI_RES_SCALE) FOLLOWED by the LayerNorm backward of the
                             // post-norm site in front of the branch: out32 = dz, out16 = bf16 masked branch gradient, colpart = d gamma | d beta | d bias partials
};

struct GemmParams {
    const void* A;      // activations, f16 [M][lda]
    const void* W;      // weights, f16 [N][ldw]
    const float* bias;  // [N] or null
    int M, N, K;        // K % 64 == 0
    int lda, ldw, ldo;
    void* out16;        // f16 / bf16 output (see epilogue)
    void* out16b;       // second half-precision output (K for EPI_QK_HEADS)
    void* out32;        // f32 output or null
    const float* res;   // f32 residual [M][ldo] or null
    const void* res16;  // ... or the same as f16 (post-norm stacks: the residual IS the previous LayerNorm's output, whose f16
                        // copy the next MFMA reads anyway; saves the f32 stream's write + read; used when res is null)
    const float* gamma; // LN affine
    const float* beta;
    const float* pc;    // EPI_CONVERT: [C][N] per-speaker-slot constant
    const int* ilens;   // ALOAD_CONV: valid frames per sequence
    float eps;
    float alpha;        // EPI_RES_SCALE
    int Tp;             // frames per sequence slab (Tp % 64 == 0)
    int H, dh;          // heads layout
    int C;              // speaker slots (EPI_CONVERT)
    int conv_cin;       // ALOAD_CONV: input channels (multiple of 64)
    int conv_pad;       // ALOAD_CONV: left padding (taps before the centre)
    int dbg;            // ablation flags (EEND_GEMM_DBG env, perf studies only): 1 skip stores, 2 reload k-tile 0, 4 skip MFMA
    // training step
    int bf16;           // operands (A, W) and 2-byte outputs are bf16 instead of f16
    const void* mask;   // EPI_MASK_BF16: saved forward activation, 2-byte floats [M][ldmask]
    int ldmask;
    const int* mask_lens; // EPI_F32_ROWMASK: frames per sequence that receive a gradient
    float* colpart;     // EPI_RES_LNBWD: f32 [ceil(M/64)][3][256] column partials (x_hat, rstat, gamma, drop are inputs there)
    void* xhat16;       // EPI_RES_LN_TRAIN: normalised rows before the affine, [M][ldo]
    float* rstat;       // EPI_*_TRAIN: 1/sigma or 1/||x|| per row, [M]
    DropSpec drop;      // EPI_RES_LN_TRAIN: dropout of (acc + bias) before the residual add; EPI_PLAIN_RELU_F16: of the
                        // activation; EPI_MASK_BF16: only .scale is used (the saved activation already carries the zeros)
};

struct AttnParams {
    const void* Q;   // bf16 [nseq][H][Tp][64]
    const void* K;   // bf16 [nseq][H][Tp][64]
    const void* Vt;  // bf16 [nseq][H][64][Tp]
    void* O;         // f16 [nseq*Tp][ldo], head h at columns h*64..
    int nseq, H, Tp, ldo;
    int mask_delay;  // allowed(i,j) <=> j - i <= mask_delay && j < kv_len
    int kv_len;      // number of real key frames (<= Tp)
    float scale_log2;  // (1/sqrt(dh)) * log2(e)
    float* Lse;        // optional (training): f32 [nseq][H][Tp], log2-domain log-sum-exp of every query row
    DropSpec drop;     // training: dropout of the attention probabilities (a = (seq*H + h)*Tp + q, b = key)
};

struct SpkAttnParams {
    const void* qkv;  // f16 [nrows][3*D], row = (b*C + c)*Tp + t
    void* O;          // f16 [nrows][D]
    int B, C, Tp, H;  // D = H*64
    float scale;      // 1/sqrt(dh)
    DropSpec drop;    // training: dropout of the probabilities (a = ((b*Tp + t)*4 + head)*16 + query slot, b = key slot)
};


int eend_launch_emb_consistency(const float* emb, const float* tgt, const int* lens, float inv_count, float* partial_ws, float* out,
                                int B, int T, int Tp, int D, int C, hipStream_t stream);

int eend_launch_activity_median(const float* pred, int ld, int T, int S, float thr, int k, unsigned char* out, hipStream_t stream);
int eend_launch_segments(const unsigned char* act, int T, int S, int* changes, int* counts, int cap, hipStream_t stream);
int eend_launch_der_counters(const float* pred, int ldp, const float* label, int ldl, int T, int C, int delay,
                             unsigned long long* counters, hipStream_t stream);

int eend_launch_stft_logmel(const float* y, long len, long first, int n_frames, const float* dft, const float* melT, float* out,
                            hipStream_t stream);
int eend_launch_colnorm(const float* Y, float* out, int T, int F, int mode, hipStream_t stream);
int eend_launch_splice_subsample(const float* Y, int T, int F, int ctx, int sub, float* out, hipStream_t stream);

int eend_launch_pit_cost(const float* y, const float* lab, int B, int T, int C, double* cost, hipStream_t stream);
int eend_launch_pit_assign(const double* cost, const int* nspk, int B, int C, int* perm, double* loss, hipStream_t stream);

struct RetParams {
    const void* Q;    // f16 [nseq][H][Tp][64]
    const void* K;    // f16 [nseq][H][Tp][64]   (already scaled by dk^-0.5)
    const void* Kt;   // f16 [nseq][H][64][Tp]
    const void* Vt;   // f16 [nseq][H][64][Tp]
    const void* G;    // f16 [nseq*Tp][ldg] gate pre-activation
    void* O;          // f16 [nseq*Tp][ldo]
    void* St;         // f16 [nseq][H][nc][2][64 hd][64 kd]: state before each chunk, hi/lo, prescaled
    float* cscale;    // [nseq][H][nc] reference cross_scale of that state
    float* sexp;      // [nseq][H][nc] 2^e undoing the prescale
    float* kv_ws;     // [nseq][H][nc][64][64] f32 per-chunk K^T V (workspace)
    int nseq, H, Tp, L, nc, ldo, ldg;
    float gn_eps;
    const float* state_in;   // optional: unscaled chunk state S = sum k (x) v before the first chunk, f32 [nseq][H][64 kd][64 hd]
    float* state_out;        // optional: the state after the last chunk (same layout)
    // training forward (chunk-resident kernel only): what the backward needs of the per-head LayerNorm and the scale
    void* Rhat;              // optional f16 [nseq*Tp][ldo]: normalised retention rows (before the gate)
    float* Rc;               // optional f32 [nseq*Tp][H]: 1/sigma of the per-head LayerNorm times the detached row scale
                             // 1/(sqrt(i+1) * max(inner_scale, cross_scale)) (retention.py:163,185 -- no gradient flows into it)
};

int eend_launch_gemm(const GemmParams& p, int epi, hipStream_t stream);

// attn_stream.hip: in-projection + causal attention in one launch, token-owning waves and fragment-packed weights (Tp = 512, H = 4)
#define EEND_ATTN_LONG_MAX_PAIRS 64
struct InprojAttnParams {
    const void* X; int ldx;      // f16 [nseq*Tp][ldx], 256 model dims
    const void* W;               // f16 packed in_proj_weight (eend_launch_inproj_attn_pack), q rows pre-scaled (ops.QSCALE_LOG2)
    const float* bias;           // [768]
    void* O;                     // f16 [nseq*Tp][ldo]
    int nseq, H, Tp, ldo, mask_delay, kv_len;
    // windows of more than 512 frames (eend_launch_inproj_attn_long fills the table): items = (query group, key group) of 512 frames
    int npairs;
    void* Opart;                 // f16 [off-diagonal slot][nseq][512][256]: normalised partial rows
    float* lse;                  // [item of the table][nseq][4][512]: log2 of the softmax denominators
    unsigned char pq[EEND_ATTN_LONG_MAX_PAIRS], pk[EEND_ATTN_LONG_MAX_PAIRS], pslot[EEND_ATTN_LONG_MAX_PAIRS];
    // training forward (eend_launch_inproj_attn_train): bf16 head rows [seq][H][Tp][64] for the backward, lse [nseq][4][Tp], dropout of P
    void* Qh; void* Kh; void* Vh;
    DropSpec drop;
};
int eend_launch_inproj_attn_train(const InprojAttnParams& p, hipStream_t stream);
long eend_inproj_attn_packed_nelems();
int eend_inproj_attn_long_scratch(int nseq, int Tp, int mask_delay, int kv_len, long* part_elems, long* lse_elems);
int eend_launch_inproj_attn_long(const InprojAttnParams& p, hipStream_t stream);
int eend_launch_inproj_attn_pack(const void* W, void* out, hipStream_t stream);
int eend_launch_inproj_attn_stream(const InprojAttnParams& p, hipStream_t stream);

// skinny.hip: linear layers with M <= EEND_SKINNY_MAX_M rows (the frame-by-frame streaming steps); same epilogue
// semantics as the gemm.hip epilogues they replace.  EEND_SKINNY=0 in the environment keeps the tiled GEMM (A/B).
#define EEND_SKINNY_MAX_M 16
bool eend_skinny_ok(const void* A, int lda, const void* W, int ldw, int M, int K);
int eend_launch_skinny_plain(const void* A, int lda, const void* W, int ldw, const float* bias, void* out16, int ldo,
                             int M, int N, int K, int act, hipStream_t stream);
int eend_launch_skinny_glu(const void* A, int lda, const void* W, int ldw, const float* bias, void* out16, int ldo,
                           int M, int N2, int K, hipStream_t stream);
int eend_launch_skinny_res(const void* A, int lda, const void* W, int ldw, const float* bias, const float* res, float alpha,
                           const float* gamma, const float* beta, float eps, float* out32, void* out16, int M, int K,
                           int mode, hipStream_t stream);
int eend_launch_ret_state_scan(const RetParams& p, hipStream_t stream);
int eend_launch_ret_chunk(const RetParams& p, hipStream_t stream);
int eend_launch_layernorm_f16(const float* x, const float* gamma, const float* beta, float eps, void* out16,
                              long M, int D, hipStream_t stream);
int eend_launch_dwconv_bn_swish(const void* x16, const float* w, const float* bn_w, const float* bn_b,
                                const float* bn_mean, const float* bn_var, float eps, void* out16, int nseq,
                                int Tp, int D, int k, const void* halo16, hipStream_t stream);
int eend_launch_attn_causal(const AttnParams& p, hipStream_t stream);
int eend_launch_spk_attn(const SpkAttnParams& p, hipStream_t stream);
int eend_launch_bn_cast_pad(const float* x, const float* bn_w, const float* bn_b, const float* bn_mean,
                            const float* bn_var, float eps, void* out16, int B, int T, int Tp, int Fin,
                            int Fpad, int apply_bn, hipStream_t stream);
int eend_launch_head(const float* emb, const void* attr, int attr_is_f16, float* attr_out, float* logits, int B, int T,
                     int Tp, int C, int D, hipStream_t stream);
int eend_launch_ret_step(const void* qkvg, float* kv, const float* scale_in, float* scale_out, void* out16, int N,
                         int H, float eps, hipStream_t stream);
int eend_launch_ret_step_f32in(const float* qkvg, float* kv, const float* scale_in, float* scale_out, void* out16, float* out32, int N,
                               int H, float eps, hipStream_t stream);
int eend_launch_spk_attn_step_f32(const float* qkv, float* out, int B, int C, float scale, hipStream_t stream);
int eend_launch_l2norm_rows_f32(const float* x, float* y, int rows, hipStream_t stream);
int eend_launch_skinny_plain_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* out32, int ldo, int M, int N,
                                 int K, int act, hipStream_t stream);
int eend_launch_skinny_res_f32(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, float alpha,
                               const float* gamma, const float* beta, float eps, float* out32, void* out16, int M, int K, int mode,
                               hipStream_t stream, float* ln_out32 = nullptr);
// gemm_f32.hip: f32 linear layers for M > 16 rows on the exact-f32 MFMA (multi-stream frame steps)
bool eend_linear_f32_mfma_ok(const float* A, int lda, const float* W, int ldw, int M, int N, int K, int ldo);
int eend_launch_linear_f32_mfma(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldres, float alpha,
                                int res_mode, int act, float* out32, int ldo, int M, int N, int K, hipStream_t stream);
int eend_launch_layernorm_rows_f32(const float* x, const float* gamma, const float* beta, float eps, float* out32, int M, hipStream_t stream);
int eend_launch_convert_step_f32(const float* emb, const float* W, int ldw, const float* pc, float* out32, void* out16, int B, int C,
                                 hipStream_t stream);
int eend_launch_ret_proj_step(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias,
                              float* out, int N, hipStream_t stream);
int eend_launch_dwconv_step(const void* x16, float* cache, const float* w, const float* bn_w, const float* bn_b,
                            const float* bn_mean, const float* bn_var, float eps, void* out16, int B, int D, int k,
                            hipStream_t stream);
int eend_launch_attn_decode(const void* qkv, void* Kc, void* Vc, void* out16, int N, int H, int cap, int t, const int* t_dev,
                            float scale, hipStream_t stream);
int eend_launch_counter_add(int* c, int inc, hipStream_t stream);
int eend_launch_attn_decode_split(const void* qkv, void* Kc, void* Vc, void* out16, float* part, long part_floats, int N, int H, int cap,
                                  const int* t_dev, float scale, hipStream_t stream);
int eend_launch_gather_bn_cast_pad(const float* const* x_ptrs, const int* lens, float pad_value, const float* bn_w,
                                   const float* bn_b, const float* bn_mean, const float* bn_var, float eps, void* out16,
                                   int B, int T, int Tp, int Fin, int Fpad, int apply_bn, hipStream_t stream);

enum FfnEpilogue {
    FFN_EPI_RES_LN = 0,          // out32 = out16 = LN((h W2^T + b2) * alpha + res)
    FFN_EPI_RES_SCALE_LN16 = 1,  // out32 = (h W2^T + b2) * alpha + res ; out16 = LN(out32)
};

struct FfnParams {
    const void* X;      // f16 [M][ldx], K = 256
    const void* W1;     // f16 [F][256]
    const float* b1;    // [F]
    const void* W2;     // f16 [256][F]
    const float* b2;    // [256]
    const float* res;   // f32 [M][256] or null
    const void* res16;  // the out-projection residual as f16 [M][256] instead (A != null, used when res is null); out32 may then be null
    const float* gamma; // LN affine [256]
    const float* beta;
    float* out32;       // f32 [M][256]
    void* out16;        // f16 [M][256]
    void* out16lo;      // optional f16 [M][256]: f16(v - f16(v)) of what out16 holds (the next retention's query path, ret_stream.hip)
    float alpha, eps;
    int M, F, ldx;
    // optional fused producer (null A = off): X = LayerNorm1(A Wo^T + bo + res), which also becomes the residual
    const void* A;      // f16 [M][lda] (attention output)
    const void* Wo;     // f16 [256][256]
    const void* Wo_lo;  // optional f16 [256][256]: f16(W - f16(W)), a second product on the same A fragments (plain PRE kernel only)
    const float* bo;    // [256]
    const float* g1;    // LayerNorm1 affine
    const float* be1;
    float eps1;
    int lda;
    // training forward of a post-norm FFN block (round 5, eend_ffn_train_f16; null hid16 = off): h = drop1(relu(X W1^T + b1)) is ALSO written to
    // hid16 (the saved activation of the backward; its zeros are the ReLU-and-dropout mask), y = drop2(h W2^T + b2) * alpha + res,
    // out32 / out16 = LayerNorm(y), xhat16 = the normalised pre-affine rows, rstat = 1/sigma per row (what eend_layernorm_bwd_f32 reads)
    void* hid16;        // f16 [M][F]
    void* z16;          // f16 [M][F], Swish blocks only: the pre-activation X W1^T + b1 (the backward's swish'(z))
    void* xhat16;       // f16 [M][256]
    float* rstat;       // [M]
    DropSpec drop1, drop2;
    // data-gradient backward of the same block (MODE 4, eend_ffn_bwd_data_bf16; null hidmask = off): X = dY bf16, W1 = W2^T bf16 [F][256],
    // W2 = W1^T bf16 [256][F], hidmask = the saved activations f16 [M][F]; hid16 receives dH = drop1.scale * (dY W2) where hidmask != 0
    // (bf16), out32 = dH W1 * alpha + res (f32, may alias res)
    const void* hidmask;
    int dbg;            // ablation flags, only honoured by -DEEND_FFN_ABLATE builds (perf studies): 1 no output stores, 2 no residual read, 4 no in-loop weight DMA
};
int eend_launch_ffn_fused(const FfnParams& p, int act, int epi, hipStream_t stream);

// ffn_stream.hip: the same row-local layer tail on a packed weight stream (4 waves x 48 rows, wave-private hidden units)
struct FfnStreamParams {
    const void* A;        // mode 1: attention output f16 [M][lda]; mode 0: X f16 [M][lda]
    int lda;
    const void* wstream;  // eend_ffn_stream_pack_f16 output
    const float* bo;      // mode 1: out-projection bias, LayerNorm1 affine
    const float* g1;
    const float* be1;
    float eps1;
    const float* res32;   // residual f32 [M][256] (mode 0: required) ...
    const void* res16;    // ... or, mode 1, f16 [M][256]
    const float* b1;      // [F]
    const float* b2;      // [256]
    const float* gamma;   // final LayerNorm affine
    const float* beta;
    float eps, alpha;
    float* out32;         // may be null in mode 1 / FFN_EPI_RES_LN
    void* out16;
    int wo_lo;            // mode 1 on f32 residual rows only (LO form): the stream carries Wo's f16 remainder behind Wo (ffn_stream_pack with Wo_lo)
    void* out16lo;        // LO form, optional: f16(y - f16(y))
    int M, F;
};
long eend_ffn_stream_nelems(int F, int with_wo);
long eend_ffn_stream_debug_row_cap();                    // test hook (eend_debug_ffn_stream_set), 0 = none
int eend_launch_ffn_stream_pack(const void* Wo, const void* Wo_lo, const void* W1, const void* W2, void* out, int F, int k_permuted, hipStream_t stream);
int eend_launch_ffn_stream(const FfnStreamParams& p, int mode, int act, int epi, hipStream_t stream);
// ffn_train_stream.hip: training forward (tr = 1) / data gradient (tr = 2) of the post-norm ReLU FFN block on a packed weight stream
struct FfnTrainStreamParams {
    const void* X;        // tr 1: x f16 [M][ldx]; tr 2: dY bf16 [M][ldx]
    int ldx;
    const void* wstream;  // eend_ffn_train_stream_pack output (tr 1: of W1, W2 f16; tr 2: of W2^T, W1^T bf16)
    const float* b1;      // tr 1: [F]
    const float* b2;      // tr 1: [256]
    const float* gamma;   // tr 1: LayerNorm affine
    const float* beta;
    const float* res;     // f32 [M][256]: tr 1 the residual, tr 2 the incoming gradient stream
    float alpha, eps;
    float* out32;         // f32 [M][256]: tr 1 LayerNorm output, tr 2 the outgoing gradient stream (may alias res)
    void* out16;          // tr 1: f16 [M][256]
    void* xhat16;         // tr 1: f16 [M][256] normalised pre-affine rows
    float* rstat;         // tr 1: [M] 1/sigma
    void* hid;            // f16 [M][F]: tr 1 written (saved activations), tr 2 read (their zeros are the mask)
    void* dH;             // tr 2: bf16 [M][F], written
    DropSpec drop1, drop2; // tr 1: hidden / output dropout; tr 2: drop1.scale only
    int M, F;
};
long eend_ffn_train_stream_nelems(int F);
bool eend_ffn_train_stream_fits(int M, int F, int ldx);
int eend_launch_ffn_train_stream_pack(const void* W1, const void* W2, void* out, int F, hipStream_t stream);
int eend_launch_ffn_train_stream(const FfnTrainStreamParams& p, int tr, hipStream_t stream);
// spk_stream.hip: x1 = LN11(A Wo1^T + bo1 + res16), O = speaker-axis MHA(x1 Win2^T + bin2) in one launch (C in {3, 6, 12})
struct SpkStreamParams {
    const void* A;        // time-axis attention output f16 [B*C*Tp][lda], row = (b*C + c)*Tp + t
    int lda;
    const void* wstream;  // eend_spk_stream_pack_f16 output
    const float* bo;      // out-projection bias, LayerNorm11 affine
    const float* g1;
    const float* be1;
    float eps1;
    const void* res16;    // residual f16 [B*C*Tp][256]
    void* x16;            // x1 f16 [B*C*Tp][256] (may be res16)
    const float* res32;   // ... or the f32 form of both (round 5): residual f32 in, x1 f32 out (may be res32), no f16 x1
    float* x32;
    const float* bin;     // [768] in-projection bias (q, k, v)
    void* O;              // attention output f16 [B*C*Tp][256] (may be A when lda == 256)
    int B, C, Tp;
    float scale;          // 1/sqrt(dh)
};
long eend_spk_stream_nelems();
int eend_spk_stream_supported(int C, int Tp);
int eend_launch_spk_stream_pack(const void* Wo, const void* Win, void* out, hipStream_t stream);
int eend_launch_spk_stream(const SpkStreamParams& p, hipStream_t stream);
// convert_rows.hip: the decoder input fan-out as a store-shaped kernel (weights stationary)
int eend_launch_convert_fanout_rows(const void* E, const void* W1, const float* pc, float* out32, void* out16, int B, int Tp, int C,
                                    hipStream_t stream);
// conv_stream.hip: look-ahead Conv1d(256 -> 256) + bias + L2 norm on a packed weight stream
struct ConvStreamParams {
    const void* X;        // f16 [nseq][Tp][256]
    const void* wstream;  // eend_conv_stream_pack_f16 output
    const float* bias;    // [256]
    const int* ilens;     // [nseq]: frames >= ilens[seq] read as zero
    float* out32;         // f32 [nseq*Tp][256]
    void* out16;          // f16 [nseq*Tp][256]
    float* inv_norm;      // optional [nseq*Tp]
    int nseq, Tp, ktaps, pad;
};
long eend_conv_stream_nelems(int ktaps);
int eend_conv_stream_supported(int cin, int ktaps, int pad);
int eend_launch_conv_stream_pack(const void* Wr, void* out, int ktaps, hipStream_t stream);
int eend_launch_conv_stream(const ConvStreamParams& p, hipStream_t stream);
// encin.hip: pad_sequence + BatchNorm + cast + input projection + LayerNorm of the encoder input in one launch
struct EncInParams {
    const float* const* x_ptrs;   // device table of B utterance pointers, each (len_b, Fin) f32, 16-byte aligned
    const int* lens;              // device, [B]
    float pad_value;
    const float *bn_w, *bn_b, *bn_mean, *bn_var;
    float bn_eps;
    const void* W;                // f16 [256][ldw], columns >= Fin zero
    int ldw;
    const float *bias, *gamma, *beta;
    float eps;
    float* out32;                 // optional f32 [B*Tp][256]
    void* out16;                  // f16 [B*Tp][256]
    int B, T, Tp, Fin;
};
int eend_encin_supported(int Fin, int Tp, int ldw);
int eend_launch_encin(const EncInParams& p, hipStream_t stream);
// convert_f32.hip: decoder input in f32 on the exact-f32 MFMA (LS-EEND batch forward)
int eend_launch_convert_fanout_f32(const float* emb, const float* W, int ldw, const float* pc, float* out32, void* out16, void* out16lo,
                                   int B, int Tp, int C, hipStream_t stream);
int eend_launch_attn_causal_full(const AttnParams& p, hipStream_t stream);

enum ProjKind { PROJ_ROWMAJOR = 0, PROJ_HEADS = 1, PROJ_HEADS_T = 2, PROJ_HEADS_BOTH = 3 };
struct ProjParams {       // proj.hip: up to four 256-feature output groups
    const void* X;        // f16 [M][ldx], 256 features
    const void* W;        // f16 [N][256]
    const float* bias;    // [N]
    int M, N, ldx, Tp, H;
    int kind[4];          // ProjKind per group
    int is_bf16[4];       // output element type per group (else f16)
    void* out[4];         // ROWMAJOR: [M][ld] (+ column offset applied by the caller); HEADS: [seq][H][Tp][64]; HEADS_T: [seq][H][64][Tp]
    void* out2[4];        // HEADS_BOTH: the transposed copy
    int ld[4];
};
int eend_launch_proj_xres(const ProjParams& p, hipStream_t stream);
// proj_stream.hip (round 6): the same projections on a packed weight stream, wave-owned rows, up to three destinations per 256-feature group
struct ProjStreamParams {
    const void* X; int ldx;   // f16 [M][ldx], 256 features
    const void* wstream;      // eend_launch_proj_stream_pack output for W f16 [N][256]
    const float* bias;        // [N]
    int M, N, Tp, H;
    int kind_a[4];            // 0 none, 1 row-major [M][ld_a] (pointer at the group's first column), 2 head rows [seq][H][Tp][64]
    int bf_a[4];              // element type of out_a (else f16)
    int ld_a[4];
    void* out_a[4];
    void* out_b[4];           // optional: bf16 head rows (second copy)
    void* out_t[4];           // optional: transposed head rows [seq][H][64][Tp]
    int bf_t[4];
};
// gemm_acc_stream.hip (round 6): g[M][256] += A[M][K] Wt^T on a packed weight stream (data gradient of a K -> 256 linear layer)
struct GemmAccStreamParams {
    const void* A; int lda;   // bf16 [M][lda]
    const void* wstream;      // eend_launch_gemm_acc_stream_pack output for Wt bf16 [256][K]
    float* g;                 // f32 [M][256], in place
    int M, K;
};
long eend_gemm_acc_stream_nelems(int K);
int eend_launch_gemm_acc_stream_pack(const void* Wt, int ldw, void* out, int K, hipStream_t stream);
bool eend_gemm_acc_stream_fits(int M, int K, int lda);
int eend_launch_gemm_acc_stream(const GemmAccStreamParams& p, hipStream_t stream);
long eend_proj_stream_nelems(int N);
int eend_launch_proj_stream_pack(const void* W, void* out, int N, hipStream_t stream);
bool eend_proj_stream_fits(const ProjStreamParams& p);
int eend_launch_proj_stream(const ProjStreamParams& p, hipStream_t stream);
int eend_launch_ret_chunk_full(const RetParams& p, hipStream_t stream);

// ret_stream.hip (round 5): the retention with its q / k / v / g projections fused on chip (pass 1: chunk K^T V products, pass 2: rows)
struct RetStreamParams {
    const void* X; int ldx;      // f16 [nseq*Tp][ldx]: the retention's input rows (256 model dims)
    const void* Xlo;             // optional f16 rows, same layout: f16(x - f16(x)) of the f32 residual stream (query path only)
    const void* W;               // packed weight stream (eend_launch_ret_stream_pack)
    const float* bias;           // [1024]: q, k * dk^-0.5, v, g
    void* O; int ldo;            // f16 [nseq*Tp][ldo]: gated, normalised retention rows (input of the out-projection)
    const void* St;              // scan outputs (retention.hip): hi/lo state before each chunk, cross_scale, prescale exponent
    const float* cscale;
    const float* sexp;
    float* kv_ws;                // pass 1: [nseq][4][nc][64][64] f32 chunk products
    int nseq, Tp, L, nc;
    int nkv;                     // pass 1: chunks 0 .. nkv-1
    float gn_eps;
    int has_state_in;            // a state is carried into chunk 0 (long-form walk)
};
long eend_ret_stream_packed_nelems();
bool eend_ret_stream_ok(int L, int Tp, int ldx, int ldo);
int eend_launch_ret_stream_pack(const float* W, void* out, hipStream_t stream);
int eend_launch_ret_stream(const RetStreamParams& p, bool kv, hipStream_t stream);
int eend_launch_ret_state_scan_only(const RetParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------
// training step (backward kernels, optimiser)
// ---------------------------------------------------------------------------------------------------
struct WgradParams {          // wgrad.hip: partial[s][n][k] = sum_{m in split s} A[m][n] * B[m][k]
    const void* A;            // dY, bf16 [M][lda]
    const void* B;            // X, f16 (b_is_f16) or bf16 [M][ldb]
    float* partial;           // f32 [nsplit][N * K]: per split, the output tiles in the accumulator order of wgrad_tr_kernel
                              // (summed and scattered to [N][K] by eend_launch_wgrad_reduce_tiles)
    float* bias_partial;      // optional f32 [nsplit][N]: column sums of A per split (the bias gradient on the side)
    long M;
    int N, K, lda, ldb;
    int nsplit;
    long m_per_split;         // multiple of 64
    int tile;                 // output tile edge: 128 or 256 (N, K, conv_cin multiples of it)
    int b_is_f16;
    int a_blocked, b_blocked; // the operand is stored [M/16][ld/32][16 rows][32 features] (hid / dH of ffn_train_stream.hip; ld = its full row width)
    // Conv1d weight gradient: B rows of k-tile (tap, c_in block) are read at frame t + tap - conv_pad (zero outside [0, ilen))
    int conv, conv_cin, conv_pad, Tp;
    const int* ilens;
};
int eend_launch_wgrad(const WgradParams& p, hipStream_t stream);
int eend_launch_wgrad_reduce_tiles(const float* partial, int tile, int nsplit, int N, int K, int K_out, float* out, int ld_out,
                                   float scale, int accumulate, hipStream_t stream, int group_rows = 0, long group_gap = 0);
int eend_launch_wgrad_reduce(const float* partial, long split_stride, int nsplit, int N, int K, int K_out, float* out,
                             int ld_out, float scale, int accumulate, hipStream_t stream);
int eend_launch_wgrad_reduce_multi(const float* partial, long split_stride, int nsplit, int W, float* out0, float* out1, float* out2,
                                   hipStream_t stream);
int eend_launch_colsum_partial(const void* Y, int ld, long M, int N, int is_bf16, int nsplit, float* partial,
                               hipStream_t stream);
int eend_launch_conv_wgrad_unpermute(const float* tmp, float* g, int cout, int cin, int ktaps, hipStream_t stream);

struct AttnBwdParams {        // attn_bwd.hip
    const void *Q, *Qt;       // bf16 [nseq][H][Tp][64] / [nseq][H][64][Tp]  (Q as the forward kernel saw it)
    const void *K, *Kt;       // bf16, same two layouts
    const void* V;            // bf16 [nseq][H][Tp][64]
    const void* dO;           // bf16 [nseq*Tp][ldo], head h at columns h*64
    const void* dOt;          // bf16 [nseq][H][64][Tp]
    const float* Lse;         // f32 [nseq][H][Tp] from the forward (log2 domain)
    const float* Dh;          // f32 [nseq][H][Tp]: <dO_i, O_i> per head
    void* dQKV;               // bf16 [nseq*Tp][ldg]: dQ at column h*64, dK at 256 + h*64, dV at 512 + h*64
    int nseq, H, Tp, ldo, ldg;
    int mask_delay, kv_len, q_len;
    float scale_log2;         // as the forward
    float sq, sk;             // output scales of dQ and dK
    DropSpec drop;            // the forward's probability dropout
    // retention variant (eend_launch_ret_bwd): chunk length, number of valid chunks, states of ret_bwd_scan_kernel
    int L, nc;
    const void* St;           // bf16 [nseq][H][nc][6][64][64]: Spre hi/lo [kd][hd], R hi/lo [kd][hd], R^T hi/lo [hd][kd]
};
int eend_launch_attn_bwd(const AttnBwdParams& p, hipStream_t stream);
// attn_bwd_fused.hip: windows of up to 512 frames in one launch (Qt / Kt / dOt are not read); the launchers above dispatch to it
bool eend_attn_bwd_fused_ok(const AttnBwdParams& p, bool ret);
int eend_launch_attn_bwd_fused(const AttnBwdParams& p, bool ret, hipStream_t stream);
int eend_launch_ret_bwd(const AttnBwdParams& p, hipStream_t stream);
int eend_launch_ret_bwd_states(const void* Kt, const void* Vt, const void* Qt, const void* dOt, float* kv_ws, float* g_ws, void* St,
                               int nseq, int H, int Tp, int L, int nc, hipStream_t stream);
int eend_launch_heads_transpose(const void* in, int ld, void* out, int nseq, int H, int Tp, hipStream_t stream);
int eend_launch_ret_bwd_states_rm(const void* K, const void* V, const void* Q, const void* dO, int ldo, float* kv_ws, float* g_ws, void* St,
                                  int nseq, int H, int Tp, int L, int nc, hipStream_t stream);

int eend_launch_ln_bwd(const float* g, const void* xhat16, const float* rstd, const float* gamma, float* ds32, void* ds16,
                       float* partial, int* nblocks_out, long M, DropSpec drop, hipStream_t stream);
int eend_launch_head_bce(const float* emb, const float* attr, const float* labels, const int* ilens, const int* ncols,
                         float inv_frames, const float* dlogits_in, float* logits, float* da, float* de, float* loss_partial, int B,
                         int T, int Tp, int C, hipStream_t stream);
int eend_launch_l2norm_bwd(const float* y, const float* dy, const float* inv_norm, void* dx16, int B, int T, int Tp, hipStream_t stream);
int eend_launch_slot_sum(const float* g0, void* gsum16, float* partial, int* nblocks_out, int B, int Tp, int C, hipStream_t stream);
int eend_launch_convert_const(int mode, const float* W, const float* bias, const float* pe, float* pc, const float* dpc, float* dW,
                              float* dbias, int C, hipStream_t stream);
int eend_launch_spk_attn_bwd(const void* qkv16, const void* dO16, void* dqkv16, int B, int C, int Tp, float scale, DropSpec drop,
                             hipStream_t stream);
int eend_launch_bn_colstats(const float* const* x_ptrs, const int* lens, float pad_value, const float* shift, float* partial,
                            int B, int T, int F, int nsplit, hipStream_t stream);
int eend_launch_bn_finalize(int pass, const float* sums, float n, float* mean, float* var, float* run_mean, float* run_var,
                            float momentum, int F, hipStream_t stream);
int eend_launch_bn_bwd(const float* const* x_ptrs, const int* lens, float pad_value, const float* mean, const float* var, float eps,
                       const void* dy16, int ld, float* partial, int B, int T, int Tp, int F, int nsplit, hipStream_t stream);
int eend_launch_attn_rowdot(const void* dO16, const void* O16, float* Dh, int nseq, int H, int Tp, hipStream_t stream);
int eend_launch_emb_consistency_bwd(const void* emb16, const float* tgt, const int* lens, float inv_count, float* de,
                                    int B, int T, int Tp, int D, int C, hipStream_t stream);
typedef eend_prep_entry PrepEntry;
int eend_launch_grad_sumsq(const float* g, long n, float* partial_ws, float* out, hipStream_t stream);
int eend_launch_scalar_sum(const float* partial, long n, float scale, float* out, hipStream_t stream);
int eend_launch_adam(float* p, const float* g, float* m, float* v, long n, const float* hp, const float* gsumsq, float b1, float b2,
                     float eps, hipStream_t stream);
int eend_launch_prep_weights(const PrepEntry* tab, int n_entries, hipStream_t stream);
int eend_launch_grad_accumulate(float* acc, const float* g, float scale, int first, long n, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------
// LS-EEND training step (ls_train.hip, retention_bwd.hip)
// ---------------------------------------------------------------------------------------------------
int eend_launch_swish_drop_fwd(const void* z16, void* a16, long M, int F, DropSpec drop, hipStream_t stream);
int eend_launch_swish_bwd(void* dz16, const void* z16, long M, int F, DropSpec drop, hipStream_t stream);
int eend_launch_layernorm_train(const float* x, const float* gamma, const float* beta, float eps, void* y16, void* xhat16, float* rstd,
                                long M, hipStream_t stream);
int eend_launch_ln_bwd2(const void* g, int g_is_bf16, const void* xhat16, const float* rstd, const float* gamma, float* ds32,
                        int accumulate, void* ds16, float alpha16, float* partial, int* nblocks_out, long M, DropSpec drop,
                        hipStream_t stream);
int eend_launch_resgrad_cast(const float* g, void* ds16, float alpha, float* partial, int* nblocks_out, long M, DropSpec drop,
                             hipStream_t stream);
int eend_launch_glu_dwconv_fwd(const void* P16, const float* w, void* c16, int nseq, int Tp, int Tv, int k, hipStream_t stream);
int eend_launch_bn_colstats16(const void* c16, const float* shift, float* partial, int nseq, int Tp, int Tv, int nblocks,
                              hipStream_t stream);
int eend_launch_bn_local_mean(const float* sum, float n, float* stats, hipStream_t stream);
int eend_launch_bn_merge(const float* stats, int R, float* mean, float* var, float* n_out, float* run_mean, float* run_var, float momentum,
                         hipStream_t stream);
int eend_launch_bn_swish_fwd(const void* c16, const float* mean, const float* var, float eps, const float* gamma, const float* beta,
                             void* s16, long M, hipStream_t stream);
int eend_launch_bn_swish_bwd_stats(const void* ds16, const void* c16, const float* mean, const float* var, float eps, const float* gamma,
                                   const float* beta, float* partial, int nseq, int Tp, int Tv, int nblocks, hipStream_t stream);
int eend_launch_bn_swish_bwd_apply(void* ds16, const void* c16, const float* mean, const float* var, float eps, const float* gamma,
                                   const float* beta, const float* sums, const float* n_dev, int nseq, int Tp, int Tv, hipStream_t stream);
int eend_launch_dwconv_glu_bwd(const void* dc16, const void* P16, const float* w, void* dP16, float* partial, int nseq, int Tp, int Tv,
                               int k, hipStream_t stream);
int eend_launch_ret_gate_gn_bwd(const float* dctx32, const void* g16, int ldg, const void* rhat16, const float* rc, void* dg16, int ldq,
                                void* ot16, int nseq, int Tp, int Tv, hipStream_t stream);
