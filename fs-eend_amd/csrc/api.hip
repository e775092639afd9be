// C-ABI shim: validates arguments, fills the launch structs, forwards to the kernels.
// The exported surface is exactly include/eend_hip.h.
#include "../../include/eend_hip.h"
#include "kernels.h"
#include <stdlib.h>
#include <string.h>

namespace {
GemmParams base_params(const void* A, int lda, const void* W, int ldw, const float* bias, int M, int N, int K) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.ldo = N; p.Tp = 64; p.H = 4; p.dh = 64; p.C = 1;
    p.alpha = 1.0f; p.eps = 1e-5f; p.conv_cin = 64; p.conv_pad = 0; p.drop.scale = 1.0f;
#ifdef EEND_GEMM_ABLATE                               // perf-study build only (tools/gemm_ablate.py)
    const char* d = getenv("EEND_GEMM_DBG");
    p.dbg = d ? atoi(d) : 0;
#endif
    return p;
}
}  // namespace

extern "C" {

int eend_abi_version(void) { return 5; }

int eend_bn_cast_pad_f16(const float* x, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                         const float* bn_var, float eps, void* out_f16, int B, int T, int Tp, int Fin,
                         int Fpad, int apply_bn, void* stream) {
    if (!x || !out_f16 || Tp < T || (Fpad % 64) != 0) return EEND_EINVAL;
    if (apply_bn && (!bn_weight || !bn_bias || !bn_mean || !bn_var)) return EEND_EINVAL;
    return eend_launch_bn_cast_pad(x, bn_weight, bn_bias, bn_mean, bn_var, eps, out_f16, B, T, Tp, Fin, Fpad,
                                   apply_bn, (hipStream_t)stream);
}

int eend_gather_bn_cast_pad_f16(const void* const* x_ptrs, const int* lens, float pad_value, const float* bn_weight,
                                const float* bn_bias, const float* bn_mean, const float* bn_var, float eps,
                                void* out_f16, int B, int T, int Tp, int Fin, int Fpad, int apply_bn, void* stream) {
    if (!x_ptrs || !lens || !out_f16 || Tp < T || (Fpad % 64) != 0) return EEND_EINVAL;
    if (apply_bn && (!bn_weight || !bn_bias || !bn_mean || !bn_var)) return EEND_EINVAL;
    return eend_launch_gather_bn_cast_pad((const float* const*)x_ptrs, lens, pad_value, bn_weight, bn_bias, bn_mean,
                                          bn_var, eps, out_f16, B, T, Tp, Fin, Fpad, apply_bn, (hipStream_t)stream);
}

int eend_linear_f16(const void* A, int lda, const void* W, int ldw, const float* bias, void* out_f16, int ldo,
                    int M, int N, int K, int act, void* stream) {
    if (!A || !W || !out_f16 || (ldo & 3) || act < 0 || act > 2) return EEND_EINVAL;
    if (eend_skinny_ok(A, lda, W, ldw, M, K))        // streaming steps: a few rows, weights spread over the chip
        return eend_launch_skinny_plain(A, lda, W, ldw, bias, out_f16, ldo, M, N, K, act, (hipStream_t)stream);
    if (act == 0 && K == 256 && ldw == 256 && bias && (N % 256) == 0 && N <= 1024 && (ldo & 7) == 0) {
        ProjParams q;                                     // X-resident projection kernel (proj.hip)
        memset(&q, 0, sizeof(q));
        q.X = A; q.ldx = lda; q.W = W; q.bias = bias; q.M = M; q.N = N; q.Tp = 64; q.H = 4;
        for (int g = 0; g < N / 256; ++g) { q.kind[g] = PROJ_ROWMAJOR; q.out[g] = (char*)out_f16 + (size_t)g * 256 * 2; q.ld[g] = ldo; }
        return eend_launch_proj_xres(q, (hipStream_t)stream);
    }
    GemmParams p = base_params(A, lda, W, ldw, bias, M, N, K);
    p.out16 = out_f16; p.ldo = ldo;
    const int epi = act == 1 ? EPI_PLAIN_RELU_F16 : act == 2 ? EPI_PLAIN_SWISH_F16 : EPI_PLAIN_F16;
    return eend_launch_gemm(p, epi, (hipStream_t)stream);
}

int eend_linear_glu_f16(const void* A, int lda, const void* Wi, int ldw, const float* bias_i, void* out_f16,
                        int ldo, int M, int N2, int K, void* stream) {
    if (!A || !Wi || !bias_i || !out_f16 || (ldo & 1) || (N2 & 1)) return EEND_EINVAL;
    if (eend_skinny_ok(A, lda, Wi, ldw, M, K))
        return eend_launch_skinny_glu(A, lda, Wi, ldw, bias_i, out_f16, ldo, M, N2, K, (hipStream_t)stream);
    GemmParams p = base_params(A, lda, Wi, ldw, bias_i, M, N2, K);
    p.out16 = out_f16; p.ldo = ldo;
    return eend_launch_gemm(p, EPI_GLU_F16, (hipStream_t)stream);
}

int eend_inproj_heads_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* Q_bf16,
                           void* K_bf16, void* Vt_bf16, int nseq, int Tp, int H, int dh, int K, void* stream) {
    if (!A || !W || !bias || !Q_bf16 || !K_bf16 || !Vt_bf16) return EEND_EINVAL;
    if (nseq <= 0 || Tp <= 0 || (Tp % 64) != 0 || dh != 64 || H <= 0 || ((H * dh) % 128) != 0) return EEND_EINVAL;
    const int D = H * dh;
    if (K == 256 && ldw == 256 && H == 4) {
        ProjParams q;
        memset(&q, 0, sizeof(q));
        q.X = A; q.ldx = lda; q.W = W; q.bias = bias; q.M = nseq * Tp; q.N = 768; q.Tp = Tp; q.H = H;
        q.kind[0] = PROJ_HEADS; q.out[0] = Q_bf16; q.kind[1] = PROJ_HEADS; q.out[1] = K_bf16;
        q.kind[2] = PROJ_HEADS_T; q.out[2] = Vt_bf16;
        q.is_bf16[0] = q.is_bf16[1] = q.is_bf16[2] = 1;
        return eend_launch_proj_xres(q, (hipStream_t)stream);
    }
    GemmParams p = base_params(A, lda, W, ldw, bias, nseq * Tp, 2 * D, K);
    p.Tp = Tp; p.H = H; p.dh = dh; p.out16 = Q_bf16; p.out16b = K_bf16;
    int rc = eend_launch_gemm(p, EPI_QK_HEADS, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    GemmParams v = base_params(A, lda, (const char*)W + (size_t)2 * D * ldw * 2, ldw, bias + 2 * D, nseq * Tp, D, K);
    v.Tp = Tp; v.H = H; v.dh = dh; v.out16 = Vt_bf16;
    return eend_launch_gemm(v, EPI_VT_HEADS, (hipStream_t)stream);
}

int eend_linear_res_ln_f16(const void* A, int lda, const void* W, int ldw, const float* bias, const float* res,
                           float alpha, const float* gamma, const float* beta, float eps, float* out_f32,
                           void* out_f16, int M, int K, void* stream) {
    if (!A || !W || (!out_f32 && !out_f16) || ((gamma == nullptr) != (beta == nullptr))) return EEND_EINVAL;
    if (out_f32 && eend_skinny_ok(A, lda, W, ldw, M, K))
        return eend_launch_skinny_res(A, lda, W, ldw, bias, res, alpha, gamma, beta, eps, out_f32, out_f16, M, K, 1, (hipStream_t)stream);
    GemmParams p = base_params(A, lda, W, ldw, bias, M, 256, K);
    p.res = res; p.alpha = alpha; p.gamma = gamma; p.beta = beta; p.eps = eps; p.out32 = out_f32; p.out16 = out_f16;
    return eend_launch_gemm(p, EPI_RES_LN, (hipStream_t)stream);
}

int eend_linear_res16_ln_f16(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res_f16,
                             float alpha, const float* gamma, const float* beta, float eps, float* out_f32,
                             void* out_f16, int M, int K, void* stream) {
    if (!A || !W || !res_f16 || !out_f16 || !gamma || !beta) return EEND_EINVAL;
    GemmParams p = base_params(A, lda, W, ldw, bias, M, 256, K);
    p.res16 = res_f16; p.alpha = alpha; p.gamma = gamma; p.beta = beta; p.eps = eps; p.out32 = out_f32; p.out16 = out_f16;
    return eend_launch_gemm(p, EPI_RES_LN, (hipStream_t)stream);
}

int eend_linear_res_scale_ln16_f16(const void* A, int lda, const void* W, int ldw, const float* bias,
                                   const float* res, float alpha, const float* gamma, const float* beta, float eps,
                                   float* out_f32, void* out_f16, int M, int K, void* stream) {
    if (!A || !W || !out_f32 || !out_f16 || !gamma || !beta) return EEND_EINVAL;
    if (eend_skinny_ok(A, lda, W, ldw, M, K))
        return eend_launch_skinny_res(A, lda, W, ldw, bias, res, alpha, gamma, beta, eps, out_f32, out_f16, M, K, 2, (hipStream_t)stream);
    GemmParams p = base_params(A, lda, W, ldw, bias, M, 256, K);
    p.res = res; p.alpha = alpha; p.gamma = gamma; p.beta = beta; p.eps = eps; p.out32 = out_f32; p.out16 = out_f16;
    return eend_launch_gemm(p, EPI_RES_SCALE_LN16, (hipStream_t)stream);
}

int eend_ffn_fused_f16(const void* X, int ldx, const void* W1, const float* b1, const void* W2, const float* b2,
                       const float* res, float alpha, const float* gamma, const float* beta, float eps,
                       float* out_f32, void* out_f16, int M, int F, int act, int residual_stream_unnormalised,
                       void* stream) {
    FfnParams p;
    memset(&p, 0, sizeof(p));
    p.X = X; p.ldx = ldx; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.res = res; p.alpha = alpha; p.gamma = gamma;
    p.beta = beta; p.eps = eps; p.out32 = out_f32; p.out16 = out_f16; p.M = M; p.F = F; p.dbg = 0;
#ifdef EEND_FFN_ABLATE
    if (const char* d = getenv("EEND_FFN_DBG")) p.dbg = atoi(d);
#endif
    return eend_launch_ffn_fused(p, act, residual_stream_unnormalised ? FFN_EPI_RES_SCALE_LN16 : FFN_EPI_RES_LN,
                                 (hipStream_t)stream);
}

int eend_attnout_ffn_fused_f16(const void* A, int lda, const void* Wo, const float* bo, const float* res,
                               const float* g1, const float* be1, float eps1, const void* W1, const float* b1,
                               const void* W2, const float* b2, const float* g2, const float* be2, float eps2,
                               float* out_f32, void* out_f16, void* out_lo_f16, const void* Wo_lo, int M, int F, void* stream) {
    if (!A || !Wo || !bo || !g1 || !be1) return EEND_EINVAL;
    FfnParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.Wo = Wo; p.Wo_lo = Wo_lo; p.bo = bo; p.g1 = g1; p.be1 = be1; p.eps1 = eps1;
    p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.res = res; p.alpha = 1.0f; p.gamma = g2; p.beta = be2; p.eps = eps2;
    p.out32 = out_f32; p.out16 = out_f16; p.out16lo = out_lo_f16; p.M = M; p.F = F;
    return eend_launch_ffn_fused(p, 1, FFN_EPI_RES_LN, (hipStream_t)stream);
}

int eend_attnout_ffn_fused_res16_f16(const void* A, int lda, const void* Wo, const float* bo, const void* res_f16,
                                     const float* g1, const float* be1, float eps1, const void* W1, const float* b1,
                                     const void* W2, const float* b2, const float* g2, const float* be2, float eps2,
                                     float* out_f32, void* out_f16, int M, int F, void* stream) {
    if (!A || !Wo || !bo || !g1 || !be1 || !res_f16) return EEND_EINVAL;
    FfnParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.Wo = Wo; p.bo = bo; p.g1 = g1; p.be1 = be1; p.eps1 = eps1;
    p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.res16 = res_f16; p.alpha = 1.0f; p.gamma = g2; p.beta = be2; p.eps = eps2;
    p.out32 = out_f32; p.out16 = out_f16; p.M = M; p.F = F;
    return eend_launch_ffn_fused(p, 1, FFN_EPI_RES_LN, (hipStream_t)stream);
}

int eend_convert_fanout_f32(const float* E_f32, const float* W_f32, int ldw, const float* pc, float* out_f32, void* out_f16,
                            void* out_lo_f16, int B, int Tp, int C, void* stream) {
    return eend_launch_convert_fanout_f32(E_f32, W_f32, ldw, pc, out_f32, out_f16, out_lo_f16, B, Tp, C, (hipStream_t)stream);
}

int eend_ffn_stream_elems(int F, int with_wo) { return (int)eend_ffn_stream_nelems(F, with_wo); }

int eend_ffn_stream_pack_f16(const void* Wo, const void* W1, const void* W2, void* stream_out, int F, void* stream) {
    return eend_launch_ffn_stream_pack(Wo, nullptr, W1, W2, stream_out, F, Wo ? 1 : 0, (hipStream_t)stream);
}

int eend_ffn_stream_pack_lo_f16(const void* Wo, const void* Wo_lo, const void* W1, const void* W2, void* stream_out, int F, void* stream) {
    if (!Wo || !Wo_lo) return EEND_EINVAL;
    return eend_launch_ffn_stream_pack(Wo, Wo_lo, W1, W2, stream_out, F, 1, (hipStream_t)stream);
}

// The stream kernels address rows with 32-bit buffer offsets (and prefetch one grid of tiles ahead): a launch takes at most
// eend_ffn_stream_max_rows(lda) rows.  Larger batches are served in several launches over row ranges (rows are independent; the range
// size is a multiple of both tile sizes, so every row sits at the same tile position as in one launch).  ADVICE r04: the un-chunked
// entry returned EEND_EINVAL beyond ~2.03 M rows (B = 512, C = 12, Tp = 512) where the kernels it replaced had no limit.
int eend_ffn_stream_max_rows(int lda) {
    const long row_bytes = (long)(lda > 512 ? lda : 512) * 2;       // widest row the kernel addresses: f32 residual / output rows are 1 KB
    long m = ((1L << 31) - 1) / row_bytes - 65536 - 1;
    const long cap = eend_ffn_stream_debug_row_cap();               // tests: exercise the multi-launch path at small sizes
    if (cap > 0 && cap < m) m = cap;
    m = m / 384 * 384;
    return m > 0 ? (int)m : 0;
}

static int ffn_stream_chunked(FfnStreamParams p, int mode, int act, int epi, hipStream_t stream) {
    const int cap = eend_ffn_stream_max_rows(p.lda);
    if (cap <= 0 || p.M <= 0) return EEND_EINVAL;
    const char* A = (const char*)p.A;
    const char* r16 = (const char*)p.res16;
    const float* r32 = p.res32;
    float* o32 = p.out32;
    char* o16 = (char*)p.out16;
    char* o16l = (char*)p.out16lo;
    const int M = p.M;
    for (int m0 = 0; m0 < M; m0 += cap) {
        FfnStreamParams q = p;
        q.M = M - m0 < cap ? M - m0 : cap;
        q.A = A + (size_t)m0 * p.lda * 2;
        q.res16 = r16 ? r16 + (size_t)m0 * 512 : nullptr;
        q.res32 = r32 ? r32 + (size_t)m0 * 256 : nullptr;
        q.out32 = o32 ? o32 + (size_t)m0 * 256 : nullptr;
        q.out16 = o16 + (size_t)m0 * 512;
        q.out16lo = o16l ? o16l + (size_t)m0 * 512 : nullptr;
        const int rc = eend_launch_ffn_stream(q, mode, act, epi, stream);
        if (rc != EEND_OK) return rc;
    }
    return EEND_OK;
}

int eend_ffn_stream_f16(const void* X, int ldx, const void* wstream, const float* b1, const float* b2,
                        const float* res, float alpha, const float* gamma, const float* beta, float eps,
                        float* out_f32, void* out_f16, int M, int F, int act, int residual_stream_unnormalised,
                        void* stream) {
    FfnStreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = X; p.lda = ldx; p.wstream = wstream; p.b1 = b1; p.b2 = b2; p.res32 = res; p.alpha = alpha; p.gamma = gamma;
    p.beta = beta; p.eps = eps; p.out32 = out_f32; p.out16 = out_f16; p.M = M; p.F = F;
    return ffn_stream_chunked(p, 0, act, residual_stream_unnormalised ? FFN_EPI_RES_SCALE_LN16 : FFN_EPI_RES_LN, (hipStream_t)stream);
}

int eend_attnout_ffn_stream_f16(const void* A, int lda, const void* wstream, const float* bo, const float* res,
                                const void* res_f16, const float* g1, const float* be1, float eps1, const float* b1,
                                const float* b2, const float* g2, const float* be2, float eps2,
                                float* out_f32, void* out_f16, int M, int F, void* stream) {
    if ((res != nullptr) == (res_f16 != nullptr)) return EEND_EINVAL;
    FfnStreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.wstream = wstream; p.bo = bo; p.g1 = g1; p.be1 = be1; p.eps1 = eps1; p.res32 = res; p.res16 = res_f16;
    p.b1 = b1; p.b2 = b2; p.alpha = 1.0f; p.gamma = g2; p.beta = be2; p.eps = eps2; p.out32 = out_f32; p.out16 = out_f16;
    p.M = M; p.F = F;
    return ffn_stream_chunked(p, 1, 1, FFN_EPI_RES_LN, (hipStream_t)stream);
}

int eend_attnout_ffn_stream_lo_f16(const void* A, int lda, const void* wstream, const float* bo, const float* res, const float* g1,
                                   const float* be1, float eps1, const float* b1, const float* b2, const float* g2, const float* be2,
                                   float eps2, float* out_f32, void* out_f16, void* out_lo_f16, int M, int F, void* stream) {
    if (!res) return EEND_EINVAL;
    FfnStreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.wstream = wstream; p.bo = bo; p.g1 = g1; p.be1 = be1; p.eps1 = eps1; p.res32 = res; p.wo_lo = 1;
    p.b1 = b1; p.b2 = b2; p.alpha = 1.0f; p.gamma = g2; p.beta = be2; p.eps = eps2; p.out32 = out_f32; p.out16 = out_f16; p.out16lo = out_lo_f16;
    p.M = M; p.F = F;
    return ffn_stream_chunked(p, 1, 1, FFN_EPI_RES_LN, (hipStream_t)stream);
}

int eend_conv_stream_elems(int ktaps) { return (int)eend_conv_stream_nelems(ktaps); }

int eend_conv_stream_ok(int cin, int ktaps, int pad) { return eend_conv_stream_supported(cin, ktaps, pad); }

int eend_conv_stream_pack_f16(const void* Wr, void* stream_out, int ktaps, void* stream) {
    return eend_launch_conv_stream_pack(Wr, stream_out, ktaps, (hipStream_t)stream);
}

int eend_conv1d_l2norm_stream_f16(const void* X, const void* wstream, const float* bias, const int* ilens, float* out_f32,
                                  void* out_f16, int nseq, int Tp, int ktaps, int pad, void* stream) {
    ConvStreamParams p;
    memset(&p, 0, sizeof(p));
    p.X = X; p.wstream = wstream; p.bias = bias; p.ilens = ilens; p.out32 = out_f32; p.out16 = out_f16; p.nseq = nseq; p.Tp = Tp;
    p.ktaps = ktaps; p.pad = pad;
    return eend_launch_conv_stream(p, (hipStream_t)stream);
}

int eend_encoder_input_ok(int Fin, int Tp, int ldw) { return eend_encin_supported(Fin, Tp, ldw); }

int eend_encoder_input_f16(const float* const* x_ptrs, const int* lens, float pad_value, const float* bn_w, const float* bn_b,
                           const float* bn_mean, const float* bn_var, float bn_eps, const void* W_f16, int ldw, const float* bias,
                           const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, int B, int T, int Tp,
                           int Fin, void* stream) {
    EncInParams p;
    memset(&p, 0, sizeof(p));
    p.x_ptrs = x_ptrs; p.lens = lens; p.pad_value = pad_value; p.bn_w = bn_w; p.bn_b = bn_b; p.bn_mean = bn_mean; p.bn_var = bn_var;
    p.bn_eps = bn_eps; p.W = W_f16; p.ldw = ldw; p.bias = bias; p.gamma = gamma; p.beta = beta; p.eps = eps; p.out32 = out_f32;
    p.out16 = out_f16; p.B = B; p.T = T; p.Tp = Tp; p.Fin = Fin;
    return eend_launch_encin(p, (hipStream_t)stream);
}

int eend_spk_stream_elems(void) { return (int)eend_spk_stream_nelems(); }

int eend_spk_stream_ok(int C, int Tp) { return eend_spk_stream_supported(C, Tp); }

int eend_spk_stream_pack_f16(const void* Wo, const void* W_in, void* stream_out, void* stream) {
    return eend_launch_spk_stream_pack(Wo, W_in, stream_out, (hipStream_t)stream);
}

int eend_attnout_spk_stream_f16(const void* A, int lda, const void* wstream, const float* bo, const void* res_f16,
                                const float* g1, const float* be1, float eps1, void* x_f16, const float* b_in, void* O_f16,
                                int B, int C, int Tp, float scale, void* stream) {
    SpkStreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.wstream = wstream; p.bo = bo; p.res16 = res_f16; p.g1 = g1; p.be1 = be1; p.eps1 = eps1;
    p.x16 = x_f16; p.bin = b_in; p.O = O_f16; p.B = B; p.C = C; p.Tp = Tp; p.scale = scale;
    return eend_launch_spk_stream(p, (hipStream_t)stream);
}

int eend_attnout_spk_stream_res32_f16(const void* A, int lda, const void* wstream, const float* bo, const float* res_f32,
                                      const float* g1, const float* be1, float eps1, float* x_f32, const float* b_in, void* O_f16,
                                      int B, int C, int Tp, float scale, void* stream) {
    SpkStreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.wstream = wstream; p.bo = bo; p.res32 = res_f32; p.g1 = g1; p.be1 = be1; p.eps1 = eps1;
    p.x32 = x_f32; p.bin = b_in; p.O = O_f16; p.B = B; p.C = C; p.Tp = Tp; p.scale = scale;
    return eend_launch_spk_stream(p, (hipStream_t)stream);
}

int eend_emb_consistency_f32(const float* emb, const float* labels, const int* lens, float inv_count,
                             float* partial_ws, float* out, int B, int T, int Tp, int D, int C, void* stream) {
    return eend_launch_emb_consistency(emb, labels, lens, inv_count, partial_ws, out, B, T, Tp, D, C, (hipStream_t)stream);
}

int eend_activity_median_u8(const float* pred, int ld, int T, int S, float threshold, int median,
                            unsigned char* act, void* stream) {
    return eend_launch_activity_median(pred, ld, T, S, threshold, median, act, (hipStream_t)stream);
}

int eend_activity_segments_i32(const unsigned char* act, int T, int S, int* changes, int* counts, int cap,
                               void* stream) {
    return eend_launch_segments(act, T, S, changes, counts, cap, (hipStream_t)stream);
}

int eend_der_counters_u64(const float* pred, int ldp, const float* label, int ldl, int T, int C, int label_delay,
                          unsigned long long* counters, void* stream) {
    return eend_launch_der_counters(pred, ldp, label, ldl, T, C, label_delay, counters, (hipStream_t)stream);
}

int eend_stft_logmel23_f32(const float* y, long len, long first, int n_frames, const float* dft, const float* melT,
                           float* out, void* stream) {
    return eend_launch_stft_logmel(y, len, first, n_frames, dft, melT, out, (hipStream_t)stream);
}

int eend_feature_meannorm_f32(const float* Y, float* out, int T, int F, int mode, void* stream) {
    return eend_launch_colnorm(Y, out, T, F, mode, (hipStream_t)stream);
}

int eend_splice_subsample_f32(const float* Y, int T, int F, int ctx, int sub, float* out, void* stream) {
    return eend_launch_splice_subsample(Y, T, F, ctx, sub, out, (hipStream_t)stream);
}

int eend_pit_cost_f64(const float* y, const float* labels, int B, int T, int C, double* cost, void* stream) {
    return eend_launch_pit_cost(y, labels, B, T, C, cost, (hipStream_t)stream);
}

int eend_pit_assign_i32(const double* cost, const int* nspk, int B, int C, int* perm, double* loss, void* stream) {
    return eend_launch_pit_assign(cost, nspk, B, C, perm, loss, (hipStream_t)stream);
}

int eend_retention_proj_f16(const void* A, int lda, const void* Wqkvg, int ldw, const float* bias, void* Q, void* K,
                            void* Kt, void* Vt, void* G, int nseq, int Tp, int H, int dh, int Kdim, void* stream) {
    if (!A || !Wqkvg || !bias || !Q || !K || !Kt || !Vt || !G) return EEND_EINVAL;
    if (nseq <= 0 || Tp <= 0 || (Tp % 64) != 0 || dh != 64 || H <= 0 || ((H * dh) % 128) != 0) return EEND_EINVAL;
    const int D = H * dh, M = nseq * Tp;
    if (Kdim == 256 && ldw == 256 && H == 4) {
        ProjParams q;
        memset(&q, 0, sizeof(q));
        q.X = A; q.ldx = lda; q.W = Wqkvg; q.bias = bias; q.M = M; q.N = 1024; q.Tp = Tp; q.H = H;
        q.kind[0] = PROJ_HEADS; q.out[0] = Q;
        q.kind[1] = PROJ_HEADS_BOTH; q.out[1] = K; q.out2[1] = Kt;
        q.kind[2] = PROJ_HEADS_T; q.out[2] = Vt;
        q.kind[3] = PROJ_ROWMAJOR; q.out[3] = G; q.ld[3] = D;
        return eend_launch_proj_xres(q, (hipStream_t)stream);
    }
    const char* W = (const char*)Wqkvg;
    const size_t rowb = (size_t)ldw * 2;
    GemmParams qk = base_params(A, lda, W, ldw, bias, M, 2 * D, Kdim);                      // rows [0, 2D): q, k
    qk.Tp = Tp; qk.H = H; qk.dh = dh; qk.out16 = Q; qk.out16b = K;
    int rc = eend_launch_gemm(qk, EPI_QK_HEADS_F16, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    GemmParams kv = base_params(A, lda, W + (size_t)D * rowb, ldw, bias + D, M, 2 * D, Kdim);  // rows [D, 3D): k, v
    kv.Tp = Tp; kv.H = H; kv.dh = dh; kv.out16 = Kt; kv.out16b = Vt;
    rc = eend_launch_gemm(kv, EPI_KTVT_HEADS_F16, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    GemmParams g = base_params(A, lda, W + (size_t)3 * D * rowb, ldw, bias + 3 * D, M, D, Kdim);  // rows [3D, 4D): g
    g.out16 = G; g.ldo = D;
    return eend_launch_gemm(g, EPI_PLAIN_F16, (hipStream_t)stream);
}

int eend_retention_chunk_f16(const void* Q, const void* K, const void* Kt, const void* Vt, const void* G,
                             void* O_f16, void* St_ws, float* kv_ws, float* cscale_ws, float* sexp_ws, int nseq, int H,
                             int Tp, int L, int ldo, int ldg, float gn_eps, int T_valid, const float* state_in,
                             float* state_out, void* stream) {
    if (!Q || !K || !Kt || !Vt || !G || !O_f16 || !St_ws || !kv_ws || !cscale_ws || !sexp_ws || L <= 0) return EEND_EINVAL;
    RetParams p;
    memset(&p, 0, sizeof(p));
    p.Q = Q; p.K = K; p.Kt = Kt; p.Vt = Vt; p.G = G; p.O = O_f16; p.St = St_ws; p.cscale = cscale_ws; p.sexp = sexp_ws; p.kv_ws = kv_ws; p.kv_ws = kv_ws;
    // chunk sizes that fit on chip (500 in every shipped config) take the chunk-resident kernel, one block per
    // (chunk, head, sequence): there, chunks that start at or beyond T_valid (pure slab padding) are skipped
    // altogether.  The tiled kernel's waves span chunk boundaries, so it always sees every chunk.
    const bool use_full = L <= 512 && (L & 3) == 0 && (ldo & 7) == 0;
    const int Tv = (use_full && T_valid > 0 && T_valid < Tp) ? T_valid : Tp;
    p.nseq = nseq; p.H = H; p.Tp = Tp; p.L = L; p.nc = (Tv + L - 1) / L; p.ldo = ldo; p.ldg = ldg; p.gn_eps = gn_eps;
    p.state_in = state_in; p.state_out = state_out;
    int rc = eend_launch_ret_state_scan(p, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    if (use_full) return eend_launch_ret_chunk_full(p, (hipStream_t)stream);
    return eend_launch_ret_chunk(p, (hipStream_t)stream);
}

int eend_retention_stream_elems(void) { return (int)eend_ret_stream_packed_nelems(); }

int eend_retention_stream_ok(int L, int Tp, int ldx, int ldo) { return eend_ret_stream_ok(L, Tp, ldx, ldo) ? 1 : 0; }

int eend_retention_stream_pack_f16(const float* Wqkvg_f32, void* packed_out, void* stream) {
    return eend_launch_ret_stream_pack(Wqkvg_f32, packed_out, (hipStream_t)stream);
}

int eend_retention_stream_f16(const void* X_f16, int ldx, const void* Xlo_f16, const void* W_packed, const float* bias, void* O_f16, int ldo,
                              void* St_ws, float* kv_ws, float* cscale_ws, float* sexp_ws, int nseq, int Tp, int L, float gn_eps,
                              int T_valid, const float* state_in, float* state_out, void* stream) {
    if (!X_f16 || !W_packed || !bias || !O_f16 || !St_ws || !kv_ws || !cscale_ws || !sexp_ws || L <= 0 || nseq <= 0 || nseq > 65535)
        return EEND_EINVAL;
    if (!eend_ret_stream_ok(L, Tp, ldx, ldo)) return EEND_EINVAL;
    const int Tv = (T_valid > 0 && T_valid < Tp) ? T_valid : Tp;
    const int nc = (Tv + L - 1) / L;
    RetStreamParams q;
    memset(&q, 0, sizeof(q));
    q.X = X_f16; q.ldx = ldx; q.Xlo = Xlo_f16; q.W = W_packed; q.bias = bias; q.O = O_f16; q.ldo = ldo;
    q.St = St_ws; q.cscale = cscale_ws; q.sexp = sexp_ws; q.kv_ws = kv_ws;
    q.nseq = nseq; q.Tp = Tp; q.L = L; q.nc = nc; q.nkv = state_out ? nc : nc - 1; q.gn_eps = gn_eps; q.has_state_in = state_in ? 1 : 0;
    int rc = eend_launch_ret_stream(q, true, (hipStream_t)stream);           // pass 1: chunk K^T V products
    if (rc != EEND_OK) return rc;
    RetParams p;
    memset(&p, 0, sizeof(p));
    p.St = St_ws; p.cscale = cscale_ws; p.sexp = sexp_ws; p.kv_ws = kv_ws;
    p.nseq = nseq; p.H = 4; p.Tp = Tp; p.L = L; p.nc = nc; p.gn_eps = gn_eps; p.state_in = state_in; p.state_out = state_out;
    rc = eend_launch_ret_state_scan_only(p, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_ret_stream(q, false, (hipStream_t)stream);             // pass 2: the rows
}

int eend_retention_proj_step_f32(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* Wqkvg,
                                 const float* bias, float* qkvg_f32, int N, void* stream) {
    return eend_launch_ret_proj_step(x, ln_gamma, ln_beta, ln_eps, Wqkvg, bias, qkvg_f32, N, (hipStream_t)stream);
}

int eend_convert_fanout_step_f32(const float* emb_f32, const float* W_f32, int ldw, const float* pc, float* out_f32, void* out_f16,
                                 int B, int C, void* stream) {
    return eend_launch_convert_step_f32(emb_f32, W_f32, ldw, pc, out_f32, out_f16, B, C, (hipStream_t)stream);
}

int eend_retention_step_f32(const float* qkvg, float* kv_state, const float* scale_in, float* scale_out, void* out_f16, float* out_f32,
                            int N, int H, float gn_eps, void* stream) {
    if (!qkvg || !kv_state || !scale_in || !scale_out || (!out_f16 && !out_f32)) return EEND_EINVAL;
    return eend_launch_ret_step_f32in(qkvg, kv_state, scale_in, scale_out, out_f16, out_f32, N, H, gn_eps, (hipStream_t)stream);
}

int eend_linear_step_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* out_f32, int ldo, int M, int N,
                         int K, int act, void* stream) {
    return eend_launch_skinny_plain_f32(A, lda, W, ldw, bias, out_f32, ldo, M, N, K, act, (hipStream_t)stream);
}

int eend_linear_res_ln_step_f32(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, float alpha,
                                const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, int M, int K,
                                void* stream) {
    if (!gamma || !beta) return EEND_EINVAL;
    return eend_launch_skinny_res_f32(A, lda, W, ldw, bias, res, alpha, gamma, beta, eps, out_f32, out_f16, M, K, 1, (hipStream_t)stream);
}

int eend_linear_res_scale_ln_step_f32(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, float alpha,
                                      const float* gamma, const float* beta, float eps, float* out_f32, float* ln_out_f32,
                                      void* ln_out_f16, int M, int K, void* stream) {
    if (!gamma || !beta || !out_f32) return EEND_EINVAL;
    return eend_launch_skinny_res_f32(A, lda, W, ldw, bias, res, alpha, gamma, beta, eps, out_f32, ln_out_f16, M, K, 2, (hipStream_t)stream,
                                      ln_out_f32);
}

int eend_layernorm_rows_f32(const float* x, const float* gamma, const float* beta, float eps, float* out_f32, int M, void* stream) {
    return eend_launch_layernorm_rows_f32(x, gamma, beta, eps, out_f32, M, (hipStream_t)stream);
}

int eend_l2norm_rows_f32(const float* x, float* y, int rows, void* stream) {
    return eend_launch_l2norm_rows_f32(x, y, rows, (hipStream_t)stream);
}

int eend_spk_attn_step_f32(const float* qkv, float* out_f32, int B, int C, float scale, void* stream) {
    return eend_launch_spk_attn_step_f32(qkv, out_f32, B, C, scale, (hipStream_t)stream);
}

int eend_layernorm_f16(const float* x, const float* gamma, const float* beta, float eps, void* out_f16, int M,
                       int D, void* stream) {
    if (!x || !gamma || !beta || !out_f16) return EEND_EINVAL;
    return eend_launch_layernorm_f16(x, gamma, beta, eps, out_f16, M, D, (hipStream_t)stream);
}

int eend_dwconv_bn_swish_f16(const void* x_f16, const float* w, const float* bn_weight, const float* bn_bias,
                             const float* bn_mean, const float* bn_var, float eps, void* out_f16, int nseq, int Tp,
                             int D, int k, const void* halo_f16, void* stream) {
    if (!x_f16 || !w || !bn_weight || !bn_bias || !bn_mean || !bn_var || !out_f16) return EEND_EINVAL;
    return eend_launch_dwconv_bn_swish(x_f16, w, bn_weight, bn_bias, bn_mean, bn_var, eps, out_f16, nseq, Tp, D, k, halo_f16,
                                       (hipStream_t)stream);
}

int eend_linear_res_scale_f16(const void* A, int lda, const void* W, int ldw, const float* bias,
                              const float* res, float alpha, float* out_f32, void* out_f16, int M, int K,
                              void* stream) {
    if (!A || !W || (!out_f32 && !out_f16)) return EEND_EINVAL;
    if (eend_skinny_ok(A, lda, W, ldw, M, K))
        return eend_launch_skinny_res(A, lda, W, ldw, bias, res, alpha, nullptr, nullptr, 0.f, out_f32, out_f16, M, K, 0, (hipStream_t)stream);
    GemmParams p = base_params(A, lda, W, ldw, bias, M, 256, K);
    p.res = res; p.alpha = alpha; p.out32 = out_f32; p.out16 = out_f16;
    return eend_launch_gemm(p, EPI_RES_SCALE, (hipStream_t)stream);
}

int eend_conv1d_l2norm_f16(const void* X, const void* Wr, const float* bias, const int* ilens, float* out_f32,
                           void* out_f16, int nseq, int Tp, int cin, int ktaps, int pad, void* stream) {
    if (!X || !Wr || !ilens || (!out_f32 && !out_f16)) return EEND_EINVAL;
    if (nseq <= 0 || Tp <= 0 || (Tp % 64) != 0 || cin <= 0 || (cin % 64) != 0 || ktaps <= 0 || pad < 0 ||
        pad >= ktaps)
        return EEND_EINVAL;
    GemmParams p = base_params(X, cin, Wr, ktaps * cin, bias, nseq * Tp, 256, ktaps * cin);
    p.Tp = Tp; p.ilens = ilens; p.conv_cin = cin; p.conv_pad = pad; p.out32 = out_f32; p.out16 = out_f16;
    return eend_launch_gemm(p, EPI_L2NORM, (hipStream_t)stream);
}

int eend_convert_fanout_f16(const void* E, const void* W1, const float* pc, float* out_f32, void* out_f16,
                            int B, int Tp, int C, void* stream) {
    if (!E || !W1 || !pc || !out_f16 || B <= 0 || C <= 0 || Tp <= 0) return EEND_EINVAL;     // out_f32 may be NULL (f16 stream only)
    if (C <= 32 && (long)B * Tp * 512 < (1L << 31))
        return eend_launch_convert_fanout_rows(E, W1, pc, out_f32, out_f16, B, Tp, C, (hipStream_t)stream);
    GemmParams p = base_params(E, 256, W1, 256, nullptr, B * Tp, 256, 256);
    p.Tp = Tp; p.C = C; p.pc = pc; p.out32 = out_f32; p.out16 = out_f16;
    return eend_launch_gemm(p, EPI_CONVERT, (hipStream_t)stream);
}

int eend_attn_causal_bf16(const void* Q, const void* K, const void* Vt, void* O_f16, int nseq, int H, int Tp,
                          int ldo, int mask_delay, int kv_len, float scale, void* stream) {
    if (!Q || !K || !Vt || !O_f16 || nseq > 65535 || H > 65535) return EEND_EINVAL;
    AttnParams p;
    p.Q = Q; p.K = K; p.Vt = Vt; p.O = O_f16; p.nseq = nseq; p.H = H; p.Tp = Tp; p.ldo = ldo;
    p.mask_delay = mask_delay; p.kv_len = kv_len; p.scale_log2 = scale * 1.4426950408889634f; p.Lse = nullptr;
    p.drop = DropSpec{0u, 0u, 1.0f};
    return eend_launch_attn_causal(p, (hipStream_t)stream);
}

int eend_inproj_attn_packed_elems(void) { return (int)eend_inproj_attn_packed_nelems(); }

int eend_inproj_attn_pack_f16(const void* W_in, void* packed_out, void* stream) {
    return eend_launch_inproj_attn_pack(W_in, packed_out, (hipStream_t)stream);
}

int eend_inproj_attn_causal_packed_f16(const void* X_f16, int ldx, const void* W_packed, const float* b_in, void* O_f16,
                                       int nseq, int H, int Tp, int ldo, int mask_delay, int kv_len, void* stream) {
    if (!X_f16 || !W_packed || !b_in || !O_f16 || nseq > 16383) return EEND_EINVAL;
    InprojAttnParams p;
    p.X = X_f16; p.ldx = ldx; p.W = W_packed; p.bias = b_in; p.O = O_f16;
    p.nseq = nseq; p.H = H; p.Tp = Tp; p.ldo = ldo; p.mask_delay = mask_delay; p.kv_len = kv_len;
    return eend_launch_inproj_attn_stream(p, (hipStream_t)stream);
}

int eend_inproj_attn_long_scratch_elems(int nseq, int Tp, int mask_delay, int kv_len, long long* part_f16_elems, long long* lse_f32_elems) {
    if (!part_f16_elems || !lse_f32_elems) return EEND_EINVAL;
    long a = 0, b = 0;
    const int rc = eend_inproj_attn_long_scratch(nseq, Tp, mask_delay, kv_len, &a, &b);
    *part_f16_elems = a; *lse_f32_elems = b;
    return rc;
}

int eend_inproj_attn_causal_long_f16(const void* X_f16, int ldx, const void* W_packed, const float* b_in, void* O_f16, void* part_f16,
                                     float* lse_f32, int nseq, int H, int Tp, int ldo, int mask_delay, int kv_len, void* stream) {
    if (!X_f16 || !W_packed || !b_in || !O_f16 || !lse_f32 || nseq > 16383) return EEND_EINVAL;
    InprojAttnParams p;
    memset(&p, 0, sizeof(p));
    p.X = X_f16; p.ldx = ldx; p.W = W_packed; p.bias = b_in; p.O = O_f16; p.Opart = part_f16; p.lse = lse_f32;
    p.nseq = nseq; p.H = H; p.Tp = Tp; p.ldo = ldo; p.mask_delay = mask_delay; p.kv_len = kv_len;
    return eend_launch_inproj_attn_long(p, (hipStream_t)stream);
}

int eend_spk_attn_f16(const void* qkv, void* O_f16, int B, int C, int Tp, int H, float scale, void* stream) {
    if (!qkv || !O_f16) return EEND_EINVAL;
    SpkAttnParams p;
    p.qkv = qkv; p.O = O_f16; p.B = B; p.C = C; p.Tp = Tp; p.H = H; p.scale = scale; p.drop = DropSpec{0u, 0u, 1.0f};
    return eend_launch_spk_attn(p, (hipStream_t)stream);
}

int eend_head_l2dot_f32(const float* emb, const float* attr, float* attr_out, float* logits, int B, int T,
                        int Tp, int C, int D, void* stream) {
    if (!emb || !attr || !attr_out || !logits) return EEND_EINVAL;
    return eend_launch_head(emb, attr, 0, attr_out, logits, B, T, Tp, C, D, (hipStream_t)stream);
}

int eend_head_l2dot_a16_f32(const float* emb, const void* attr_f16, float* attr_out, float* logits, int B, int T,
                            int Tp, int C, int D, void* stream) {
    if (!emb || !attr_f16 || !attr_out || !logits) return EEND_EINVAL;
    return eend_launch_head(emb, attr_f16, 1, attr_out, logits, B, T, Tp, C, D, (hipStream_t)stream);
}

int eend_attn_decode_f16(const void* qkv, void* K_cache, void* V_cache, void* out_f16, int N, int H, int cap, int t,
                         float scale, void* stream) {
    if (!qkv || !K_cache || !V_cache || !out_f16) return EEND_EINVAL;
    return eend_launch_attn_decode(qkv, K_cache, V_cache, out_f16, N, H, cap, t, nullptr, scale, (hipStream_t)stream);
}

int eend_attn_decode_dev_f16(const void* qkv, void* K_cache, void* V_cache, void* out_f16, int N, int H, int cap,
                             const int* t_dev, float scale, void* stream) {
    if (!qkv || !K_cache || !V_cache || !out_f16 || !t_dev) return EEND_EINVAL;
    return eend_launch_attn_decode(qkv, K_cache, V_cache, out_f16, N, H, cap, 0, t_dev, scale, (hipStream_t)stream);
}

int eend_attn_decode_split_f16(const void* qkv, void* K_cache, void* V_cache, void* out_f16, float* ws, long ws_floats, int N,
                               int H, int cap, const int* t_dev, float scale, void* stream) {
    if (!qkv || !K_cache || !V_cache || !out_f16 || !ws || !t_dev) return EEND_EINVAL;
    return eend_launch_attn_decode_split(qkv, K_cache, V_cache, out_f16, ws, ws_floats, N, H, cap, t_dev, scale, (hipStream_t)stream);
}

int eend_counter_add_i32(int* counter, int inc, void* stream) {
    return eend_launch_counter_add(counter, inc, (hipStream_t)stream);
}

int eend_retention_step_f16(const void* qkvg, float* kv_state, const float* scale_in, float* scale_out,
                            void* out_f16, int N, int H, float gn_eps, void* stream) {
    if (!qkvg || !kv_state || !scale_in || !scale_out || !out_f16) return EEND_EINVAL;
    return eend_launch_ret_step(qkvg, kv_state, scale_in, scale_out, out_f16, N, H, gn_eps, (hipStream_t)stream);
}

int eend_dwconv_step_f16(const void* x_f16, float* cache, const float* w, const float* bn_weight, const float* bn_bias,
                         const float* bn_mean, const float* bn_var, float eps, void* out_f16, int B, int D, int k,
                         void* stream) {
    if (!x_f16 || !cache || !w || !bn_weight || !bn_bias || !bn_mean || !bn_var || !out_f16) return EEND_EINVAL;
    return eend_launch_dwconv_step(x_f16, cache, w, bn_weight, bn_bias, bn_mean, bn_var, eps, out_f16, B, D, k,
                                   (hipStream_t)stream);
}

}  // extern "C"
