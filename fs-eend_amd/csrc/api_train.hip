// C-ABI shim of the training-step entry points (include/eend_hip.h, "TRAINING STEP"): argument validation,
// workspace partitioning, launch sequencing.  No allocation, no synchronisation.
#include "../../include/eend_hip.h"
#include "kernels.h"
#include <string.h>

namespace {

GemmParams gemm_base(const void* A, int lda, const void* W, int ldw, const float* bias, int M, int N, int K) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.ldo = N; p.Tp = 64; p.H = 4; p.dh = 64; p.C = 1;
    p.alpha = 1.0f; p.eps = 1e-5f; p.conv_cin = 64; p.conv_pad = 0; p.drop.scale = 1.0f;
    return p;
}

// NULL / p == 0 -> no dropout (thresh24 == 0 switches every kernel's mask off, scale 1 keeps the backward scaling neutral)
DropSpec drop_spec(const eend_dropout* d) {
    DropSpec s{0u, 0u, 1.0f};
    if (d && d->thresh24) { s.seed = d->seed; s.thresh24 = d->thresh24; s.scale = d->scale; }
    return s;
}

// Output tile and token split of a weight gradient (wgrad.hip).  256-aligned shapes take the 256 x 256 tile (one 8-wave
// workgroup per CU, half the operand bytes per flop), the others the 128 x 128 tile (two 4-wave workgroups per CU); the token
// axis is split so that about one round of workgroups runs, in multiples of 8 so that every XCD owns whole splits, within the
// workspace.
int plan_wgrad(long M, int N, int K, int conv_cin, long ws_floats, int* tile, int* nsplit, long* m_per_split, bool with_bias = false) {
    const long tile_floats = (long)N * K + (with_bias ? N : 0);      // per split: the partial tile (+ the partial column sums)
    if (tile_floats <= 0 || ws_floats < tile_floats) return EEND_EINVAL;
    // (a single 256 x 256 output keeps the small tile: four tiles x half the splits halve the partials the reduction reads --
    //  same box, [196608, 256, 256]: 50 + 21 us vs 48 + 12 us; [32768, 256, 256]: 19 + 21 vs 16 + 11)
    const bool big = (N % 256) == 0 && (K % 256) == 0 && (long)N * K > 65536 && (conv_cin == 0 || (conv_cin % 256) == 0) && M >= 16384;
#ifdef EEND_WG_FORCE128                                   // study build (tools/ab_one_source.sh)
    const int bt = 128;
    const int slots = eend_cu_count() * 2;
#else
    const int bt = big ? 256 : 128;
    const int slots = eend_cu_count() * (big ? 1 : 2);
#endif
    *tile = bt;
    const int ntiles = (N / bt) * (K / bt);
    long want = (slots + ntiles - 1) / ntiles;
    const long cap = ws_floats / tile_floats;
    if (want > cap) want = cap;
    if (want >= 8 && (want & 7)) {
        const long down = want & ~7L, up = down + 8;
        want = (up <= cap && up * ntiles <= slots) ? up : down;
    }
    const long steps = (M + 63) / 64;
    if (want > steps) want = steps;
    if (want < 1) want = 1;
    const long sps = (steps + want - 1) / want;          // 64-row steps per split
    *m_per_split = sps * 64;
    *nsplit = (int)((steps + sps - 1) / sps);
    return EEND_OK;
}

}  // namespace

extern "C" {

int eend_linear_res_ln_train_f16(const void* A, int lda, const void* W, int ldw, const float* bias, const float* res,
                                 float alpha, const float* gamma, const float* beta, float eps, float* out_f32,
                                 void* out_f16, void* xhat_f16, float* rstd, int M, int K, const eend_dropout* drop,
                                 void* stream) {
    if (!A || !W || !out_f32 || !out_f16 || !xhat_f16 || !rstd || !gamma || !beta) return EEND_EINVAL;
    GemmParams p = gemm_base(A, lda, W, ldw, bias, M, 256, K);
    p.res = res; p.alpha = alpha; p.gamma = gamma; p.beta = beta; p.eps = eps; p.out32 = out_f32; p.out16 = out_f16;
    p.xhat16 = xhat_f16; p.rstat = rstd; p.drop = drop_spec(drop);
    return eend_launch_gemm(p, EPI_RES_LN_TRAIN, (hipStream_t)stream);
}

int eend_linear_relu_train_f16(const void* A, int lda, const void* W, int ldw, const float* bias, void* out_f16, int ldo,
                               int M, int N, int K, const eend_dropout* drop, void* stream) {
    if (!A || !W || !out_f16 || (ldo & 3)) return EEND_EINVAL;
    GemmParams p = gemm_base(A, lda, W, ldw, bias, M, N, K);
    p.out16 = out_f16; p.ldo = ldo; p.drop = drop_spec(drop);
    return eend_launch_gemm(p, EPI_PLAIN_RELU_F16, (hipStream_t)stream);
}

int eend_ffn_train_f16(const void* X, int ldx, const void* W1, const float* b1, const void* W2, const float* b2, const float* res,
                       float alpha, const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, void* hid_f16,
                       void* xhat_f16, float* rstd, int M, int F, const eend_dropout* drop_hidden, const eend_dropout* drop_out,
                       void* stream) {
    if (!X || !W1 || !b1 || !W2 || !b2 || !res || !gamma || !beta || !out_f32 || !out_f16 || !hid_f16 || !xhat_f16 || !rstd) return EEND_EINVAL;
    FfnParams p;
    memset(&p, 0, sizeof(p));
    p.X = X; p.ldx = ldx; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.res = res; p.alpha = alpha; p.gamma = gamma; p.beta = beta;
    p.eps = eps; p.out32 = out_f32; p.out16 = out_f16; p.M = M; p.F = F; p.hid16 = hid_f16; p.xhat16 = xhat_f16; p.rstat = rstd;
    p.drop1 = drop_spec(drop_hidden); p.drop2 = drop_spec(drop_out);
    return eend_launch_ffn_fused(p, 1, FFN_EPI_RES_LN, (hipStream_t)stream);
}

int eend_ffn_swish_train_f16(const void* X, int ldx, const void* W1, const float* b1, const void* W2, const float* b2, const float* res,
                             float alpha, const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, void* z_f16,
                             void* a_f16, void* xhat_f16, float* rstd, int M, int F, int residual_stream_unnormalised,
                             const eend_dropout* drop_hidden, const eend_dropout* drop_out, void* stream) {
    if (!X || !W1 || !b1 || !W2 || !b2 || !res || !gamma || !beta || !out_f32 || !out_f16 || !z_f16 || !a_f16 || !xhat_f16 || !rstd) return EEND_EINVAL;
    FfnParams p;
    memset(&p, 0, sizeof(p));
    p.X = X; p.ldx = ldx; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.res = res; p.alpha = alpha; p.gamma = gamma; p.beta = beta;
    p.eps = eps; p.out32 = out_f32; p.out16 = out_f16; p.M = M; p.F = F; p.hid16 = a_f16; p.z16 = z_f16; p.xhat16 = xhat_f16; p.rstat = rstd;
    p.drop1 = drop_spec(drop_hidden); p.drop2 = drop_spec(drop_out);
    return eend_launch_ffn_fused(p, 2, residual_stream_unnormalised ? FFN_EPI_RES_SCALE_LN16 : FFN_EPI_RES_LN, (hipStream_t)stream);
}

int eend_ffn_bwd_data_bf16(const void* dY, int ldy, const void* W2T, const void* hid_f16, const void* W1T, float drop_scale,
                           void* dH_bf16, float* g_f32, int M, int F, void* stream) {
    if (!dY || !W2T || !hid_f16 || !W1T || !dH_bf16 || !g_f32) return EEND_EINVAL;
    FfnParams p;
    memset(&p, 0, sizeof(p));
    p.X = dY; p.ldx = ldy; p.W1 = W2T; p.W2 = W1T; p.hidmask = hid_f16; p.hid16 = dH_bf16; p.res = g_f32; p.out32 = g_f32; p.alpha = 1.0f;
    p.M = M; p.F = F; p.drop1 = DropSpec{0u, 0u, drop_scale};
    return eend_launch_ffn_fused(p, 0, FFN_EPI_RES_LN, (hipStream_t)stream);
}

int eend_ffn_train_stream_elems(int F) { return (F < 64 || (F % 64) != 0 || F > 2048) ? 0 : (int)eend_ffn_train_stream_nelems(F); }

int eend_ffn_train_stream_ok(int M, int F, int ldx) { return eend_ffn_train_stream_fits(M, F, ldx) ? 1 : 0; }

int eend_ffn_train_stream_pack(const void* W1, const void* W2, void* stream_out, int F, void* stream) {
    return eend_launch_ffn_train_stream_pack(W1, W2, stream_out, F, (hipStream_t)stream);
}

int eend_ffn_train_stream_f16(const void* X, int ldx, const void* wstream, const float* b1, const float* b2, const float* res, float alpha,
                              const float* gamma, const float* beta, float eps, float* out_f32, void* out_f16, void* hid_f16, void* xhat_f16,
                              float* rstd, int M, int F, const eend_dropout* drop_hidden, const eend_dropout* drop_out, void* stream) {
    FfnTrainStreamParams p;
    memset(&p, 0, sizeof(p));
    p.X = X; p.ldx = ldx; p.wstream = wstream; p.b1 = b1; p.b2 = b2; p.gamma = gamma; p.beta = beta; p.res = res; p.alpha = alpha; p.eps = eps;
    p.out32 = out_f32; p.out16 = out_f16; p.xhat16 = xhat_f16; p.rstat = rstd; p.hid = hid_f16; p.M = M; p.F = F;
    p.drop1 = drop_spec(drop_hidden); p.drop2 = drop_spec(drop_out);
    return eend_launch_ffn_train_stream(p, 1, (hipStream_t)stream);
}

int eend_ffn_bwd_data_stream_bf16(const void* dY, int ldy, const void* wstream, const void* hid_f16, float drop_scale, void* dH_bf16,
                                  float* g_f32, int M, int F, void* stream) {
    FfnTrainStreamParams p;
    memset(&p, 0, sizeof(p));
    p.X = dY; p.ldx = ldy; p.wstream = wstream; p.res = g_f32; p.out32 = g_f32; p.hid = (void*)hid_f16; p.dH = dH_bf16; p.alpha = 1.0f;
    p.M = M; p.F = F; p.drop1 = DropSpec{0u, 0u, drop_scale > 0.f ? drop_scale : 1.0f};
    return eend_launch_ffn_train_stream(p, 2, (hipStream_t)stream);
}

int eend_spk_attn_train_f16(const void* qkv, void* O_f16, int B, int C, int Tp, int H, float scale,
                            const eend_dropout* drop, void* stream) {
    if (!qkv || !O_f16 || H != 4) return EEND_EINVAL;
    SpkAttnParams p;
    p.qkv = qkv; p.O = O_f16; p.B = B; p.C = C; p.Tp = Tp; p.H = H; p.scale = scale; p.drop = drop_spec(drop);
    return eend_launch_spk_attn(p, (hipStream_t)stream);
}

int eend_conv1d_l2norm_train_f16(const void* X, const void* Wr, const float* bias, const int* ilens, float* out_f32,
                                 void* out_f16, float* inv_norm, int nseq, int Tp, int cin, int ktaps, int pad,
                                 void* stream) {
    if (!X || !Wr || !ilens || !out_f32 || !out_f16 || !inv_norm) return EEND_EINVAL;
    if (nseq <= 0 || Tp <= 0 || (Tp % 64) != 0 || cin <= 0 || (cin % 64) != 0 || ktaps <= 0 || pad < 0 || pad >= ktaps)
        return EEND_EINVAL;
    GemmParams p = gemm_base(X, cin, Wr, ktaps * cin, bias, nseq * Tp, 256, ktaps * cin);
    p.Tp = Tp; p.ilens = ilens; p.conv_cin = cin; p.conv_pad = pad; p.out32 = out_f32; p.out16 = out_f16; p.rstat = inv_norm;
    return eend_launch_gemm(p, EPI_L2NORM_TRAIN, (hipStream_t)stream);
}

int eend_inproj_heads_train_bf16(const void* A, int lda, const void* W, const float* bias, void* Q, void* Qt,
                                 void* K, void* Kt, void* V, void* Vt, int nseq, int Tp, int H, void* stream) {
    // Qt / Kt may be NULL: the one-launch backward (windows up to 512 frames, attn_bwd_fused.hip) reads its transposed operands from the
    // row-major head images through the LDS, so the [d][t] copies need not exist
    if (!A || !W || !bias || !Q || !K || !V) return EEND_EINVAL;
    if (nseq <= 0 || Tp <= 0 || (Tp % 64) != 0 || H != 4) return EEND_EINVAL;
    ProjParams q;
    memset(&q, 0, sizeof(q));
    q.X = A; q.ldx = lda; q.W = W; q.bias = bias; q.M = nseq * Tp; q.N = 768; q.Tp = Tp; q.H = H;
    q.kind[0] = Qt ? PROJ_HEADS_BOTH : PROJ_HEADS; q.out[0] = Q; q.out2[0] = Qt;
    q.kind[1] = Kt ? PROJ_HEADS_BOTH : PROJ_HEADS; q.out[1] = K; q.out2[1] = Kt;
    q.kind[2] = Vt ? PROJ_HEADS_BOTH : PROJ_HEADS; q.out[2] = V; q.out2[2] = Vt;
    q.is_bf16[0] = q.is_bf16[1] = q.is_bf16[2] = 1;
    return eend_launch_proj_xres(q, (hipStream_t)stream);
}

int eend_gemm_acc_stream_elems(int K) { return (int)eend_gemm_acc_stream_nelems(K); }

int eend_gemm_acc_stream_ok(int M, int K, int lda) { return eend_gemm_acc_stream_fits(M, K, lda) ? 1 : 0; }

int eend_gemm_acc_stream_pack_bf16(const void* Wt, int ldw, void* stream_out, int K, void* stream) {
    return eend_launch_gemm_acc_stream_pack(Wt, ldw, stream_out, K, (hipStream_t)stream);
}

int eend_gemm_acc_stream_bf16(const void* A, int lda, const void* wstream, float* g_f32, int M, int K, void* stream) {
    GemmAccStreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.wstream = wstream; p.g = g_f32; p.M = M; p.K = K;
    return eend_launch_gemm_acc_stream(p, (hipStream_t)stream);
}

int eend_proj_stream_elems(int N) { return (int)eend_proj_stream_nelems(N); }

int eend_proj_stream_pack_f16(const void* W, void* stream_out, int N, void* stream) {
    return eend_launch_proj_stream_pack(W, stream_out, N, (hipStream_t)stream);
}

static bool proj_stream_params(ProjStreamParams& p, const void* X, int ldx, const void* wstream, const float* bias, int M, int N, int Tp, int H,
                               const eend_proj_group* groups) {
    if (!groups || N <= 0 || (N % 256) != 0 || N > 1024) return false;
    memset(&p, 0, sizeof(p));
    p.X = X; p.ldx = ldx; p.wstream = wstream; p.bias = bias; p.M = M; p.N = N; p.Tp = Tp; p.H = H;
    for (int g = 0; g < N / 256; ++g) {
        p.kind_a[g] = groups[g].rows ? groups[g].rows_kind : 0; p.bf_a[g] = groups[g].rows_bf16; p.ld_a[g] = groups[g].rows_ld;
        p.out_a[g] = groups[g].rows; p.out_b[g] = groups[g].rows2_bf16_heads; p.out_t[g] = groups[g].heads_t; p.bf_t[g] = groups[g].heads_t_bf16;
    }
    return true;
}

int eend_proj_stream_ok(int ldx, int M, int N, int Tp, int H, const eend_proj_group* groups) {
    ProjStreamParams p;
    if (!proj_stream_params(p, (const void*)16, ldx, (const void*)16, (const float*)16, M, N, Tp, H, groups)) return 0;
    return eend_proj_stream_fits(p) ? 1 : 0;
}

int eend_proj_stream_f16(const void* X, int ldx, const void* wstream, const float* bias, int M, int N, int Tp, int H,
                         const eend_proj_group* groups, void* stream) {
    ProjStreamParams p;
    if (!proj_stream_params(p, X, ldx, wstream, bias, M, N, Tp, H, groups)) return EEND_EINVAL;
    return eend_launch_proj_stream(p, (hipStream_t)stream);
}

int eend_inproj_attn_train_bf16(const void* X_f16, int ldx, const void* W_packed, const float* b_in, void* O_f16, int ldo, void* Q_bf16,
                                void* K_bf16, void* V_bf16, float* lse, int nseq, int H, int Tp, int mask_delay, int kv_len,
                                const eend_dropout* drop, void* stream) {
    if (!X_f16 || !W_packed || !b_in || !O_f16 || !Q_bf16 || !K_bf16 || !V_bf16 || !lse || nseq > 16383) return EEND_EINVAL;
    InprojAttnParams p;
    memset(&p, 0, sizeof(p));
    p.X = X_f16; p.ldx = ldx; p.W = W_packed; p.bias = b_in; p.O = O_f16; p.ldo = ldo; p.Qh = Q_bf16; p.Kh = K_bf16; p.Vh = V_bf16; p.lse = lse;
    p.nseq = nseq; p.H = H; p.Tp = Tp; p.mask_delay = mask_delay; p.kv_len = kv_len; p.drop = drop_spec(drop);
    return eend_launch_inproj_attn_train(p, (hipStream_t)stream);
}

int eend_attn_causal_lse_bf16(const void* Q, const void* K, const void* Vt, void* O_f16, float* lse, int nseq, int H,
                              int Tp, int ldo, int mask_delay, int kv_len, float scale, const eend_dropout* drop,
                              void* stream) {
    if (!Q || !K || !Vt || !O_f16 || !lse || nseq > 65535 || H > 65535) return EEND_EINVAL;
    AttnParams p;
    p.Q = Q; p.K = K; p.Vt = Vt; p.O = O_f16; p.nseq = nseq; p.H = H; p.Tp = Tp; p.ldo = ldo;
    p.mask_delay = mask_delay; p.kv_len = kv_len; p.scale_log2 = scale * 1.4426950408889634f; p.Lse = lse;
    p.drop = drop_spec(drop);
    return eend_launch_attn_causal(p, (hipStream_t)stream);
}

int eend_attn_causal_bwd_bf16(const void* Q, const void* Qt, const void* K, const void* Kt, const void* V,
                              const void* dO, int ldo, const void* O_f16, int ldout, const float* lse, void* dOt_ws,
                              float* dh_ws, void* dQKV, int ldg, int nseq, int H, int Tp, int mask_delay, int kv_len,
                              int q_len, float scale_log2, float sq, float sk, const eend_dropout* drop, void* stream) {
    if (!dO || !O_f16 || !dOt_ws || !dh_ws || H != 4 || ldo != 256 || ldout != 256) return EEND_EINVAL;
    int rc = eend_launch_attn_rowdot(dO, O_f16, dh_ws, nseq, H, Tp, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    AttnBwdParams p;
    memset(&p, 0, sizeof(p));
    p.Q = Q; p.Qt = Qt; p.K = K; p.Kt = Kt; p.V = V; p.dO = dO; p.dOt = dOt_ws; p.Lse = lse; p.Dh = dh_ws; p.dQKV = dQKV;
    p.nseq = nseq; p.H = H; p.Tp = Tp; p.ldo = ldo; p.ldg = ldg; p.mask_delay = mask_delay; p.kv_len = kv_len; p.q_len = q_len;
    p.scale_log2 = scale_log2; p.sq = sq; p.sk = sk; p.drop = drop_spec(drop);
    if (!eend_attn_bwd_fused_ok(p, false)) {              // windows beyond 512 frames: the two-kernel form reads dO^T (attn_bwd.hip)
        rc = eend_launch_heads_transpose(dO, ldo, dOt_ws, nseq, H, Tp, (hipStream_t)stream);
        if (rc != EEND_OK) return rc;
    }
    return eend_launch_attn_bwd(p, (hipStream_t)stream);
}

int eend_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* out_bf16, int ldo,
                   int M, int N, int K, void* stream) {
    if (!A || !W || !out_bf16 || (ldo & 7)) return EEND_EINVAL;
    GemmParams p = gemm_base(A, lda, W, ldw, bias, M, N, K);
    p.bf16 = 1; p.out16 = out_bf16; p.ldo = ldo;
    return eend_launch_gemm(p, EPI_PLAIN_BF16, (hipStream_t)stream);
}

int eend_gemm_relu_bwd_bf16(const void* A, int lda, const void* W, int ldw, const void* act, int ldact,
                            void* out_bf16, int ldo, int M, int N, int K, float drop_scale, void* stream) {
    if (!A || !W || !act || !out_bf16 || (ldo & 7) || (ldact & 7)) return EEND_EINVAL;
    GemmParams p = gemm_base(A, lda, W, ldw, nullptr, M, N, K);
    p.bf16 = 1; p.out16 = out_bf16; p.ldo = ldo; p.mask = act; p.ldmask = ldact;
    p.drop.scale = drop_scale > 0.f ? drop_scale : 1.0f;    // `act` is the dropped activation: its zeros are the mask
    return eend_launch_gemm(p, EPI_MASK_BF16, (hipStream_t)stream);
}

int eend_gemm_acc_bf16(const void* A, int lda, const void* W, int ldw, const float* res_f32, float alpha,
                       float* out_f32, void* out_bf16, int M, int K, void* stream) {
    if (!A || !W || (!out_f32 && !out_bf16)) return EEND_EINVAL;
    GemmParams p = gemm_base(A, lda, W, ldw, nullptr, M, 256, K);
    p.bf16 = 1; p.res = res_f32; p.alpha = alpha; p.out32 = out_f32; p.out16 = out_bf16;
    return eend_launch_gemm(p, EPI_RES_SCALE, (hipStream_t)stream);
}

int eend_gemm_acc_lnbwd_bf16(const void* A, int lda, const void* W, int ldw, const float* g_f32, const void* xhat_f16, const float* rstd,
                             const float* gamma, float* ds_f32, void* ds_bf16, float* ws, long ws_floats, float* dgamma, float* dbeta,
                             float* dbias, int M, int K, const eend_dropout* drop, void* stream) {
    if (!A || !W || !g_f32 || !xhat_f16 || !rstd || !gamma || !ds_f32 || !ds_bf16 || !ws || !dgamma || !dbeta || M <= 0) return EEND_EINVAL;
    const long nb = ((long)M + 63) / 64;
    if (ws_floats < nb * 768) return EEND_EINVAL;
    GemmParams p = gemm_base(A, lda, W, ldw, nullptr, M, 256, K);
    p.bf16 = 1; p.res = g_f32; p.alpha = 1.0f; p.out32 = ds_f32; p.out16 = ds_bf16; p.ldo = 256;
    p.xhat16 = (void*)xhat_f16; p.rstat = (float*)rstd; p.gamma = gamma; p.colpart = ws; p.drop = drop_spec(drop);
    int rc = eend_launch_gemm(p, EPI_RES_LNBWD, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce_multi(ws, 768, (int)nb, 256, dgamma, dbeta, dbias, (hipStream_t)stream);
}

int eend_conv1d_dgrad_bf16(const void* dY, const void* Wd, const int* src_lens, const int* mask_lens, float* out_f32,
                           int nseq, int Tp, int cout, int ktaps, int pad, void* stream) {
    if (!dY || !Wd || !src_lens || !mask_lens || !out_f32) return EEND_EINVAL;
    if (nseq <= 0 || Tp <= 0 || (Tp % 64) != 0 || cout != 256 || ktaps <= 0 || pad < 0 || pad >= ktaps) return EEND_EINVAL;
    GemmParams p = gemm_base(dY, cout, Wd, ktaps * cout, nullptr, nseq * Tp, 256, ktaps * cout);
    p.bf16 = 1; p.Tp = Tp; p.ilens = src_lens; p.mask_lens = mask_lens; p.conv_cin = cout; p.conv_pad = pad; p.out32 = out_f32;
    return eend_launch_gemm(p, EPI_F32_ROWMASK, (hipStream_t)stream);
}

int eend_wgrad_bf16(const void* dY, int lda, const void* X, int ldb, int x_is_f16, long M, int N, int K, float* ws,
                    long ws_floats, float* out, int ld_out, int K_out, float scale, int accumulate, void* stream) {
    if (!dY || !X || !ws || !out || M <= 0 || N <= 0 || K <= 0 || (N % 128) || (K % 128)) return EEND_EINVAL;
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.A = dY; p.B = X; p.partial = ws; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.b_is_f16 = (x_is_f16 & 1) ? 1 : 0; p.b_blocked = (x_is_f16 & 2) ? 1 : 0; p.a_blocked = (x_is_f16 & 4) ? 1 : 0;
    int rc = plan_wgrad(M, N, K, 0, ws_floats, &p.tile, &p.nsplit, &p.m_per_split);
    if (rc != EEND_OK) return rc;
    rc = eend_launch_wgrad(p, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce_tiles(ws, p.tile, p.nsplit, N, K, K_out, out, ld_out, scale, accumulate, (hipStream_t)stream);
}

int eend_wgrad_bias_bf16(const void* dY, int lda, const void* X, int ldb, int x_is_f16, long M, int N, int K, float* ws,
                         long ws_floats, float* out, int ld_out, int K_out, float* bias_out, float scale, int accumulate, void* stream) {
    if (!dY || !X || !ws || !out || !bias_out || M <= 0 || N <= 0 || K <= 0 || (N % 128) || (K % 128)) return EEND_EINVAL;
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.A = dY; p.B = X; p.partial = ws; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.b_is_f16 = (x_is_f16 & 1) ? 1 : 0; p.b_blocked = (x_is_f16 & 2) ? 1 : 0; p.a_blocked = (x_is_f16 & 4) ? 1 : 0;
    int rc = plan_wgrad(M, N, K, 0, ws_floats, &p.tile, &p.nsplit, &p.m_per_split, true);
    if (rc != EEND_OK) return rc;
    p.bias_partial = ws + (size_t)p.nsplit * N * K;
    rc = eend_launch_wgrad(p, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    rc = eend_launch_wgrad_reduce_tiles(ws, p.tile, p.nsplit, N, K, K_out, out, ld_out, scale, accumulate, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce(p.bias_partial, N, p.nsplit, 1, N, N, bias_out, N, scale, accumulate, (hipStream_t)stream);
}

int eend_wgrad_bias_grouped_bf16(const void* dY, int lda, const void* X, int ldb, int x_is_f16, long M, int N, int K, float* ws,
                                 long ws_floats, float* out, float* bias_out, int group_rows, long group_stride, float scale, void* stream) {
    if (!dY || !X || !ws || !out || !bias_out || M <= 0 || N <= 0 || K <= 0 || (N % 128) || (K % 128) || group_rows <= 0 || (N % group_rows) ||
        group_stride < (long)group_rows * K)
        return EEND_EINVAL;
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.A = dY; p.B = X; p.partial = ws; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.b_is_f16 = (x_is_f16 & 1) ? 1 : 0; p.b_blocked = (x_is_f16 & 2) ? 1 : 0; p.a_blocked = (x_is_f16 & 4) ? 1 : 0;
    int rc = plan_wgrad(M, N, K, 0, ws_floats, &p.tile, &p.nsplit, &p.m_per_split, true);
    if (rc != EEND_OK) return rc;
    p.bias_partial = ws + (size_t)p.nsplit * N * K;
    rc = eend_launch_wgrad(p, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    rc = eend_launch_wgrad_reduce_tiles(ws, p.tile, p.nsplit, N, K, K, out, K, scale, 0, (hipStream_t)stream, group_rows,
                                        group_stride - (long)group_rows * K);
    if (rc != EEND_OK) return rc;
    // bias_out[g * group_stride + r]: the column sums of dY as N / group_rows rows of group_rows
    return eend_launch_wgrad_reduce(p.bias_partial, N, p.nsplit, N / group_rows, group_rows, group_rows, bias_out, (int)group_stride, scale, 0,
                                    (hipStream_t)stream);
}

int eend_conv1d_wgrad_bf16(const void* dY, const void* X_f16, const int* ilens, int nseq, int Tp, int cin, int ktaps,
                           int pad, float* ws, long ws_floats, float* tmp, float* out, void* stream) {
    if (!dY || !X_f16 || !ilens || !ws || !tmp || !out || nseq <= 0 || Tp <= 0 || cin != 256 || ktaps <= 0 || pad < 0 || pad >= ktaps)
        return EEND_EINVAL;
    const int N = 256, K = ktaps * cin;
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.A = dY; p.B = X_f16; p.partial = ws; p.M = (long)nseq * Tp; p.N = N; p.K = K; p.lda = 256; p.ldb = cin; p.b_is_f16 = 1;
    p.conv = 1; p.conv_cin = cin; p.conv_pad = pad; p.Tp = Tp; p.ilens = ilens;
    int rc = plan_wgrad(p.M, N, K, cin, ws_floats, &p.tile, &p.nsplit, &p.m_per_split);
    if (rc != EEND_OK) return rc;
    rc = eend_launch_wgrad(p, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    rc = eend_launch_wgrad_reduce_tiles(ws, p.tile, p.nsplit, N, K, K, tmp, K, 1.0f, 0, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_conv_wgrad_unpermute(tmp, out, N, cin, ktaps, (hipStream_t)stream);
}

int eend_colsum_f32(const void* Y, int ld, long M, int N, int is_bf16, float* ws, long ws_floats, float* out,
                    float scale, int accumulate, void* stream) {
    if (!Y || !ws || !out || M <= 0 || N <= 0 || ws_floats < N) return EEND_EINVAL;
    long ns = ws_floats / N;
    if (ns > 1024) ns = 1024;
    if (ns > (M + 63) / 64) ns = (M + 63) / 64;          // at least 64 rows per split
    int rc = eend_launch_colsum_partial(Y, ld, M, N, is_bf16, (int)ns, ws, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce(ws, N, (int)ns, 1, N, N, out, N, scale, accumulate, (hipStream_t)stream);
}

int eend_layernorm_bwd_f32(const float* g, const void* xhat_f16, const float* rstd, const float* gamma, float* ds_f32,
                           void* ds_bf16, float* ws, long ws_floats, float* dgamma, float* dbeta, float* dbias, long M,
                           const eend_dropout* drop, void* stream) {
    if (!ws || !dgamma || !dbeta || ws_floats < 1024L * 768 || (dbias && !ds_bf16)) return EEND_EINVAL;
    int nb = 0;
    int rc = eend_launch_ln_bwd(g, xhat_f16, rstd, gamma, ds_f32, ds_bf16, ws, &nb, M, drop_spec(drop), (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    // d gamma | d beta | d bias: the three column blocks of the row pass's partials, one reduction launch
    return eend_launch_wgrad_reduce_multi(ws, 768, nb, 256, dgamma, dbeta, dbias, (hipStream_t)stream);
}

int eend_head_bce_f32(const float* emb, const float* attr, const float* labels, const int* ilens, const int* ncols,
                      float inv_frames, const float* dlogits_in, float* logits, float* da, float* de, float* ws, long ws_floats,
                      float* loss_out, int B, int T, int Tp, int C, void* stream) {
    const long nb = ((long)B * Tp + 3) / 4;
    if (!ws || !loss_out || ws_floats < nb) return EEND_EINVAL;
    int rc = eend_launch_head_bce(emb, attr, labels, ilens, ncols, inv_frames, dlogits_in, logits, da, de, ws, B, T, Tp, C,
                                  (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_scalar_sum(ws, nb, 1.0f, loss_out, (hipStream_t)stream);
}

int eend_l2norm_bwd_bf16(const float* y, const float* dy, const float* inv_norm, void* dx_bf16, int B, int T, int Tp,
                         void* stream) {
    return eend_launch_l2norm_bwd(y, dy, inv_norm, dx_bf16, B, T, Tp, (hipStream_t)stream);
}

int eend_convert_fanout_bwd_f32(const float* g0, void* gsum_bf16, float* ws, long ws_floats, float* dpc, int B, int Tp,
                                int C, void* stream) {
    if (!ws || !dpc || C < 1 || C > 12 || ws_floats < 256L * C * 256) return EEND_EINVAL;
    int nb = 0;
    int rc = eend_launch_slot_sum(g0, gsum_bf16, ws, &nb, B, Tp, C, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce(ws, (long)C * 256, nb, C, 256, 256, dpc, 256, 1.0f, 0, (hipStream_t)stream);
}

int eend_convert_const_f32(int mode, const float* W, const float* bias, const float* pe, float* pc, const float* dpc,
                           float* dW, float* dbias, int C, void* stream) {
    return eend_launch_convert_const(mode, W, bias, pe, pc, dpc, dW, dbias, C, (hipStream_t)stream);
}

int eend_spk_attn_bwd_bf16(const void* qkv_f16, const void* dO_bf16, void* dqkv_bf16, int B, int C, int Tp, int H,
                           float scale, const eend_dropout* drop, void* stream) {
    if (H != 4) return EEND_EINVAL;
    return eend_launch_spk_attn_bwd(qkv_f16, dO_bf16, dqkv_bf16, B, C, Tp, scale, drop_spec(drop), (hipStream_t)stream);
}

int eend_bn_train_stats_f32(const void* const* x_ptrs, const int* lens, float pad_value, float* ws, long ws_floats,
                            float* mean, float* var, float* run_mean, float* run_var, float momentum, int B, int T,
                            int F, void* stream) {
    if (!ws || !mean || !var || B <= 0 || T <= 0 || F <= 0) return EEND_EINVAL;
    long ns = ((long)B * T + 255) / 256;
    if (ns > 512) ns = 512;
    if (ws_floats < (ns + 1) * 2L * F) return EEND_EINVAL;
    float* sums = ws + ns * 2L * F;
    const float n = (float)((long)B * T);
    for (int pass = 0; pass < 2; ++pass) {
        int rc = eend_launch_bn_colstats((const float* const*)x_ptrs, lens, pad_value, pass ? mean : nullptr, ws, B, T, F, (int)ns,
                                         (hipStream_t)stream);
        if (rc != EEND_OK) return rc;
        rc = eend_launch_wgrad_reduce(ws, 2L * F, (int)ns, 1, 2 * F, 2 * F, sums, 2 * F, 1.0f, 0, (hipStream_t)stream);
        if (rc != EEND_OK) return rc;
        rc = eend_launch_bn_finalize(pass, sums, n, mean, var, run_mean, run_var, momentum, F, (hipStream_t)stream);
        if (rc != EEND_OK) return rc;
    }
    return EEND_OK;
}

int eend_bn_bwd_f32(const void* const* x_ptrs, const int* lens, float pad_value, const float* mean, const float* var,
                    float eps, const void* dy_bf16, int ld, float* ws, long ws_floats, float* dgamma, float* dbeta,
                    int B, int T, int Tp, int F, void* stream) {
    if (!ws || !dgamma || !dbeta || B <= 0 || T <= 0 || F <= 0) return EEND_EINVAL;
    long ns = ((long)B * T + 255) / 256;
    if (ns > 512) ns = 512;
    if (ws_floats < ns * 2L * F) return EEND_EINVAL;
    int rc = eend_launch_bn_bwd((const float* const*)x_ptrs, lens, pad_value, mean, var, eps, dy_bf16, ld, ws, B, T, Tp, F, (int)ns,
                                (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    if ((F & 31) == 0) return eend_launch_wgrad_reduce_multi(ws, 2L * F, (int)ns, F, dgamma, dbeta, nullptr, (hipStream_t)stream);
    rc = eend_launch_wgrad_reduce(ws, 2L * F, (int)ns, 1, 2 * F, F, dgamma, F, 1.0f, 0, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce(ws + F, 2L * F, (int)ns, 1, 2 * F, F, dbeta, F, 1.0f, 0, (hipStream_t)stream);
}

int eend_emb_consistency_bwd_f16(const void* emb_f16, const float* labels, const int* lens, float inv_count, float* de,
                                 int B, int T, int Tp, int D, int C, void* stream) {
    return eend_launch_emb_consistency_bwd(emb_f16, labels, lens, inv_count, de, B, T, Tp, D, C, (hipStream_t)stream);
}

int eend_grad_sumsq_f32(const float* g, long n, float* ws, long ws_floats, float* out, void* stream) {
    if (!ws || ws_floats < 1024) return EEND_EINVAL;
    return eend_launch_grad_sumsq(g, n, ws, out, (hipStream_t)stream);
}

int eend_adam_step_f32(float* p, const float* g, float* m, float* v, long n, const float* hp, const float* gsumsq,
                       float beta1, float beta2, float eps, void* stream) {
    return eend_launch_adam(p, g, m, v, n, hp, gsumsq, beta1, beta2, eps, (hipStream_t)stream);
}

int eend_grad_accumulate_f32(float* acc, const float* g, float scale, int first, long n, void* stream) {
    return eend_launch_grad_accumulate(acc, g, scale, first, n, (hipStream_t)stream);
}

int eend_prep_weights(const eend_prep_entry* table, int n, void* stream) {
    return eend_launch_prep_weights(table, n, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// LS-EEND training step
// ---------------------------------------------------------------------------------------------------------------
int eend_swish_dropout_f16(const void* z_f16, void* a_f16, long M, int F, const eend_dropout* drop, void* stream) {
    return eend_launch_swish_drop_fwd(z_f16, a_f16, M, F, drop_spec(drop), (hipStream_t)stream);
}

int eend_swish_bwd_bf16(void* dz_bf16, const void* z_f16, long M, int F, const eend_dropout* drop, void* stream) {
    return eend_launch_swish_bwd(dz_bf16, z_f16, M, F, drop_spec(drop), (hipStream_t)stream);
}

int eend_layernorm_train_f16(const float* x, const float* gamma, const float* beta, float eps, void* y_f16,
                             void* xhat_f16, float* rstd, long M, void* stream) {
    return eend_launch_layernorm_train(x, gamma, beta, eps, y_f16, xhat_f16, rstd, M, (hipStream_t)stream);
}

int eend_layernorm_bwd2_f32(const void* g, int g_is_bf16, const void* xhat_f16, const float* rstd, const float* gamma,
                            float* ds_f32, int accumulate, void* ds_bf16, float alpha16, float* ws, long ws_floats,
                            float* dgamma, float* dbeta, float* dbias, long M, const eend_dropout* drop, void* stream) {
    if (!ws || !dgamma || !dbeta || ws_floats < 1024L * 768 || (dbias && !ds_bf16)) return EEND_EINVAL;
    int nb = 0;
    int rc = eend_launch_ln_bwd2(g, g_is_bf16, xhat_f16, rstd, gamma, ds_f32, accumulate, ds_bf16, alpha16, ws, &nb, M, drop_spec(drop),
                                 (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    // d gamma | d beta | d bias: the three column blocks of the row pass's partials, one reduction launch
    return eend_launch_wgrad_reduce_multi(ws, 768, nb, 256, dgamma, dbeta, dbias, (hipStream_t)stream);
}

int eend_resgrad_cast_bf16(const float* g, void* ds_bf16, float alpha, float* ws, long ws_floats, float* dbias, long M,
                           const eend_dropout* drop, void* stream) {
    if (!ws || !dbias || ws_floats < 1024L * 256) return EEND_EINVAL;
    int nb = 0;
    int rc = eend_launch_resgrad_cast(g, ds_bf16, alpha, ws, &nb, M, drop_spec(drop), (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce(ws, 256, nb, 1, 256, 256, dbias, 256, 1.0f, 0, (hipStream_t)stream);
}

int eend_linear_res_scale_ln_train_f16(const void* A, int lda, const void* W, int ldw, const float* bias,
                                       const float* res, float alpha, const float* gamma, const float* beta, float eps,
                                       float* out_f32, void* out_f16, void* xhat_f16, float* rstd, int M, int K,
                                       const eend_dropout* drop, void* stream) {
    if (!A || !W || !out_f32 || !out_f16 || !xhat_f16 || !rstd || !gamma || !beta) return EEND_EINVAL;
    GemmParams p = gemm_base(A, lda, W, ldw, bias, M, 256, K);
    p.res = res; p.alpha = alpha; p.gamma = gamma; p.beta = beta; p.eps = eps; p.out32 = out_f32; p.out16 = out_f16;
    p.xhat16 = xhat_f16; p.rstat = rstd; p.drop = drop_spec(drop);
    return eend_launch_gemm(p, EPI_RES_SCALE_LN16_TRAIN, (hipStream_t)stream);
}

int eend_glu_dwconv_f16(const void* P_f16, const float* w, void* c_f16, int nseq, int Tp, int Tv, int k, void* stream) {
    return eend_launch_glu_dwconv_fwd(P_f16, w, c_f16, nseq, Tp, Tv, k, (hipStream_t)stream);
}

int eend_bn_batch_stats_f16(const void* c_f16, float* ws, long ws_floats, float* stats, int nseq, int Tp, int Tv,
                            void* stream) {
    if (!c_f16 || !ws || !stats || nseq <= 0 || Tv <= 0) return EEND_EINVAL;
    long nb = ((long)nseq * Tv + 31) / 32;             // 32 rows per block: 8 iterations of four rows in flight per thread
    if (nb > 4096) nb = 4096;
    if (ws_floats < (nb + 1) * 256L) return EEND_EINVAL;
    float* sum = ws + nb * 256L;
    const float n = (float)((long)nseq * Tv);
    hipStream_t st = (hipStream_t)stream;
    int rc = eend_launch_bn_colstats16(c_f16, nullptr, ws, nseq, Tp, Tv, (int)nb, st);
    if (rc != EEND_OK) return rc;
    rc = eend_launch_wgrad_reduce(ws, 256, (int)nb, 1, 256, 256, sum, 256, 1.0f, 0, st);
    if (rc != EEND_OK) return rc;
    rc = eend_launch_bn_local_mean(sum, n, stats, st);                  // stats[0..255] = mean, stats[512] = n
    if (rc != EEND_OK) return rc;
    rc = eend_launch_bn_colstats16(c_f16, stats, ws, nseq, Tp, Tv, (int)nb, st);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce(ws, 256, (int)nb, 1, 256, 256, stats + 256, 256, 1.0f, 0, st);
}

int eend_bn_merge_f32(const float* stats, int R, float* mean, float* var, float* n_out, float* run_mean, float* run_var,
                      float momentum, void* stream) {
    return eend_launch_bn_merge(stats, R, mean, var, n_out, run_mean, run_var, momentum, (hipStream_t)stream);
}

int eend_bn_swish_f16(const void* c_f16, const float* mean, const float* var, float eps, const float* gamma,
                      const float* beta, void* s_f16, long M, void* stream) {
    return eend_launch_bn_swish_fwd(c_f16, mean, var, eps, gamma, beta, s_f16, M, (hipStream_t)stream);
}

int eend_bn_swish_bwd_stats_bf16(const void* ds_bf16, const void* c_f16, const float* mean, const float* var, float eps,
                                 const float* gamma, const float* beta, float* ws, long ws_floats, float* sums,
                                 float* dgamma, float* dbeta, int nseq, int Tp, int Tv, void* stream) {
    if (!ws || !sums || !dgamma || !dbeta || nseq <= 0 || Tv <= 0) return EEND_EINVAL;
    long nb = ((long)nseq * Tv + 31) / 32;             // 32 rows per block: 8 iterations of four rows in flight per thread
    if (nb > 4096) nb = 4096;
    if (ws_floats < nb * 512L) return EEND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int rc = eend_launch_bn_swish_bwd_stats(ds_bf16, c_f16, mean, var, eps, gamma, beta, ws, nseq, Tp, Tv, (int)nb, st);
    if (rc != EEND_OK) return rc;
    rc = eend_launch_wgrad_reduce(ws, 512, (int)nb, 1, 512, 512, sums, 512, 1.0f, 0, st);
    if (rc != EEND_OK) return rc;
    rc = eend_launch_wgrad_reduce(ws, 512, (int)nb, 1, 512, 256, dbeta, 256, 1.0f, 0, st);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce(ws + 256, 512, (int)nb, 1, 512, 256, dgamma, 256, 1.0f, 0, st);
}

int eend_bn_swish_bwd_apply_bf16(void* ds_bf16, const void* c_f16, const float* mean, const float* var, float eps,
                                 const float* gamma, const float* beta, const float* sums, const float* n_dev, int nseq,
                                 int Tp, int Tv, void* stream) {
    return eend_launch_bn_swish_bwd_apply(ds_bf16, c_f16, mean, var, eps, gamma, beta, sums, n_dev, nseq, Tp, Tv, (hipStream_t)stream);
}

int eend_dwconv_glu_bwd_bf16(const void* dc_bf16, const void* P_f16, const float* w, void* dP_bf16, float* ws,
                             long ws_floats, float* dw, int nseq, int Tp, int Tv, int k, void* stream) {
    if (!ws || !dw || nseq <= 0 || Tp <= 0 || k <= 0) return EEND_EINVAL;
    const long nblk = (long)nseq * ((Tp + 63) / 64);
    if (ws_floats < nblk * 256L * k) return EEND_EINVAL;
    int rc = eend_launch_dwconv_glu_bwd(dc_bf16, P_f16, w, dP_bf16, ws, nseq, Tp, Tv, k, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_wgrad_reduce(ws, 256L * k, (int)nblk, 256, k, k, dw, k, 1.0f, 0, (hipStream_t)stream);
}

int eend_retention_chunk_train_f16(const void* Q, const void* K, const void* Kt, const void* Vt, const void* G,
                                   void* O_f16, void* rhat_f16, float* rc_out, void* St_ws, float* kv_ws, float* cscale_ws,
                                   float* sexp_ws, int nseq, int H, int Tp, int L, int ldo, int ldg, float gn_eps,
                                   int T_valid, void* stream) {
    if (!Q || !K || !Kt || !Vt || !G || !O_f16 || !rhat_f16 || !rc_out || !St_ws || !kv_ws || !cscale_ws || !sexp_ws) return EEND_EINVAL;
    if (L <= 0 || L > 512 || (L & 3) || (ldo & 7) || T_valid <= 0 || T_valid > Tp || (T_valid % L) != 0) return EEND_EINVAL;
    RetParams p;
    memset(&p, 0, sizeof(p));
    p.Q = Q; p.K = K; p.Kt = Kt; p.Vt = Vt; p.G = G; p.O = O_f16; p.St = St_ws; p.cscale = cscale_ws; p.sexp = sexp_ws; p.kv_ws = kv_ws;
    p.nseq = nseq; p.H = H; p.Tp = Tp; p.L = L; p.nc = T_valid / L; p.ldo = ldo; p.ldg = ldg; p.gn_eps = gn_eps;
    p.Rhat = rhat_f16; p.Rc = rc_out;
    int rc = eend_launch_ret_state_scan(p, (hipStream_t)stream);
    if (rc != EEND_OK) return rc;
    return eend_launch_ret_chunk_full(p, (hipStream_t)stream);
}

int eend_retention_bwd_bf16(const void* Q, const void* Qt, const void* K, const void* Kt, const void* V, const void* Vt,
                            const float* dctx_f32, const void* g_f16, int ldg, const void* rhat_f16, const float* rc_in,
                            void* ot_ws, void* ott_ws, float* kv_ws, float* g_ws, void* St_ws, void* dqkvg_bf16, int ldq,
                            int nseq, int H, int Tp, int L, int T_valid, float sk, void* stream) {
    if (!Q || !K || !V || !dctx_f32 || !g_f16 || !rhat_f16 || !rc_in || !ot_ws || !kv_ws || !g_ws || !St_ws || !dqkvg_bf16) return EEND_EINVAL;
    if (H != 4 || L <= 0 || T_valid <= 0 || T_valid > Tp || (T_valid % L) != 0 || ldq < 1024 || (ldq & 7)) return EEND_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nc = T_valid / L;
    AttnBwdParams p;
    memset(&p, 0, sizeof(p));
    p.H = H; p.nseq = nseq; p.Tp = Tp; p.ldo = 256; p.ldg = ldq; p.L = L; p.nc = nc;
    // chunk lengths up to 512: the one-launch backward and the row-major state kernel read no [d][t] copies (Qt / Kt / Vt / ott_ws may be NULL)
    const bool row_major = eend_attn_bwd_fused_ok(p, true);
    if (!row_major && (!Qt || !Kt || !Vt || !ott_ws)) return EEND_EINVAL;
    int rc = eend_launch_ret_gate_gn_bwd(dctx_f32, g_f16, ldg, rhat_f16, rc_in, (__bf16*)dqkvg_bf16 + 768, ldq, ot_ws, nseq, Tp, T_valid, st);
    if (rc != EEND_OK) return rc;
    if (row_major) {
        rc = eend_launch_ret_bwd_states_rm(K, V, Q, ot_ws, 256, kv_ws, g_ws, St_ws, nseq, H, Tp, L, nc, st);
    } else {
        rc = eend_launch_heads_transpose(ot_ws, 256, ott_ws, nseq, H, Tp, st);
        if (rc != EEND_OK) return rc;
        rc = eend_launch_ret_bwd_states(Kt, Vt, Qt, ott_ws, kv_ws, g_ws, St_ws, nseq, H, Tp, L, nc, st);
    }
    if (rc != EEND_OK) return rc;
    memset(&p, 0, sizeof(p));
    p.Q = Q; p.Qt = Qt; p.K = K; p.Kt = Kt; p.V = V; p.dO = ot_ws; p.dOt = ott_ws; p.dQKV = dqkvg_bf16;
    p.nseq = nseq; p.H = H; p.Tp = Tp; p.ldo = 256; p.ldg = ldq; p.mask_delay = 0; p.kv_len = T_valid; p.q_len = T_valid;
    p.scale_log2 = 0.f; p.sq = 1.0f; p.sk = sk; p.drop = drop_spec(nullptr); p.L = L; p.nc = nc; p.St = St_ws;
    return eend_launch_ret_bwd(p, st);
}

}  // extern "C"
