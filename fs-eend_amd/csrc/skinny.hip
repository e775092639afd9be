// Skinny linear layers (M <= 16 rows): the frame-by-frame streaming paths (FS-EEND/streaming_infer_dia.py,
// LS-EEND/streaming_infer_dia.py: one new frame x up to max_nspks attractor slots per step) run every nn.Linear
// with 1..10 rows.  The tiled MFMA GEMM (gemm.hip) then keeps N/128 (or M/64 = 1) workgroups busy and its launch is
// pure latency (measured 15-20 us per call, ~45 calls per FS frame, ~100 per LS frame).  Here the weight matrix is
// the only traffic that matters (<= 2.5 MB, L2 / MALL resident): it is spread over the whole chip -- one wave per
// output feature (K <= 512) or one workgroup per output feature with the contraction split over its 4 waves
// (K > 512) -- each lane streams 16-byte pieces of the weight row against the <= 16 activation rows and the partial
// dot products are reduced with DPP-free shuffles.  HBM/L2-bound by construction: no MFMA (2*M*N*K <= 17 MFLOP).
//
//   y[m][n] = sum_k A[m][k] W[n][k] + bias[n]          A f16 [M][lda], W f16 [N][ldw] (torch nn.Linear layout)
//
// Epilogues (same semantics as the gemm.hip ones they stand in for):
//   SK_PLAIN : out16 = act(y)                                  (EPI_PLAIN_F16 / _RELU / _SWISH)
//   SK_GLU   : out16[m][n] = y[m][2n] * sigmoid(y[m][2n+1])    (EPI_GLU_F16: value / gate rows interleaved)
//   SK_RES   : out32 = y * alpha + res (and out16 = f16 of it) (EPI_RES_SCALE; first half of EPI_RES_LN /
//              EPI_RES_SCALE_LN16, whose LayerNorm over the 256 features is skinny_ln_kernel below)
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SK_PLAIN = 0, SK_GLU = 1, SK_RES = 2;

struct SkinnyParams {
    const void* A; int lda;      // f16 (F32 = false) or f32 (the all-f32 LS decoder frame step)
    const void* W; int ldw;
    const float* bias;
    int M, N, K;                 // N = output features (GLU: pairs)
    int act;                     // 0 none, 1 relu, 2 swish
    float alpha;
    const float* res; int ldres;
    float* out32; _Float16* out16; int ldo;
};

// 8 consecutive operand elements as floats
template <bool F32>
DEV void load8(const void* base, size_t off, float (&v)[8]) {
    if constexpr (F32) {
        const float4 a = *(const float4*)((const float*)base + off), b = *(const float4*)((const float*)base + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const f16x8 h = *(const f16x8*)((const _Float16*)base + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    }
}

DEV float act_apply(float v, int act) {
    if (act == 1) return __builtin_fmaxf(v, 0.f);
    if (act == 2) return v / (1.0f + __expf(-v));
    return v;
}

// MB = row bucket (rows >= M are clamped duplicates, never stored); ROWS = weight rows per output (2 for GLU);
// SPLITK: one workgroup per output, its 4 waves take the 512-wide k chunks round-robin.
template <int MB, int EPI, bool SPLITK, bool F32>
__global__ __launch_bounds__(256)
void skinny_linear_kernel(const SkinnyParams p) {
    constexpr int ROWS = EPI == SK_GLU ? 2 : 1;
    __shared__ float red[4][ROWS][MB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = SPLITK ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const int m0 = (int)blockIdx.y * MB;               // row group (the f32 frame steps of multi-stream sessions: M > 16 rows)
    const bool live = n < p.N;
    const int nn = live ? n : p.N - 1;
    float acc[ROWS][MB];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
    const int k_first = (SPLITK ? wave : 0) * 512 + lane * 8;
    const int k_step = SPLITK ? 2048 : 512;
    for (int k = k_first; k < p.K; k += k_step) {
        float w[ROWS][8];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) load8<F32>(p.W, (size_t)(nn * ROWS + r) * p.ldw + k, w[r]);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int mm = m0 + m < p.M ? m0 + m : p.M - 1;
            float a[8];
            load8<F32>(p.A, (size_t)mm * p.lda + k, a);
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r][m] = __builtin_fmaf(w[r][e], a[e], acc[r][m]);
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float v = acc[r][m];
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
            acc[r][m] = v;
        }
    if constexpr (SPLITK) {
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int m = 0; m < MB; ++m) red[wave][r][m] = acc[r][m];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[r][m] = red[0][r][m] + red[1][r][m] + red[2][r][m] + red[3][r][m];
    }
    if (!live) return;
    // lane m finishes row m (every lane holds all the sums after the butterfly)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int row = m0 + m;
        if (lane != m || row >= p.M) continue;
        if constexpr (EPI == SK_PLAIN) {
            const float v = act_apply(acc[0][m] + (p.bias ? p.bias[n] : 0.f), p.act);
            if (p.out32) p.out32[(size_t)row * p.ldo + n] = v;
            else p.out16[(size_t)row * p.ldo + n] = to_f16_sat(v);
        } else if constexpr (EPI == SK_GLU) {
            const float a = acc[0][m] + p.bias[2 * n], g = acc[1][m] + p.bias[2 * n + 1];
            p.out16[(size_t)row * p.ldo + n] = to_f16_sat(a / (1.0f + __expf(-g)));
        } else {
            float v = (acc[0][m] + (p.bias ? p.bias[n] : 0.f)) * p.alpha;
            if (p.res) v += p.res[(size_t)row * p.ldres + n];
            if (p.out32) p.out32[(size_t)row * p.ldo + n] = v;
            if (p.out16) p.out16[(size_t)row * p.ldo + n] = to_f16_sat(v);
        }
    }
}

// LayerNorm over the 256 features of each row of x32 [M][256] (one wave per row, 4 features per lane).
// normalise_out32: out32 = LN(x) (EPI_RES_LN) -- else out32 keeps the un-normalised stream (EPI_RES_SCALE_LN16).
__global__ __launch_bounds__(64)
void skinny_ln_kernel(float* __restrict__ x32, _Float16* __restrict__ out16, const float* __restrict__ gamma,
                      const float* __restrict__ beta, float eps, int normalise_out32, float* __restrict__ ln_out32 = nullptr) {
    const int m = blockIdx.x, lane = threadIdx.x;
    float4 v = *(const float4*)(x32 + (size_t)m * 256 + lane * 4);
    float s = v.x + v.y + v.z + v.w;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) s += __shfl_xor(s, k, 64);
    const float mean = s * (1.0f / 256.0f);
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    float q = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) q += __shfl_xor(q, k, 64);
    const float rstd = 1.0f / __builtin_sqrtf(q * (1.0f / 256.0f) + eps);
    float4 g = make_float4(1, 1, 1, 1), b = make_float4(0, 0, 0, 0);
    if (gamma) { g = *(const float4*)(gamma + lane * 4); b = *(const float4*)(beta + lane * 4); }
    const float4 y = make_float4(d0 * rstd * g.x + b.x, d1 * rstd * g.y + b.y, d2 * rstd * g.z + b.z, d3 * rstd * g.w + b.w);
    if (normalise_out32) *(float4*)(x32 + (size_t)m * 256 + lane * 4) = y;
    if (ln_out32) *(float4*)(ln_out32 + (size_t)m * 256 + lane * 4) = y;          // f32 copy of LN(x) beside the un-normalised stream
    if (out16) {
        f16x4 o;
        o[0] = to_f16_sat(y.x); o[1] = to_f16_sat(y.y); o[2] = to_f16_sat(y.z); o[3] = to_f16_sat(y.w);
        *(f16x4*)(out16 + (size_t)m * 256 + lane * 4) = o;
    }
}

template <int EPI, bool SPLITK, bool F32>
int launch_bucket(const SkinnyParams& p, hipStream_t stream) {
    const dim3 grid(SPLITK ? p.N : (p.N + 3) / 4, p.M > 16 ? (p.M + 15) / 16 : 1), block(256);
    if (p.M <= 1) hipLaunchKernelGGL((skinny_linear_kernel<1, EPI, SPLITK, F32>), grid, block, 0, stream, p);
    else if (p.M <= 2) hipLaunchKernelGGL((skinny_linear_kernel<2, EPI, SPLITK, F32>), grid, block, 0, stream, p);
    else if (p.M <= 4) hipLaunchKernelGGL((skinny_linear_kernel<4, EPI, SPLITK, F32>), grid, block, 0, stream, p);
    else if (p.M <= 8) hipLaunchKernelGGL((skinny_linear_kernel<8, EPI, SPLITK, F32>), grid, block, 0, stream, p);
    else if (p.M <= 12) hipLaunchKernelGGL((skinny_linear_kernel<12, EPI, SPLITK, F32>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((skinny_linear_kernel<16, EPI, SPLITK, F32>), grid, block, 0, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

template <int EPI, bool F32 = false>
int launch_skinny(const SkinnyParams& p, hipStream_t stream) {
    return p.K > 512 ? launch_bucket<EPI, true, F32>(p, stream) : launch_bucket<EPI, false, F32>(p, stream);
}

}  // namespace

bool eend_skinny_ok(const void* A, int lda, const void* W, int ldw, int M, int K) {
    return M >= 1 && M <= EEND_SKINNY_MAX_M && (K & 7) == 0 && (lda & 7) == 0 && (ldw & 7) == 0 &&
           (((size_t)A | (size_t)W) & 15) == 0;
}

int eend_launch_skinny_plain(const void* A, int lda, const void* W, int ldw, const float* bias, void* out16, int ldo,
                             int M, int N, int K, int act, hipStream_t stream) {
    SkinnyParams p{A, lda, W, ldw, bias, M, N, K, act, 1.0f, nullptr, 0, nullptr, (_Float16*)out16, ldo};
    return launch_skinny<SK_PLAIN>(p, stream);
}

int eend_launch_skinny_glu(const void* A, int lda, const void* W, int ldw, const float* bias, void* out16, int ldo,
                           int M, int N2, int K, hipStream_t stream) {
    SkinnyParams p{A, lda, W, ldw, bias, M, N2 / 2, K, 0, 1.0f, nullptr, 0, nullptr, (_Float16*)out16, ldo};
    return launch_skinny<SK_GLU>(p, stream);
}

// mode 0: out32 / out16 = v (EPI_RES_SCALE); 1: out32 = LN(v), out16 = f16 LN(v) (EPI_RES_LN);
// 2: out32 = v, out16 = f16 LN(v) (EPI_RES_SCALE_LN16).  N = 256.  Modes 1 / 2 need out32 as the staging row buffer.
int eend_launch_skinny_res(const void* A, int lda, const void* W, int ldw, const float* bias, const float* res, float alpha,
                           const float* gamma, const float* beta, float eps, float* out32, void* out16, int M, int K,
                           int mode, hipStream_t stream) {
    if (mode != 0 && !out32) return EEND_EINVAL;
    SkinnyParams p{A, lda, W, ldw, bias, M, 256, K, 0, alpha, res, 256, out32, mode == 0 ? (_Float16*)out16 : nullptr, 256};
    int rc = launch_skinny<SK_RES>(p, stream);
    if (rc != EEND_OK || mode == 0) return rc;
    hipLaunchKernelGGL(skinny_ln_kernel, dim3(M), dim3(64), 0, stream, out32, (_Float16*)out16, gamma, beta, eps, mode == 1 ? 1 : 0);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// The same kernels on f32 activations AND f32 weights (torch nn.Linear layout), f32 out: the LS decoder's frame step.
int eend_launch_skinny_plain_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* out32, int ldo, int M, int N,
                                 int K, int act, hipStream_t stream) {
    if (!A || !W || !out32 || M < 1 || (K & 7) || (lda & 3) || (ldw & 3) || (((size_t)A | (size_t)W) & 15))
        return EEND_EINVAL;
    // more than 16 rows (multi-stream sessions): the f32 MFMA kernel of gemm_f32.hip instead of serial 16-row groups
    if (eend_linear_f32_mfma_ok(A, lda, W, ldw, M, N, K, ldo) && !((size_t)out32 & 15) && !(bias && ((size_t)bias & 15)))
        return eend_launch_linear_f32_mfma(A, lda, W, ldw, bias, nullptr, 0, 1.0f, 0, act, out32, ldo, M, N, K, stream);
    SkinnyParams p{A, lda, W, ldw, bias, M, N, K, act, 1.0f, nullptr, 0, out32, nullptr, ldo};
    return launch_skinny<SK_PLAIN, true>(p, stream);
}

int eend_launch_layernorm_rows_f32(const float* x, const float* gamma, const float* beta, float eps, float* out32, int M, hipStream_t stream) {
    if (!x || !gamma || !beta || !out32 || M <= 0) return EEND_EINVAL;
    hipLaunchKernelGGL(skinny_ln_kernel, dim3(M), dim3(64), 0, stream, (float*)x, (_Float16*)nullptr, gamma, beta, eps, 0, out32);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_skinny_res_f32(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, float alpha,
                               const float* gamma, const float* beta, float eps, float* out32, void* out16, int M, int K, int mode,
                               hipStream_t stream, float* ln_out32) {
    if (!A || !W || !out32 || M < 1 || (K & 7) || (lda & 3) || (ldw & 3) || (((size_t)A | (size_t)W) & 15))
        return EEND_EINVAL;
    int rc;
    if (mode != 0 && eend_linear_f32_mfma_ok(A, lda, W, ldw, M, 256, K, 256) && !((size_t)out32 & 15) && !(bias && ((size_t)bias & 15)) &&
        !(res && ((size_t)res & 15))) {
        rc = eend_launch_linear_f32_mfma(A, lda, W, ldw, bias, res, 256, alpha, 1, 0, out32, 256, M, 256, K, stream);
    } else {
        SkinnyParams p{A, lda, W, ldw, bias, M, 256, K, 0, alpha, res, 256, out32, mode == 0 ? (_Float16*)out16 : nullptr, 256};
        rc = launch_skinny<SK_RES, true>(p, stream);
    }
    if (rc != EEND_OK || mode == 0) return rc;
    hipLaunchKernelGGL(skinny_ln_kernel, dim3(M), dim3(64), 0, stream, out32, (_Float16*)out16, gamma, beta, eps, mode == 1 ? 1 : 0, ln_out32);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
