// f32 linear layers for a few hundred rows on the exact-f32 MFMA (round 5): the all-f32 LS-EEND frame step of a multi-stream session
// (LS-EEND/streaming_infer_dia.py:52-97 run for B streams at once: B rows in the encoder, B * max_nspks in the decoder; DESIGN 9a: the
// frame steps run in f32 because the retention's per-head LayerNorm amplifies f16 operand rounding over an hour-long stream).
//
//   y[m][n] = sum_k A[m][k] W[n][k] + bias[n]        A f32 [M][lda], W f32 [N][ldw] (torch nn.Linear layout), all f32
//
// Round 4 served M > 16 rows with the wave-per-output-feature kernel of skinny.hip in serial 16-row groups: every group re-read the
// whole weight matrix from L2 and every wave re-read its group's activations (1.3 GB of cache traffic for one 640 x 2048 x 256 layer);
// 64 streams cost 1.74 ms per frame.  Here a wave owns a 16-feature x 64-row block: per 16-wide k block one float4 of W and four of A
// per lane feed 16 v_mfma_f32_16x16x4_f32 -- the k index inside a block is permuted the same way for both operands
// (lane (f, kk) holds k = kb + 4 kk + s at step s), which a contraction does not care about.
//   K <= 768 : the 4 waves of a workgroup take 4 neighbouring feature blocks (64 features x 64 rows per workgroup)
//   K  > 768 : they split K (FFN2, the look-ahead conv) and reduce through LDS (16 features x 64 rows per workgroup)
// Epilogues: act(y) -> out32, or y * alpha + res -> out32 (the LayerNorm that follows is skinny.hip's row kernel).
#include "common.h"
#include "kernels.h"

namespace {

struct LinF32Params {
    const float* A; int lda;
    const float* W; int ldw;
    const float* bias;
    int M, N, K;
    int act;                     // 0 none, 1 relu, 2 swish (plain epilogue)
    float alpha;
    const float* res; int ldres; // residual epilogue when res_mode
    int res_mode;
    float* out32; int ldo;
};

DEV float act_f32(float v, int act) {
    if (act == 1) return __builtin_fmaxf(v, 0.f);
    if (act == 2) return v / (1.0f + __expf(-v));
    return v;
}

template <bool SPLITK>
__global__ __launch_bounds__(256)
void linear_f32_mfma_kernel(const LinF32Params p) {
    __shared__ float red[SPLITK ? 3 : 1][4][4][64];         // [wave 1..3][row block][r][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f = lane & 15, kk = lane >> 4;
    const int n0 = SPLITK ? (int)blockIdx.x * 16 : ((int)blockIdx.x * 4 + wave) * 16;
    const int m0 = (int)blockIdx.y * 64;
    const bool live = n0 < p.N;
    const int nr = live ? n0 + f : p.N - 16 + f;            // N is a multiple of 16
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wrow = p.W + (size_t)nr * p.ldw + kk * 4;
    const float* arow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + j * 16 + f;
        m = m < p.M ? m : p.M - 1;                          // rows beyond M: clamped duplicates, never stored
        arow[j] = p.A + (size_t)m * p.lda + kk * 4;
    }
    // K is a multiple of 16; SPLITK: wave w takes the k blocks w, w + 4, ...
    const int kstep = SPLITK ? 64 : 16;
    for (int kb = SPLITK ? wave * 16 : 0; kb < p.K; kb += kstep) {
        const float4 w4 = *(const float4*)(wrow + kb);
        float4 a4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a4[j] = *(const float4*)(arow[j] + kb);
        const float ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float as = s == 0 ? a4[j].x : s == 1 ? a4[j].y : s == 2 ? a4[j].z : a4[j].w;
                // D[feature][row] += W[feature][k] A[row][k]: lane holds features n0 + kk*4 + r of row m0 + j*16 + f
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[s], as, acc[j], 0, 0, 0);
            }
    }
    if constexpr (SPLITK) {
        if (wave != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave - 1][j][r][lane] = acc[j][r];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] += red[0][j][r][lane] + red[1][j][r][lane] + red[2][j][r][lane];
    }
    if (!live) return;
    const int nf = n0 + kk * 4;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) b4 = *(const float4*)(p.bias + nf);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + j * 16 + f;
        if (m >= p.M) continue;
        float4 v = make_float4(acc[j][0] + b4.x, acc[j][1] + b4.y, acc[j][2] + b4.z, acc[j][3] + b4.w);
        if (p.res_mode) {
            v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
            if (p.res) {
                const float4 r4 = *(const float4*)(p.res + (size_t)m * p.ldres + nf);
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            }
        } else {
            v.x = act_f32(v.x, p.act); v.y = act_f32(v.y, p.act); v.z = act_f32(v.z, p.act); v.w = act_f32(v.w, p.act);
        }
        *(float4*)(p.out32 + (size_t)m * p.ldo + nf) = v;
    }
}

}  // namespace

bool eend_linear_f32_mfma_ok(const float* A, int lda, const float* W, int ldw, int M, int N, int K, int ldo) {
    return M > EEND_SKINNY_MAX_M && (N & 15) == 0 && (K & 15) == 0 && (lda & 3) == 0 && (ldw & 3) == 0 && (ldo & 3) == 0 &&
           (((size_t)A | (size_t)W) & 15) == 0;
}

// res_mode 0: out32 = act(y + bias); 1: out32 = (y + bias) * alpha + res
int eend_launch_linear_f32_mfma(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldres, float alpha,
                                int res_mode, int act, float* out32, int ldo, int M, int N, int K, hipStream_t stream) {
    if (!A || !W || !out32 || !eend_linear_f32_mfma_ok(A, lda, W, ldw, M, N, K, ldo) || (bias && ((size_t)bias & 15)) ||
        (res && (((size_t)res & 15) || (ldres & 3))) || ((size_t)out32 & 15))
        return EEND_EINVAL;
    LinF32Params p{A, lda, W, ldw, bias, M, N, K, act, alpha, res, ldres, res_mode, out32, ldo};
    const int mb = (M + 63) / 64;
    if (K > 768) hipLaunchKernelGGL(linear_f32_mfma_kernel<true>, dim3(N / 16, mb), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(linear_f32_mfma_kernel<false>, dim3((N / 16 + 3) / 4, mb), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
