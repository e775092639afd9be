// Decoder input in f32:  attr0[(b,c)][t][:] = W1 emb[b][t][:] + pc[c][:]   (LS model :216-217, `convert(cat(emb, pe_c))` in the
// factored form of DESIGN 3) on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32), emb and W1 as f32.
//
// Why: the LS-EEND decoder's retention normalises by a near-zero-mean statistic with eps 1e-6 (merge_retnet_layer.py /
// retention.py group_norm), which amplifies the f16 operand rounding of THIS linear up to ~30x on frames whose attractor slots are
// nearly constant (7.8e-4 in the logits from this linear alone on a long stream).  With 12 speaker slots
// (conf/*dihard*.yaml: max_speakers 10 + 2) the batch forward left the 1e-3 bar (golden ls_c12_T1000: 1.3e-3) -- the f32 form costs
// 4.3 GFLOP per 32768 frames (~40 us) next to the 0.5 GB this stage writes anyway.
//
// One workgroup (4 waves) per 64 frames: the emb rows sit in LDS (64 KB), a wave owns 64 output features (4 x 4 accumulator
// fragments), W1 fragments come straight from global memory (256 KB, L2-resident, one float per lane per fragment); the result
// tile is staged through LDS and written once per speaker slot with pc[c] added (f32 + f16 copies, whole rows).
#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256)
void convert_fanout_f32_kernel(const float* __restrict__ emb, const float* __restrict__ W, int ldw, const float* __restrict__ pc,
                               float* __restrict__ out32, _Float16* __restrict__ out16, _Float16* __restrict__ out16lo, int B, int Tp, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = (float*)smem;                            // [64 rows][256 + 4] (padded against bank conflicts of the fragment reads)
    constexpr int LD = 260;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * 64, M = (long)B * Tp;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {     // 64 rows x 64 float4
        const int row = i >> 6, c4 = i & 63;
        const long r = r0 + row < M ? r0 + row : M - 1;
        *(float4*)(tile + row * LD + c4 * 4) = *(const float4*)(emb + r * 256 + c4 * 4);
    }
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;              // fragment row / k index of this lane
    f32x4 acc[4][4];                                       // [token fragment][feature fragment]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wrow = W + (size_t)(wave * 64 + fr) * ldw + fk;
#pragma unroll 4
    for (int k = 0; k < 256; k += 4) {
        float af[4], wf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = tile[(a * 16 + fr) * LD + k + fk];
#pragma unroll
        for (int b = 0; b < 4; ++b) wf[b] = wrow[(size_t)b * 16 * ldw + k];
        // D[feature][token] = sum_k W[feature][k] emb[token][k]: lane holds token = lane & 15, features (lane >> 4) * 4 + r
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b], af[a], acc[a][b], 0, 0, 0);
    }
    __syncthreads();                                       // every wave is done with the emb tile: it becomes the result tile
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) *(f32x4*)(tile + (a * 16 + fr) * LD + wave * 64 + b * 16 + fk * 4) = acc[a][b];
    __syncthreads();
    // fan out over the speaker slots: row (b, t) -> rows (b*C + c, t), + pc[c]
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int row = i >> 6, c4 = i & 63;
        const long r = r0 + row;
        if (r >= M) continue;
        const long b = r / Tp, t = r - b * Tp;
        const float4 y = *(const float4*)(tile + row * LD + c4 * 4);
        for (int c = 0; c < C; ++c) {
            const float4 p4 = *(const float4*)(pc + (size_t)c * 256 + c4 * 4);
            const float4 v = make_float4(y.x + p4.x, y.y + p4.y, y.z + p4.z, y.w + p4.w);
            const size_t o = (((size_t)b * C + c) * Tp + t) * 256 + c4 * 4;
            if (out32) *(float4*)(out32 + o) = v;
            f16x4 h;
            h[0] = to_f16_sat(v.x); h[1] = to_f16_sat(v.y); h[2] = to_f16_sat(v.z); h[3] = to_f16_sat(v.w);
            *(f16x4*)(out16 + o) = h;
            if (out16lo) {                                  // the f16 remainder of the f32 row: the retention's query path reads hi + lo
                f16x4 l;
                l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
                l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
                *(f16x4*)(out16lo + o) = l;
            }
        }
    }
}

}  // namespace

int eend_launch_convert_fanout_f32(const float* emb, const float* W, int ldw, const float* pc, float* out32, void* out16, void* out16lo,
                                   int B, int Tp, int C, hipStream_t stream) {
    if (!emb || !W || !pc || !out16 || B <= 0 || Tp <= 0 || C <= 0 || ldw < 256) return EEND_EINVAL;
    static EendOncePerDevice attr_once;
    constexpr int smem_bytes = 64 * 260 * 4;
    if (!eend_set_dynamic_lds(attr_once, (const void*)convert_fanout_f32_kernel, smem_bytes)) return EEND_ELAUNCH;
    const long M = (long)B * Tp;
    hipLaunchKernelGGL(convert_fanout_f32_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), smem_bytes, stream, emb, W, ldw, pc, out32,
                       (_Float16*)out16, (_Float16*)out16lo, B, Tp, C);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
