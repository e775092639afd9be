// Decoder input fan-out (round 4): attr0[(b, c), t, :] = emb[(b, t), :] W1^T + pc[c, :] for the C speaker slots -- the operator of the
// EPI_CONVERT epilogue of gemm.hip (FS model :113-114 factored: convert([emb; pe_c]) = W[:, :D] emb + (W[:, D:] pe_c + b)), as a
// store-shaped kernel: the GEMM is 4 GFLOP, the output C x 16 MB.  The generic epilogue wrote it in 8-byte pieces (43.6 us for
// 100 MB = 2.3 TB/s).  Here the weights are stationary (a wave owns 64 of the 256 output features, 32 A fragments = 128 VGPRs for
// the whole launch, rows permuted so that a lane ends up with 16 CONSECUTIVE features of its token), a tile is 32 embedding rows,
// and every slot's row leaves as 32 contiguous bytes per lane, 128 contiguous bytes per token and wave.  No LDS tile, no barrier
// in the loop.
#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256, 1)
void convert_fanout_rows_kernel(const _Float16* __restrict__ E, const _Float16* __restrict__ W1, const float* __restrict__ pc,
                                float* __restrict__ out32, _Float16* __restrict__ out16, int B, int Tp, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* pcl = (float*)smem;                            // [C][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, g = lane >> 4;
    for (int i = tid; i < C * 256; i += 256) pcl[i] = pc[i];
    // w[ks][nf]: A fragment row rho = (g', r) <-> feature 64 wave + g'*16 + nf*4 + r
    f16x8 w[8][4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
        const _Float16* src = W1 + (size_t)(64 * wave + (frow >> 2) * 16 + nf * 4 + (frow & 3)) * 256 + g * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w[ks][nf] = *(const f16x8*)(src + ks * 32);
    }
    __syncthreads();
    const int M = B * Tp, ntiles = (M + 31) / 32;
    const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)E, 0, M * 512, 0x00020000);
    f16x8 xf[8][2], xn[8][2];
    auto load_x = [&](int tile, f16x8 (&x)[8][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = (tile * 32 + j * 16 + frow) * 512 + g * 16;     // rows beyond M read as zeros
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) x[ks][j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(re, off + ks * 64, 0, 0));
        }
    };
    if ((int)blockIdx.x < ntiles) load_x(blockIdx.x, xf);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x4 acc[4][2];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) { acc[nf][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[nf][1] = acc[nf][0]; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[nf][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks][nf], xf[ks][j], acc[nf][j], 0, 0, 0);
        const int ntile = tile + (int)gridDim.x;
        if (ntile < ntiles) load_x(ntile, xn);            // in flight under this tile's stores
        const int f0 = 64 * wave + g * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = tile * 32 + j * 16 + frow;
            if (m < M) {
                const int b = m / Tp, t = m - b * Tp;
                for (int c = 0; c < C; ++c) {
                    const size_t row = ((size_t)b * C + c) * Tp + t;
                    f32x4 y[4];
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) y[nf] = acc[nf][j] + *(const f32x4*)(pcl + c * 256 + f0 + nf * 4);
                    f16x8 o0, o1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o0[q] = to_f16_sat(y[0][q]); o0[4 + q] = to_f16_sat(y[1][q]);
                        o1[q] = to_f16_sat(y[2][q]); o1[4 + q] = to_f16_sat(y[3][q]);
                    }
                    *(f16x8*)(out16 + row * 256 + f0) = o0;
                    *(f16x8*)(out16 + row * 256 + f0 + 8) = o1;
                    if (out32) {
#pragma unroll
                        for (int nf = 0; nf < 4; ++nf) *(f32x4*)(out32 + row * 256 + f0 + nf * 4) = y[nf];
                    }
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { xf[ks][0] = xn[ks][0]; xf[ks][1] = xn[ks][1]; }
    }
}

}  // namespace

int eend_launch_convert_fanout_rows(const void* E, const void* W1, const float* pc, float* out32, void* out16, int B, int Tp, int C,
                                    hipStream_t stream) {
    if (!E || !W1 || !pc || !out16 || B <= 0 || Tp <= 0 || C <= 0 || C > 32 || (long)B * Tp * 512 >= (1L << 31)) return EEND_EINVAL;
    const int ncu = eend_cu_count();
    const int ntiles = (B * Tp + 31) / 32;
    hipLaunchKernelGGL(convert_fanout_rows_kernel, dim3(ntiles < 2 * ncu ? ntiles : 2 * ncu), dim3(256), C * 1024, stream, (const _Float16*)E,
                       (const _Float16*)W1, pc, out32, (_Float16*)out16, B, Tp, C);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
