// Fused position-wise feed-forward block:
//     y = epilogue( act(X W1^T + b1) W2^T + b2 )        X: [M][256] f16, W1: [F][256], W2: [256][F]
// in ONE launch.  The F-wide hidden activations (2048 for FS-EEND / the decoders, 1024 for the
// Conformer FFNs) never leave the CU: they are produced 64 columns at a time into LDS and
// consumed immediately by the second GEMM.  Versus linear1 + linear2 launches this removes the
// M x F f16 round trip through HBM (805 MB written + 805 MB read per decoder layer at B=64, C=6,
// measured HBM-bound at 5.7 TB/s) and one of the two X-tile staging passes.
//
// Replaces: nn.TransformerEncoderLayer's linear1/ReLU/linear2 + residual + norm2 (FS model :147),
// _ff_block + norm22 of the fusion layers (FS merge_tfm_encoder.py:374,397-399; LS
// merge_retnet_layer.py:252,309-311) and FeedForwardModule + half-step residual of the Conformer
// block (LS conformer/feed_forward.py:47-57, encoder.py:76-110).
//
// Block = 128 rows, 8 waves (512 threads), 1 block/CU, LDS 144 KB:
//   Xs  [4 k-tiles][128][64]  64 KB   resident for the whole block
//   W1s [4 k-tiles][ 64][64]  32 KB   hidden units f0..f0+63 (all K)
//   Hs  [128][64]             16 KB   act(X W1c^T + b1c), f16
//   W2s [256][64]             32 KB   W2[:, f0..f0+63]
// all as XOR-swizzled [rows][128 B] tiles (common.h swz128: conflict-free ds_read_b128 fragments).
// Per 64-wide hidden chunk: GEMM1 (waves 4(m) x 2(f), 32x32 each) -> Hs -> GEMM2 (waves 2(m) x 4(n),
// 64x64 each, fp32 accumulators stay in registers across the whole F loop); the next chunk's
// weights are prefetched into registers during the chunk and stored to LDS behind a barrier.
// Both GEMMs feed the weights as the MFMA "A" operand, so a lane owns 4 consecutive output features
// of one token (8-byte Hs writes, float4 / f16x4 epilogue, lane-local LayerNorm partials).
#include "common.h"
#include "kernels.h"
#include <type_traits>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int KD = 256;            // model dim (K of GEMM1, N of GEMM2)
constexpr int FC = 64;             // hidden chunk
constexpr int NT = 512;
constexpr int XS_BYTES = 4 * BM * 128;          // 65536
constexpr int W1_BYTES = 4 * FC * 128;          // 32768
constexpr int HS_BYTES = BM * 128;              // 16384
constexpr int W2_BYTES = KD * 128;              // 32768
constexpr int SMEM_BYTES = XS_BYTES + W1_BYTES + HS_BYTES + W2_BYTES;   // 147456

template <int ACT, int EPI>
__global__ __launch_bounds__(NT)
void ffn_fused_kernel(const FfnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;
    char* W1s = smem + XS_BYTES;
    char* Hs = W1s + W1_BYTES;
    char* W2s = Hs + HS_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fkg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int nF = p.F / FC;

    const _Float16* __restrict__ X = (const _Float16*)p.X;
    const _Float16* __restrict__ W1 = (const _Float16*)p.W1;
    const _Float16* __restrict__ W2 = (const _Float16*)p.W2;

    // ---- X tile: 128 rows x 512 B, once (rows >= M clamp to a valid row; never stored)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + i * NT;                       // 4096 chunks: row = q / 32, k-chunk = q % 32
        const int row = q >> 5, c32 = q & 31;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        const u32x4 v = *(const u32x4*)(X + (size_t)m * p.ldx + c32 * 8);
        *(u32x4*)(Xs + (c32 >> 3) * (BM * 128) + swz128(row, c32 & 7)) = v;
    }
    // ---- weight chunk staging: 4 + 4 16-byte pieces per thread
    u32x4 w1r[4], w2r[4];
    auto wload = [&](int f0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * NT;                   // W1 chunk: 64 rows x 32 k-chunks
            w1r[i] = *(const u32x4*)(W1 + (size_t)(f0 + (q >> 5)) * KD + (q & 31) * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * NT;                   // W2 chunk: 256 rows x 8 f-chunks
            w2r[i] = *(const u32x4*)(W2 + (size_t)(q >> 3) * p.F + f0 + (q & 7) * 8);
        }
    };
    auto wstore1 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * NT;
            const int row = q >> 5, c32 = q & 31;
            *(u32x4*)(W1s + (c32 >> 3) * (FC * 128) + swz128(row, c32 & 7)) = w1r[i];
        }
    };
    auto wstore2 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * NT;
            *(u32x4*)(W2s + swz128(q >> 3, q & 7)) = w2r[i];
        }
    };

    // GEMM1 wave tile: 32 tokens x 32 hidden; GEMM2 wave tile: 64 tokens x 64 outputs
    const int g1m = (wave >> 1) * 32, g1f = (wave & 1) * 32;
    const int g2m = (wave >> 2) * 64, g2n = (wave & 3) * 64;

    f32x4 acc[4][4];                                        // [n frag][m frag]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto chunk = [&](int f0, bool store_next_w1) __attribute__((always_inline)) {
        // ---- GEMM1: h[f][m] = sum_k W1c[f][k] X[m][k]   (A = W1 rows, B = X rows)
        f32x4 h[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) h[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i] = *(const f16x8*)(W1s + kt * (FC * 128) + swz128(g1f + i * 16 + frow, ks * 4 + fkg));
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    b[j] = *(const f16x8*)(Xs + kt * (BM * 128) + swz128(g1m + j * 16 + frow, ks * 4 + fkg));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        h[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], h[i][j], 0, 0, 0);
            }
        // bias + activation, f16, into Hs[m][f] (lane: 4 consecutive f of token m)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int fl = g1f + i * 16 + fkg * 4;          // hidden index inside the chunk
            const float4 bb = *(const float4*)(p.b1 + f0 + fl);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v0 = h[i][j][0] + bb.x, v1 = h[i][j][1] + bb.y, v2 = h[i][j][2] + bb.z, v3 = h[i][j][3] + bb.w;
                if (ACT == 1) {
                    v0 = __builtin_fmaxf(v0, 0.f); v1 = __builtin_fmaxf(v1, 0.f);
                    v2 = __builtin_fmaxf(v2, 0.f); v3 = __builtin_fmaxf(v3, 0.f);
                } else if (ACT == 2) {
                    v0 = v0 / (1.0f + __expf(-v0)); v1 = v1 / (1.0f + __expf(-v1));
                    v2 = v2 / (1.0f + __expf(-v2)); v3 = v3 / (1.0f + __expf(-v3));
                }
                f16x4 o;
                o[0] = to_f16_sat(v0); o[1] = to_f16_sat(v1); o[2] = to_f16_sat(v2); o[3] = to_f16_sat(v3);
                const int row = g1m + j * 16 + frow;
                // element offset fl inside the 64-wide row: 16-B chunk fl>>3, 8-B half (fl>>2)&1
                *(f16x4*)(Hs + swz128(row, fl >> 3) + ((fl >> 2) & 1) * 8) = o;
            }
        }
        __syncthreads();             // barrier A: Hs visible; every wave is done with W1s of this chunk
        if (store_next_w1) wstore1();   // next chunk's W1 slice (prefetched in registers) -> W1s, consumed after barrier B
        // ---- GEMM2: acc[n][m] += sum_f W2c[n][f] H[m][f]   (A = W2 rows, B = H rows)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const f16x8*)(W2s + swz128(g2n + i * 16 + frow, ks * 4 + fkg));
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(const f16x8*)(Hs + swz128(g2m + j * 16 + frow, ks * 4 + fkg));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    // Two barriers per 64-wide chunk: A (inside chunk(): Hs complete, W1s free -> store next W1 slice)
    // and B (GEMM2 done by every wave: Hs and W2s free -> store next W2 slice; it becomes visible at the
    // next barrier A, before the next GEMM2 reads it; the W1 slice stored after A is visible after B).
    wload(0);
    wstore1();
    wstore2();
    __syncthreads();
    for (int c = 0; c < nF - 1; ++c) {
        wload((c + 1) * FC);                 // next chunk's weights in flight during this chunk
        chunk(c * FC, true);
        __syncthreads();                     // barrier B
        wstore2();
    }
    chunk((nF - 1) * FC, false);
    __syncthreads();

    // ---- epilogue: y = (acc + b2) * alpha + res ; LayerNorm over the 256 features of each token
    // acc[i][j][r]: n = g2n + i*16 + fkg*4 + r ; m = m0 + g2m + j*16 + frow
    float* red = (float*)smem;                               // [4 n-waves][128 rows]
    const int wn = wave & 3;
    const int N0 = g2n + fkg * 4;
    const int M0 = m0 + g2m + frow;
    float4 b4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b4[i] = *(const float4*)(p.b2 + N0 + i * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = M0 + j * 16;
        const bool ok = m < p.M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 r = make_float4(0, 0, 0, 0);
            if (p.res && ok) r = *(const float4*)(p.res + (size_t)m * KD + N0 + i * 16);
            acc[i][j][0] = (acc[i][j][0] + b4[i].x) * p.alpha + r.x;
            acc[i][j][1] = (acc[i][j][1] + b4[i].y) * p.alpha + r.y;
            acc[i][j][2] = (acc[i][j][2] + b4[i].z) * p.alpha + r.z;
            acc[i][j][3] = (acc[i][j][3] + b4[i].w) * p.alpha + r.w;
        }
    }
    auto block_rowsum = [&](float (&part)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            part[j] = wave_xor_add(part[j], 16);
            part[j] = wave_xor_add(part[j], 32);
        }
        __syncthreads();
        if (fkg == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wn * BM + g2m + j * 16 + frow] = part[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = g2m + j * 16 + frow;
            part[j] = red[row] + red[BM + row] + red[2 * BM + row] + red[3 * BM + row];
        }
    };
    float part[4], mean[4], rstd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        part[j] = s;
    }
    block_rowsum(part);
#pragma unroll
    for (int j = 0; j < 4; ++j) mean[j] = part[j] * (1.0f / KD);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[j]; s += d * d; }
        part[j] = s;
    }
    block_rowsum(part);
#pragma unroll
    for (int j = 0; j < 4; ++j) rstd[j] = 1.0f / __builtin_sqrtf(part[j] * (1.0f / KD) + p.eps);

    float* __restrict__ o32 = p.out32;
    _Float16* __restrict__ o16 = (_Float16*)p.out16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = N0 + i * 16;
        const float4 g = *(const float4*)(p.gamma + n), be = *(const float4*)(p.beta + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = M0 + j * 16;
            if (m >= p.M) continue;
            const float v0 = (acc[i][j][0] - mean[j]) * rstd[j] * g.x + be.x;
            const float v1 = (acc[i][j][1] - mean[j]) * rstd[j] * g.y + be.y;
            const float v2 = (acc[i][j][2] - mean[j]) * rstd[j] * g.z + be.z;
            const float v3 = (acc[i][j][3] - mean[j]) * rstd[j] * g.w + be.w;
            if (EPI == FFN_EPI_RES_SCALE_LN16)               // residual stream stays un-normalised
                *(float4*)(o32 + (size_t)m * KD + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            else
                *(float4*)(o32 + (size_t)m * KD + n) = make_float4(v0, v1, v2, v3);
            f16x4 o;
            o[0] = to_f16_sat(v0); o[1] = to_f16_sat(v1); o[2] = to_f16_sat(v2); o[3] = to_f16_sat(v3);
            *(f16x4*)(o16 + (size_t)m * KD + n) = o;
        }
    }
}

template <int ACT, int EPI>
int launch(const FfnParams& p, hipStream_t stream) {
    static bool attr_done = false;
    auto kern = ffn_fused_kernel<ACT, EPI>;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES) != hipSuccess)
            return EEND_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((p.M + BM - 1) / BM), dim3(NT), SMEM_BYTES, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

}  // namespace

int eend_launch_ffn_fused(const FfnParams& p, int act, int epi, hipStream_t stream) {
    if (p.M <= 0 || p.F <= 0 || (p.F % FC) != 0 || (p.ldx & 7) || !p.X || !p.W1 || !p.W2 || !p.b1 || !p.b2 || !p.gamma ||
        !p.beta || !p.out32 || !p.out16)
        return EEND_EINVAL;
    if (epi == FFN_EPI_RES_LN) {
        if (act == 1) return launch<1, FFN_EPI_RES_LN>(p, stream);
        if (act == 2) return launch<2, FFN_EPI_RES_LN>(p, stream);
    } else if (epi == FFN_EPI_RES_SCALE_LN16) {
        if (act == 1) return launch<1, FFN_EPI_RES_SCALE_LN16>(p, stream);
        if (act == 2) return launch<2, FFN_EPI_RES_SCALE_LN16>(p, stream);
    }
    return EEND_EINVAL;
}
