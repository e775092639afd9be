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
// Persistent kernel (grid = #CUs, one 512-thread block per CU, 160 KB LDS, see the MODE comments below):
// a tile is 128 rows; the wave's X fragments (32 tokens x 256) live in registers; W1 / W2 slices of 64 hidden
// units arrive by LDS-DMA into double buffers ([rows][128 B] XOR-swizzled images, common.h swz128:
// conflict-free ds_read_b128 fragments); Hs is double-buffered, so one barrier per hidden chunk separates
// "GEMM1 of chunk c+1 (waves 4(m) x 2(f), 32 x 32) + GEMM2 of chunk c (waves 2(m) x 4(n), 64 x 64, fp32
// accumulators live across the whole F loop)".  Both GEMMs feed the weights as the MFMA "A" operand, so a
// lane owns 4 consecutive output features of one token (8-byte Hs writes, lane-local LayerNorm partials).
#include "common.h"
#include "kernels.h"
#include <type_traits>
#include <utility>
#include <cstdlib>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifndef EEND_FFN_PRE_SRC
#define EEND_FFN_PRE_SRC 2       // how the plain PRE kernel fetches its A rows (see proj_ln_phase); 0 = the older per-wave form (A/B)
#endif

// Perf-study build (-DEEND_FFN_TRACE, tools/ffn_trace.py): s_memtime stamps of the tile phases of thread 0 of every
// workgroup, read back through eend_debug_ffn_trace; never defined in the shipped library.
#ifdef EEND_FFN_TRACE
__device__ unsigned long long g_ffn_trace[256 * 16 * 12];
#define FFN_STAMP(k)                                                                                              \
    do {                                                                                                          \
        if (threadIdx.x == 0 && (tile - (int)blockIdx.x) / (int)gridDim.x < 16)                                    \
            g_ffn_trace[((size_t)blockIdx.x * 16 + (tile - (int)blockIdx.x) / (int)gridDim.x) * 12 + (k)] =        \
                __builtin_amdgcn_s_memtime();                                                                     \
    } while (0)
#else
#define FFN_STAMP(k) do {} while (0)
#endif

// dropout without the "off" branch (thresh24 == 0 keeps every element: (h >> 8) >= 0, and the host passes scale = 1 with it): the uniform
// branch of drop_apply, taken 16 times per hidden chunk, splits the MFMA / LDS schedule of the chunk loop into pieces
DEV float drop_nb(const DropSpec d, float v, unsigned a, unsigned b) { return drop_keep(d, a, b) ? v * d.scale : 0.f; }

constexpr int BM = 128;
constexpr int KD = 256;            // model dim (K of GEMM1, N of GEMM2)
constexpr int FC = 64;             // hidden chunk
constexpr int NT = 512;
constexpr int W1_BYTES = 4 * FC * 128;          // 32768
constexpr int HS_BYTES = BM * 128;              // 16384
constexpr int W2_BYTES = KD * 128;              // 32768

// Epilogue: the accumulators hold res / alpha + b2 + h W2^T (they were seeded in the prologue); scale, LayerNorm
// over the 256 features of each token, and both output tiles go through LDS so that every global store
// instruction writes whole rows (64 lanes x 16 B contiguous) instead of 16 token-row segments of 64 B (fp32) /
// 32 B (f16): measured, the segment stores drained at 2.7 TB/s and were the largest fixed cost of the kernel.
struct RowsContiguous {              // tile row -> global row (or -1): plain 128-row tiles
    int m0, M;
    __device__ __forceinline__ long operator()(int r) const { return m0 + r < M ? (long)(m0 + r) : -1L; }
};

// TRAIN (round 5): the accumulators hold b2 + h W2^T only; dropout of the sub-layer output, scale and residual happen here, and the LayerNorm
// also saves 1/sigma per row and the normalised pre-affine rows (third staged tile) for the backward.
template <int EPI, bool TRAIN, class RowMap>
__device__ __forceinline__ void ffn_epilogue(const FfnParams& p, f32x4 (&acc)[4][4], char* smem, int wave, int frow, int fkg,
                                             int g2m, int g2n, const RowMap rowmap) {
    // acc[i][j][r]: feature n = g2n + i*16 + fkg*4 + r ; tile row = g2m + j*16 + frow
    float* red = (float*)(smem + 131072);                    // [4 n-waves][128 rows]; clear of the staging tiles
    const int wn = wave & 3;
    const int N0 = g2n + fkg * 4;
    if constexpr (TRAIN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long gr = rowmap(g2m + j * 16 + frow);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned n = (unsigned)(N0 + i * 16);
                float4 r4 = make_float4(0, 0, 0, 0);
                if (gr >= 0) r4 = *(const float4*)(p.res + (size_t)gr * KD + n);
                const unsigned m = (unsigned)(gr >= 0 ? gr : 0);
                acc[i][j][0] = drop_nb(p.drop2, acc[i][j][0], m, n) * p.alpha + r4.x;
                acc[i][j][1] = drop_nb(p.drop2, acc[i][j][1], m, n + 1) * p.alpha + r4.y;
                acc[i][j][2] = drop_nb(p.drop2, acc[i][j][2], m, n + 2) * p.alpha + r4.z;
                acc[i][j][3] = drop_nb(p.drop2, acc[i][j][3], m, n + 3) * p.alpha + r4.w;
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] *= p.alpha;
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
    if constexpr (TRAIN) {
        if (wn == 0 && fkg == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long gr = rowmap(g2m + j * 16 + frow);
                if (gr >= 0) p.rstat[gr] = rstd[j];
            }
        }
    }

    float* __restrict__ o32 = p.out32;
    _Float16* __restrict__ o16 = (_Float16*)p.out16;
    {
        const int lane = frow + fkg * 16;
        f16x4 h16[4][4], l16[4][4];
        // fp32 tile -> LDS [128 rows][1 KB], 16-B chunk c of row r at chunk c ^ (r & 7)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = N0 + i * 16;
            const float4 g = *(const float4*)(p.gamma + n), be = *(const float4*)(p.beta + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = g2m + j * 16 + frow;
                f32x4 v;
                v[0] = (acc[i][j][0] - mean[j]) * rstd[j] * g.x + be.x;
                v[1] = (acc[i][j][1] - mean[j]) * rstd[j] * g.y + be.y;
                v[2] = (acc[i][j][2] - mean[j]) * rstd[j] * g.z + be.z;
                v[3] = (acc[i][j][3] - mean[j]) * rstd[j] * g.w + be.w;
                h16[i][j][0] = to_f16_sat(v[0]); h16[i][j][1] = to_f16_sat(v[1]);
                h16[i][j][2] = to_f16_sat(v[2]); h16[i][j][3] = to_f16_sat(v[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    l16[i][j][r] = TRAIN ? to_f16_sat((acc[i][j][r] - mean[j]) * rstd[j]) : (_Float16)(v[r] - (float)h16[i][j][r]);
                *(f32x4*)(smem + row * 1024 + (((n >> 2) ^ (row & 7)) << 4)) =
                    EPI == FFN_EPI_RES_SCALE_LN16 ? acc[i][j] : v;       // LS: the residual stream stays un-normalised
            }
        }
        __syncthreads();
        if (o32) {                                           // (null: the caller keeps only the f16 stream)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int row = k * 8 + wave;
                const f32x4 v = *(const f32x4*)(smem + row * 1024 + ((lane ^ (row & 7)) << 4));
                const long gr = rowmap(row);
                if (gr >= 0) *(f32x4*)(o32 + (size_t)gr * KD + lane * 4) = v;
            }
        }
        __syncthreads();
        // f16 tile -> LDS [128 rows][512 B], 16-B chunk c of row r at chunk c ^ ((r >> 1) & 7)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = N0 + i * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = g2m + j * 16 + frow;
                *(f16x4*)(smem + row * 512 + (((n >> 3) ^ ((row >> 1) & 7)) << 4) + ((n >> 2) & 1) * 8) = h16[i][j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = (k * 8 + wave) * 2 + (lane >> 5), c = lane & 31;
            const u32x4 v = *(const u32x4*)(smem + row * 512 + ((c ^ ((row >> 1) & 7)) << 4));
            const long gr = rowmap(row);
            if (gr >= 0) *(u32x4*)(o16 + (size_t)gr * KD + c * 8) = v;
        }
        if (TRAIN || p.out16lo) {                            // the f16 remainder tile (TRAIN: the normalised pre-affine rows), same way
            _Float16* __restrict__ o16l = (_Float16*)(TRAIN ? p.xhat16 : p.out16lo);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = N0 + i * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = g2m + j * 16 + frow;
                    *(f16x4*)(smem + row * 512 + (((n >> 3) ^ ((row >> 1) & 7)) << 4) + ((n >> 2) & 1) * 8) = l16[i][j];
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = (k * 8 + wave) * 2 + (lane >> 5), c = lane & 31;
                const u32x4 v = *(const u32x4*)(smem + row * 512 + ((c ^ ((row >> 1) & 7)) << 4));
                const long gr = rowmap(row);
                if (gr >= 0) *(u32x4*)(o16l + (size_t)gr * KD + c * 8) = v;
            }
        }
    }
}

// MODE 4 epilogue: out32 = acc + res (the f32 residual-gradient stream, in place), whole-row stores through the staging tile; no LayerNorm.
template <class RowMap>
__device__ __forceinline__ void ffn_epilogue_acc(const FfnParams& p, f32x4 (&acc)[4][4], char* smem, int wave, int frow, int fkg,
                                                 int g2m, int g2n, const RowMap rowmap) {
    const int N0 = g2n + fkg * 4;
    const int lane = frow + fkg * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = g2m + j * 16 + frow;
        const long gr = rowmap(row);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = N0 + i * 16;
            float4 r4 = make_float4(0, 0, 0, 0);
            if (gr >= 0) r4 = *(const float4*)(p.res + (size_t)gr * KD + n);
            const f32x4 v = f32x4{acc[i][j][0] * p.alpha + r4.x, acc[i][j][1] * p.alpha + r4.y, acc[i][j][2] * p.alpha + r4.z, acc[i][j][3] * p.alpha + r4.w};
            *(f32x4*)(smem + row * 1024 + (((n >> 2) ^ (row & 7)) << 4)) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int row = k * 8 + wave;
        const f32x4 v = *(const f32x4*)(smem + row * 1024 + ((lane ^ (row & 7)) << 4));
        const long gr = rowmap(row);
        if (gr >= 0) *(f32x4*)(p.out32 + (size_t)gr * KD + lane * 4) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// The kernel: one barrier per hidden chunk.
//   * the wave's X fragments (32 tokens x 256) live in registers for the whole block (64 VGPRs), so
//     the 64 KB X tile is only a prologue staging area and GEMM1 reads just the W1 fragments;
//   * W1s, W2s and Hs are double-buffered (2 x 32 + 2 x 32 + 2 x 16 KB = 160 KB): between two
//     barriers a wave runs GEMM1 of chunk c+1 and GEMM2 of chunk c -- two independent MFMA chains,
//     so LDS latency of one hides behind the other;
//   * weight slices arrive by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
//     The DMA destination is lane-linear (wave base + lane*16), so the swz128 image is produced by
//     permuting the per-lane SOURCE address (8 rows x 128 B per instruction; every global row is
//     still read as one full 128-B line).  The barrier's vmcnt(0) is the completion wait; b1 is
//     fetched one chunk ahead so no ordinary load result is consumed while a DMA is in flight.
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const char glb_char;

constexpr int V2_W1 = 0;                       // 2 x 32 KB
constexpr int V2_W2 = 2 * W1_BYTES;            // 2 x 32 KB (prologue: X staging, 64 KB)
constexpr int V2_HS = V2_W2 + 2 * W2_BYTES;    // 2 x 16 KB
constexpr int V2_SMEM = V2_HS + 2 * HS_BYTES;  // 163840

// PRE: the block first computes its X tile itself, X = LayerNorm1(A Wo^T + bo + res) (attention
// out-projection + residual + norm1 of the layer), so that the normalised activations between the two
// sub-layers never travel through HBM: Wo (128 KB) sits in the still-unused weight buffers, the A
// fragments come straight from global memory, and the result lands in the same accumulator layout that
// seeds GEMM2 (fp32 residual) and, as f16, in the X staging tile.
//

template <int ACT, int EPI, int MODE>
__global__ __launch_bounds__(NT)
void ffn_fused_kernel(const FfnParams p) {
    // MODE 3: training forward of the plain FFN; MODE 4: its data-gradient backward (bf16 operands: X = dY, W1 = W2^T, W2 = W1^T, the
    // "activation" is the ReLU-and-dropout mask read from the saved hidden activations, the "hidden" tile is dH and leaves for HBM too, the
    // epilogue accumulates into the f32 residual-gradient stream)
    constexpr bool PRE = MODE == 1, TRAIN = MODE == 3, BWD = MODE == 4, HSTORE = TRAIN || BWD;
    auto mfma = [](f16x8 a, f16x8 b, f32x4 c) __attribute__((always_inline)) {
        if constexpr (BWD) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nF = p.F / FC;
    const int ntiles = (p.M + BM - 1) / BM;
    // Persistent over row tiles (grid = #CUs, 1 block/CU): the output stores of a tile are never waited
    // for, so they drain under the next tile's loads instead of being an exposed phase of every block.
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * BM;
    const RowsContiguous rows_c{m0, p.M};
    auto rowmap = [&](int r) __attribute__((always_inline)) -> long {
        return rows_c(r);
    };
    auto rowclamp = [&](int r) __attribute__((always_inline)) -> long {      // always a valid row (for loads)
        return m0 + r < p.M ? (long)(m0 + r) : (long)p.M - 1;
    };
    FFN_STAMP(0);
    if (tile != (int)blockIdx.x) {
        // every wave is done reading the LN scratch at the start of LDS before the weight DMA lands there
        // (raw barrier: __syncthreads() would also wait for the previous tile's global stores)
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), vmcnt/expcnt untouched
        __builtin_amdgcn_s_barrier();
    }
    // The thread index is laundered per tile so that everything derived from it (LDS addresses, DMA
    // offsets, epilogue columns) is recomputed here instead of being hoisted out of the tile loop and
    // kept live -- or spilled -- across the register-tight chunk loop.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, fkg = lane >> 4;

    const _Float16* __restrict__ X = (const _Float16*)p.X;
    const _Float16* __restrict__ W1 = (const _Float16*)p.W1;
    const _Float16* __restrict__ W2 = (const _Float16*)p.W2;

    // DMA one 64-hidden-unit slice of W1 ([64][256] -> 4 k-tiles of [64][128 B]) / W2 ([256][64]).
    // (buffer_load ... lds rather than global_load_lds: the latter is a FLAT encoding and, while one is
    // pending, hipcc degrades every LDS wait to lgkmcnt(0), which would serialise the fragment pipeline.)
    int drow = lane >> 3, dslot = lane & 7;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)W1, 0, p.F * KD * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)W2, 0, p.F * KD * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(HSTORE ? p.hid16 : nullptr, 0, HSTORE ? (unsigned)((size_t)p.M * p.F * 2) : 0u, 0x00020000);
    // TRAIN + Swish (LS-EEND Macaron FFN): the pre-activation z is saved too (its backward needs swish'(z)); four 8-byte buffer stores per lane
    // and chunk, straight from the accumulators (rows beyond M dropped by the bounds check, so the count the chunk barrier relies on is exact)
    constexpr bool ZSTORE = TRAIN && ACT == 2;
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(ZSTORE ? p.z16 : nullptr, 0, ZSTORE ? (unsigned)((size_t)p.M * p.F * 2) : 0u, 0x00020000);
    // (the activation is then taken of the ROUNDED z, as the two-launch form does: the backward differentiates at exactly the saved point)
    auto store_z = [&](int row, int col, float& z0, float& z1, float& z2, float& z3) __attribute__((always_inline)) {
        if constexpr (ZSTORE) {
            f16x4 zz;
            zz[0] = to_f16_sat(z0); zz[1] = to_f16_sat(z1); zz[2] = to_f16_sat(z2); zz[3] = to_f16_sat(z3);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, zz), rsZ, (unsigned)(m0 + row) * (unsigned)(p.F * 2) + col * 2, 0, 0);
            z0 = (float)zz[0]; z1 = (float)zz[1]; z2 = (float)zz[2]; z3 = (float)zz[3];
        }
    };
    int vo1[4], vo2[4];                                      // per-lane byte offsets of the 4 pieces this wave moves
    auto dma_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;                  // 32 pieces of 8 rows x 128 B
            const int kt = piece >> 3, row1 = (piece & 7) * 8 + drow;
            vo1[i] = (row1 * KD + kt * 64 + (dslot ^ ((row1 >> 1) & 7)) * 8) * 2;
            const int row2 = piece * 8 + drow;
            vo2[i] = (row2 * p.F + (dslot ^ ((row2 >> 1) & 7)) * 8) * 2;
        }
    };
    dma_offsets();
    auto dma_w1 = [&](int f0, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_char*)(smem + V2_W1 + buf * W1_BYTES + (wave * 4 + i) * 1024), 16, vo1[i],
                                                     f0 * KD * 2, 0, 0);
    };
    auto dma_w2 = [&](int f0, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_char*)(smem + V2_W2 + buf * W2_BYTES + (wave * 4 + i) * 1024), 16, vo2[i],
                                                     f0 * 2, 0, 0);
    };

    const int g1m = (wave >> 1) * 32, g1f = (wave & 1) * 32;   // GEMM1 wave tile: 32 tokens x 32 hidden
    const int g2m = (wave >> 2) * 64, g2n = (wave & 3) * 64;   // GEMM2 wave tile: 64 tokens x 64 outputs
    int bofs = g1f + fkg * 4;                                  // this lane's b1 offsets inside a chunk: bofs, bofs+16

    char* Xst = smem + V2_W2;
    f32x4 acc[4][4];                                         // GEMM2 accumulators [n frag][m frag]
    if constexpr (!PRE) {
        // ---- prologue: W1 slices 0/1 by DMA; X tile -> staging (W2 region) -> fragments
        dma_w1(0, 0);
        if (nF > 1) dma_w1(FC, 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = tid + i * NT;
            const int row = q >> 5, c32 = q & 31;
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            const u32x4 v = *(const u32x4*)(X + (size_t)m * p.ldx + c32 * 8);
            *(u32x4*)(Xst + (c32 >> 3) * (BM * 128) + swz128(row, c32 & 7)) = v;
        }
    } else {
        // ---- projection + residual + LayerNorm phase: x = LN(A Wo^T + bo + res)
        //   SRC_LDS: the A operand is the f16 tile in the staging region (else fragments come from global p.A)
        //   LAST:    x feeds the FFN (f16 -> staging tile, fp32 + b2 -> GEMM2 seed, W1 slices 0/1 requested);
        //            otherwise x feeds the speaker attention (f16 -> staging tile, fp32 -> out32 stream,
        //            attention weight slices 0/1 requested)
        const int N0 = g2n + fkg * 4;
        float* red = (float*)(smem + 131072);
        const int wn = wave & 3;
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
        // SRC: 0 = A fragments straight from global (each of the 4 n-waves of a token group fetches the same rows),
        //      1 = A is the f16 tile already in the staging region (the round-2 whole-layer form; no caller left),
        //      2 = plain PRE tiles: A rows are fetched ONCE (compact, 8 x 16 B per thread) and shared through the staging
        //          tile, and Wo arrives in two halves -- k-tiles 0/1 with the inputs, k-tiles 2/3 (whose LDS image
        //          overlaps the staging tile) behind the fragment reads, under the first half of the GEMM.  The input
        //          wait of a tile is ingest-bound (~12 B/clk/CU with every CU fetching at once: s_memtime trace,
        //          tools/ffn_trace.py); this form ingests 320 KB per tile instead of 512 KB.
        auto proj_ln_phase = [&](const void* Wo, const float* bo, const float* g, const float* be, const float eps,
                                 const float* resp, auto SRC, auto LAST) __attribute__((always_inline)) {
            constexpr int src = decltype(SRC)::value;
            constexpr bool src_lds = src == 1, compact = src == 2, last = decltype(LAST)::value;
            f16x8 af[8][4];                                  // A fragments [k-step][token frag]
            const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)Wo, 0, KD * KD * 2, 0x00020000);
            auto dma_wo_piece = [&](int piece) __attribute__((always_inline)) {      // Wo -> LDS [4 k-tiles][256 n][128 B], 128 pieces of 1 KB
                const int kt = piece >> 5, row = (piece & 31) * 8 + drow;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rso, (lds_char*)(smem + piece * 1024), 16,
                                                         (row * KD + kt * 64 + (dslot ^ ((row >> 1) & 7)) * 8) * 2, 0, 0, 0);
            };
            u32x4 ar[compact ? 8 : 1];
            if constexpr (src_lds) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        af[ks][j] = *(const f16x8*)(Xst + (ks >> 1) * (BM * 128) + swz128(g2m + j * 16 + frow, (ks & 1) * 4 + fkg));
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave holds its fragments: the tile may be overwritten
            }
            if constexpr (compact) {
#pragma unroll
                for (int i = 0; i < 8; ++i) dma_wo_piece(wave * 8 + i);              // k-tiles 0, 1
                const _Float16* __restrict__ A = (const _Float16*)p.A;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = tid + i * NT;
                    ar[i] = *(const u32x4*)(A + (size_t)rowclamp(q >> 5) * p.lda + (q & 31) * 8);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) dma_wo_piece(wave * 16 + i);
            }
            if constexpr (src == 0) {
                const _Float16* __restrict__ A = (const _Float16*)p.A;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long m = rowclamp(g2m + j * 16 + frow);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) af[ks][j] = *(const f16x8*)(A + (size_t)m * p.lda + ks * 32 + fkg * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                    // accumulators start at bo + res
                const long m = rowmap(g2m + j * 16 + frow);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 b4 = *(const float4*)(bo + N0 + i * 16);
                    float4 r = make_float4(0, 0, 0, 0);
                    if (m >= 0) {
                        if (resp) r = *(const float4*)(resp + (size_t)m * KD + N0 + i * 16);
                        else if (src != 1 && p.res16) {          // (the second out-projection of the layer tail reads its own out32 stream)
                            const f16x4 h = *(const f16x4*)((const _Float16*)p.res16 + (size_t)m * KD + N0 + i * 16);
                            r = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
                        }
                    }
                    acc[i][j] = f32x4{r.x + b4.x, r.y + b4.y, r.z + b4.z, r.w + b4.w};
                }
            }
            auto gemm_ks = [&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                f16x8 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = *(const f16x8*)(smem + (ks >> 1) * (KD * 128) + swz128(g2n + i * 16 + frow, (ks & 1) * 4 + fkg));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], af[ks][j], acc[i][j], 0, 0, 0);
            };
            if constexpr (compact) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = tid + i * NT;
                    *(u32x4*)(Xst + ((q & 31) >> 3) * (BM * 128) + swz128(q >> 5, q & 7)) = ar[i];
                }
                FFN_STAMP(1);
                __syncthreads();                             // A tile visible; Wo k-tiles 0/1 and the residual have landed
                FFN_STAMP(2);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        af[ks][j] = *(const f16x8*)(Xst + (ks >> 1) * (BM * 128) + swz128(g2m + j * 16 + frow, (ks & 1) * 4 + fkg));
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // staging tile free: Wo k-tiles 2/3 land there
#pragma unroll
                for (int i = 0; i < 8; ++i) dma_wo_piece(64 + wave * 8 + i);
                FFN_STAMP(3);
                static_for<4>([&](auto KS) __attribute__((always_inline)) { gemm_ks(KS); });
                FFN_STAMP(4);
                __syncthreads();                             // (vmcnt(0): k-tiles 2/3 complete in every wave)
                FFN_STAMP(5);
                // Wo as a hi / lo f16 pair (LS-EEND decoder, round 5: the f16 rounding of the speaker-attention out-projection weight was
                // the largest single contributor to the worst logit of the 12-slot golden): the lo part's k-tiles follow the hi
                // part's through the same two LDS halves, each requested as soon as every wave is done with the half it overwrites --
                // the second product costs its 32 MFMAs per wave and half a DMA latency, the A fragments stay in registers
                const bool lo = p.Wo_lo != nullptr && Wo == p.Wo;
                const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(lo ? p.Wo_lo : Wo), 0, KD * KD * 2, 0x00020000);
                auto dma_wl_piece = [&](int piece) __attribute__((always_inline)) {
                    const int kt = piece >> 5, row = (piece & 31) * 8 + drow;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsl, (lds_char*)(smem + piece * 1024), 16,
                                                             (row * KD + kt * 64 + (dslot ^ ((row >> 1) & 7)) * 8) * 2, 0, 0, 0);
                };
                if (lo) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) dma_wl_piece(wave * 8 + i);          // lo k-tiles 0, 1 (every wave is past the hi ones: barrier above)
                }
                static_for<4>([&](auto KS) __attribute__((always_inline)) { gemm_ks(std::integral_constant<int, decltype(KS)::value + 4>{}); });
                if (lo) {
                    __syncthreads();                         // lo k-tiles 0 / 1 complete in every wave; every wave past the hi k-tiles 2 / 3
#pragma unroll
                    for (int i = 0; i < 8; ++i) dma_wl_piece(64 + wave * 8 + i);
                    static_for<4>([&](auto KS) __attribute__((always_inline)) { gemm_ks(KS); });
                    __syncthreads();
                    static_for<4>([&](auto KS) __attribute__((always_inline)) { gemm_ks(std::integral_constant<int, decltype(KS)::value + 4>{}); });
                }
            } else {
                __syncthreads();
                static_for<8>([&](auto KS) __attribute__((always_inline)) { gemm_ks(KS); });
            }
            float part[4], mean[4], rstd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
                part[j] = sum;
            }
            block_rowsum(part);
            // every wave is past its Wo reads: the next phase's first two weight slices can land in the first
            // 64 KB now (the staging barrier below, with its vmcnt(0), is the completion wait)
            dma_w1(0, 0);
            if (nF > 1) dma_w1(FC, 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) mean[j] = part[j] * (1.0f / KD);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[j]; sum += d * d; }
                part[j] = sum;
            }
            block_rowsum(part);                              // (its barriers also retire every wave's Wo reads)
#pragma unroll
            for (int j = 0; j < 4; ++j) rstd[j] = 1.0f / __builtin_sqrtf(part[j] * (1.0f / KD) + eps);
            const float ralpha = 1.0f / p.alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = N0 + i * 16;
                const float4 gg = *(const float4*)(g + n), bb = *(const float4*)(be + n), b4 = *(const float4*)(p.b2 + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = g2m + j * 16 + frow;
                    const float v0 = (acc[i][j][0] - mean[j]) * rstd[j] * gg.x + bb.x;
                    const float v1 = (acc[i][j][1] - mean[j]) * rstd[j] * gg.y + bb.y;
                    const float v2 = (acc[i][j][2] - mean[j]) * rstd[j] * gg.z + bb.z;
                    const float v3 = (acc[i][j][3] - mean[j]) * rstd[j] * gg.w + bb.w;
                    f16x4 o;
                    o[0] = to_f16_sat(v0); o[1] = to_f16_sat(v1); o[2] = to_f16_sat(v2); o[3] = to_f16_sat(v3);
                    *(f16x4*)(Xst + (n >> 6) * (BM * 128) + swz128(row, (n & 63) >> 3) + ((n >> 2) & 1) * 8) = o;
                    if constexpr (last) {
                        acc[i][j] = f32x4{v0 * ralpha + b4.x, v1 * ralpha + b4.y, v2 * ralpha + b4.z, v3 * ralpha + b4.w};   // GEMM2 seed
                    } else {
                        const long m = rowmap(row);
                        if (m >= 0) *(float4*)(p.out32 + (size_t)m * KD + n) = make_float4(v0, v1, v2, v3);
                    }
                }
            }
        };

        proj_ln_phase(p.Wo, p.bo, p.g1, p.be1, p.eps1, p.res, std::integral_constant<int, EEND_FFN_PRE_SRC>{}, std::true_type{});
    }
    FFN_STAMP(6);
    float4 bcur[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if constexpr (!BWD) {
        bcur[0] = *(const float4*)(p.b1 + bofs);
        bcur[1] = *(const float4*)(p.b1 + bofs + 16);
    }
    // BWD: the 4 x 4 saved hidden activations this lane's dH values are masked with (hm[i*2 + j]: row g1m + j*16 + frow, units fl .. fl + 3 of
    // the chunk), requested a chunk ahead like the bias
    f16x4 hm[BWD ? 4 : 1];
    auto load_mask = [&](int chunk, f16x4 (&dst)[BWD ? 4 : 1]) __attribute__((always_inline)) {
        if constexpr (BWD) {
            const _Float16* hid = (const _Float16*)p.hidmask;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int m = m0 + g1m + j * 16 + frow;
                    m = m < p.M ? m : p.M - 1;
                    dst[i * 2 + j] = *(const f16x4*)(hid + (size_t)m * p.F + chunk * FC + g1f + i * 16 + fkg * 4);
                }
        }
    };
    load_mask(0, hm);
    __syncthreads();
    f16x8 xf[4][2][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                xf[kt][ks][j] = *(const f16x8*)(Xst + kt * (BM * 128) + swz128(g1m + j * 16 + frow, ks * 4 + fkg));
    __syncthreads();

    if constexpr (!PRE) {
        // GEMM2 accumulators seeded with res / alpha + b2: the residual is fetched here, together with the
        // X tile, instead of as a second exposed HBM phase in the epilogue.
        const float ralpha = 1.0f / p.alpha;
        const int N0 = g2n + fkg * 4, M0 = m0 + g2m + frow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 b4 = BWD ? make_float4(0, 0, 0, 0) : *(const float4*)(p.b2 + N0 + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = M0 + j * 16;
                float4 r = make_float4(0, 0, 0, 0);
#ifdef EEND_FFN_ABLATE
                if (p.dbg & 2) {} else
#endif
                if (!TRAIN && !BWD && p.res && m < p.M) r = *(const float4*)(p.res + (size_t)m * KD + N0 + i * 16);      // (TRAIN: the residual joins behind the dropout, in the epilogue)
                acc[i][j] = f32x4{r.x * ralpha + b4.x, r.y * ralpha + b4.y, r.z * ralpha + b4.z, r.w * ralpha + b4.w};
            }
        }
    }

    // GEMM1 of one chunk: h[f][m] = sum_k W1c[f][k] X[m][k] ; bias + activation ; f16 -> Hs[m][f]
    auto gemm1 = [&](const char* W1s, char* Hs, const float4 (&bb)[2]) __attribute__((always_inline)) {
        f32x4 h[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) h[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 a[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i] = *(const f16x8*)(W1s + kt * (FC * 128) + swz128(g1f + i * 16 + frow, ks * 4 + fkg));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        h[i][j] = mfma(a[i], xf[kt][ks][j], h[i][j]);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int fl = g1f + i * 16 + fkg * 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v0 = h[i][j][0] + bb[i].x, v1 = h[i][j][1] + bb[i].y, v2 = h[i][j][2] + bb[i].z, v3 = h[i][j][3] + bb[i].w;
                store_z(g1m + j * 16 + frow, fl, v0, v1, v2, v3);
                if (ACT == 1) {
                    v0 = __builtin_fmaxf(v0, 0.f); v1 = __builtin_fmaxf(v1, 0.f);
                    v2 = __builtin_fmaxf(v2, 0.f); v3 = __builtin_fmaxf(v3, 0.f);
                } else if (ACT == 2) {
                    v0 = v0 / (1.0f + __expf(-v0)); v1 = v1 / (1.0f + __expf(-v1));
                    v2 = v2 / (1.0f + __expf(-v2)); v3 = v3 / (1.0f + __expf(-v3));
                }
                const int row = g1m + j * 16 + frow;
                if constexpr (BWD) {                         // (ACT == 0) dH = scale * (dY W2) where the saved activation is non-zero, bf16
                    const f16x4 mk = hm[i * 2 + j];
                    const float sc = p.drop1.scale;
                    bf16x4 ob;
                    ob[0] = (__bf16)(mk[0] != (_Float16)0 ? v0 * sc : 0.f); ob[1] = (__bf16)(mk[1] != (_Float16)0 ? v1 * sc : 0.f);
                    ob[2] = (__bf16)(mk[2] != (_Float16)0 ? v2 * sc : 0.f); ob[3] = (__bf16)(mk[3] != (_Float16)0 ? v3 * sc : 0.f);
                    *(bf16x4*)(Hs + swz128(row, fl >> 3) + ((fl >> 2) & 1) * 8) = ob;
                    continue;
                }
                if constexpr (TRAIN) {                       // chunk 0: dropout after the activation, indices (row, hidden unit)
                    const unsigned m = (unsigned)(m0 + row), n = (unsigned)fl;
                    v0 = drop_nb(p.drop1, v0, m, n); v1 = drop_nb(p.drop1, v1, m, n + 1);
                    v2 = drop_nb(p.drop1, v2, m, n + 2); v3 = drop_nb(p.drop1, v3, m, n + 3);
                }
                f16x4 o;
                o[0] = to_f16_sat(v0); o[1] = to_f16_sat(v1); o[2] = to_f16_sat(v2); o[3] = to_f16_sat(v3);
                *(f16x4*)(Hs + swz128(row, fl >> 3) + ((fl >> 2) & 1) * 8) = o;
            }
        }
    };
    // GEMM2 of one chunk: acc[n][m] += sum_f W2c[n][f] H[m][f]
    auto gemm2 = [&](const char* W2s, const char* Hs) __attribute__((always_inline)) {
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
                    acc[i][j] = mfma(a[i], b[j], acc[i][j]);
        }
    };

    // chunk 0's hidden units; W2 slice 0 arrives meanwhile
    dma_w2(0, 0);
    gemm1(smem + V2_W1, smem + V2_HS, bcur);
    if (nF > 1) {
        if constexpr (!BWD) {
            bcur[0] = *(const float4*)(p.b1 + FC + bofs);
            bcur[1] = *(const float4*)(p.b1 + FC + bofs + 16);
        }
        load_mask(1, hm);
    }
    __syncthreads();
    FFN_STAMP(7);

    // iteration c: DMA W1(c+2), W2(c+1) | GEMM1(c+1) -> Hs[(c+1)&1] | GEMM2(c) | barrier
    // The 64 MFMAs of an iteration are issued as 16 items of 4 (one GEMM1 k-step or one GEMM2 W2
    // fragment x 4 token fragments) in a fixed interleaved order; the LDS fragments of item t+2 are
    // requested right behind the MFMAs of item t, and the bias/activation/Hs-store of the new hidden
    // chunk rides on the last GEMM2 items.  sched_barrier pins that order.
    constexpr int KIND[16] = {1, 1, 2, 1, 1, 2, 1, 1, 2, 1, 1, 2, 2, 2, 2, 2};     // 1 = GEMM1 k-step, 2 = GEMM2 item
    constexpr int ARG[16]  = {0, 1, 0, 2, 3, 1, 4, 5, 2, 6, 7, 3, 4, 5, 6, 7};
    constexpr int HB1_T = 8, ACT_T0 = 12, ACT_T1 = 13;       // item after which the ks=1 H fragments / the activation parts are issued
    // (a strictly alternating G1/G2 order with the activation after the last item measured the same, same-box A/B)
    for (int c = 0; c < nF - 1; ++c) {
        const int cb = c & 1, nb = cb ^ 1;
        float4 bnext[2] = {bcur[0], bcur[1]};
        const bool more = c + 2 < nF;
        f16x4 hmn[BWD ? 4 : 1];
        if constexpr (BWD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hmn[q] = hm[q];
        }
        if (more) {
            // b1 of chunk c+2, requested a whole iteration before the barrier's vmcnt(0) has to cover it
            if constexpr (!BWD) {
                bnext[0] = *(const float4*)(p.b1 + (c + 2) * FC + bofs);
                bnext[1] = *(const float4*)(p.b1 + (c + 2) * FC + bofs + 16);
            }
            load_mask(c + 2, hmn);
        }
        // The 8 DMA pieces of this iteration (W2 slice c+1, W1 slice c+2) are issued one per early item instead
        // of as a burst in front of the first MFMA: an LDS-DMA instruction occupies the issuing wave for
        // ~60-180 cycles, which hides behind the matrix pipe once MFMAs are in flight.
        auto dma_piece = [&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
#ifdef EEND_FFN_ABLATE
            if (p.dbg & 4) return;                       // study: no in-loop weight stream (stale slices; results are garbage)
#endif
            if constexpr (i < 4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_char*)(smem + V2_W2 + nb * W2_BYTES + (wave * 4 + i) * 1024), 16, vo2[i],
                                                         (c + 1) * FC * 2, 0, 0);
            } else if constexpr (i < 8) {
                if (more)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_char*)(smem + V2_W1 + cb * W1_BYTES + (wave * 4 + i - 4) * 1024), 16, vo1[i - 4],
                                                             (c + 2) * FC * KD * 2, 0, 0);
            }
        };
        const char* W1n = smem + V2_W1 + nb * W1_BYTES;
        char* Hn = smem + V2_HS + nb * HS_BYTES;
        const char* W2c = smem + V2_W2 + cb * W2_BYTES;
        const char* Hc = smem + V2_HS + cb * HS_BYTES;

        f16x8 wa[3][2], w2a[3], hb[2][4];
        f32x4 h[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)                          // accumulators start at the bias
#pragma unroll
            for (int j = 0; j < 2; ++j) h[i][j] = f32x4{bcur[i].x, bcur[i].y, bcur[i].z, bcur[i].w};

        auto ld_g1 = [&](auto K) __attribute__((always_inline)) {
            constexpr int k = decltype(K)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wa[k % 3][i] = *(const f16x8*)(W1n + (k >> 1) * (FC * 128) + swz128(g1f + i * 16 + frow, (k & 1) * 4 + fkg));
        };
        auto ld_w2 = [&](auto S) __attribute__((always_inline)) {
            constexpr int sidx = decltype(S)::value;
            w2a[sidx % 3] = *(const f16x8*)(W2c + swz128(g2n + (sidx & 3) * 16 + frow, (sidx >> 2) * 4 + fkg));
        };
        auto ld_hb = [&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) hb[ks][j] = *(const f16x8*)(Hc + swz128(g2m + j * 16 + frow, ks * 4 + fkg));
        };
        auto ld_item = [&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            if constexpr (t < 16) {
                if constexpr (KIND[t] == 1) ld_g1(std::integral_constant<int, ARG[t]>{});
                else ld_w2(std::integral_constant<int, ARG[t]>{});
            }
        };
        auto act_part = [&](auto Q) __attribute__((always_inline)) {
            constexpr int i = decltype(Q)::value >> 1, j = decltype(Q)::value & 1;
            const int fl = g1f + i * 16 + fkg * 4;
            float v0 = h[i][j][0], v1 = h[i][j][1], v2 = h[i][j][2], v3 = h[i][j][3];
            store_z(g1m + j * 16 + frow, (c + 1) * FC + fl, v0, v1, v2, v3);
            if (ACT == 1) {
                v0 = __builtin_fmaxf(v0, 0.f); v1 = __builtin_fmaxf(v1, 0.f);
                v2 = __builtin_fmaxf(v2, 0.f); v3 = __builtin_fmaxf(v3, 0.f);
            } else if (ACT == 2) {
                v0 = v0 / (1.0f + __expf(-v0)); v1 = v1 / (1.0f + __expf(-v1));
                v2 = v2 / (1.0f + __expf(-v2)); v3 = v3 / (1.0f + __expf(-v3));
            }
            const int row = g1m + j * 16 + frow;
            if constexpr (BWD) {
                const f16x4 mk = hm[i * 2 + j];
                const float sc = p.drop1.scale;
                bf16x4 ob;
                ob[0] = (__bf16)(mk[0] != (_Float16)0 ? v0 * sc : 0.f); ob[1] = (__bf16)(mk[1] != (_Float16)0 ? v1 * sc : 0.f);
                ob[2] = (__bf16)(mk[2] != (_Float16)0 ? v2 * sc : 0.f); ob[3] = (__bf16)(mk[3] != (_Float16)0 ? v3 * sc : 0.f);
                *(bf16x4*)(Hn + swz128(row, fl >> 3) + ((fl >> 2) & 1) * 8) = ob;
                return;
            }
            if constexpr (TRAIN) {
                const unsigned m = (unsigned)(m0 + row), n = (unsigned)((c + 1) * FC + fl);
                v0 = drop_nb(p.drop1, v0, m, n); v1 = drop_nb(p.drop1, v1, m, n + 1);
                v2 = drop_nb(p.drop1, v2, m, n + 2); v3 = drop_nb(p.drop1, v3, m, n + 3);
            }
            f16x4 o;
            o[0] = to_f16_sat(v0); o[1] = to_f16_sat(v1); o[2] = to_f16_sat(v2); o[3] = to_f16_sat(v3);
            *(f16x4*)(Hn + swz128(row, fl >> 3) + ((fl >> 2) & 1) * 8) = o;
        };
        // TRAIN: the hidden chunk this iteration consumes leaves for HBM as whole 128-byte row pieces, read back from its LDS tile (stable for
        // the whole iteration).  Buffer stores: rows beyond M are dropped by the bounds check, so exactly two stores per thread are issued
        // BEHIND the iteration's weight DMA and the barrier can wait with vmcnt(2).
        auto store_hidden = [&](auto PASS) __attribute__((always_inline)) {
            if constexpr (HSTORE) {
                constexpr int pass = decltype(PASS)::value;
                const int row = pass * 64 + (tid >> 3), c8 = tid & 7;
                const u32x4 v = *(const u32x4*)(Hc + swz128(row, c8));
                __builtin_amdgcn_raw_buffer_store_b128(v, rsH, (unsigned)(m0 + row) * (unsigned)(p.F * 2) + (c * FC + c8 * 8) * 2, 0, 0);
            }
        };

        // fragments of items 0, 1 and of GEMM2 ks = 0
        ld_item(std::integral_constant<int, 0>{});
        ld_item(std::integral_constant<int, 1>{});
        ld_hb(std::integral_constant<int, 0>{});
        static_for<16>([&](auto T) __attribute__((always_inline)) {
            constexpr int t = decltype(T)::value;
            constexpr int a = ARG[t];
            if constexpr (KIND[t] == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        h[i][j] = mfma(wa[a % 3][i], xf[a >> 1][a & 1][j], h[i][j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[a & 3][j] = mfma(w2a[a % 3], hb[a >> 2][j], acc[a & 3][j]);
            }
            ld_item(std::integral_constant<int, t + 2>{});
            if constexpr (t < 8) dma_piece(std::integral_constant<int, t>{});
            if constexpr (t == 9) store_hidden(std::integral_constant<int, 0>{});
            if constexpr (t == 10) store_hidden(std::integral_constant<int, 1>{});
            if constexpr (t == HB1_T) ld_hb(std::integral_constant<int, 1>{});
            if constexpr (t == ACT_T0) { act_part(std::integral_constant<int, 0>{}); act_part(std::integral_constant<int, 1>{}); }
            if constexpr (t == ACT_T1) { act_part(std::integral_constant<int, 2>{}); act_part(std::integral_constant<int, 3>{}); }
            __builtin_amdgcn_sched_barrier(0);
        });
        bcur[0] = bnext[0]; bcur[1] = bnext[1];
        if constexpr (BWD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hm[q] = hmn[q];
        }
        if constexpr (ZSTORE) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... and the four z stores of the activation parts
        else if constexpr (HSTORE) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // weight DMA landed; the two hidden-row stores stay in flight
        else __syncthreads();
    }
    {
        const int cb = (nF - 1) & 1;
        if constexpr (HSTORE) {                              // the last hidden chunk
            const char* Hl = smem + V2_HS + cb * HS_BYTES;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int row = pass * 64 + (tid >> 3), c8 = tid & 7;
                const u32x4 v = *(const u32x4*)(Hl + swz128(row, c8));
                __builtin_amdgcn_raw_buffer_store_b128(v, rsH, (unsigned)(m0 + row) * (unsigned)(p.F * 2) + ((nF - 1) * FC + c8 * 8) * 2, 0, 0);
            }
        }
        gemm2(smem + V2_W2 + cb * W2_BYTES, smem + V2_HS + cb * HS_BYTES);
    }
    __syncthreads();

    FFN_STAMP(8);
    if constexpr (BWD) ffn_epilogue_acc(p, acc, smem, wave, frow, fkg, g2m, g2n, rows_c);
    else ffn_epilogue<EPI, TRAIN>(p, acc, smem, wave, frow, fkg, g2m, g2n, rows_c);
    FFN_STAMP(9);
    }
}

template <int ACT, int EPI, int MODE>
int launch(const FfnParams& p, hipStream_t stream) {
    static EendOncePerDevice attr_once;
    auto kern = ffn_fused_kernel<ACT, EPI, MODE>;
    const int smem_bytes = V2_SMEM;
    if (!eend_set_dynamic_lds(attr_once, (const void*)kern, smem_bytes)) return EEND_ELAUNCH;
    const int ncu = eend_cu_count();
    const int ntiles = (p.M + BM - 1) / BM;
    hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(NT), smem_bytes, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

}  // namespace

#ifdef EEND_FFN_TRACE
extern "C" int eend_debug_ffn_trace(void* dst, void* stream) {
    return hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(g_ffn_trace), sizeof(g_ffn_trace), 0, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}
#endif

int eend_launch_ffn_fused(const FfnParams& p, int act, int epi, hipStream_t stream) {
    if (p.M <= 0 || p.F <= 0 || (p.F % FC) != 0 || (p.ldx & 7) || !p.W1 || !p.W2) return EEND_EINVAL;
    if (p.hidmask) {                                         // data-gradient backward of the block (bf16 operands)
        if (!p.X || p.A || !p.hid16 || !p.res || !p.out32 || act != 0 || ((size_t)p.M + 128) * p.F * 2 >= (1ull << 32) ||
            (((size_t)p.hid16 | (size_t)p.hidmask | (size_t)p.res | (size_t)p.out32) & 15))
            return EEND_EINVAL;
        return launch<0, FFN_EPI_RES_LN, 4>(p, stream);
    }
    if (!p.b1 || !p.b2 || !p.gamma || !p.beta || !p.out16) return EEND_EINVAL;
    if (!p.out32 && (!p.A || epi != FFN_EPI_RES_LN)) return EEND_EINVAL;   // f16-only output: the attnout + FFN form only
    if (p.A) {                                               // fused attention out-projection + norm1 producer
        if (!p.Wo || !p.bo || !p.g1 || !p.be1 || (p.lda & 7) || epi != FFN_EPI_RES_LN || act != 1) return EEND_EINVAL;
        return launch<1, FFN_EPI_RES_LN, 1>(p, stream);
    }
    if (!p.X) return EEND_EINVAL;
    if (p.hid16) {                                           // training forward (post-norm ReLU block)
        if (!p.res || !p.out32 || !p.xhat16 || !p.rstat || p.out16lo || ((size_t)p.M + 128) * p.F * 2 >= (1ull << 32) ||
            (((size_t)p.hid16 | (size_t)p.xhat16 | (size_t)p.z16) & 15))
            return EEND_EINVAL;
        if (act == 1 && epi == FFN_EPI_RES_LN && !p.z16) return launch<1, FFN_EPI_RES_LN, 3>(p, stream);
        if (act == 2 && p.z16) {                             // LS-EEND Macaron FFN (Swish; pre-activation saved)
            if (epi == FFN_EPI_RES_LN) return launch<2, FFN_EPI_RES_LN, 3>(p, stream);
            if (epi == FFN_EPI_RES_SCALE_LN16) return launch<2, FFN_EPI_RES_SCALE_LN16, 3>(p, stream);
        }
        return EEND_EINVAL;
    }
    if (epi == FFN_EPI_RES_LN) {
        if (act == 1) return launch<1, FFN_EPI_RES_LN, 0>(p, stream);
        if (act == 2) return launch<2, FFN_EPI_RES_LN, 0>(p, stream);
    } else if (epi == FFN_EPI_RES_SCALE_LN16) {
        if (act == 1) return launch<1, FFN_EPI_RES_SCALE_LN16, 0>(p, stream);
        if (act == 2) return launch<2, FFN_EPI_RES_SCALE_LN16, 0>(p, stream);
    }
    return EEND_EINVAL;
}
