// X-resident projection kernel for the K = 256 input projections whose N is a multiple of 256:
//   packed MHA in-proj (N = 768 -> Q, K head rows + V^T), speaker-MHA in-proj (N = 768, row-major),
//   retention q/k/v/g projections (N = 1024 -> Q, K, K^T, V^T heads + G row-major).
//
// Block = 128 tokens, 8 waves (512 threads), 1 block/CU.  The X tile (128 x 256 f16, 64 KB) is
// loaded ONCE and stays in LDS while the weight matrix streams through in 64-output-feature chunks
// (32 KB, register-prefetched one chunk ahead).  Per chunk: 128 x 64 outputs on
// v_mfma_f32_16x16x32_f16 (waves 4(m) x 2(n), 32 x 32 each), bias, convert, stage in LDS, then
// cooperative 16-B/lane stores:
//   row-major      : 128-byte rows of out[m][ld]
//   head rows      : [seq][H][Tp][64] -- 64 features = one head, so a chunk is one 16 KB contiguous piece
//   head transposed: [seq][H][64][Tp] -- the MFMA operands are swapped for these chunks so a lane owns 4
//                    consecutive tokens of one feature; rows of 256 bytes
// Versus the generic 128 x 128 GEMM tiles: X is read once instead of N/128 times, two barriers per
// 64 columns with the whole K extent resident, and every global store is a full cache line.
#include "common.h"
#include "kernels.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;
constexpr int KD = 256;
constexpr int FC = 64;
constexpr int NT = 512;
constexpr int XS_BYTES = 4 * BM * 128;            // 64 KB
constexpr int WS_BYTES = 4 * FC * 128;            // 32 KB
constexpr int SROW_MN = FC + 8;                   // staged [m][n] row stride (elements)
constexpr int SROW_NM = BM + 8;                   // staged [n][m] row stride (elements)
constexpr int ST_BYTES = BM * SROW_MN * 2;        // 18432 (>= FC * SROW_NM * 2 = 17408)
constexpr int SMEM_BYTES = XS_BYTES + 2 * WS_BYTES + ST_BYTES;     // 146 KB: weight chunk double-buffered

DEV unsigned pack2(float a, float b, bool bf) {
    if (bf) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
        return __builtin_bit_cast(unsigned, v);
    }
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v; v[0] = to_f16_sat(a); v[1] = to_f16_sat(b);
    return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(NT)
void proj_xres_kernel(const ProjParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;
    char* Ws0 = smem + XS_BYTES;
    unsigned short* St = (unsigned short*)(Ws0 + 2 * WS_BYTES);   // [m][n] staging ...
    unsigned short* St2 = St;                                     // ... or [n][m] (transposed outputs): one per pass

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fkg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int nchunks = p.N / FC;
    const _Float16* __restrict__ X = (const _Float16*)p.X;
    const _Float16* __restrict__ W = (const _Float16*)p.W;

#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + i * NT;
        const int row = q >> 5, c32 = q & 31;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        const u32x4 v = *(const u32x4*)(X + (size_t)m * p.ldx + c32 * 8);
        *(u32x4*)(Xs + (c32 >> 3) * (BM * 128) + swz128(row, c32 & 7)) = v;
    }
    u32x4 wr[4];
    auto wload = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * NT;                   // 64 rows x 32 k-chunks
            wr[i] = *(const u32x4*)(W + (size_t)(n0 + (q >> 5)) * KD + (q & 31) * 8);
        }
    };
    auto wstore = [&](int buf) __attribute__((always_inline)) {
        char* Ws = Ws0 + buf * WS_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * NT;
            const int row = q >> 5, c32 = q & 31;
            *(u32x4*)(Ws + (c32 >> 3) * (FC * 128) + swz128(row, c32 & 7)) = wr[i];
        }
    };

    const int gm = (wave >> 1) * 32, gn = (wave & 1) * 32;     // wave tile: 32 tokens x 32 features

    // The wave's X fragments (32 tokens x 256 features) are the same for every weight chunk: read
    // them from LDS once and keep them in registers (16 fragments, 64 VGPRs) -- halves the LDS reads
    // of a chunk.  A- and B-operand fragments of v_mfma_f32_16x16x32_f16 have the same lane layout,
    // so the same registers serve both operand orders.
    wload(0);
    __syncthreads();
    f16x8 xf[4][2][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                xf[kt][ks][j] = *(const f16x8*)(Xs + kt * (BM * 128) + swz128(gm + j * 16 + frow, ks * 4 + fkg));

    // One pass = GEMM of chunk c in one orientation + staging.  `tr` (block-uniform): transposed pass.
    auto chunk = [&](int c, bool tr) __attribute__((always_inline)) {
        const int n0 = c * FC;
        const int grp = n0 >> 8;                               // 256-feature group
        const bool bf = p.is_bf16[grp] != 0;
        const char* Ws = Ws0 + (c & 1) * WS_BYTES;
        f32x4 h[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) h[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!tr) {
            // A = W rows (features), B = X rows (tokens): lane owns 4 consecutive features of a token
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f16x8 a[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i] = *(const f16x8*)(Ws + kt * (FC * 128) + swz128(gn + i * 16 + frow, ks * 4 + fkg));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) h[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], xf[kt][ks][j], h[i][j], 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nl = gn + i * 16 + fkg * 4;
                const float4 bb = *(const float4*)(p.bias + n0 + nl);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ml = gm + j * 16 + frow;
                    const float v0 = h[i][j][0] + bb.x, v1 = h[i][j][1] + bb.y, v2 = h[i][j][2] + bb.z, v3 = h[i][j][3] + bb.w;
                    uint2 o; o.x = pack2(v0, v1, bf); o.y = pack2(v2, v3, bf);
                    *(uint2*)(St + ml * SROW_MN + nl) = o;
                }
            }
        } else {
            // transposed heads: A = X rows (tokens), B = W rows (features): lane owns 4 consecutive tokens
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f16x8 b[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) b[i] = *(const f16x8*)(Ws + kt * (FC * 128) + swz128(gn + i * 16 + frow, ks * 4 + fkg));
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i) h[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[kt][ks][j], b[i], h[i][j], 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nl = gn + i * 16 + frow;
                const float bb = p.bias[n0 + nl];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ml = gm + j * 16 + fkg * 4;
                    uint2 o; o.x = pack2(h[i][j][0] + bb, h[i][j][1] + bb, bf); o.y = pack2(h[i][j][2] + bb, h[i][j][3] + bb, bf);
                    *(uint2*)(St2 + nl * SROW_NM + ml) = o;
                }
            }
        }
    };

    auto store_out = [&](int c, bool tr) __attribute__((always_inline)) {
        const int n0 = c * FC;
        const int grp = n0 >> 8;
        const int kind = p.kind[grp];
        const int nn = n0 & 255;                                   // feature inside the 256-group
        const int head = nn >> 6;                                  // FC == dh == 64: chunk == one head
        if (!tr) {
            // [m][n] staging: 128 rows x 8 chunks of 16 B
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = tid + it * NT;
                const int row = idx >> 3, ch = idx & 7;
                const int m = m0 + row;
                if (m >= p.M) continue;
                const uint4 v = *(const uint4*)(St + row * SROW_MN + ch * 8);
                unsigned short* dst;
                if (kind == PROJ_ROWMAJOR) {
                    dst = (unsigned short*)p.out[grp] + (size_t)m * p.ld[grp] + nn + ch * 8;
                } else {
                    const int seq = m / p.Tp, t = m - seq * p.Tp;
                    dst = (unsigned short*)p.out[grp] + (((size_t)seq * p.H + head) * p.Tp + t) * 64 + ch * 8;
                }
                *(uint4*)dst = v;
            }
        }
        if (tr) {
            // [n][m] staging: 64 rows (d) x 16 chunks of 8 tokens
            unsigned short* base = (unsigned short*)(kind == PROJ_HEADS_T ? p.out[grp] : p.out2[grp]);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = tid + it * NT;
                const int d = idx >> 4, ch = idx & 15;
                const int m = m0 + ch * 8;
                if (m >= p.M) continue;
                const uint4 v = *(const uint4*)(St2 + d * SROW_NM + ch * 8);
                const int seq = m / p.Tp, t = m - seq * p.Tp;
                *(uint4*)(base + (((size_t)seq * p.H + head) * 64 + d) * p.Tp + t) = v;
            }
        }
    };

    // Weight chunks are double-buffered in LDS and prefetched into registers one iteration ahead of
    // their LDS store, i.e. two chunks ahead of their use: the loads have a whole chunk to land.
    wstore(0);                               // chunk 0 was requested before the X fragments were read
    if (nchunks > 1) wload(FC);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) wstore((c + 1) & 1);        // chunk c+1 (loaded during chunk c-1) -> other buffer
        if (c + 2 < nchunks) wload((c + 2) * FC);
        const int kind = p.kind[(c * FC) >> 8];
        if (kind != PROJ_HEADS_T) {
            chunk(c, false);
            __syncthreads();
            store_out(c, false);
            __syncthreads();
        }
        if (kind == PROJ_HEADS_T || kind == PROJ_HEADS_BOTH) {   // BOTH (retention K): second pass, swapped operands
            chunk(c, true);
            __syncthreads();
            store_out(c, true);
            __syncthreads();
        }
    }
}

}  // namespace

int eend_launch_proj_xres(const ProjParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || (p.N % 256) != 0 || p.N > 1024 || (p.ldx & 7) || !p.X || !p.W || !p.bias || p.H != 4)
        return EEND_EINVAL;
    for (int g = 0; g < p.N / 256; ++g) {
        if (!p.out[g]) return EEND_EINVAL;
        if (p.kind[g] == PROJ_HEADS_BOTH && !p.out2[g]) return EEND_EINVAL;
        if (p.kind[g] != PROJ_ROWMAJOR && (p.Tp <= 0 || (p.Tp % 64) != 0 || (p.M % p.Tp) != 0)) return EEND_EINVAL;
        if (p.kind[g] == PROJ_ROWMAJOR && (p.ld[g] & 7)) return EEND_EINVAL;
    }
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)proj_xres_kernel, SMEM_BYTES)) return EEND_ELAUNCH;
    hipLaunchKernelGGL(proj_xres_kernel, dim3((p.M + BM - 1) / BM), dim3(NT), SMEM_BYTES, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
