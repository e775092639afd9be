// Weight gradients of the training step:  dW[n][k] = sum_m dY[m][n] * X[m][k]   ("TN" GEMM: both operands are
// stored token-major, the contraction runs over the token axis m).
//
// MFMA fragments want the contraction index contiguous per lane, the operands have it strided.  The transpose
// happens once, in registers, on the way from HBM to LDS: a thread loads an 8-token x 8-feature block (8 coalesced
// 16-byte loads), transposes it with 32 byte-permutes (transpose8x8_b16) and writes 8 feature rows of 8 tokens.
// After that the LDS tiles are [feature][64 tokens] -- exactly the K-contiguous image of gemm.hip -- and the inner
// loop is the same conflict-free ds_read_b128 + v_mfma_f32_16x16x32_bf16 stream.
//
// dY is bf16 (gradients need the exponent range).  X is a saved forward activation: f16 (converted to bf16 in the
// staging registers) or bf16.  Accumulation is f32.  The token range is split over `nsplit` workgroups per output
// tile; every workgroup writes its partial tile to a workspace and wgrad_reduce_kernel sums the partials in a fixed
// order (deterministic: no atomics), optionally scales, and writes the f32 gradient with the destination's own row
// stride (the flat gradient buffer of the optimiser).
//
// Conv1d weight gradient (FS model :30,:40): the same kernel with the X rows of k-tile (tap, c_in block) read at
// frame t + tap - pad of the same sequence, zero outside [0, ilen) -- the implicit-GEMM loader of gemm.hip mirrored.
#include "train_common.h"
#include "kernels.h"

namespace {

#ifndef EEND_WGRAD_PF
#define EEND_WGRAD_PF 2         // register sets of the global -> LDS staging pipeline (prefetch distance in 64-token steps)
#endif
#ifndef EEND_WGRAD_XCD
#define EEND_WGRAD_XCD 1        // 0: the plain block order (kept for the same-box A/B, tools/ab_variants.sh)
#endif
constexpr int TN_BM = 64;

// BT = output tile edge: 128 (4 waves, 64 KB LDS, two workgroups per CU) or 256 (8 waves, 128 KB LDS, one per CU).
// The kernel is bound by operand delivery, not by HBM or the MFMA pipe (PMC on [393216, 256, 2048]: FETCH_SIZE = the
// algorithmic 1.8 GB, MFMA pipe 19 % busy, L2 serving 6.5 GB = every tile's own copy of its dY / X rows at ~7.5 TB/s,
// ~30 GB/s per CU, for every shape tried), so the lever is bytes per flop: a 256 x 256 tile needs half of them.
// Measured, same box: 15-18 % faster on the shapes with more than one 256-tile (e.g. [393216, 256, 2048] 866 -> 737 us,
// 559 TFLOP/s); deeper register prefetch (2 / 3 sets, branch-free loads so that the waits are partial) changed nothing.
template <bool B_F16, bool CONV, int BT, bool BIAS>
__global__ __launch_bounds__(BT * 2)
void wgrad_tn_kernel(const WgradParams p) {
    constexpr int TN_BN = BT, TN_BK = BT;
    constexpr int HT = BT;                                            // threads staging one operand (BT / 8 feature groups x 8)
    constexpr int JN = BT / 32;                                       // 16-wide n fragments per wave (wave tile 64 k x BT / 2 n)
    constexpr int PF = BT == 128 ? EEND_WGRAD_PF : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 x (A^T [BT][128 B] + B^T [BT][128 B])
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wn = wave & 1;                          // wave tile: 64 k-rows x BT / 2 n-cols
    const int ntk = p.K / TN_BK, ntn = p.N / TN_BN;
    // block -> (output tile, token split).  XCD-aware when the split count allows it: workgroup b runs on XCD b % 8
    // (hardware round-robin), and ALL output tiles of a token split go to one XCD, so that the split's dY / X rows are
    // fetched from HBM once and the ntk / ntn-fold re-reads by the other tiles are hits in that XCD's L2 (same-box A/B:
    // 5-8 % on every shape; the plain b % ntiles order put the four tiles of a [M, 256, 256] split on four XCDs).
    int tile, split;
    if (EEND_WGRAD_XCD && (p.nsplit & 7) == 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        split = (i / (ntk * ntn)) * 8 + xcd;
        tile = i % (ntk * ntn);
    } else {
        tile = blockIdx.x % (ntk * ntn);
        split = blockIdx.x / (ntk * ntn);
    }
    const int n0 = (tile / ntk) * TN_BN, k0 = (tile % ntk) * TN_BK;
    const long m_begin = (long)split * p.m_per_split;
    long m_end = m_begin + p.m_per_split;
    if (m_end > p.M) m_end = p.M;
    const int nsteps = m_end > m_begin ? (int)((m_end - m_begin + TN_BM - 1) / TN_BM) : 0;

    // staging role: the first half of the threads transposes the dY tile, the second the X tile; unit = (8 tokens mg, 8 features fc)
    const bool isB = tid >= HT;
    const int u = tid & (HT - 1), fc = u & (BT / 8 - 1), mg = u / (BT / 8);
    const unsigned short* src = isB ? (const unsigned short*)p.B + (CONV ? 0 : k0) + fc * 8
                                    : (const unsigned short*)p.A + n0 + fc * 8;
    const int ld = isB ? p.ldb : p.lda;
    int conv_shift = 0, conv_c0 = 0;
    if constexpr (CONV) {
        const int tap = k0 / p.conv_cin;
        conv_c0 = k0 - tap * p.conv_cin;
        conv_shift = tap - p.conv_pad;
    }

    // Bias gradient on the side (p.bias_partial): the column sums of dY are accumulated from the staging registers of the dY
    // stagers -- by the workgroups of k-tile 0 only, every dY element is seen there exactly once -- instead of by a separate
    // pass over dY (eend_colsum_f32: 3-5 % of a training step, HBM-bound).
    const bool do_bias = BIAS && (tile % ntk) == 0;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // PF register sets: the rows of step s are requested PF steps ahead (an MFMA phase is ~0.2 us, a loaded-HBM round trip
    // ten times that: with one set -- request at s-1, transpose at the end of s-1 -- every step waited for its own loads),
    // transposed into the LDS buffer one step ahead, consumed at step s.  The loads are branch-free (clamped address +
    // select) so that the compiler can wait with vmcnt(8 * (PF - 1)) instead of vmcnt(0).
    u32x4 reg[PF][8];
    auto gload = [&](int step, u32x4 (&rg)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const long m = m_begin + (long)step * TN_BM + mg * 8 + r;
            if constexpr (CONV) {
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (m < m_end) {
                    if (isB) {
                        const int seq = (int)(m / p.Tp), t = (int)(m - (long)seq * p.Tp);
                        const int ts = t + conv_shift;
                        if (ts >= 0 && ts < p.ilens[seq])
                            v = *(const u32x4*)(src + ((long)seq * p.Tp + ts) * ld + conv_c0);
                    } else {
                        v = *(const u32x4*)(src + m * ld);
                    }
                }
                rg[r] = v;
            } else {
                const bool ok = m < m_end;
                const u32x4 v = *(const u32x4*)(src + (ok ? m : m_begin) * ld);
                rg[r] = ok ? v : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto lstore = [&](int buf, const u32x4 (&rg)[8]) __attribute__((always_inline)) {
        if (do_bias && !isB) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned v = rg[r][j];
                    cs[2 * j] += bf16_lo(v);
                    cs[2 * j + 1] += bf16_hi(v);
                }
        }
        u32x4 in[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (B_F16 && isB) ? f16x8_to_bf16x8(rg[r]) : rg[r];
        char* base = smem + buf * (2 * TN_BN * 128) + (isB ? TN_BN * 128 : 0);
        if constexpr (BT == 256) {
            // 128 accumulator registers: one transposed row at a time (4 live registers instead of 32), addresses recomputed
            int fc_ = fc, mg_ = mg;
            asm volatile("" : "+v"(fc_), "+v"(mg_));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned a = in[2 * j][e >> 1], b = in[2 * j + 1][e >> 1];
                    o[j] = (e & 1) ? ((a >> 16) | (b & 0xFFFF0000u)) : ((a & 0xFFFFu) | (b << 16));
                }
                *(u32x4*)(base + swzT(fc_ * 8 + e, mg_)) = o;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            u32x4 out[8];
            transpose8x8_b16(in, out);
#pragma unroll
            for (int e = 0; e < 8; ++e) *(u32x4*)(base + swzT(fc * 8 + e, mg)) = out[e];
        }
    };

    f32x4 acc[4][JN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow_c = lane & 15, fkg_c = lane >> 4;

    auto mma_step = [&](int buf) __attribute__((always_inline)) {
        int frow = frow_c, fkg = fkg_c;
        if (BT == 256) asm volatile("" : "+v"(frow), "+v"(fkg));     // 128 accumulator registers: recompute the 24 fragment
                                                                      // addresses per step instead of keeping them (and spilling)
        const char* at = smem + buf * (2 * TN_BN * 128);              // dY^T: [n][64 m]
        const char* bt = at + TN_BN * 128;                           // X^T : [k][64 m]
        if constexpr (BT == 256) {
            // 128 accumulator registers: fragments are fetched in small groups (4 n-fragments, then one k-fragment per 4 MFMAs)
            // and the groups are fenced, so that at most 20 fragment registers are live
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8 lf[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) lf[j] = *(const bf16x8*)(at + swzT(wn * 128 + (jh * 4 + j) * 16 + frow, ks * 4 + fkg));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bf16x8 rf = *(const bf16x8*)(bt + swzT(wk * 64 + i * 16 + frow, ks * 4 + fkg));
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][jh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf, lf[j], acc[i][jh * 4 + j], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 rf[4], lf[JN];
#pragma unroll
            for (int i = 0; i < 4; ++i) rf[i] = *(const bf16x8*)(bt + swzT(wk * 64 + i * 16 + frow, ks * 4 + fkg));
#pragma unroll
            for (int j = 0; j < JN; ++j) lf[j] = *(const bf16x8*)(at + swzT(wn * (BT / 2) + j * 16 + frow, ks * 4 + fkg));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf[i], lf[j], acc[i][j], 0, 0, 0);
        }
        }
    };
    // steps beyond nsteps load nothing (all rows >= m_end select zero) and add zero: the trip count is rounded up to the
    // unroll factor instead of breaking out of the unrolled body
#pragma unroll
    for (int d = 0; d < PF; ++d) gload(d, reg[d]);
    lstore(0, reg[0]);
    __syncthreads();
    constexpr int UN = (PF & 1) ? 2 * PF : PF;                        // register set AND LDS buffer static inside the body
    for (int st0 = 0; st0 < nsteps; st0 += UN) {
#pragma unroll
        for (int dd = 0; dd < UN; ++dd) {
            const int st = st0 + dd;
            gload(st + PF, reg[dd % PF]);                             // reg[dd % PF] (step st) went to LDS during step st - 1
            mma_step(dd & 1);
            lstore((dd & 1) ^ 1, reg[(dd + 1) % PF]);
            __syncthreads();
        }
    }

    if (do_bias) {                                                    // the loop ended on a barrier: the LDS tiles are free
        float* red = (float*)smem;                                    // [8 token groups][BT]
        if (!isB) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[mg * BT + fc * 8 + e] = cs[e];
        }
        __syncthreads();
        if (tid < BT) {
            float t = 0.f;
#pragma unroll
            for (int gsum = 0; gsum < 8; ++gsum) t += red[gsum * BT + tid];
            p.bias_partial[(size_t)split * p.N + n0 + tid] = t;
        }
    }
    // acc[i][j][r]: k = k0 + wk*64 + i*16 + fkg*4 + r (4 consecutive k per lane), n = n0 + wn*(BT/2) + j*16 + frow
    const int frow = frow_c, fkg = fkg_c;
    float* __restrict__ out = p.partial + (size_t)split * p.N * p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int k = k0 + wk * 64 + i * 16 + fkg * 4, n = n0 + wn * (BT / 2) + j * 16 + frow;
            *(f32x4*)(out + (size_t)n * p.K + k) = acc[i][j];
        }
}

// out[n][k] (row stride ld_out, k < K_out) = scale * sum_s partial[s][n][k]   (+ out if accumulate)
__global__ __launch_bounds__(256)
void wgrad_reduce_kernel(const float* __restrict__ partial, long split_stride, int nsplit, int N, int K, int K_out,
                         float* __restrict__ out, int ld_out, float scale, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * K_out) return;
    const int n = (int)(idx / K_out), k = (int)(idx - (long)n * K_out);
    const float* src = partial + (size_t)n * K + k;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        s0 += src[(size_t)(s + 0) * split_stride];
        s1 += src[(size_t)(s + 1) * split_stride];
        s2 += src[(size_t)(s + 2) * split_stride];
        s3 += src[(size_t)(s + 3) * split_stride];
    }
    for (; s < nsplit; ++s) s0 += src[(size_t)s * split_stride];
    float v = ((s0 + s1) + (s2 + s3)) * scale;
    float* o = out + (size_t)n * ld_out + k;
    if (accumulate) v += *o;
    *o = v;
}

// Few outputs, many partials (LayerNorm / bias gradients: <= a few thousand outputs, up to 1024 partials): 8 lane
// groups walk the partials in parallel (coalesced over 32 consecutive outputs) and are combined in a fixed order.
__global__ __launch_bounds__(256)
void wgrad_reduce_small_kernel(const float* __restrict__ partial, long split_stride, int nsplit, int N, int K, int K_out,
                               float* __restrict__ out, int ld_out, float scale, int accumulate) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long idx = (long)blockIdx.x * 32 + o;
    const bool live = idx < (long)N * K_out;
    float s = 0.f;
    if (live) {
        const int n = (int)(idx / K_out), k = (int)(idx - (long)n * K_out);
        const float* src = partial + (size_t)n * K + k;
        // eight loads in flight per thread (the walk is latency-bound: 20 us per call with two); fixed combination order
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sp = g;
        for (; sp + 56 < nsplit; sp += 64) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += src[(size_t)(sp + 8 * j) * split_stride];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (sp + 8 * j < nsplit) a[j] += src[(size_t)(sp + 8 * j) * split_stride];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[g][o] = s;
    __syncthreads();
    if (g == 0 && live) {
        const int n = (int)(idx / K_out), k = (int)(idx - (long)n * K_out);
        float v = (((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]))) * scale;
        float* dst = out + (size_t)n * ld_out + k;
        if (accumulate) v += *dst;
        *dst = v;
    }
}

// Column sums of a 2-byte-float matrix [M][N] (bias gradients): partial[s][n] = sum over the split's rows.
// A block covers 256 columns (32 threads x 16-byte loads) in 8 row phases; N % 8 == 0.
template <bool IS_BF16>
__global__ __launch_bounds__(256)
void colsum_partial_kernel(const unsigned short* __restrict__ Y, int ld /* in 2-byte elements */, long M, int N, long rows_per_split,
                           float* __restrict__ partial) {
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + cg * 8;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > M) r1 = M;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        for (long r = r0 + ph; r < r1; r += 8) {
            const u32x4 v = *(const u32x4*)(Y + r * ld + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned x = v[j];
                if (IS_BF16) { a[2 * j] += bf16_lo(x); a[2 * j + 1] += bf16_hi(x); }
                else { a[2 * j] += f16_lo(x); a[2 * j + 1] += f16_hi(x); }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ph][cg * 8 + e] = a[e];
    __syncthreads();
    const int c = threadIdx.x;
    if (blockIdx.x * 256 + c < N)
        partial[(size_t)blockIdx.y * N + blockIdx.x * 256 + c] =
            ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
}

// cnn.weight gradient back to the parameter's own layout: g[co][ci][tap] = tmp[co][tap*cin + ci]
__global__ __launch_bounds__(256)
void conv_wgrad_unpermute_kernel(const float* __restrict__ tmp, float* __restrict__ g, int cout, int cin, int ktaps) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)cout * cin * ktaps) return;
    const int tap = (int)(idx % ktaps);
    const long r = idx / ktaps;
    const int ci = (int)(r % cin), co = (int)(r / cin);
    g[idx] = tmp[((size_t)co * ktaps + tap) * cin + ci];
}

}  // namespace

int eend_launch_wgrad(const WgradParams& p, hipStream_t stream) {
    if (!p.A || !p.B || !p.partial || p.M <= 0 || p.N <= 0 || p.K <= 0) return EEND_EINVAL;
    const int bt = p.tile == 256 ? 256 : 128;
    if ((p.N % bt) || (p.K % bt) || (p.lda & 7) || (p.ldb & 7) || p.nsplit <= 0 || p.m_per_split <= 0 || (p.m_per_split % TN_BM))
        return EEND_EINVAL;
    if (p.conv && (!p.ilens || p.conv_cin <= 0 || (p.conv_cin % bt) || p.Tp <= 0)) return EEND_EINVAL;
    const int smem = 2 * 2 * bt * 128;
    const dim3 grid((unsigned)((p.N / bt) * (p.K / bt) * p.nsplit));
#define WG_LAUNCH(F16, CV, BT, BS)                                                                                      \
    do {                                                                                                                \
        static EendOncePerDevice attr_once;                                                                             \
        if (!eend_set_dynamic_lds(attr_once, (const void*)wgrad_tn_kernel<F16, CV, BT, BS>, smem)) return EEND_ELAUNCH;  \
        hipLaunchKernelGGL((wgrad_tn_kernel<F16, CV, BT, BS>), grid, dim3(BT * 2), smem, stream, p);                    \
    } while (0)
#define WG_PICK(BT)                                                                                                     \
    do {                                                                                                                \
        if (p.conv) { if (p.b_is_f16) WG_LAUNCH(true, true, BT, false); else WG_LAUNCH(false, true, BT, false); }       \
        else if (p.bias_partial) { if (p.b_is_f16) WG_LAUNCH(true, false, BT, (BT == 128)); else WG_LAUNCH(false, false, BT, (BT == 128)); } \
        else { if (p.b_is_f16) WG_LAUNCH(true, false, BT, false); else WG_LAUNCH(false, false, BT, false); }            \
    } while (0)
    if (p.bias_partial && (p.conv || bt != 128)) return EEND_EINVAL;      // the column sums ride on the 128-tile kernel only
    if (bt == 256) WG_PICK(256); else WG_PICK(128);
#undef WG_PICK
#undef WG_LAUNCH
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_wgrad_reduce(const float* partial, long split_stride, int nsplit, int N, int K, int K_out, float* out,
                             int ld_out, float scale, int accumulate, hipStream_t stream) {
    if (!partial || !out || nsplit <= 0 || N <= 0 || K <= 0 || K_out <= 0 || K_out > K || ld_out < K_out) return EEND_EINVAL;
    const long n = (long)N * K_out;
    if (n <= 16384 && nsplit >= 16)
        hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, stream, partial, split_stride,
                           nsplit, N, K, K_out, out, ld_out, scale, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, split_stride,
                           nsplit, N, K, K_out, out, ld_out, scale, accumulate);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_colsum_partial(const void* Y, int ld, long M, int N, int is_bf16, int nsplit, float* partial,
                               hipStream_t stream) {
    if (!Y || !partial || M <= 0 || N <= 0 || (N & 7) || (ld & 7) || nsplit <= 0 || nsplit > 65535) return EEND_EINVAL;
    const long rps = (M + nsplit - 1) / nsplit;
    const dim3 grid((N + 255) / 256, nsplit);
    if (is_bf16)
        hipLaunchKernelGGL(colsum_partial_kernel<true>, grid, dim3(256), 0, stream, (const unsigned short*)Y, ld, M, N, rps, partial);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<false>, grid, dim3(256), 0, stream, (const unsigned short*)Y, ld, M, N, rps, partial);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_conv_wgrad_unpermute(const float* tmp, float* g, int cout, int cin, int ktaps, hipStream_t stream) {
    if (!tmp || !g || cout <= 0 || cin <= 0 || ktaps <= 0) return EEND_EINVAL;
    const long n = (long)cout * cin * ktaps;
    hipLaunchKernelGGL(conv_wgrad_unpermute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tmp, g, cout, cin, ktaps);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
