// Look-ahead Conv1d(256 -> 256, ktaps, padding) + bias + L2 normalisation on a packed weight stream (round 4): the operator of
// eend_conv1d_l2norm_f16 (FS model :38-41 / LS model :80-87; frames >= ilens[seq] read as zero) with the decomposition of
// ffn_stream.hip instead of the generic implicit GEMM (119 us for 81 GFLOP = 0.27 of the f16 peak):
//   * one workgroup per CU, one wave per SIMD; a tile = 128 consecutive frames of one sequence; a wave owns 64 of them and 128 of the
//     256 output features (128 accumulator registers): per 16-KB weight item a wave reads 8 weight and 4 input fragments for 32 MFMAs,
//     48 KB of LDS reads per item for the workgroup (with 32 frames x 256 features per wave it was 72 KB and LDS-read-bound);
//   * the tile's input rows (128 + ktaps - 1, halo included) are staged ONCE in LDS by LDS-DMA (chunk index XORed with the row via
//     the per-lane source address; rows outside [0, min(ilen, Tp)) are zero-filled by the buffer bounds check); the B fragment of
//     tap tau is the same tile read tau rows further down -- no im2col, no re-read;
//   * the weights, pre-packed per (tap, 32-wide channel block) in MFMA fragment order (eend_conv_stream_pack_f16), flow by LDS-DMA
//     through a 5-slot ring, one barrier per 16-KB item (32 MFMAs per wave), continuously across tiles;
//   * bias, sum of squares (wave-local + one exchange with the wave holding the other half of the features), x / ||x||; f32 and f16
//     half rows (256 / 512 contiguous bytes) leave through a staging tile as full 128-byte lines.
#include "common.h"
#include "kernels.h"
#include <type_traits>
#include <utility>

namespace {

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }
template <int V> using IC = std::integral_constant<int, V>;

typedef __attribute__((address_space(3))) char lds_char;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NR = 4;                // token fragments per wave (64 rows)
constexpr int NF = 8;                // feature fragments per wave (128 features)
constexpr int TM = 128, WM = 64;     // rows per tile / per wave
constexpr int SLOT = 16384;          // one stream item: 16 fragments of 1 KB = the weights of one (tap, 32-channel block)
constexpr int NSLOT = 5;
constexpr int MAXTAPS = 24;
constexpr int XROWS = TM + MAXTAPS;  // staged input rows (halo included), 512 B each
constexpr int L_RING = 0;
constexpr int L_X = NSLOT * SLOT;                 // 81920
constexpr int L_VEC = L_X + XROWS * 512;          // bias
constexpr int L_SS = L_VEC + 1024;                // [2 row groups][64 rows][2 halves] partial sums of squares
constexpr int SMEM = L_SS + 1024;                 // 161792
constexpr int NB = 8, PD = 4;
constexpr int INFL = 4 * (NSLOT - 3);             // this wave's pieces younger than the ones a barrier needs (2 items x 4)

// weight stream: item q = tau*8 + kc, fragment i, lane (f = l & 15, g = l >> 4): Wr[n(i,f)][tau*256 + kc*32 + g*8 + e] with
// n(i,f) = (i>>3)*128 + (f>>2)*32 + (i&7)*4 + (f&3): fragments 0..7 / 8..15 are the two feature halves, a lane's 32 features contiguous
__global__ void conv_stream_pack_kernel(const _Float16* __restrict__ Wr, _Float16* __restrict__ out, int ktaps) {
    const long total = (long)ktaps * 8 * 1024;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int item = (int)(t >> 10), w = (int)(t & 1023), i = w >> 6, l = w & 63, f = l & 15, g = l >> 4;
        const int tau = item >> 3, kc = item & 7;
        const _Float16* src = Wr + (size_t)((i >> 3) * 128 + (f >> 2) * 32 + (i & 7) * 4 + (f & 3)) * (ktaps * 256) + tau * 256 + kc * 32 + g * 8;
        _Float16* dst = out + t * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = src[e];
    }
}

__global__ __launch_bounds__(256, 1)
void conv_stream_kernel(const ConvStreamParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = p.ktaps * 8;                            // stream items per tile
    const int TPS = (p.Tp + TM - 1) / TM;                 // tiles per sequence
    const int ntiles = p.nseq * TPS;
    const int xrows = TM + p.ktaps - 1;

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, g = lane >> 4;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, S * SLOT, 0x00020000);
    int nxt = 0, slot = 0;
    auto dma_piece = [&](int sd, auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_char*)(smem + L_RING + sd * SLOT + wave * 4096 + i * 1024), 16, lane * 16 + wave * 4096,
                                                 nxt * SLOT + i * 1024, 0, 0);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) { nxt = nxt + 1 == S ? 0 : nxt + 1; };
    sfor<NSLOT - 1>([&](auto IT) __attribute__((always_inline)) {
        sfor<4>([&](auto I) __attribute__((always_inline)) { dma_piece(decltype(IT)::value, I); });
        dma_advance();
    });
    float* vb = (float*)(smem + L_VEC);
    vb[tid] = p.bias[tid];

    // the tile's input rows -> LDS: staging row r holds frame t0 - pad + r of the sequence, chunk c at (c ^ (r & 7))
    auto stage_x = [&](int tile) __attribute__((always_inline)) {
        const int seq = __builtin_amdgcn_readfirstlane(tile / TPS), t0 = (tile - seq * TPS) * TM;
        int len = p.ilens[seq];
        len = len < p.Tp ? len : p.Tp;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.X + (size_t)seq * p.Tp * 512), 0, len * 512, 0x00020000);
        const int npc = (xrows + 1) >> 1;
        for (int pc = wave; pc < npc; pc += 4) {
            const int r = 2 * pc + (lane >> 5);
            const int off = (t0 - p.pad + r) * 512 + (((lane & 31) ^ (r & 7)) << 4);     // negative / beyond the sequence: out of bounds -> zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_char*)(smem + L_X + pc * 1024), 16, off, 0, 0, 0);
        }
    };

    const int rg = wave >> 1, nh = wave & 1;             // row group (64 rows), feature half (128 features)
    f16x8 wf[NB];
    f32x4 acc[NF][NR];
    bool cold = true;                                     // the fragment rotation is empty (first item of the launch only)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; frow = lane & 15; g = lane >> 4;
        const int seq = __builtin_amdgcn_readfirstlane(tile / TPS), t0 = (tile - seq * TPS) * TM;
        // every wave is done with the staging region (input rows of the previous tile, then its output staging)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stage_x(tile);
        // the rows and item 0 have landed: everything this wave requested so far (the ring keeps streaming behind it)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const f32x4 b4 = *(const f32x4*)(vb + nh * 128 + g * 32 + i * 4);
#pragma unroll
            for (int j = 0; j < NR; ++j) acc[i][j] = b4;
        }
        const char* xl = smem + L_X;
        auto read_x = [&](int q, f16x8 (&xo)[NR]) __attribute__((always_inline)) {
            const int tau = q >> 3, kc = q & 7;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = rg * WM + j * 16 + frow + tau;
                xo[j] = *(const f16x8*)(xl + r * 512 + (((kc * 4 + g) ^ (r & 7)) << 4));
            }
        };
        f16x8 xf[NR], xn[NR];
        read_x(0, xf);
        for (int q = 0; q < S; ++q) {
            // this wave's pieces of the NEXT item have landed (2 younger items x 4 pieces may stay in flight) -- so behind the barrier
            // every wave's have, and its first fragments can be requested under this item's last MFMAs
            if (q > 0) {
                __builtin_amdgcn_s_waitcnt(0x0F70 | INFL);
                __builtin_amdgcn_s_barrier();
            }
            // accumulators pinned to the accumulator file, the fragment rotation to the vector file, once per item: left alone the
            // compiler renames them around the loop (124 v_accvgpr_mov per item, each waiting for the MFMA that wrote its source)
            const char* wc = smem + L_RING + slot * SLOT + nh * (NF * 1024) + lane * 16;
            const char* wn = smem + L_RING + (slot + 1 == NSLOT ? 0 : slot + 1) * SLOT + nh * (NF * 1024) + lane * 16;
            const int sd = slot == 0 ? NSLOT - 1 : slot - 1;
            if (cold) {
                sfor<PD>([&](auto Q) __attribute__((always_inline)) { wf[decltype(Q)::value % NB] = *(const f16x8*)(wc + decltype(Q)::value * 1024); });
                cold = false;
            }
            sfor<NF>([&](auto PI) __attribute__((always_inline)) {
                constexpr int pi = decltype(PI)::value;
                const f16x8 w = wf[pi % NB];
#pragma unroll
                for (int j = 0; j < NR; ++j) acc[pi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xf[j], acc[pi][j], 0, 0, 0);
                if constexpr (pi + PD < NF) wf[(pi + PD) % NB] = *(const f16x8*)(wc + (pi + PD) * 1024);
                else wf[(pi + PD) % NB] = *(const f16x8*)(wn + (pi + PD - NF) * 1024);
                if constexpr (pi < 4) dma_piece(sd, IC<pi>{});
                if constexpr (pi == 2) { if (q + 1 < S) read_x(q + 1, xn); }
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int j = 0; j < NR; ++j) xf[j] = xn[j];
            dma_advance();
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
        // every wave has read its last input fragments: the staging region becomes the output staging
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        asm volatile("" : "+v"(tid));
        lane = tid & 63; frow = lane & 15; g = lane >> 4;
        // sum of squares: this wave's 128 features, then the other half's through LDS
        float* ssl = (float*)(smem + L_SS);
        float ss[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            f32x2 sq2 = f32x2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const f32x2 x0 = f32x2{acc[i][j][0], acc[i][j][1]}, x1 = f32x2{acc[i][j][2], acc[i][j][3]};
                sq2 = x1 * x1 + (x0 * x0 + sq2);
            }
            ss[j] = wave_g_allreduce_add(sq2[0] + sq2[1]);
            if (g == 0) ssl[(rg * WM + j * 16 + frow) * 2 + nh] = ss[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        char* st = smem + L_X + wave * 4096;              // 8 rows x 512 B (f32 half rows) / 8 rows x 256 B (f16 half rows)
        const int tlim = p.Tp - t0;                       // rows of this tile inside the sequence
        sfor<NR>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const float rinv = 1.0f / __builtin_sqrtf(ss[j] + ssl[(rg * WM + j * 16 + frow) * 2 + (nh ^ 1)]);
            if (p.inv_norm) { const int rr = rg * WM + j * 16 + frow; if (g == 0 && nh == 0 && rr < tlim) p.inv_norm[(size_t)seq * p.Tp + t0 + rr] = rinv; }
#pragma unroll
            for (int half = 0; half < 2; ++half) {          // token rows 0..7 / 8..15 of the fragment
                const int rb = rg * WM + j * 16 + half * 8;
                // f16 half rows: the lane's 32 features = 4 chunks of 16 B at chunk g*4 + e of a 256-byte row
                if ((frow >> 3) == half) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f16x8 o;
#pragma unroll
                        for (int q = 0; q < 8; ++q) o[q] = to_f16_sat(acc[e * 2 + (q >> 2)][j][q & 3] * rinv);
                        *(f16x8*)(st + (frow & 7) * 256 + (((g * 4 + e) ^ (frow & 7)) << 4)) = o;
                    }
                }
                wave_lds_sync();
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int rr = lane >> 3, cc = (lane & 7) + 8 * ps;
                    const f16x8 v = *(const f16x8*)(st + rr * 256 + ((cc ^ rr) << 4));
                    if (rb + rr < tlim) *(f16x8*)((_Float16*)p.out16 + ((size_t)seq * p.Tp + t0 + rb + rr) * 256 + nh * 128 + cc * 8) = v;
                }
                wave_lds_sync();
                // f32 half rows: 8 chunks of 16 B at chunk g*8 + e of a 512-byte row
                if ((frow >> 3) == half) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) *(f32x4*)(st + (frow & 7) * 512 + (((g * 8 + e) ^ (frow & 7)) << 4)) = acc[e][j] * rinv;
                }
                wave_lds_sync();
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int rr = lane >> 3, cc = (lane & 7) + 8 * ps;
                    const f32x4 v = *(const f32x4*)(st + rr * 512 + ((cc ^ rr) << 4));
                    if (rb + rr < tlim) *(f32x4*)(p.out32 + ((size_t)seq * p.Tp + t0 + rb + rr) * 256 + nh * 128 + cc * 4) = v;
                }
                wave_lds_sync();
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the workgroup
}

}  // namespace

long eend_conv_stream_nelems(int ktaps) { return (long)ktaps * 8 * (SLOT / 2); }

int eend_conv_stream_supported(int cin, int ktaps, int pad) { return cin == 256 && ktaps >= 1 && ktaps <= MAXTAPS && pad >= 0 && pad < ktaps; }

int eend_launch_conv_stream_pack(const void* Wr, void* out, int ktaps, hipStream_t stream) {
    if (!Wr || !out || ktaps < 1 || ktaps > MAXTAPS) return EEND_EINVAL;
    hipLaunchKernelGGL(conv_stream_pack_kernel, dim3(ktaps * 8 * 4), dim3(256), 0, stream, (const _Float16*)Wr, (_Float16*)out, ktaps);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_conv_stream(const ConvStreamParams& p, hipStream_t stream) {
    if (!p.X || !p.wstream || !p.bias || !p.ilens || !p.out32 || !p.out16 || p.nseq <= 0 || p.Tp <= 0 || !eend_conv_stream_supported(256, p.ktaps, p.pad) ||
        (long)p.Tp * 512 >= (1L << 31))
        return EEND_EINVAL;
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)conv_stream_kernel, SMEM)) return EEND_ELAUNCH;
    const int ncu = eend_cu_count();
    const int ntiles = p.nseq * ((p.Tp + TM - 1) / TM);
    hipLaunchKernelGGL(conv_stream_kernel, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), SMEM, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
