// First half of a speaker-fusion decoder layer on a packed weight stream (round 4), one launch instead of two:
//     x1 = LayerNorm11(A Wo1^T + bo1 + res)                     (time-axis attention out-projection, residual, norm11)
//     O  = MHA_over_slots(x1 Win2^T + bin2)                     (speaker-axis in-projection + the C x C attention of every frame)
// Reference sites: FS merge_tfm_encoder.py:356-394 (_sa_block1 tail + norm11, _sa_block2), LS merge_retnet_layer.py:301-306.
// Replaces eend_linear_res16_ln_f16 + eend_spk_qkv_attn_f16 on the hot path for every slot count C <= 12 (3 / 6 / 12 fill the tiling
// exactly; the others leave phantom slot positions that are masked and never stored).
//
// Same machinery as ffn_stream.hip (one wave per SIMD, 48 token rows per wave, weight fragments streamed by LDS-DMA through an
// 8-slot ring, one barrier per 16-KB item, LayerNorm output == next GEMM's B operand), with two differences:
//   * a wave's 48 rows are the C slots of 48/C consecutive frames (row = (b*C + c)*Tp + t), token index = c*G + t', so every
//     frame's slots sit in ONE wave: in the MFMA output layout (lane = token column, 4 rows per 16-lane group) the keys and
//     values of the other slots of a lane's frame are in the same lane (other token fragment) or a fixed rotation away inside
//     the 16-lane row -- DPP row_ror, no LDS, no barrier.  The whole attention is register arithmetic on the projection's
//     accumulators; q, k, v never exist in memory in any form (and are never rounded to f16).
//   * the in-projection rows are permuted inside each head (MFMA row rho of feature fragment ff <-> feature (rho>>2)*16 + ff*4
//     + (rho&3)) so that a lane holds 16 CONSECUTIVE head features of its tokens: the head's output leaves as full 128-byte lines.
// The key bias is dropped: q . b_k is the same for every key of a query and cancels in the softmax.
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

constexpr int NJ = 3;                // token fragments per wave (48 rows)
constexpr int SLOT = 16384;          // one stream item: 16 fragments of 1 KB
constexpr int NSLOT = 8;
constexpr int STAGE = NSLOT * SLOT;  // 4 x 4 KB wave-private output staging
constexpr int VECS = STAGE + 4 * 4096;   // f32 vectors: bo, g1, be1 (3 x 256), bin (768)
constexpr int SMEM = VECS + 6 * 1024;    // 153600
constexpr int NB = 8;                // weight-fragment registers in rotation
constexpr int PD = 6;                // fragment prefetch distance
constexpr int INFL = 4 * (NSLOT - 3);
constexpr int NITEMS = 8 + 24;       // Wo1: 8 items; in-projection: 4 heads x {q, k, v} x 2 halves of 32 features

// ---------------------------------------------------------------------------------------------------------------
// weight stream packing, one thread per 16 bytes.  lane = (f = l & 15, g = l >> 4):
//   items 0..7   Wo (kc = item >> 1, sl = item & 1), fragment i : Wo[(f>>2)*64 + i*4 + (f&3)][kc*64 + sl*32 + g*8 + e]
//   item 8 + h*6 + t*2 + u (t = 0 q, 1 k, 2 v), fragment p = s*2 + hf :
//                Win[t*256 + h*64 + (f>>2)*16 + (u*2+hf)*4 + (f&3)][g*64 + 8s + e]
__global__ void spk_stream_pack_kernel(const _Float16* __restrict__ Wo, const _Float16* __restrict__ Win, _Float16* __restrict__ out) {
    const long total = (long)NITEMS * (SLOT / 16);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int item = (int)(t >> 10), w = (int)(t & 1023);
        const int pfrag = w >> 6, l = w & 63, f = l & 15, g = l >> 4;
        const _Float16* src;
        if (item < 8) {
            const int kc = item >> 1, sl = item & 1;
            src = Wo + (size_t)((f >> 2) * 64 + pfrag * 4 + (f & 3)) * 256 + kc * 64 + sl * 32 + g * 8;
        } else {
            const int q = item - 8, h = q / 6, tt = (q % 6) >> 1, u = q & 1;
            const int s_ = pfrag >> 1, hf = pfrag & 1;
            src = Win + (size_t)(tt * 256 + h * 64 + (f >> 2) * 16 + (u * 2 + hf) * 4 + (f & 3)) * 256 + g * 64 + 8 * s_;
        }
        _Float16* dst = out + t * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = src[e];
    }
}

// Perf-study build (-DEEND_SPK_TRACE, tools/spk_stream_trace.py): s_memtime stamps of the tile phases of wave 0 of every
// workgroup (first 4 tiles), read back through eend_debug_spk_stream_trace; never defined in the shipped library.
#ifdef EEND_SPK_TRACE
__device__ unsigned long long g_spks_trace[256 * 4 * 12];
#define SPK_STAMP(k) do { ts[k] = __builtin_amdgcn_s_memtime(); } while (0)      /* kept in scalar registers, written at the tile's end */
#define SPK_STAMP_H(base, head)                                                                                        \
    do {                                                                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                    \
        if ((head) == 0) ts[base] = t_; else if ((head) == 1) ts[(base) + 2] = t_; else if ((head) == 2) ts[(base) + 4] = t_; else ts[(base) + 6] = t_; \
    } while (0)
#else
#define SPK_STAMP_H(base, head) do {} while (0)
#define SPK_STAMP(k) do {} while (0)
#endif

template <int N>
__device__ __forceinline__ float row_rot(float x) {          // value of the lane N places away inside the 16-lane row
    if constexpr (N == 0) return x;
    else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xF, 0xF, false));
}

// ---------------------------------------------------------------------------------------------------------------
// G frames per wave, R = 16/G slot positions per token fragment, C = 3R positions of which the first CC hold the model's slots.
// CC < C: the phantom positions read the last slot's rows (any valid rows do), are masked as keys (a -1e30 score bias that is rotated
// with the keys, so it always describes the lane it came from) and are never stored.
// R32 (round 5, LS-EEND's decoder): the residual stream is f32 -- res32 rows in, x1 rows out as f32 (x32; may be res32), no f16 copy of x1
// (the layer tail behind reads the f32 stream; merge_retnet_layer.py:301-306 with the f32 residual of DESIGN 4).
template <int G, int CC, bool R32>
__global__ __launch_bounds__(256, 1)
void spk_stream_kernel(const SpkStreamParams p) {
    constexpr int R = 16 / G, C = 3 * R;
    constexpr bool FULL = CC == C;
    static_assert(CC >= 1 && CC <= C, "slot count beyond the positions of this tiling");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TPB = p.Tp / (4 * G);                       // tiles per utterance
    const int ntiles = p.B * TPB;

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, g = lane >> 4;
    int fo = g * 64;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, NITEMS * SLOT, 0x00020000);
    int dvo = lane * 16 + wave * 4096;
    int nxt = 0;
    int slot = 0;

    auto dma_piece = [&](int sd, auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_char*)(smem + sd * SLOT + wave * 4096 + i * 1024), 16, dvo,
                                                 nxt * SLOT + i * 1024, 0, 0);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) { nxt = nxt + 1 == NITEMS ? 0 : nxt + 1; };

    sfor<NSLOT - 1>([&](auto IT) __attribute__((always_inline)) {
        sfor<4>([&](auto I) __attribute__((always_inline)) { dma_piece(decltype(IT)::value, I); });
        dma_advance();
    });

    float* vecs = (float*)(smem + VECS);                  // [0..767]: bo, g1, be1; [768..1535]: bin (q, k, v)
    {
        vecs[0 * 256 + tid] = p.bo[tid];
        vecs[1 * 256 + tid] = p.g1[tid];
        vecs[2 * 256 + tid] = p.be1[tid];
        vecs[3 * 256 + tid] = p.bin[tid];
        vecs[4 * 256 + tid] = p.bin[256 + tid];
        vecs[5 * 256 + tid] = p.bin[512 + tid];
    }
    auto vec4 = [&](int which, int i) __attribute__((always_inline)) { return *(const f32x4*)(vecs + which * 256 + fo + i * 4); };

    const char* wl = smem + lane * 16;
    f16x8 wf[NB];
    f32x4 acc[16][NJ];                                    // out-projection accumulators, features fo + i*4 + r
    f32x4 qkv[12][NJ];                                    // one head: [t*4 + ff], features g*16 + ff*4 + r of the head
    f16x8 xf[8][NJ];

    // memory row of token (fragment j, column fr) of a tile
    auto slot_of = [&](int j, int fr) __attribute__((always_inline)) { return j * R + fr / G; };
    auto row_tok = [&](int tile, int j, int fr) __attribute__((always_inline)) {
        const int b = tile / TPB, tt = tile - b * TPB;
        int c = slot_of(j, fr);
        if constexpr (!FULL) c = c < CC ? c : CC - 1;
        return (b * CC + c) * p.Tp + tt * (4 * G) + wave * G + (fr % G);
    };
    auto load_in_frags = [&](int tile, auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        const _Float16* src = (const _Float16*)p.A + (size_t)row_tok(tile, j, frow) * p.lda + g * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) xf[s][j] = *(const f16x8*)(src + s * 32);
    };
    f16x8 r8[R32 ? 1 : NJ][8];
    auto load_res16 = [&](int tile, auto J) __attribute__((always_inline)) {
        if constexpr (!R32) {
            constexpr int j = decltype(J)::value;
            const _Float16* src = (const _Float16*)p.res16 + (size_t)row_tok(tile, j, frow) * 256 + fo;
#pragma unroll
            for (int e = 0; e < 8; ++e) r8[j][e] = *(const f16x8*)(src + e * 8);
        }
    };
    // R32: two 64-register buffers; a fragment's buffer receives its x1 values in place (LayerNorm pass 2) and is the source of its
    // f32 row stores, then takes the residual rows of the fragment after next
    f32x4 t4[R32 ? 2 : 1][16];
    auto load_res32 = [&](int tile, auto J, auto BUF) __attribute__((always_inline)) {
        if constexpr (R32) {
            constexpr int j = decltype(J)::value, b = decltype(BUF)::value;
            const float* src = p.res32 + (size_t)row_tok(tile, j, frow) * 256 + fo;
#pragma unroll
            for (int i = 0; i < 16; ++i) t4[b][i] = *(const f32x4*)(src + i * 4);
        }
    };

    __builtin_amdgcn_s_waitcnt(0x0070 | ((4 * (NSLOT - 2)) & 15) | (((4 * (NSLOT - 2)) >> 4) << 14));   // item 0 has landed; lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (blockIdx.x < ntiles) sfor<NJ>([&](auto J) __attribute__((always_inline)) { load_in_frags(blockIdx.x, J); });

    // One stream item.  KIND 0: acc += Wo(item) x xf[src]; KIND 1: qkv[TU*2 + hf] = Win2 fragments x xf (TU = t*2 + u).
    // vmcnt(INFL + VWX): the VMEM operations of this wave that are certainly younger than its pieces of the NEXT item (those were
    // requested six items earlier): the INFL pieces of the five items in between plus VWX loads / stores issued since.  Compile-time
    // (a run-time choice between two s_waitcnt compiles to both); undercounting is safe, it only waits for more.
    auto step = [&](auto KIND, auto SRCc, auto COLDc, auto PFNc, auto VWXc) __attribute__((always_inline)) {
        constexpr int kind = decltype(KIND)::value, src = decltype(SRCc)::value, vw = INFL + decltype(VWXc)::value;
        constexpr bool cold = decltype(COLDc)::value, pfn = decltype(PFNc)::value;
        static_assert(vw <= 63, "vmcnt is a 6-bit field");
        __builtin_amdgcn_s_waitcnt(0x0F70 | (vw & 15) | ((vw >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        const char* wc = wl + slot * SLOT;
        const char* wn = wl + ((slot + 1) & (NSLOT - 1)) * SLOT;
        const int sd = (slot + NSLOT - 1) & (NSLOT - 1);
        if constexpr (cold) {
            sfor<PD>([&](auto Q) __attribute__((always_inline)) {
                wf[decltype(Q)::value % NB] = *(const f16x8*)(wc + decltype(Q)::value * 1024);
            });
        }
        sfor<8>([&](auto P2) __attribute__((always_inline)) {
            sfor<2>([&](auto PH) __attribute__((always_inline)) {
                constexpr int pi = decltype(P2)::value * 2 + decltype(PH)::value;
                const f16x8 w = wf[pi % NB];
                if constexpr (kind == 0) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[pi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xf[src][j], acc[pi][j], 0, 0, 0);
                } else {
                    constexpr int s_ = pi >> 1, hf = pi & 1, idx = src * 2 + hf;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        qkv[idx][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xf[s_][j], s_ == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : qkv[idx][j], 0, 0, 0);
                }
                if constexpr (pi + PD < 16) wf[(pi + PD) % NB] = *(const f16x8*)(wc + (pi + PD) * 1024);
                else if constexpr (pfn) wf[(pi + PD) % NB] = *(const f16x8*)(wn + (pi + PD - 16) * 1024);
                if constexpr (pi < 4) dma_piece(sd, IC<pi>{});
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        dma_advance();
        slot = (slot + 1) & (NSLOT - 1);
    };

    char* st = smem + STAGE + wave * 4096;
#ifdef EEND_SPK_TRACE
    int tix = -1;
#endif
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; frow = lane & 15; g = lane >> 4; fo = g * 64;
        dvo = lane * 16 + wave * 4096;
        wl = smem + lane * 16;
        st = smem + STAGE + wave * 4096;
        const int ntile = tile + (int)gridDim.x;
#ifdef EEND_SPK_TRACE
        ++tix;
        unsigned long long ts[11];
#endif
        SPK_STAMP(0);
        using T = std::true_type;
        using Fa = std::false_type;

        // ---- x1 = LN11(A Wo1^T + bo1 + res)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 b4 = vec4(0, i);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = b4;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i][j]));
        // items 0..4: the tile's 24 input-row loads are younger than the pieces they wait for (first tile: issued behind the ring
        // prime; later: behind the previous tile's last item, with 6 output stores on top).  The residual rows are requested
        // three, two and one item ahead of the LayerNorm that adds them, into registers the consumed input fragments freed.
        step(IC<0>{}, IC<0>{}, T{}, T{}, IC<24>{});
        step(IC<0>{}, IC<1>{}, Fa{}, T{}, IC<24>{});
        step(IC<0>{}, IC<2>{}, Fa{}, T{}, IC<24>{});
        step(IC<0>{}, IC<3>{}, Fa{}, T{}, IC<24>{});
        step(IC<0>{}, IC<4>{}, Fa{}, T{}, IC<24>{});
        step(IC<0>{}, IC<5>{}, Fa{}, T{}, IC<24>{});
        load_res32(tile, IC<0>{}, IC<0>{});              // (R32: 16 loads into the registers of the input fragments 0 .. 5)
        step(IC<0>{}, IC<6>{}, Fa{}, T{}, IC<(R32 ? 16 : 0)>{});
        load_res16(tile, IC<0>{});
        load_res32(tile, IC<1>{}, IC<1>{});
        step(IC<0>{}, IC<7>{}, Fa{}, Fa{}, IC<(R32 ? 32 : 8)>{});
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i][j]));
        SPK_STAMP(1);
        _Float16* x16 = (_Float16*)p.x16;
        sfor<NJ>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            constexpr int tb = R32 ? (j & 1) : 0;
            auto resv = [&](int i, int q) __attribute__((always_inline)) { return (float)r8[R32 ? 0 : j][i >> 1][(i & 1) * 4 + q]; };
            // Two passes over the accumulators (AGPR reads are cheap) instead of a 64-value buffer: the buffer next to the residual
            // rows of this and the next fragment overflowed the register file, and a scratch reload waits for every VMEM
            // operation in flight (stores, weight DMA).  Statistics: sum and sum of squares in one pass (f32; |x| = O(10)).
            if constexpr (R32) {
                // f32 stream: the rows carry the decoder's state at full f32 precision, so the statistics are the two-pass form
                // (mean, then centred squares -- no E[x^2] - mean^2 cancellation); the row values replace the residual buffer in place
                f32x2 sm = f32x2{0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 v = acc[i][j] + t4[tb][i];
                    t4[tb][i] = v;
                    sm += f32x2{v[0], v[1]} + f32x2{v[2], v[3]};
                }
                const float mean = wave_g_allreduce_add(sm[0] + sm[1]) * (1.0f / 256);
                f32x2 sq2 = f32x2{0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 d = t4[tb][i] - mean;
                    t4[tb][i] = d;
                    sq2 = f32x2{d[2], d[3]} * f32x2{d[2], d[3]} + (f32x2{d[0], d[1]} * f32x2{d[0], d[1]} + sq2);
                }
                const float rstd = 1.0f / __builtin_sqrtf(wave_g_allreduce_add(sq2[0] + sq2[1]) * (1.0f / 256) + p.eps1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 g4 = vec4(1, i), b4 = vec4(2, i);
                    const f32x4 d = t4[tb][i] * rstd;
                    f32x4 x;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        x[q] = __builtin_fmaf(d[q], g4[q], b4[q]);
                        xf[i >> 1][j][(i & 1) * 4 + q] = (_Float16)x[q];
                    }
                    t4[tb][i] = x;
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            f32x2 sm = f32x2{0.f, 0.f}, sq2 = f32x2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 a4 = acc[i][j];
                const f32x2 x0 = f32x2{a4[0] + resv(i, 0), a4[1] + resv(i, 1)};
                const f32x2 x1 = f32x2{a4[2] + resv(i, 2), a4[3] + resv(i, 3)};
                sm += x0 + x1;
                sq2 = x1 * x1 + (x0 * x0 + sq2);
            }
            const float sum = wave_g_allreduce_add(sm[0] + sm[1]);
            const float sqs = wave_g_allreduce_add(sq2[0] + sq2[1]);
            const float mean = sum * (1.0f / 256);
            const float var = __builtin_fmaxf(sqs * (1.0f / 256) - mean * mean, 0.f);
            const float rstd = 1.0f / __builtin_sqrtf(var + p.eps1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j + 1 < NJ) load_res16(tile, IC<j + 1>{});
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 gg = vec4(1, i) * rstd, bb = vec4(2, i) - vec4(1, i) * (rstd * mean);
                const f32x4 a4 = acc[i][j];
#pragma unroll
                for (int q = 0; q < 4; ++q) xf[i >> 1][j][(i & 1) * 4 + q] = (_Float16)__builtin_fmaf(a4[q] + resv(i, q), gg[q], bb[q]);
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            }
            if constexpr (R32) {
                // x1 rows leave as f32 through the staging tile: 8 token rows x one 32-feature half of each 64-feature block per pass
                float* x32 = p.x32;
#pragma unroll
                for (int fh = 0; fh < 2; ++fh)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if ((frow >> 3) == half) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) *(f32x4*)(st + (frow & 7) * 512 + (((g * 8 + e) ^ (frow & 7)) << 4)) = t4[tb][fh * 8 + e];
                        }
                        wave_lds_sync();
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const int rr = 2 * q4 + (lane >> 5), cc = lane & 31;
                            const f32x4 v4 = *(const f32x4*)(st + rr * 512 + ((cc ^ rr) << 4));
                            if ((FULL || slot_of(j, half * 8 + rr) < CC))
                                *(f32x4*)(x32 + (size_t)row_tok(tile, j, half * 8 + rr) * 256 + (cc >> 3) * 64 + fh * 32 + (cc & 7) * 4) = v4;
                        }
                        wave_lds_sync();
                    }
                if constexpr (j == 0 && NJ > 2) load_res32(tile, IC<2>{}, IC<0>{});
            }
            // x1 rows leave through the staging tile as whole 512-byte rows
#pragma unroll
            for (int half = 0; half < (R32 ? 0 : 2); ++half) {
                if ((frow >> 3) == half) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) *(f16x8*)(st + (frow & 7) * 512 + (((g * 8 + e) ^ (frow & 7)) << 4)) = xf[e][j];
                }
                wave_lds_sync();
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int rr = 2 * q4 + (lane >> 5), cc = lane & 31;
                    const f16x8 v4 = *(const f16x8*)(st + rr * 512 + ((cc ^ rr) << 4));
                    if ((FULL || slot_of(j, half * 8 + rr) < CC)) *(f16x8*)(x16 + (size_t)row_tok(tile, j, half * 8 + rr) * 256 + cc * 8) = v4;
                }
                wave_lds_sync();
            }
            __builtin_amdgcn_sched_barrier(0);
        });

        SPK_STAMP(2);
        // ---- per head: q, k, v of the wave's 48 tokens (6 items), then the C x C attention of its frames in registers
        _Float16* O = (_Float16*)p.O;
        // (the last head is peeled: the next tile's input loads issued there would otherwise look pending at every iteration's top)
        // score bias of the key each (fragment, rotation) delivers to this lane: 0 for a real slot, -1e30 for a phantom position
        float kbias[FULL ? 1 : C];
        if constexpr (!FULL) {
            sfor<NJ>([&](auto J2) __attribute__((always_inline)) {
                constexpr int j2 = decltype(J2)::value;
                const float own = slot_of(j2, frow) < CC ? 0.f : -1e30f;
                sfor<R>([&](auto D) __attribute__((always_inline)) { kbias[j2 * R + decltype(D)::value] = row_rot<decltype(D)::value * G>(own); });
            });
        }
        // YNG: this wave's stores certainly younger than the pieces the head's six waits need -- the 6 output stores of the previous
        // head, or (head 0) the x1 row stores: 8 per fragment, or 16 as f32 (capped by the 6-bit counter)
        auto head_body = [&](int head, auto LAST, auto YNG) __attribute__((always_inline)) {
            constexpr int yng = decltype(YNG)::value;
            step(IC<1>{}, IC<0>{}, T{}, T{}, IC<yng>{});
            step(IC<1>{}, IC<1>{}, Fa{}, T{}, IC<yng>{});
            step(IC<1>{}, IC<2>{}, Fa{}, T{}, IC<yng>{});
            step(IC<1>{}, IC<3>{}, Fa{}, T{}, IC<yng>{});
            step(IC<1>{}, IC<4>{}, Fa{}, T{}, IC<yng>{});
            step(IC<1>{}, IC<5>{}, Fa{}, Fa{}, IC<yng>{});
            SPK_STAMP_H(3, head);
            if (decltype(LAST)::value && ntile < ntiles)              // x1 is dead: the next tile's input rows travel under the last attention
                sfor<NJ>([&](auto J) __attribute__((always_inline)) { load_in_frags(ntile, J); });

            // packed f32 arithmetic (v_pk_fma_f32: two FMAs per lane and instruction) on register pairs of the accumulator quads
            const float* bq = vecs + 3 * 256 + head * 64 + g * 16;
            const float* bv = vecs + 5 * 256 + head * 64 + g * 16;
            f32x4 bnext = *(const f32x4*)bq;              // bias quads are requested one fragment ahead of their use
            f32x2 s2[NJ][C];
#pragma unroll
            for (int a = 0; a < NJ; ++a)
#pragma unroll
                for (int c = 0; c < C; ++c) s2[a][c] = f32x2{0.f, 0.f};
            f16x8 of[NJ][2];
            sfor<4>([&](auto FF) __attribute__((always_inline)) {
                constexpr int ff = decltype(FF)::value;
                const f32x4 b4 = bnext;
                bnext = ff < 3 ? *(const f32x4*)(bq + (ff + 1) * 4) : *(const f32x4*)bv;
                f32x2 q[NJ][2];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const f32x4 t = (qkv[ff][j] + b4) * p.scale;
                    q[j][0] = f32x2{t[0], t[1]}; q[j][1] = f32x2{t[2], t[3]};
                }
                sfor<NJ>([&](auto J2) __attribute__((always_inline)) {
                    constexpr int j2 = decltype(J2)::value;
                    const f32x4 k = qkv[4 + ff][j2];
                    sfor<R>([&](auto D) __attribute__((always_inline)) {
                        constexpr int d = decltype(D)::value;
                        const f32x2 k0 = f32x2{row_rot<d * G>(k[0]), row_rot<d * G>(k[1])};
                        const f32x2 k1 = f32x2{row_rot<d * G>(k[2]), row_rot<d * G>(k[3])};
#pragma unroll
                        for (int j1 = 0; j1 < NJ; ++j1) {
                            s2[j1][j2 * R + d] = q[j1][1] * k1 + (q[j1][0] * k0 + s2[j1][j2 * R + d]);
                        }
                    });
                });
            });
            float s[NJ][C];
#pragma unroll
            for (int a = 0; a < NJ; ++a) {
                float mx = -INFINITY, den = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    s[a][c] = s2[a][c][0] + s2[a][c][1];
                    if constexpr (!FULL) s[a][c] += kbias[c];
                    s[a][c] = wave_g_allreduce_add(s[a][c]);
                    mx = __builtin_fmaxf(mx, s[a][c]);
                }
#pragma unroll
                for (int c = 0; c < C; ++c) { s[a][c] = __expf(s[a][c] - mx); den += s[a][c]; }
                const float inv = __builtin_amdgcn_rcpf(den);
#pragma unroll
                for (int c = 0; c < C; ++c) s[a][c] *= inv;
            }
            sfor<4>([&](auto FF) __attribute__((always_inline)) {
                constexpr int ff = decltype(FF)::value;
                const f32x4 b4 = bnext;
                if constexpr (ff < 3) bnext = *(const f32x4*)(bv + (ff + 1) * 4);
                f32x2 o[NJ][2];
#pragma unroll
                for (int j = 0; j < NJ; ++j) { o[j][0] = f32x2{b4[0], b4[1]}; o[j][1] = f32x2{b4[2], b4[3]}; }
                sfor<NJ>([&](auto J2) __attribute__((always_inline)) {
                    constexpr int j2 = decltype(J2)::value;
                    const f32x4 vv = qkv[8 + ff][j2];
                    sfor<R>([&](auto D) __attribute__((always_inline)) {
                        constexpr int d = decltype(D)::value;
                        const f32x2 v0 = f32x2{row_rot<d * G>(vv[0]), row_rot<d * G>(vv[1])};
                        const f32x2 v1 = f32x2{row_rot<d * G>(vv[2]), row_rot<d * G>(vv[3])};
#pragma unroll
                        for (int j1 = 0; j1 < NJ; ++j1) {
                            const f32x2 pw = f32x2{s[j1][j2 * R + d], s[j1][j2 * R + d]};
                            o[j1][0] = pw * v0 + o[j1][0];
                            o[j1][1] = pw * v1 + o[j1][1];
                        }
                    });
                });
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    of[j][ff >> 1][(ff & 1) * 4 + 0] = to_f16_sat(o[j][0][0]); of[j][ff >> 1][(ff & 1) * 4 + 1] = to_f16_sat(o[j][0][1]);
                    of[j][ff >> 1][(ff & 1) * 4 + 2] = to_f16_sat(o[j][1][0]); of[j][ff >> 1][(ff & 1) * 4 + 3] = to_f16_sat(o[j][1][1]);
                }
            });
            // the head's 64 features of 16 tokens = 16 full 128-byte lines per token fragment, through the staging tile (2 KB per
            // fragment: fragments 0 and 1 in one LDS round trip, fragment 2 in a second)
            auto stage_out = [&](auto J0, auto NF) __attribute__((always_inline)) {
                constexpr int j0 = decltype(J0)::value, nf = decltype(NF)::value;
#pragma unroll
                for (int jj = 0; jj < nf; ++jj) {
                    char* sj = st + jj * 2048;
                    *(f16x8*)(sj + frow * 128 + (((g * 2) ^ (frow & 7)) << 4)) = of[j0 + jj][0];
                    *(f16x8*)(sj + frow * 128 + (((g * 2 + 1) ^ (frow & 7)) << 4)) = of[j0 + jj][1];
                }
                wave_lds_sync();
                f16x8 v4[nf][2];
                const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
                for (int jj = 0; jj < nf; ++jj)
#pragma unroll
                    for (int half = 0; half < 2; ++half) v4[jj][half] = *(const f16x8*)(st + jj * 2048 + (half * 8 + rr) * 128 + ((cc ^ rr) << 4));
#pragma unroll
                for (int jj = 0; jj < nf; ++jj)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        if ((FULL || slot_of(j0 + jj, half * 8 + rr) < CC))
                            *(f16x8*)(O + (size_t)row_tok(tile, j0 + jj, half * 8 + rr) * 256 + head * 64 + cc * 8) = v4[jj][half];
                wave_lds_sync();
            };
            stage_out(IC<0>{}, IC<2>{});
            stage_out(IC<2>{}, IC<1>{});
            SPK_STAMP_H(4, head);
            __builtin_amdgcn_sched_barrier(0);
        };
        head_body(0, Fa{}, IC<(R32 ? 63 - INFL : 8 * NJ)>{});
        for (int head = 1; head < 3; ++head) head_body(head, Fa{}, IC<6>{});
        head_body(3, T{}, IC<6>{});
#ifdef EEND_SPK_TRACE
        if (tix < 4 && threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 11; ++k) g_spks_trace[((size_t)blockIdx.x * 4 + tix) * 12 + k] = ts[k];
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int G, int CC, bool R32>
int launch(const SpkStreamParams& p, hipStream_t stream) {
    static EendOncePerDevice attr_once;
    auto kern = spk_stream_kernel<G, CC, R32>;
    if (!eend_set_dynamic_lds(attr_once, (const void*)kern, SMEM)) return EEND_ELAUNCH;
    const int ncu = eend_cu_count();
    const int ntiles = p.B * (p.Tp / (4 * G));
    hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), SMEM, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

constexpr int frames_per_wave(int C) { return C <= 3 ? 16 : C <= 6 ? 8 : 4; }

}  // namespace

#ifdef EEND_SPK_TRACE
extern "C" int eend_debug_spk_stream_trace(void* dst, void* stream) {
    return hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(g_spks_trace), sizeof(g_spks_trace), 0, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}
#endif

long eend_spk_stream_nelems() { return (long)NITEMS * (SLOT / 2); }

int eend_launch_spk_stream_pack(const void* Wo, const void* Win, void* out, hipStream_t stream) {
    if (!Wo || !Win || !out) return EEND_EINVAL;
    const long total = eend_spk_stream_nelems() / 8;
    hipLaunchKernelGGL(spk_stream_pack_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, (const _Float16*)Wo,
                       (const _Float16*)Win, (_Float16*)out);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_spk_stream_supported(int C, int Tp) {
    if (C < 1 || C > 12) return 0;
    return Tp > 0 && Tp % (4 * frames_per_wave(C)) == 0;
}

int eend_launch_spk_stream(const SpkStreamParams& p, hipStream_t stream) {
    const bool r32 = p.res32 != nullptr;
    if (!p.A || !p.wstream || !p.bo || !p.g1 || !p.be1 || !p.bin || !p.O || p.B <= 0 || (p.lda & 7) || !eend_spk_stream_supported(p.C, p.Tp) ||
        (r32 ? (!p.x32 || p.res16 || p.x16 || (((size_t)p.res32 | (size_t)p.x32) & 15)) : (!p.res16 || !p.x16 || p.x32)))
        return EEND_EINVAL;
    switch (p.C) {
#define SPK_CASE(n) case n: return r32 ? launch<frames_per_wave(n), n, true>(p, stream) : launch<frames_per_wave(n), n, false>(p, stream);
        SPK_CASE(1) SPK_CASE(2) SPK_CASE(3) SPK_CASE(4) SPK_CASE(5) SPK_CASE(6) SPK_CASE(7) SPK_CASE(8) SPK_CASE(9) SPK_CASE(10)
        SPK_CASE(11) SPK_CASE(12)
#undef SPK_CASE
        default: return EEND_EINVAL;
    }
}
