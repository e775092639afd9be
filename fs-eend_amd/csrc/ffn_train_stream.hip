// Training forms of the post-norm FFN block on the packed weight stream of ffn_stream.hip (round 6):
//
//   TR = 1, forward   (eend_ffn_train_stream_f16; reference sites: nn.TransformerEncoderLayer._ff_block + norm2 of FS model :147,
//                      merge_tfm_encoder.py:395-399 linear1 / dropout / linear2 / dropout2 / norm22, LS merge_retnet_layer.py:249-253)
//       h   = drop1(relu(X W1^T + b1))                 f16, ALSO written to hid [M][F] (saved for the backward; its zeros are the mask)
//       y   = drop2(h W2^T + b2) * alpha + res         f32
//       out32 = LayerNorm(y), out16 = f16(out32), xhat16 = f16((y - mean) * rstd), rstat = rstd
//   TR = 2, data gradient of the same block (eend_ffn_bwd_data_stream_bf16)
//       dH  = scale * (dY W2) where hid != 0, else 0   bf16 [M][F], written once (the weight gradient of linear1 reads it)
//       g  += dH W1                                    the f32 residual-gradient stream, in place
//
// Same operators as ffn.hip MODE 3 / 4 (which stay as the shape fallback) in the decomposition of ffn_stream.hip: one 256-thread workgroup
// per CU, one wave per SIMD owning 16 NJ token rows end to end, accumulators for all 256 output features in registers, hidden units never
// in LDS, weights flowing through an 8-slot LDS-DMA ring with one barrier per 16-KB item.  What differs from the inference kernel:
//   * the hidden-unit order inside a 32-unit half-chunk is chosen so that a lane's 8 units are CONSECUTIVE (unit = 32 k + 8 g + e): the
//     activation / gradient rows leave (and the saved activations arrive) as one 16-byte buffer access per lane and token fragment,
//     straight from / to the registers that are the second GEMM's B operand.  The stream is packed accordingly
//     (eend_ffn_train_stream_pack: W1 fragment row a <-> unit (a>>2)*8 + hf*4 + (a&3)).
//   * hid and dH are BLOCKED [M/16][F/32][16 rows][32 units]: element (m, u) at ((m>>4) * F/32 + (u>>5)) * 512 + (m&15) * 32 + (u&31)
//     -- the bytes of 16 rows permuted among themselves.  A token fragment's half-chunk is then ONE contiguous KB (lane (row, g) at
//     row * 64 + g * 16: every store / mask load is 8 full lines), and a wave writes its 16 rows as one sequential 64-KB stream over
//     the tile.  Row-major [M][F], every store instruction touched 16 rows (4 KB apart), all CUs wrote the same 64-byte column of
//     different DRAM pages at any one time, and the launch was bound by that write pattern: same box, [196608, 2048], no dropout:
//     594 us row-major (624 staged through LDS for 64-byte row pieces, 571 with per-workgroup chunk order) -> 488 us blocked; 400 us
//     without the stores; data gradient 812 -> 493 us (profiles/r06_ffn_train_variants*.txt).  The weight-gradient kernel reads the
//     blocked operands through its LDS-DMA source addresses (wgrad.hip, WgradParams::a_blocked / b_blocked).
//   * stores and mask loads ride in the same in-order VMEM queue as the weight DMA: the counted vmcnt waits in front of the item barriers
//     account for them (two of the last three converting items' accesses are assumed younger than the awaited pieces: conservative by one).
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SLOT = 16384;            // one stream item: 16 fragments of 1 KB
constexpr int NSLOT = 8;
constexpr int STAGE = NSLOT * SLOT;    // 4 x 4 KB wave-private output staging (8 rows x 512 B)
constexpr int VECS = STAGE + 4 * 4096; // b2, gamma, beta
constexpr int B1L = VECS + 3 * 1024;   // b1, up to 2048 hidden units
constexpr int MAXF = 2048;
constexpr int SMEM = B1L + MAXF * 4;   // 158720
constexpr int NB = 8;                  // weight-fragment registers in rotation
constexpr int PD = 6;                  // fragment prefetch distance
constexpr int INFL = 4 * (NSLOT - 3);  // this wave's DMA pieces younger than the ones a barrier needs

// ---------------------------------------------------------------------------------------------------------------
// Stream packing, natural hidden-unit order.  Items of 16 fragments: W1h(0), { W1h(k), W2h(k-1) } k = 1 .. U-1, W2h(U-1)  (U = F / 32)
//   W1h(k) fragment p = s*2 + hf : lane (f, g) <- W1[32 k + (f>>2)*8 + hf*4 + (f&3)][32 s + 8 g + e]
//   W2h(k) fragment i            : lane (f, g) <- W2[(f>>2)*64 + 4 i + (f&3)][32 k + 8 g + e]
// 16-bit elements of either type (f16 forward operands, bf16 transposed copies for the backward).
__global__ void ffn_train_stream_pack_kernel(const unsigned short* __restrict__ W1, const unsigned short* __restrict__ W2,
                                             unsigned short* __restrict__ out, int F) {
    const int U = F / 32;
    const long total = (long)(2 * U) * (SLOT / 16);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int q = (int)(t >> 10), w = (int)(t & 1023);
        const int pfrag = w >> 6, l = w & 63, f = l & 15, g = l >> 4;
        bool is_w1;
        int k;
        if (q == 0) { is_w1 = true; k = 0; }
        else if (q == 2 * U - 1) { is_w1 = false; k = U - 1; }
        else if (q & 1) { is_w1 = true; k = (q + 1) >> 1; }
        else { is_w1 = false; k = (q >> 1) - 1; }
        const unsigned short* src;
        if (is_w1) {
            const int s_ = pfrag >> 1, hf = pfrag & 1;
            src = W1 + (size_t)(k * 32 + (f >> 2) * 8 + hf * 4 + (f & 3)) * 256 + s_ * 32 + g * 8;
        } else {
            const int n = (f >> 2) * 64 + pfrag * 4 + (f & 3);
            src = W2 + (size_t)n * F + k * 32 + g * 8;
        }
        *(uint4*)(out + t * 8) = *(const uint4*)src;
    }
}

DEV bool keep_of(const DropSpec d, unsigned rowbase, unsigned col) {       // drop_keep(d, row, col) with rowbase = row * golden
#ifdef FTS_NOHASH
    return (rowbase + col) != d.seed;
#endif
    return drop_mix((rowbase + col) ^ d.seed) >= (d.thresh24 << 8);
}

// ---------------------------------------------------------------------------------------------------------------
template <int TR, bool DROP, int NJ>
__global__ __launch_bounds__(256, 1)
void ffn_train_stream_kernel(const FfnTrainStreamParams p) {
    constexpr bool FWD = TR == 1;
    using T8 = std::conditional_t<FWD, f16x8, bf16x8>;       // MFMA operand type: f16 forward, bf16 gradients
    constexpr int TM = 64 * NJ, WM = 16 * NJ;
    // VMEM accesses a converting item issues next to its DMA pieces: one hidden-row store per token fragment (+ one mask load, TR 2)
    constexpr int EC = FWD ? NJ : 2 * NJ;
    constexpr int STEADY = INFL + 2 * EC;                     // see the header note
    constexpr int LOOSE = INFL + 8 * NJ;                      // first six items of a tile: at least the 8 NJ input-row loads are younger
    // last two items: the next tile's input rows (8 NJ loads) were requested in front of them, the first residual rows (16) in front of the last
    constexpr int LAST2 = STEADY + 8 * NJ < 63 ? STEADY + 8 * NJ : 63;
    constexpr int LAST1 = LAST2 + (FWD ? 16 : 0) < 63 ? LAST2 + (FWD ? 16 : 0) : 63;
    enum { VW_STEADY = 0, VW_LOOSE = 1, VW_LAST2 = 2, VW_LAST1 = 3 };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int U = p.F >> 5;
    const int S = 2 * U;
    const int ntiles = (p.M + TM - 1) / TM;

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, g = lane >> 4;
    int fo = g * 64;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, S * SLOT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (p.M - 1) * p.ldx * 2 + 512, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR32 = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.M * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO32 = __builtin_amdgcn_make_buffer_rsrc((void*)p.out32, 0, p.M * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO16 = __builtin_amdgcn_make_buffer_rsrc(p.out16, 0, FWD ? p.M * 512 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsXh = __builtin_amdgcn_make_buffer_rsrc(p.xhat16, 0, FWD ? p.M * 512 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsRs = __builtin_amdgcn_make_buffer_rsrc((void*)p.rstat, 0, FWD ? p.M * 4 : 0, 0x00020000);
    const unsigned hid_bytes = (unsigned)((p.M + 15) & ~15) * (unsigned)p.F * 2u;        // whole 16-row blocks
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(p.hid, 0, hid_bytes, 0x00020000);                   // TR 1: written; TR 2: the mask
    const __amdgpu_buffer_rsrc_t rsDH = __builtin_amdgcn_make_buffer_rsrc(p.dH, 0, FWD ? 0u : hid_bytes, 0x00020000);
    auto bload = [&](const __amdgpu_buffer_rsrc_t& r, int off) __attribute__((always_inline)) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); };
    int dvo = lane * 16 + wave * 4096;
    int nxt = 0;
    int slot = 0;

    auto dma_piece = [&](int sd, auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_char*)(smem + sd * SLOT + wave * 4096 + i * 1024), 16, dvo,
                                                 nxt * SLOT + i * 1024, 0, 0);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) { nxt = nxt + 1 == S ? 0 : nxt + 1; };

    sfor<NSLOT - 1>([&](auto IT) __attribute__((always_inline)) {
        sfor<4>([&](auto I) __attribute__((always_inline)) { dma_piece(decltype(IT)::value, I); });
        dma_advance();
    });

    float* vecs = (float*)(smem + VECS);                  // [3][256]: b2, gamma, beta
    float* b1l = (float*)(smem + B1L);
    if constexpr (FWD) {
        vecs[0 * 256 + tid] = p.b2[tid];
        vecs[1 * 256 + tid] = p.gamma[tid];
        vecs[2 * 256 + tid] = p.beta[tid];
        for (int i = tid; i < p.F; i += 256) b1l[i] = p.b1[i];
    }
    auto vec4 = [&](int which, int i) __attribute__((always_inline)) { return *(const f32x4*)(vecs + which * 256 + fo + i * 4); };

    const char* wl = smem + lane * 16;
    T8 wf[NB];
    f32x4 acc[16][NJ];
    T8 xf[8][NJ];
    f32x4 h[2][NJ];
    T8 hbA[NJ], hbB[NJ];
    f32x4 bcv[2];
    f16x8 mk[2][TR == 2 ? NJ : 1];                        // TR 2: saved activations of the chunk being converted / the one after it
    unsigned hoff[NJ];                                    // byte offset of this lane's 16 bytes of half-chunk 0 in hid / dH (blocked), per token fragment
    unsigned rowh[FWD ? NJ : 1];                          // row * golden ratio (dropout hash), per token fragment

    auto row_of = [&](int tile, int j) __attribute__((always_inline)) { return tile * TM + wave * WM + j * 16 + frow; };
    auto load_in_frags = [&](int tile, auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        const int off = row_of(tile, j) * (p.ldx * 2) + g * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) xf[s][j] = __builtin_bit_cast(T8, bload(rsA, off + s * 64));
    };
    auto load_mask = [&](auto PAR, auto J, int kc) __attribute__((always_inline)) {
        if constexpr (TR == 2) {
            constexpr int j = decltype(J)::value;
            mk[decltype(PAR)::value][j] = __builtin_bit_cast(f16x8, bload(rsH, (int)(hoff[j] + (unsigned)kc * 1024u)));
        }
    };

    __builtin_amdgcn_s_waitcnt(0x0070 | ((4 * (NSLOT - 2)) & 15) | (((4 * (NSLOT - 2)) >> 4) << 14));   // item 0 of this wave has landed; lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    if (blockIdx.x < ntiles) sfor<NJ>([&](auto J) __attribute__((always_inline)) { load_in_frags(blockIdx.x, J); });

    // activation of the half-chunk held in h -> second GEMM's operand; part = hf * NJ + j.  After a token fragment's second part its 16
    // bytes leave for hid (TR 1) / dH (TR 2), and (TR 2) the mask registers it freed take the rows of the chunk after next.
    auto conv_part = [&](auto PART, auto PAR, T8 (&hbo)[NJ], int kc) __attribute__((always_inline)) {
        constexpr int hf = decltype(PART)::value / NJ, j = decltype(PART)::value % NJ, par = decltype(PAR)::value;
        if constexpr (FWD) {
            const unsigned cb = (unsigned)kc * 32u + (unsigned)g * 8u + hf * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v;
                if constexpr (DROP) {
                    v = __builtin_amdgcn_fmed3f(h[hf][j][r] * p.drop1.scale, 0.f, 65504.f);
                    v = keep_of(p.drop1, rowh[j], cb + r) ? v : 0.f;
                } else {
                    v = __builtin_amdgcn_fmed3f(h[hf][j][r], 0.f, 65504.f);
                }
                hbo[j][hf * 4 + r] = (_Float16)v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = mk[par][j][hf * 4 + r] != (_Float16)0 ? h[hf][j][r] * p.drop1.scale : 0.f;
                hbo[j][hf * 4 + r] = (__bf16)v;
            }
        }
        if constexpr (hf == 1) {
#ifndef FTS_NOSTORE
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hbo[j]), FWD ? rsH : rsDH, (int)(hoff[j] + (unsigned)kc * 1024u), 0, 0);
#endif
#ifndef FTS_NOMASK
            load_mask(PAR, IC<j>{}, kc + 2);
#endif
        }
    };

    // One stream item = 16 fragments, NJ MFMAs each.  KIND 1: h = W1h(k) x xf (k-step s = fragment >> 1);  2: acc += W2h x hb, and (CONV) the
    // conversion of h (half-chunk kc) into hbo rides on the fragments.  `vw` = VMEM accesses of this wave that may stay in flight at the barrier.
    auto step = [&](auto KIND, auto CONVc, auto PARc, auto COLDc, auto PFNc, int vw, int k, int kc, T8 (&hb)[NJ],
                    T8 (&hbo)[NJ]) __attribute__((always_inline)) {
        constexpr int kind = decltype(KIND)::value;
        constexpr bool conv = decltype(CONVc)::value;
        constexpr bool cold = decltype(COLDc)::value;
        constexpr bool pfn = decltype(PFNc)::value;
        // vmcnt is a 6-bit field split over bits 3:0 and 15:14; the count is wave-uniform
#ifdef FTS_WAIT63
        vw = 99;
#endif
        switch (vw) {
            case 99: __builtin_amdgcn_s_waitcnt(0x0F70 | (63 & 15) | ((63 >> 4) << 14)); break;      // (timing study only: not a valid wait)
            case VW_LOOSE: __builtin_amdgcn_s_waitcnt(0x0F70 | (LOOSE & 15) | ((LOOSE >> 4) << 14)); break;
            case VW_LAST2: __builtin_amdgcn_s_waitcnt(0x0F70 | (LAST2 & 15) | ((LAST2 >> 4) << 14)); break;
            case VW_LAST1: __builtin_amdgcn_s_waitcnt(0x0F70 | (LAST1 & 15) | ((LAST1 >> 4) << 14)); break;
            default: __builtin_amdgcn_s_waitcnt(0x0F70 | (STEADY & 15) | ((STEADY >> 4) << 14)); break;
        }
        __builtin_amdgcn_s_barrier();
        const char* wc = wl + slot * SLOT;
        const char* wn = wl + ((slot + 1) & (NSLOT - 1)) * SLOT;
        const int sd = (slot + NSLOT - 1) & (NSLOT - 1);
        if constexpr (cold) {
            sfor<PD>([&](auto Q) __attribute__((always_inline)) {
                wf[decltype(Q)::value % NB] = *(const T8*)(wc + decltype(Q)::value * 1024);
            });
        }
        if constexpr (kind == 1) {
            if constexpr (FWD) {
                bcv[0] = *(const f32x4*)(b1l + k * 32 + g * 8);
                bcv[1] = *(const f32x4*)(b1l + k * 32 + g * 8 + 4);
            } else {
                bcv[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                bcv[1] = bcv[0];
            }
        }
        sfor<8>([&](auto P2) __attribute__((always_inline)) {
            sfor<2>([&](auto PH) __attribute__((always_inline)) {
                constexpr int pi = decltype(P2)::value * 2 + decltype(PH)::value;
                const T8 w = wf[pi % NB];
                if constexpr (kind == 1) {
                    constexpr int s_ = pi >> 1, hf = pi & 1;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if constexpr (FWD) {
                            if constexpr (s_ == 0)
                                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(h[hf][j]) : "v"(w), "v"(xf[s_][j]), "v"(bcv[hf]));
                            else
                                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(h[hf][j]) : "v"(w), "v"(xf[s_][j]));
                        } else {
                            if constexpr (s_ == 0)
                                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(h[hf][j]) : "v"(w), "v"(xf[s_][j]), "v"(bcv[hf]));
                            else
                                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(h[hf][j]) : "v"(w), "v"(xf[s_][j]));
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if constexpr (FWD) acc[pi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, hb[j], acc[pi][j], 0, 0, 0);
                        else acc[pi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, hb[j], acc[pi][j], 0, 0, 0);
                    }
                }
                if constexpr (pi + PD < 16) wf[(pi + PD) % NB] = *(const T8*)(wc + (pi + PD) * 1024);
                else if constexpr (pfn) wf[(pi + PD) % NB] = *(const T8*)(wn + (pi + PD - 16) * 1024);
                if constexpr (pi < 4) dma_piece(sd, IC<pi>{});
                // conversion of the half-chunk held in h (2 NJ parts) on fragments 4, 6, ...: behind the item's DMA pieces
                if constexpr (kind == 2 && conv && pi >= 4 && pi < 4 + 4 * NJ && !(pi & 1)) conv_part(IC<(pi - 4) / 2>{}, PARc, hbo, kc);

            });
            __builtin_amdgcn_sched_barrier(0);
        });
        dma_advance();
        slot = (slot + 1) & (NSLOT - 1);
    };
    auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i][j]));
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        auto relaunder = [&]() __attribute__((always_inline)) {
            asm volatile("" : "+v"(tid));
            lane = tid & 63; frow = lane & 15; g = lane >> 4; fo = g * 64;
            dvo = lane * 16 + wave * 4096;
            wl = smem + lane * 16;
        };
        relaunder();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned row = (unsigned)row_of(tile, j);
            hoff[j] = (unsigned)(tile * TM + wave * WM + j * 16) * ((unsigned)p.F * 2u) + (unsigned)frow * 64u + (unsigned)g * 16u;
            if constexpr (FWD) rowh[j] = row * 0x9E3779B1u;
        }
        // ---- accumulators: b2 (TR 1: the residual joins behind the dropout, in the epilogue) / the gradient stream's rows (TR 2)
        if constexpr (FWD) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 b4 = vec4(0, i);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = b4;
            }
        } else {
            sfor<NJ>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const int off = row_of(tile, j) * 1024 + fo * 4;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i][j] = __builtin_bit_cast(f32x4, bload(rsR32, off + i * 16));
            });
            sfor<NJ>([&](auto J) __attribute__((always_inline)) { load_mask(IC<0>{}, J, 0); load_mask(IC<1>{}, J, 1); });
        }

        // (TR 1) residual rows of one token fragment: fragment 0's travel under the last item, fragment j + 1's under fragment j's LayerNorm
        f32x4 t4[FWD ? 16 : 1];
        auto load_res = [&](auto J) __attribute__((always_inline)) {
            if constexpr (FWD) {
                constexpr int j = decltype(J)::value;
                const int roff = row_of(tile, j) * 1024 + fo * 4;
#pragma unroll
                for (int i = 0; i < 16; ++i) t4[i] = __builtin_bit_cast(f32x4, bload(rsR32, roff + i * 16));
            }
        };
        // ---- item sequence: W1h(0) | W1h(1) | { W2h(k-2) + conv h(k-1), W1h(k), W2h(k-1) + conv h(k), W1h(k+1) } | W2h(U-2) + conv | W2h(U-1)
        {
            using T = std::true_type;
            using Fa = std::false_type;
            pin_acc();
            step(IC<1>{}, Fa{}, IC<0>{}, T{}, Fa{}, VW_LOOSE, 0, 0, hbA, hbB);
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // the hand-written MFMAs' results are read by VALU instructions next
            sfor<2 * NJ>([&](auto Q) __attribute__((always_inline)) { conv_part(Q, IC<0>{}, hbA, 0); });
            step(IC<1>{}, Fa{}, IC<0>{}, T{}, T{}, VW_LOOSE, 1, 0, hbA, hbB);
            int n = 2;                                               // items consumed so far in this tile
            for (int k = 2; k < U; k += 2) {                         // U is even
                step(IC<2>{}, T{}, IC<1>{}, Fa{}, T{}, n < 6 ? VW_LOOSE : VW_STEADY, 0, k - 1, hbA, hbB);
                step(IC<1>{}, Fa{}, IC<0>{}, Fa{}, T{}, n + 1 < 6 ? VW_LOOSE : VW_STEADY, k, 0, hbA, hbB);
                step(IC<2>{}, T{}, IC<0>{}, Fa{}, T{}, n + 2 < 6 ? VW_LOOSE : VW_STEADY, 0, k, hbB, hbA);
                step(IC<1>{}, Fa{}, IC<0>{}, Fa{}, T{}, n + 3 < 6 ? VW_LOOSE : VW_STEADY, k + 1, 0, hbA, hbB);
                n += 4;
            }
            // x is dead: the next tile's input rows travel under the last two items and the epilogue (rows beyond M read as zeros)
            sfor<NJ>([&](auto J) __attribute__((always_inline)) { load_in_frags(tile + (int)gridDim.x, J); });
            step(IC<2>{}, T{}, IC<1>{}, Fa{}, T{}, n < 6 ? VW_LOOSE : VW_LAST2, 0, U - 1, hbA, hbB);
            load_res(IC<0>{});
            step(IC<2>{}, Fa{}, IC<0>{}, Fa{}, Fa{}, n + 1 < 6 ? VW_LOOSE : VW_LAST1, 0, 0, hbB, hbA);
            pin_acc();
        }

        // ---- epilogue, one token fragment at a time; rows leave through the wave's 4-KB staging tile as whole rows
        relaunder();
        char* st = smem + STAGE + wave * 4096;
        // 16 rows x 256 f32 of `val(i, q)` (feature fo + 4 i + q of this lane's token row): four staged passes
        auto store_rows32 = [&](int rbase, auto val) __attribute__((always_inline)) {
#pragma unroll
            for (int fh = 0; fh < 2; ++fh)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if ((frow >> 3) == half) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            *(f32x4*)(st + (frow & 7) * 512 + (((g * 8 + e) ^ (frow & 7)) << 4)) =
                                f32x4{val(fh * 8 + e, 0), val(fh * 8 + e, 1), val(fh * 8 + e, 2), val(fh * 8 + e, 3)};
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int rr = 2 * q4 + (lane >> 5), cc = lane & 31;
                        const f32x4 v = *(const f32x4*)(st + rr * 512 + ((cc ^ rr) << 4));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO32,
                                                               (rbase + half * 8 + rr) * 1024 + (cc >> 3) * 256 + fh * 128 + (cc & 7) * 16, 0, 0);
                    }
                    wave_lds_sync();
                }
        };
        auto store_rows16 = [&](int rbase, const __amdgpu_buffer_rsrc_t& rsrc, const f16x8 (&o)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if ((frow >> 3) == half) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) *(f16x8*)(st + (frow & 7) * 512 + (((g * 8 + e) ^ (frow & 7)) << 4)) = o[e];
                }
                wave_lds_sync();
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int rr = 2 * q4 + (lane >> 5), cc = lane & 31;
                    const f16x8 v = *(const f16x8*)(st + rr * 512 + ((cc ^ rr) << 4));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (rbase + half * 8 + rr) * 512 + cc * 16, 0, 0);
                }
                wave_lds_sync();
            }
        };
        sfor<NJ>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const int rbase = tile * TM + wave * WM + j * 16;
            if constexpr (FWD) {
                // y = drop2(acc) * alpha + res, in place
                const float sc = p.drop2.scale * p.alpha;
                if (p.drop2.thresh24) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool keep = keep_of(p.drop2, rowh[j], (unsigned)(fo + i * 4 + q));
                            acc[i][j][q] = (keep ? acc[i][j][q] * sc : 0.f) + t4[i][q];
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i][j] = acc[i][j] * sc + t4[i];
                }
                if constexpr (j + 1 < NJ) load_res(IC<j + 1>{});
                f32x2 sm = f32x2{0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) sm += f32x2{acc[i][j][0], acc[i][j][1]} + f32x2{acc[i][j][2], acc[i][j][3]};
                const float mean = wave_g_allreduce_add(sm[0] + sm[1]) * (1.0f / 256);
                f32x2 sq2 = f32x2{0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x2 d0 = f32x2{acc[i][j][0] - mean, acc[i][j][1] - mean}, d1 = f32x2{acc[i][j][2] - mean, acc[i][j][3] - mean};
                    sq2 = d1 * d1 + (d0 * d0 + sq2);
                }
                const float rstd = 1.0f / __builtin_sqrtf(wave_g_allreduce_add(sq2[0] + sq2[1]) * (1.0f / 256) + p.eps);
                if (g == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rstd), rsRs, (rbase + frow) * 4, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                f16x8 o[8];
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[i >> 1][(i & 1) * 4 + q] = to_f16_sat((acc[i][j][q] - mean) * rstd);
                store_rows16(rbase, rsXh, o);
                // the accumulators become the LayerNorm output
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 g4 = vec4(1, i), b4 = vec4(2, i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float y = (acc[i][j][q] - mean) * rstd * g4[q] + b4[q];
                        acc[i][j][q] = y;
                        o[i >> 1][(i & 1) * 4 + q] = to_f16_sat(y);
                    }
                }
                store_rows16(rbase, rsO16, o);
            }
            store_rows32(rbase, [&](int i, int q) __attribute__((always_inline)) { return acc[i][j][q]; });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the workgroup
}

template <int TR, bool DROP, int NJ>
int launch_nj(const FfnTrainStreamParams& p, int ncu, hipStream_t stream) {
    static EendOncePerDevice attr_once;
    auto kern = ffn_train_stream_kernel<TR, DROP, NJ>;
    if (!eend_set_dynamic_lds(attr_once, (const void*)kern, SMEM)) return EEND_ELAUNCH;
    const int ntiles = (p.M + 64 * NJ - 1) / (64 * NJ);
    hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), SMEM, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

template <int TR, bool DROP>
int launch(const FfnTrainStreamParams& p, hipStream_t stream) {
    const int ncu = eend_cu_count();
    const long t3 = (p.M + 191) / 192, t2 = (p.M + 127) / 128;
    const long c3 = ((t3 + ncu - 1) / ncu) * (3 * 10 + 9), c2 = ((t2 + ncu - 1) / ncu) * (2 * 10 + 9);     // rounds x (rows + fixed part), as ffn_stream.hip
    return c2 < c3 ? launch_nj<TR, DROP, 2>(p, ncu, stream) : launch_nj<TR, DROP, 3>(p, ncu, stream);
}

}  // namespace

long eend_ffn_train_stream_nelems(int F) { return (long)(2 * (F / 32)) * (SLOT / 2); }

int eend_launch_ffn_train_stream_pack(const void* W1, const void* W2, void* out, int F, hipStream_t stream) {
    if (!W1 || !W2 || !out || F < 64 || (F % 64) != 0 || F > MAXF || (((size_t)W1 | (size_t)W2 | (size_t)out) & 15)) return EEND_EINVAL;
    const long total = eend_ffn_train_stream_nelems(F) / 8;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(ffn_train_stream_pack_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, stream, (const unsigned short*)W1,
                       (const unsigned short*)W2, (unsigned short*)out, F);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// rows one launch can take: 32-bit buffer offsets into hid / dH ([M][F] 16-bit) and into the f32 rows, with the input-row prefetch running
// one grid of tiles (< 65536 rows) past the end
bool eend_ffn_train_stream_fits(int M, int F, int ldx) {
    if (M <= 0 || F < 64 || (F % 64) != 0 || F > MAXF || ldx < 256 || (ldx & 7)) return false;
    return ((long)M + 65536) * F * 2 < (1L << 32) && ((long)M + 65536) * 1024 < (1L << 31) && ((long)M + 65536) * ldx * 2 < (1L << 31);
}

int eend_launch_ffn_train_stream(const FfnTrainStreamParams& p, int tr, hipStream_t stream) {
    if (!eend_ffn_train_stream_fits(p.M, p.F, p.ldx) || !p.X || !p.wstream || !p.res || !p.out32 || !p.hid) return EEND_EINVAL;
    if ((((size_t)p.X | (size_t)p.wstream | (size_t)p.res | (size_t)p.out32 | (size_t)p.hid | (size_t)p.dH | (size_t)p.out16 | (size_t)p.xhat16) & 15))
        return EEND_EINVAL;
    if (tr == 1) {
        if (!p.b1 || !p.b2 || !p.gamma || !p.beta || !p.out16 || !p.xhat16 || !p.rstat || !(p.alpha != 0.f)) return EEND_EINVAL;
        if (p.drop1.thresh24) return launch<1, true>(p, stream);
        return launch<1, false>(p, stream);
    }
    if (tr == 2) {
        if (!p.dH) return EEND_EINVAL;
        return launch<2, false>(p, stream);
    }
    return EEND_EINVAL;
}
