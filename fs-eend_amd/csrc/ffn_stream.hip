// Row-local layer tail on a packed weight stream (round 4):
//     x   = LayerNorm1(A Wo^T + bo + res)                        (MODE 1; attention out-projection, residual, norm1)
//     out = LayerNorm2((act(x W1^T + b1) W2^T + b2) * alpha + x)  (FFN, residual, norm2)
// or, MODE 0, just the FFN on a given X with a given residual.  Same operator as ffn.hip (which it replaces on the hot
// path; reference sites: nn.TransformerEncoderLayer of FS model :147, _sa_block/_ff_block + norm* of
// merge_tfm_encoder.py:356-399, FeedForwardModule of LS conformer/feed_forward.py:47-57) with a different decomposition:
//
//   * one 256-thread workgroup per CU, ONE wave per SIMD, so a wave owns the whole 512-entry register file.  A wave owns
//     48 token rows (3 MFMA column fragments) end to end: its out-projection / GEMM2 accumulators for all 256 output
//     features (192 registers), its X fragments (96) and its hidden activations never leave its registers -- the MFMA
//     output layout of GEMM1 (lane = token, 4 consecutive hidden units per 16-lane group) IS the B-operand layout of
//     GEMM2 once the contraction index of W2 is permuted accordingly, and the same holds between LayerNorm1's output and
//     GEMM1.  No hidden-activation tile in LDS, no LDS round trip of x, LayerNorm statistics are wave-local
//     (two xor shuffles), no workgroup reduction.
//   * the three weight matrices are packed once per parameter version (eend_ffn_stream_pack_f16) into the exact
//     sequence of 1-KB MFMA A-fragments the kernel consumes: 32-KB items [Wo k-chunk 0..3], W1(0), {W1(c), W2(c-1)}...,
//     W2(n-1).  The stream flows by LDS-DMA through a 4-slot ring (128 KB), continuously across tiles; every fragment
//     read is a lane-linear, conflict-free ds_read_b128 at base + immediate.  One barrier per 32-KB item (96 MFMAs per
//     wave); a wave waits (counted vmcnt) for its own pieces of the NEXT item before the barrier, so fragment
//     prefetch runs across item boundaries.
//   * LDS fragment bytes per row drop to 2/3 of ffn.hip's (each weight fragment feeds 3 MFMAs instead of 2 / 4 with
//     the Hs exchange on top), which was its measured bound.
#include "common.h"
#include "kernels.h"
#include <type_traits>
#include <utility>
#include <cstdlib>

#include <atomic>

namespace {
// test hooks (eend_debug_ffn_stream_set): force the tile size (2 / 3 fragments; 0 = the cost model) and cap the rows of one launch
// (0 = no cap) so that the tile-size agreement and the multi-launch path can be exercised at small sizes
std::atomic<int> g_debug_nj{0};
std::atomic<long> g_debug_row_cap{0};
}  // namespace
long eend_ffn_stream_debug_row_cap() { return g_debug_row_cap.load(std::memory_order_relaxed); }
extern "C" int eend_debug_ffn_stream_set(int tile_fragments, long max_rows_per_launch) {
    if ((tile_fragments != 0 && tile_fragments != 2 && tile_fragments != 3) || max_rows_per_launch < 0) return EEND_EINVAL;
    g_debug_nj.store(tile_fragments, std::memory_order_relaxed);
    g_debug_row_cap.store(max_rows_per_launch, std::memory_order_relaxed);
    return EEND_OK;
}

namespace {

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }
template <int V> using IC = std::integral_constant<int, V>;

typedef __attribute__((address_space(3))) char lds_char;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// rows per wave = 16 NJ (NJ token fragments), rows per tile (workgroup) = 64 NJ; NJ = 3 for large M, 2 when that fills more CUs
constexpr int SLOT = 16384;        // one stream item: 16 fragments of 1 KB
constexpr int NSLOT = 8;
constexpr int STAGE = NSLOT * SLOT;    // 4 x 4 KB wave-private output staging (8 rows x 512 B)
constexpr int VECS = STAGE + 4 * 4096; // 6 per-feature f32 vectors
constexpr int B1L = VECS + 6 * 1024;   // b1, up to 2048 hidden units
constexpr int MAXF = 2048;
constexpr int DUMP = B1L + MAXF * 4;   // 256-byte dump area of the L2-touch loads
constexpr int SMEM = DUMP + 256;       // 162048
constexpr int NB = 8;     // weight-fragment registers in rotation (divides 32)
constexpr int PD = 6;     // fragment prefetch distance (items)
constexpr int INFL = 4 * (NSLOT - 3);   // this wave's DMA pieces younger than the ones a barrier needs (5 items x 4 pieces)

// ---------------------------------------------------------------------------------------------------------------
// weight stream packing: one thread per 16 bytes of the stream.  Items of 16 fragments (16 KB), in consumption order:
//   Wo (kc = 0..3, sl = 0..1)  fragment i           : Wo[n(i,f)][kc*64 + sl*32 + g*8 + e]
//   W1h(0), then { W1h(k), W2h(k-1) } for k = 1 .. U-1, then W2h(U-1)        (U = F/32 half-chunks of 32 hidden units)
//       W1h(k) fragment p = s*2 + hf : W1[k*32 + hf*16 + f][kcol(s,g) + e]
//       W2h(k) fragment i           : W2[n(i,f)][k*32 + (e>>2)*16 + g*4 + (e&3)]
//   with lane = (f = l & 15, g = l >> 4), n(i,f) = (f>>2)*64 + i*4 + (f&3), kcol = g*64 + 8s (after LayerNorm1's register
//   layout) or s*32 + g*8 (X read from memory).
__global__ void ffn_stream_pack_kernel(const _Float16* __restrict__ Wo, const _Float16* __restrict__ Wo_lo, const _Float16* __restrict__ W1,
                                       const _Float16* __restrict__ W2, _Float16* __restrict__ out, int F, int k_permuted) {
    const int U = F / 32;
    const int nWo = Wo ? (Wo_lo ? 16 : 8) : 0;             // with Wo_lo: the eight items of f16(Wo - f16(Wo)) follow those of Wo
    const long total = (long)(nWo + 2 * U) * (SLOT / 16);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int item = (int)(t >> 10), w = (int)(t & 1023);
        const int pfrag = w >> 6, l = w & 63, f = l & 15, g = l >> 4;
        _Float16 v[8];
        if (item < nWo) {
            const int kc = (item & 7) >> 1, sl = item & 1, i = pfrag;
            const int n = (f >> 2) * 64 + i * 4 + (f & 3);
            const _Float16* src = (item < 8 ? Wo : Wo_lo) + (size_t)n * 256 + kc * 64 + sl * 32 + g * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = src[e];
        } else {
            const int q = item - nWo;                    // 0: W1h(0); 2k-1: W1h(k); 2k: W2h(k-1); 2U-1: W2h(U-1)
            bool is_w1;
            int k;
            if (q == 0) { is_w1 = true; k = 0; }
            else if (q == 2 * U - 1) { is_w1 = false; k = U - 1; }
            else if (q & 1) { is_w1 = true; k = (q + 1) >> 1; }
            else { is_w1 = false; k = (q >> 1) - 1; }
            if (is_w1) {
                const int s_ = pfrag >> 1, hf = pfrag & 1;
                const int k0 = k_permuted ? g * 64 + 8 * s_ : s_ * 32 + g * 8;
                const _Float16* src = W1 + (size_t)(k * 32 + hf * 16 + f) * 256 + k0;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = src[e];
            } else {
                const int i = pfrag;
                const int n = (f >> 2) * 64 + i * 4 + (f & 3);
                const _Float16* src = W2 + (size_t)n * F + k * 32 + g * 4;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = src[(e >> 2) * 16 + (e & 3)];
            }
        }
        _Float16* dst = out + t * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = v[e];
    }
}

// Perf-study build (-DEEND_FS_TRACE, tools/ffn_stream_trace.py): s_memtime stamps of the tile phases of lane 0 of wave 0 of
// every workgroup, read back through eend_debug_fs_trace; never defined in the shipped library.
#ifdef EEND_FS_TRACE
__device__ unsigned long long g_fs_trace[256 * 8 * 10];
#define FS_STAMP(k) do { ts[k] = __builtin_amdgcn_s_memtime(); } while (0)       /* kept in scalar registers, written at the tile's end */
#else
#define FS_STAMP(k) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// LO (MODE 1 on f32 residual rows: the LS-EEND decoder layer tail): the out-projection is followed by a second product with the f16
// remainder of its weight on the same input fragments (eight more stream items), and the rows also leave as an f16 hi / lo pair
// (out16lo = f16(y - f16(y)): the second operand of the next retention's query path).
template <int MODE, int ACT, int EPI, bool RES16, int NJ, bool LO = false>
__global__ __launch_bounds__(256, 1)
void ffn_stream_kernel(const FfnStreamParams p) {
    constexpr bool PRE = MODE == 1;
    static_assert(!LO || (PRE && !RES16), "LO: out-projection form on f32 residual rows");
    constexpr int TM = 64 * NJ, WM = 16 * NJ;
    // VMEM operations of a wave that are certainly younger than the DMA pieces the first six barriers after an epilogue wait for:
    // the INFL pieces in between plus the epilogue's own f16 row stores (8 per token fragment) and, where they are issued in or
    // behind the epilogue, the next tile's input-row loads (8 per fragment)
    constexpr int LOOSE = INFL + 8 * NJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int U = p.F >> 5;                               // half-chunks of 32 hidden units
    const int S = (PRE ? (LO ? 16 : 8) : 0) + 2 * U;      // stream items per tile
    const int ntiles = (p.M + TM - 1) / TM;

    // The thread index is laundered per tile so that everything derived from it (LDS addresses, row pointers, DMA offsets)
    // is recomputed there instead of being hoisted out of the tile loop and kept live -- or spilled -- across it.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, g = lane >> 4;
    int fo = g * 64;                                      // this lane's features: fo + i*4 + r

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, S * SLOT, 0x00020000);
    // Rows move through raw buffer resources: 32-bit offsets instead of 64-bit pointers (fewer address registers), and the
    // hardware drops / zero-fills accesses beyond the last row -- no clamps, no per-row branches in the epilogue.
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (p.M - 1) * p.lda * 2 + 512, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR16 = __builtin_amdgcn_make_buffer_rsrc((void*)p.res16, 0, p.res16 ? p.M * 512 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR32 = __builtin_amdgcn_make_buffer_rsrc((void*)p.res32, 0, p.res32 ? p.M * 1024 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO16 = __builtin_amdgcn_make_buffer_rsrc(p.out16, 0, p.M * 512, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO32 = __builtin_amdgcn_make_buffer_rsrc((void*)p.out32, 0, p.out32 ? p.M * 1024 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO16L = __builtin_amdgcn_make_buffer_rsrc(p.out16lo, 0, LO && p.out16lo ? p.M * 512 : 0, 0x00020000);
    auto bload = [&](const __amdgpu_buffer_rsrc_t& r, int off) __attribute__((always_inline)) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); };
    int dvo = lane * 16 + wave * 4096;                    // this wave moves pieces wave*4 .. wave*4+3 of every item
    int nxt = 0;                                          // next stream item to request (0 .. S-1)
    int slot = 0;                                         // ring slot of the item being consumed
#ifdef EEND_FS_TRACE
    int tix = -1;
#endif

    // piece i of stream item nxt -> ring slot sd
    auto dma_piece = [&](int sd, auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_char*)(smem + sd * SLOT + wave * 4096 + i * 1024), 16, dvo,
                                                 nxt * SLOT + i * 1024, 0, 0);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) { nxt = nxt + 1 == S ? 0 : nxt + 1; };

    // prime the ring: items 0 .. NSLOT-2
    sfor<NSLOT - 1>([&](auto IT) __attribute__((always_inline)) {
        sfor<4>([&](auto I) __attribute__((always_inline)) { dma_piece(decltype(IT)::value, I); });
        dma_advance();
    });

    // per-feature vectors and b1 live in LDS: the only VMEM traffic of the FFN loop is the weight stream
    float* vecs = (float*)(smem + VECS);                  // [6][256]: bo, g1, be1, b2, gamma, beta
    float* b1l = (float*)(smem + B1L);
    {
        vecs[0 * 256 + tid] = PRE ? p.bo[tid] : 0.f;
        vecs[1 * 256 + tid] = PRE ? p.g1[tid] : 0.f;
        vecs[2 * 256 + tid] = PRE ? p.be1[tid] : 0.f;
        vecs[3 * 256 + tid] = p.b2[tid];
        vecs[4 * 256 + tid] = p.gamma[tid];
        vecs[5 * 256 + tid] = p.beta[tid];
        for (int i = tid; i < p.F; i += 256) b1l[i] = p.b1[i];
    }
    auto vec4 = [&](int which, int i) __attribute__((always_inline)) { return *(const f32x4*)(vecs + which * 256 + fo + i * 4); };

    const char* wl = smem + lane * 16;                    // fragment p of slot s: wl + s*SLOT + p*1024 (re-derived per tile)
    f16x8 wf[NB];
    f32x4 acc[16][NJ];
    f16x8 xf[8][NJ];                                      // B operand of the first GEMM of the tile (A rows) / of GEMM1 (x)
    f32x4 h[2][NJ];
    f16x8 hbA[NJ], hbB[NJ];                               // hidden activations (GEMM2 B operand), ping-pong
    f32x4 bcv[2];                                         // b1 of the half-chunk held in h

    auto row_of = [&](int tile, int j) __attribute__((always_inline)) { return tile * TM + wave * WM + j * 16 + frow; };
    auto load_in_frags = [&](int tile, auto J) __attribute__((always_inline)) {      // xf[s][j] = In[row][s*32 + g*8 ..]
        constexpr int j = decltype(J)::value;
        const int off = row_of(tile, j) * (p.lda * 2) + g * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) xf[s][j] = __builtin_bit_cast(f16x8, bload(rsA, off + s * 64));
    };

    // first fragments of slot 0 (legal once every wave's pieces of item 0 have landed); vectors visible
    __builtin_amdgcn_s_waitcnt(0x0070 | ((4 * (NSLOT - 2)) & 15) | (((4 * (NSLOT - 2)) >> 4) << 14));   // item 0 of this wave has landed; lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    // RES16: the residual rows of token fragment 0 travel with the tile's input rows, those of fragments 1 and 2 are requested
    // once the first six k-steps' input fragments are dead (before the 7th out-projection item)
    f16x8 r8[RES16 && PRE ? NJ : 1][8];
    auto load_res16 = [&](int tile, auto J) __attribute__((always_inline)) {
        if constexpr (RES16 && PRE) {
            constexpr int j = decltype(J)::value;
            const int off = row_of(tile, j) * 512 + fo * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) r8[j][e] = __builtin_bit_cast(f16x8, bload(rsR16, off + e * 16));
        }
    };
    if (blockIdx.x < ntiles) { sfor<NJ>([&](auto J) __attribute__((always_inline)) { load_in_frags(blockIdx.x, J); }); load_res16(blockIdx.x, IC<0>{}); }

    auto act_cvt = [&](float v) __attribute__((always_inline)) -> _Float16 {
        if (ACT == 1) return (_Float16)__builtin_amdgcn_fmed3f(v, 0.f, 65504.f);       // ReLU + saturation in one instruction
        v = v / (1.0f + __expf(-v));
        return (_Float16)__builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f);
    };
    auto conv_part = [&](auto PART, f16x8 (&hbo)[NJ]) __attribute__((always_inline)) {          // part = hf * NJ + j  (2 NJ parts)
        constexpr int hf = decltype(PART)::value / NJ, j = decltype(PART)::value % NJ;
#pragma unroll
        for (int r = 0; r < 4; ++r) hbo[j][hf * 4 + r] = act_cvt(h[hf][j][r]);
    };

    // One stream item = 16 fragments, 3 MFMAs each.  KIND 0: acc += Wo(kc, sl) x xf;  1: h = W1h(k) x xf;  2: acc += W2h x hb, and
    // (CONV) the activation of h into hbn rides on the fragments.  The item requested meanwhile goes to slot-1 (the slot
    // every wave finished before this barrier); the last PD fragments' places in the register rotation are refilled from
    // slot+1 (PFN).  Before the barrier a wave waits for its own pieces of the NEXT item: INFL younger pieces (+ vwx loads
    // issued since) may stay in flight -- or, `loose`, whatever an epilogue issued behind them.
    auto step = [&](auto KIND, auto SRCc, auto CONVc, auto COLDc, auto PFNc, auto VWXc, bool loose, int k, f16x8 (&hb)[NJ],
                    f16x8 (&hbo)[NJ]) __attribute__((always_inline)) {
        constexpr int kind = decltype(KIND)::value, src = decltype(SRCc)::value, vw = INFL + decltype(VWXc)::value;
        constexpr bool conv = decltype(CONVc)::value;
        constexpr bool cold = decltype(COLDc)::value;    // the previous item did not request this item's first fragments
        constexpr bool pfn = decltype(PFNc)::value;      // request the next item's first fragments (not in front of a VALU phase)
        if (loose) __builtin_amdgcn_s_waitcnt(0x0F70 | (LOOSE & 15) | ((LOOSE >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70 | (vw & 15) | ((vw >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        const char* wc = wl + slot * SLOT;
        const char* wn = wl + ((slot + 1) & (NSLOT - 1)) * SLOT;
        const int sd = (slot + NSLOT - 1) & (NSLOT - 1);
        if constexpr (cold) {
            sfor<PD>([&](auto Q) __attribute__((always_inline)) {
                wf[decltype(Q)::value % NB] = *(const f16x8*)(wc + decltype(Q)::value * 1024);
            });
        }
        if constexpr (kind == 1) {
            bcv[0] = *(const f32x4*)(b1l + k * 32 + g * 4);
            bcv[1] = *(const f32x4*)(b1l + k * 32 + 16 + g * 4);
        }
        sfor<8>([&](auto P2) __attribute__((always_inline)) {
            sfor<2>([&](auto PH) __attribute__((always_inline)) {
                constexpr int pi = decltype(P2)::value * 2 + decltype(PH)::value;
                const f16x8 w = wf[pi % NB];
                if constexpr (kind == 0) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[pi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xf[src][j], acc[pi][j], 0, 0, 0);
                } else if constexpr (kind == 1) {
                    // GEMM1 accumulates in VGPRs (the activation reads them with VALU instructions; hipcc would put every MFMA
                    // result of a 512-register kernel in the accumulator half and copy it out): VGPR-destination MFMA by hand,
                    // the first k-step starts from the bias.  Their first VALU reader is a whole item later.
                    constexpr int s_ = pi >> 1, hf = pi & 1;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if constexpr (s_ == 0)
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(h[hf][j]) : "v"(w), "v"(xf[s_][j]), "v"(bcv[hf]));
                        else
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(h[hf][j]) : "v"(w), "v"(xf[s_][j]));
                    }
                } else {
                    // (hand-written accumulator-tied MFMAs here and in kind 0 were tried to stop hipcc permuting the 192 accumulator
                    // registers at the phase boundaries: same speed, and their operand hazards are not padded -- builtin kept)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[pi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, hb[j], acc[pi][j], 0, 0, 0);
                }
                if constexpr (pi + PD < 16) wf[(pi + PD) % NB] = *(const f16x8*)(wc + (pi + PD) * 1024);
                else if constexpr (pfn) wf[(pi + PD) % NB] = *(const f16x8*)(wn + (pi + PD - 16) * 1024);
                // the 4 DMA pieces of the item NSLOT-1 ahead, on fragments 0 .. 3
                if constexpr (pi < 4) dma_piece(sd, IC<pi>{});
                // activation of the half-chunk held in h (6 fragment parts) on fragments 2, 4, ..., 12
                if constexpr (kind == 2 && conv && pi >= 2 && pi < 2 + 4 * NJ && !(pi & 1)) conv_part(IC<(pi - 2) / 2>{}, hbo);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        dma_advance();
        slot = (slot + 1) & (NSLOT - 1);
    };
    // the accumulators sit in the accumulator half of the register file whenever matrix work is about to run on them
    auto pin_acc = [&](int where) __attribute__((always_inline)) {
        {
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i][j]));
        }
    };
    // (an L2 touch of later tiles' rows by LDS-DMA into the dump area was measured and dropped with the other study switches: round 5 hygiene;
    // the shipped choice requests the next tile's rows before the last two items)
    bool loose = false;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        float alpha_l = p.alpha;                          // laundered: 1/alpha hoisted out of the tile loop ended up in scratch
        asm volatile("" : "+s"(alpha_l));
        float ralpha = 1.0f / alpha_l;
        // (re-derived at every phase boundary: nothing lane-dependent has to stay live -- or be spilled -- across a phase)
        auto relaunder = [&]() __attribute__((always_inline)) {
            asm volatile("" : "+v"(tid));
            lane = tid & 63; frow = lane & 15; g = lane >> 4; fo = g * 64;
            dvo = lane * 16 + wave * 4096;
            wl = smem + lane * 16;
        };
        relaunder();
#ifdef EEND_FS_TRACE
        ++tix;
        unsigned long long ts[8];
#endif
        FS_STAMP(0);
        if constexpr (PRE) {
            // ---- x = LN1(A Wo^T + bo + res): accumulators start at bo, residual added behind the GEMM
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 b4 = vec4(0, i);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = b4;
            }
            using T = std::true_type;
            using Fa = std::false_type;
            pin_acc(1);
            step(IC<0>{}, IC<0>{}, Fa{}, T{}, T{}, IC<0>{}, loose, 0, hbA, hbB);
            step(IC<0>{}, IC<1>{}, Fa{}, Fa{}, T{}, IC<0>{}, loose, 0, hbA, hbB);
            step(IC<0>{}, IC<2>{}, Fa{}, Fa{}, T{}, IC<0>{}, loose, 0, hbA, hbB);
            step(IC<0>{}, IC<3>{}, Fa{}, Fa{}, T{}, IC<0>{}, loose, 0, hbA, hbB);
            step(IC<0>{}, IC<4>{}, Fa{}, Fa{}, T{}, IC<0>{}, loose, 0, hbA, hbB);
            step(IC<0>{}, IC<5>{}, Fa{}, Fa{}, T{}, IC<0>{}, loose, 0, hbA, hbB);
            loose = false;
            step(IC<0>{}, IC<6>{}, Fa{}, Fa{}, T{}, IC<0>{}, false, 0, hbA, hbB);
            // residual rows of token fragment 0 travel under the last Wo item
            f32x4 t4[RES16 ? 1 : 16];
            auto load_res = [&](auto J) __attribute__((always_inline)) {
                if constexpr (!RES16) {
                    constexpr int j = decltype(J)::value;
                    const int off = row_of(tile, j) * 1024 + fo * 4;
#pragma unroll
                    for (int i = 0; i < 16; ++i) t4[i] = __builtin_bit_cast(f32x4, bload(rsR32, off + i * 16));
                }
            };
            if constexpr (LO) {                              // the remainder weight's eight items on the same input fragments
                step(IC<0>{}, IC<7>{}, Fa{}, Fa{}, T{}, IC<0>{}, false, 0, hbA, hbB);
                step(IC<0>{}, IC<0>{}, Fa{}, Fa{}, T{}, IC<0>{}, false, 0, hbA, hbB);
                step(IC<0>{}, IC<1>{}, Fa{}, Fa{}, T{}, IC<0>{}, false, 0, hbA, hbB);
                step(IC<0>{}, IC<2>{}, Fa{}, Fa{}, T{}, IC<0>{}, false, 0, hbA, hbB);
                step(IC<0>{}, IC<3>{}, Fa{}, Fa{}, T{}, IC<0>{}, false, 0, hbA, hbB);
                step(IC<0>{}, IC<4>{}, Fa{}, Fa{}, T{}, IC<0>{}, false, 0, hbA, hbB);
                step(IC<0>{}, IC<5>{}, Fa{}, Fa{}, T{}, IC<0>{}, false, 0, hbA, hbB);
                step(IC<0>{}, IC<6>{}, Fa{}, Fa{}, T{}, IC<0>{}, false, 0, hbA, hbB);
            }
            load_res(IC<0>{});
            step(IC<0>{}, IC<7>{}, Fa{}, Fa{}, Fa{}, IC<(!RES16 ? 16 : 0)>{}, false, 0, hbA, hbB);
            pin_acc(2);
            FS_STAMP(1);
            relaunder();
            asm volatile("" : "+s"(alpha_l));
            ralpha = 1.0f / alpha_l;
            sfor<NJ>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                // Two passes over the accumulators (AGPR reads are cheap) instead of a 64-value buffer: the buffer next to the
                // residual rows overflowed the register file, and a scratch reload waits for every VMEM operation in flight.
                // Statistics in one pass: sum and sum of squares (f32; |x| = O(10)).
                auto xval = [&](int i, int q) __attribute__((always_inline)) {
                    if constexpr (RES16) return acc[i][j][q] + (float)r8[j][i >> 1][(i & 1) * 4 + q];
                    else return acc[i][j][q] + t4[i][q];
                };
                f32x2 sm = f32x2{0.f, 0.f}, sq2 = f32x2{0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x2 x0 = f32x2{xval(i, 0), xval(i, 1)}, x1 = f32x2{xval(i, 2), xval(i, 3)};
                    sm += x0 + x1;
                    sq2 = x1 * x1 + (x0 * x0 + sq2);
                }
                const float sum = wave_g_allreduce_add(sm[0] + sm[1]);
                const float sqs = wave_g_allreduce_add(sq2[0] + sq2[1]);
                const float mean = sum * (1.0f / 256);
                const float rstd = 1.0f / __builtin_sqrtf(__builtin_fmaxf(sqs * (1.0f / 256) - mean * mean, 0.f) + p.eps1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (RES16) {
                    if constexpr (j == 0) { load_res16(tile, IC<1>{}); if constexpr (NJ > 2) load_res16(tile, IC<2>{}); }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 g4 = vec4(1, i), gg = g4 * rstd, bb = vec4(2, i) - g4 * (rstd * mean), b2 = vec4(3, i);
                    f32x4 xo;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = __builtin_fmaf(xval(i, q), gg[q], bb[q]);
                        // (no saturation: |x| <= 16 |gamma| + |beta| after a LayerNorm over 256 features)
                        xf[i >> 1][j][(i & 1) * 4 + q] = (_Float16)x;
                        xo[q] = x * ralpha + b2[q];
                    }
                    acc[i][j] = xo;
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // bounds how far the vector reads are hoisted
                }
                if constexpr (!RES16) { if constexpr (j + 1 < NJ) load_res(IC<j + 1>{}); }
            });
        } else {
            // ---- plain FFN: xf already holds X (natural k order); accumulators start at res / alpha + b2
            sfor<NJ>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                const int off = row_of(tile, j) * 1024 + fo * 4;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i][j] = __builtin_bit_cast(f32x4, bload(rsR32, off + i * 16)) * ralpha + vec4(3, i);
                __builtin_amdgcn_sched_barrier(0);
            });
        }

        // ---- FFN: item k = { h = W1h(k) x | acc += W2h(k-1) hb(k-1), hb(k) = act(h + b1) }, k = 0 .. U
        FS_STAMP(2);
        relaunder();
        {
            using T = std::true_type;
            using Fa = std::false_type;
            pin_acc(4);
            // W1h(0) | { W1h(k), W2h(k-1) + activation of h(k) } k = 1 .. U-1 | W2h(U-1)
            int nloose = (!PRE && loose) ? 6 : 0;           // items after an epilogue whose wait is the loose one
            step(IC<1>{}, IC<0>{}, Fa{}, T{}, Fa{}, IC<0>{}, nloose > 0, 0, hbA, hbB);
            FS_STAMP(7);
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // the hand-written MFMAs' results are read by VALU instructions next
            sfor<2 * NJ>([&](auto Q) __attribute__((always_inline)) { conv_part(Q, hbA); });
            FS_STAMP(3);
            step(IC<1>{}, IC<0>{}, Fa{}, T{}, T{}, IC<0>{}, nloose > 1, 1, hbA, hbB);
            for (int k = 2; k < U; k += 2) {                // U is even
                step(IC<2>{}, IC<0>{}, T{}, Fa{}, T{}, IC<0>{}, nloose > 2 * k - 2, 0, hbA, hbB);      // W2h(k-2) x hbA, h(k-1) -> hbB
                step(IC<1>{}, IC<0>{}, Fa{}, Fa{}, T{}, IC<0>{}, nloose > 2 * k - 1, k, hbA, hbB);      // W1h(k)
                step(IC<2>{}, IC<0>{}, T{}, Fa{}, T{}, IC<0>{}, nloose > 2 * k, 0, hbB, hbA);          // W2h(k-1) x hbB, h(k) -> hbA
                step(IC<1>{}, IC<0>{}, Fa{}, Fa{}, T{}, IC<0>{}, nloose > 2 * k + 1, k + 1, hbA, hbB);  // W1h(k+1)
            }
            if constexpr (!PRE) loose = false;
            FS_STAMP(4);
            // x is dead: the next tile's input rows are requested here and travel under the last two items and the epilogue
            // (unconditionally -- rows beyond M read as zeros -- so that the wait counts below hold on the last tile too)
            sfor<NJ>([&](auto J) __attribute__((always_inline)) { load_in_frags(tile + (int)gridDim.x, J); });
            constexpr int VWL = 8 * NJ;
            step(IC<2>{}, IC<0>{}, T{}, Fa{}, T{}, IC<VWL>{}, false, 0, hbA, hbB);     // W2h(U-2) x hbA, h(U-1) -> hbB
            step(IC<2>{}, IC<0>{}, Fa{}, Fa{}, Fa{}, IC<VWL>{}, false, 0, hbB, hbA);    // W2h(U-1) x hbB
            pin_acc(8);
            FS_STAMP(5);
        }

        // ---- epilogue, one token fragment at a time: v = acc * alpha; LayerNorm2; rows leave through the wave's 4-KB staging
        // tile as whole rows; the accumulators a fragment frees take the next tile's input rows
        relaunder();
        char* st = smem + STAGE + wave * 4096;
        const bool o32 = p.out32 != nullptr;
        const int ntile = tile + (int)gridDim.x;
        sfor<NJ>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const int rbase = tile * TM + wave * WM + j * 16;
            // two passes over the accumulators, no 64-value buffer (see LayerNorm1)
            f32x2 sm = f32x2{0.f, 0.f}, sq2 = f32x2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 a4 = acc[i][j] * p.alpha;
                const f32x2 x0 = f32x2{a4[0], a4[1]}, x1 = f32x2{a4[2], a4[3]};
                sm += x0 + x1;
                sq2 = x1 * x1 + (x0 * x0 + sq2);
            }
            const float sum = wave_g_allreduce_add(sm[0] + sm[1]);
            const float sqs = wave_g_allreduce_add(sq2[0] + sq2[1]);
            const float mean = sum * (1.0f / 256);
            const float rstd = 1.0f / __builtin_sqrtf(__builtin_fmaxf(sqs * (1.0f / 256) - mean * mean, 0.f) + p.eps);
            __builtin_amdgcn_sched_barrier(0);
            f16x8 o[8];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 g4 = vec4(4, i), gg = g4 * (rstd * p.alpha), bb = vec4(5, i) - g4 * (rstd * mean);
                const f32x4 a4 = acc[i][j];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[i >> 1][(i & 1) * 4 + q] = (_Float16)__builtin_fmaf(a4[q], gg[q], bb[q]);      // (no saturation: a LayerNorm output)
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            // what out32 receives: the LayerNorm output (RES_LN) or the un-normalised residual stream
            auto out32_val = [&](int i, int q) __attribute__((always_inline)) {
                const float a = acc[i][j][q] * p.alpha;
                if constexpr (EPI == FFN_EPI_RES_LN) return __builtin_fmaf(a - mean, vecs[4 * 256 + fo + i * 4 + q] * rstd, vecs[5 * 256 + fo + i * 4 + q]);
                else return a;
            };
#pragma unroll
            for (int half = 0; half < 2; ++half) {          // token rows 0..7 / 8..15 of the fragment
                if ((frow >> 3) == half) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) *(f16x8*)(st + (frow & 7) * 512 + (((g * 8 + e) ^ (frow & 7)) << 4)) = o[e];
                }
                wave_lds_sync();
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int rr = 2 * q4 + (lane >> 5), cc = lane & 31;
                    const f16x8 v = *(const f16x8*)(st + rr * 512 + ((cc ^ rr) << 4));      // (same type as the writes: no type-based reordering)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO16, (rbase + half * 8 + rr) * 512 + cc * 16, 0, 0);
                }
                wave_lds_sync();
            }
            if (LO && p.out16lo) {                           // the f16 remainder of the same rows, same way
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 g4 = vec4(4, i), gg = g4 * (rstd * p.alpha), bb = vec4(5, i) - g4 * (rstd * mean);
                    const f32x4 a4 = acc[i][j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float y = __builtin_fmaf(a4[q], gg[q], bb[q]);
                        o[i >> 1][(i & 1) * 4 + q] = (_Float16)(y - (float)o[i >> 1][(i & 1) * 4 + q]);
                    }
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
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
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO16L, (rbase + half * 8 + rr) * 512 + cc * 16, 0, 0);
                    }
                    wave_lds_sync();
                }
            }
            if (o32) {
#pragma unroll
                for (int fh = 0; fh < 2; ++fh)              // features fo + 0..31 / fo + 32..63
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if ((frow >> 3) == half) {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                *(f32x4*)(st + (frow & 7) * 512 + (((g * 8 + e) ^ (frow & 7)) << 4)) =
                                    f32x4{out32_val(fh * 8 + e, 0), out32_val(fh * 8 + e, 1), out32_val(fh * 8 + e, 2), out32_val(fh * 8 + e, 3)};
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
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (ntile < ntiles) load_res16(ntile, IC<0>{});
        FS_STAMP(6);
#ifdef EEND_FS_TRACE
        if (tix < 8 && threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) g_fs_trace[((size_t)blockIdx.x * 8 + tix) * 10 + k] = ts[k];
        }
#endif
        loose = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the workgroup
}

template <int MODE, int ACT, int EPI, bool RES16, int NJ, bool LO = false>
int launch_nj(const FfnStreamParams& p, int ncu, hipStream_t stream) {
    static EendOncePerDevice attr_once;
    auto kern = ffn_stream_kernel<MODE, ACT, EPI, RES16, NJ, LO>;
    if (!eend_set_dynamic_lds(attr_once, (const void*)kern, SMEM)) return EEND_ELAUNCH;
    const int ntiles = (p.M + 64 * NJ - 1) / (64 * NJ);
    hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), SMEM, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

template <int MODE, int ACT, int EPI, bool RES16, bool LO = false>
int launch(const FfnStreamParams& p, hipStream_t stream) {
    const int ncu = eend_cu_count();
    // 192-row tiles reuse every weight fragment for three MFMAs; when they leave CUs idle in the only (or last of few) rounds,
    // 128-row tiles finish earlier: compare rounds x rows-per-tile (the time of a tile is close to linear in its rows).
    const long t3 = (p.M + 191) / 192, t2 = (p.M + 127) / 128;
    const long c3 = ((t3 + ncu - 1) / ncu) * (3 * 10 + 9), c2 = ((t2 + ncu - 1) / ncu) * (2 * 10 + 9);     // per-tile cost model: rows + fixed part
    const int forced = g_debug_nj.load(std::memory_order_relaxed);              // tests only (eend_debug_ffn_stream_set)
    const int nj = forced ? forced : (c2 < c3 ? 2 : 3);
    return nj == 2 ? launch_nj<MODE, ACT, EPI, RES16, 2, LO>(p, ncu, stream) : launch_nj<MODE, ACT, EPI, RES16, 3, LO>(p, ncu, stream);
}

}  // namespace

#ifdef EEND_FS_TRACE
extern "C" int eend_debug_fs_trace(void* dst, void* stream) {
    return hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(g_fs_trace), sizeof(g_fs_trace), 0, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}
#endif

// with_wo: 0 = FFN only, 1 = Wo items in front, 2 = Wo and its f16 remainder (the LO form)
long eend_ffn_stream_nelems(int F, int with_wo) { return (long)((with_wo == 2 ? 16 : with_wo ? 8 : 0) + 2 * (F / 32)) * (SLOT / 2); }

int eend_launch_ffn_stream_pack(const void* Wo, const void* Wo_lo, const void* W1, const void* W2, void* out, int F, int k_permuted,
                                hipStream_t stream) {
    if (!W1 || !W2 || !out || F < 64 || (F % 64) != 0 || (Wo_lo && !Wo)) return EEND_EINVAL;
    const long total = eend_ffn_stream_nelems(F, Wo ? (Wo_lo ? 2 : 1) : 0) / 8;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(ffn_stream_pack_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, stream, (const _Float16*)Wo, (const _Float16*)Wo_lo,
                       (const _Float16*)W1, (const _Float16*)W2, (_Float16*)out, F, k_permuted);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_ffn_stream(const FfnStreamParams& p, int mode, int act, int epi, hipStream_t stream) {
    if (p.M <= 0 || p.F < 64 || (p.F % 64) != 0 || p.F > MAXF || (p.lda & 7) || !p.A || !p.wstream || !p.b1 || !p.b2 || !p.gamma || !p.beta ||
        !p.out16 || !(p.alpha != 0.f) || ((long)p.M + 65536) * 1024 >= (1L << 31) || ((long)p.M + 65536) * p.lda * 2 >= (1L << 31))      // 32-bit buffer offsets (prefetch runs one grid of tiles ahead)
        return EEND_EINVAL;
    if (mode == 1) {
        if (!p.bo || !p.g1 || !p.be1 || act != 1 || epi != FFN_EPI_RES_LN || (!p.res16 && !p.res32)) return EEND_EINVAL;
        if (p.wo_lo) {                                       // LO form: the stream carries the remainder weight's items (caller's contract)
            if (p.res16 || !p.res32 || ((size_t)p.out16lo & 15)) return EEND_EINVAL;
            return launch<1, 1, FFN_EPI_RES_LN, false, true>(p, stream);
        }
        if (p.out16lo) return EEND_EINVAL;
        return p.res16 ? launch<1, 1, FFN_EPI_RES_LN, true>(p, stream) : launch<1, 1, FFN_EPI_RES_LN, false>(p, stream);
    }
    if (!p.res32) return EEND_EINVAL;
    if (epi == FFN_EPI_RES_LN) {
        if (act == 1) return launch<0, 1, FFN_EPI_RES_LN, false>(p, stream);
        if (act == 2) return launch<0, 2, FFN_EPI_RES_LN, false>(p, stream);
    } else if (epi == FFN_EPI_RES_SCALE_LN16) {
        if (!p.out32) return EEND_EINVAL;
        if (act == 1) return launch<0, 1, FFN_EPI_RES_SCALE_LN16, false>(p, stream);
        if (act == 2) return launch<0, 2, FFN_EPI_RES_SCALE_LN16, false>(p, stream);
    }
    return EEND_EINVAL;
}
