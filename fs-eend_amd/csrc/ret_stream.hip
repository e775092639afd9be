// LS-EEND multi-scale retention with its four projections fused on chip (round 5): `MultiScaleRetention.forward`
// (LS-EEND/nnet/modules/retention.py:196-228: q / k / v / g projections, k * dk^-0.5) -> `chunk_recurrent_forward` (:146-194) ->
// per-head LayerNorm (:222) -> swish gate (:224), for the Conformer blocks (conformer/attention.py) and the decoder layers
// (merge_retnet_layer.py:233-253).  Replaces proj_xres_kernel + ret_kv_chunk_kernel + ret_chunk_full_kernel: q, k, k^T, v^T and g
// (2.5 KB per frame and layer, written and read back) never exist in HBM.
//
// Two persistent launches around the (unchanged) chunk-state scan of retention.hip:
//   pass 1  ret_stream_kernel<true>   one item per (sequence, head, chunk): project K^T, V^T of the chunk's frames from the X rows into
//                                     LDS and reduce KV_c = K_c^T V_c (f32) into the scan's workspace; the last chunk is skipped
//                                     unless the state is carried out.
//   scan    ret_state_scan_kernel     prefix states (hi/lo f16) + cross_scale per chunk.
//   pass 2  ret_stream_kernel<false>  one item per (sequence, head, chunk, query half): the 8 waves project K / V^T of the keys the
//                                     half can see (256 or 512 frames: 2 or 4 token fragments per wave) into LDS, Q and the gate G
//                                     only for their own 32 queries, then run the masked linear-attention loop of
//                                     ret_chunk_full_kernel (retention_full.hip: same scales, per-head LayerNorm, gate).
// The attn_stream.hip machinery: a wave keeps its tokens' X rows in registers as MFMA operand fragments (whole-row requests,
// wave-private LDS transposition), the head's weights arrive pre-packed in fragment order by LDS-DMA into regions that are free
// at that time, every 1-KB fragment read feeds 2 - 4 MFMAs, Q and G never leave the wave.
//
// Why query halves: with G next to Q the whole-chunk ownership of attn_stream.hip (64 tokens = 128 fragment registers per wave)
// does not fit the 256-register budget of two waves per SIMD.  A half item keeps Q / G for 32 tokens; the upper half recomputes
// the lower half's K / V (+2 of 10 weight items).
//
// Precision: the error of a retention row is dq . S_t with S_t the running sum of k (x) v -- it does not average
// out over keys like a rounding of k or v, and the per-head LayerNorm (eps 1e-6) amplifies it on near-cancelling rows.  So the
// QUERY path carries ~22 significand bits: q = (W_hi + W_lo)(x_hi + x_lo) with three f16 MFMA products (x_lo = the f16 remainder of the
// f32 residual stream, optional input), q itself is a hi/lo f16 pair in the score product (2 MFMAs per tile) and the cross-chunk
// term.  K, V, G, the probabilities and the state stay as before (f16 operands, f32 accumulation, hi/lo state).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }
template <int V> using IC = std::integral_constant<int, V>;

constexpr int KB = 64;
constexpr int TILE = KB * 128;            // one [64][64] f16 tile
constexpr int NW = 8;
constexpr int OSTG = 32 * 128;            // per-wave staging: 32 rows x 128 B (Q / G in, O out)
constexpr int WITEM = 16384;              // one weight item: 16 fragments of 1 KB (8 k-steps x 2 feature fragments)
constexpr int NITEM = 10;
enum { I_Q0H = 0, I_Q1H = 1, I_Q0L = 2, I_Q1L = 3, I_G0 = 4, I_G1 = 5, I_K0 = 6, I_K1 = 7, I_V0 = 8, I_V1 = 9 };
constexpr int L_K = 0, L_V = 8 * TILE, L_X = 16 * TILE;
constexpr int SMEM = L_X + NW * OSTG;     // 160 KB

typedef __attribute__((address_space(3))) char lds_char;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

DEV int swap23(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

DEV _Float16 f16_sat(float x) { return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }      // one clamp instruction
DEV u32x2 pack_f16x4(const f32x4 v) {
    f16x4 o;
    o[0] = f16_sat(v[0]); o[1] = f16_sat(v[1]); o[2] = f16_sat(v[2]); o[3] = f16_sat(v[3]);
    return __builtin_bit_cast(u32x2, o);
}

// weight packing, one thread per 16 bytes of the stream: [head h][item n][fragment p = ks*2 + hf][lane (f = l & 15, g = l >> 4)][8]
//   = part(W[t*256 + h*64 + (half*2 + hf)*16 + f][ks*32 + g*8 + e]);  W = the packed f32 [q; k * dk^-0.5; v; g] rows (t = 0 .. 3),
//   items Q0H Q1H Q0L Q1L (t = 0, half 0 / 1, hi then lo part), G0 G1 (t = 3), K0 K1 (t = 1), V0 V1 (t = 2); hi = f16(w), lo = f16(w - hi).
__global__ void ret_stream_pack_kernel(const float* __restrict__ W, _Float16* __restrict__ out) {
    const int total = 4 * NITEM * 1024;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int h = t / (NITEM * 1024), r = t - h * (NITEM * 1024), n = r >> 10, w = r & 1023;
        const int pfrag = w >> 6, l = w & 63, f = l & 15, g = l >> 4, ks = pfrag >> 1, hf = pfrag & 1;
        int tt, half;
        bool lo = false;
        if (n < 4) { tt = 0; half = n & 1; lo = n >= 2; }
        else if (n < 6) { tt = 3; half = n - 4; }
        else if (n < 8) { tt = 1; half = n - 6; }
        else { tt = 2; half = n - 8; }
        const float* src = W + (size_t)(tt * 256 + h * 64 + (half * 2 + hf) * 16 + f) * 256 + ks * 32 + g * 8;
        _Float16* dst = out + (size_t)t * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 hi = (_Float16)src[e];
            dst[e] = lo ? (_Float16)(src[e] - (float)hi) : hi;
        }
    }
}

// Perf-study build (-DEEND_RS_TRACE, tools/ret_stream_trace.py): s_memtime stamps of wave 0 of every workgroup, first 8 items of pass 2
#ifdef EEND_RS_TRACE
__device__ unsigned long long g_rs_trace[256 * 8 * 32];
#define RS_STAMP(k) do { if (!KV_ONLY) ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RS_STAMP(k) do {} while (0)
#endif

template <bool KV_ONLY>
__global__ __launch_bounds__(512)
void ret_stream_kernel(const RetStreamParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem + L_K;                           // [8][64 keys][128 B]   (KV_ONLY: K^T, [8][64 d][128 B])
    char* Vs = smem + L_V;                           // [8][64 d][128 B]; before that: weight items
    char* Xs = smem + L_X;                           // weight items V0, V1; afterwards the 8 x 4 KB staging tiles

    int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, fkg = lane >> 4;
    auto relaunder = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; frow = lane & 15; fkg = lane >> 4;
    };
    char* Ow = Xs + wave * OSTG;

    // ---- items.  pass 2: units = (query half, chunk, sequence), upper halves (the heavy ones) first; pass 1: units = (chunk, sequence).
    const int nseq = p.nseq, L = p.L;
    int ncB = 0;                                      // chunks with more than 256 frames
    if (!KV_ONLY) {
        for (int c = 0; c < p.nc; ++c) { int n = p.Tp - c * L; n = n < L ? n : L; if (n > 256) ++ncB; }
    }
    const int nunits = KV_ONLY ? p.nkv * nseq : (ncB + p.nc) * nseq;
    const int nitems = nunits * 4;
    const bool xcd_map = (nunits & 7) == 0 && ((gridDim.x & 7) == 0 || (int)gridDim.x >= nitems);

    // X rows of 32 tokens (two fragments) as requested: row r = 2 i + (lane >> 5), 16-byte chunk lane & 31
    auto request_rows = [&](const void* base, int seq_, int tok0, u32x4 (&xr)[2][8]) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)base + (size_t)seq_ * p.Tp * p.ldx), 0,
                                                                            p.Tp * p.ldx * 2, 0x00020000);
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) {
            const int off = (tok0 + jl * 16 + (lane >> 5)) * p.ldx * 2 + (lane & 31) * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[jl][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + i * 2 * p.ldx * 2, 0, 0);
        }
    };
    // rows -> operand fragments x[jl][ks] = X[tok][ks*32 + fkg*8 .. +8] through the wave-private 8-KB tile in the K region
    auto to_frags = [&](u32x4 (&xr)[2][8], f16x8 (&x)[2][8]) __attribute__((always_inline)) {
        relaunder();                                 // the 16 tile addresses are recomputed per call, not kept live across the item
        char* xt = Ks + wave * 8192;
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 2 * i + (lane >> 5);
                *(u32x4*)(xt + r * 512 + (((lane & 31) ^ r) << 4)) = xr[jl][i];
            }
            wave_lds_sync();
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) x[jl][ks] = __builtin_bit_cast(f16x8, *(const u32x4*)(xt + frow * 512 + (((ks * 4 + fkg) ^ frow) << 4)));
            wave_lds_sync();
        }
    };
    // the same through a 4-KB tile (8 rows at a time) in the lower half of the K region: for the upper items' second K / V pass, when
    // the upper half already holds K rows
    auto to_frags_half = [&](u32x4 (&xr)[2][8], f16x8 (&x)[2][8]) __attribute__((always_inline)) {
        relaunder();
        char* xt = Ks + wave * 4096;
#pragma unroll
        for (int jl = 0; jl < 2; ++jl)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r8 = 2 * i + (lane >> 5), r = sub * 8 + r8;
                    *(u32x4*)(xt + r8 * 512 + (((lane & 31) ^ r) << 4)) = xr[jl][sub * 4 + i];
                }
                wave_lds_sync();
                if ((frow >> 3) == sub) {                // the lanes whose fragment row is in this half read it (exec-masked: no selects)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        x[jl][ks] = __builtin_bit_cast(f16x8, *(const u32x4*)(xt + (frow & 7) * 512 + (((ks * 4 + fkg) ^ frow) << 4)));
                }
                wave_lds_sync();
            }
    };
    auto decode = [&](int item, int& seq_, int& h_, int& c_, int& upper_) __attribute__((always_inline)) {
        int u;
        if (xcd_map) { u = (item >> 5) * 8 + (item & 7); h_ = (item >> 3) & 3; }
        else { u = item >> 2; h_ = item & 3; }
        if (KV_ONLY) { c_ = u / nseq; seq_ = u - c_ * nseq; upper_ = 1; return; }
        if (u < ncB * nseq) { upper_ = 1; c_ = u / nseq; seq_ = u - c_ * nseq; }
        else { u -= ncB * nseq; upper_ = 0; c_ = u / nseq; seq_ = u - c_ * nseq; }
    };

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    int seq, h, c, upper;
    decode(item, seq, h, c, upper);
    seq = __builtin_amdgcn_readfirstlane(seq); h = __builtin_amdgcn_readfirstlane(h);
    c = __builtin_amdgcn_readfirstlane(c); upper = __builtin_amdgcn_readfirstlane(upper);
    relaunder();
    const int f0 = c * L;
    int n = p.Tp - f0;
    n = n < L ? n : L;                               // frames of this chunk
    // token fragments of this wave (local frame index in the chunk):
    //   pass 2: "query" fragments = 32 w' + 16 jl in the item's half, "other" fragments (upper half only) = the lower half's 32 w + 16 jl
    //   pass 1: fragments 0, 1 = 32 w + 16 jl, fragments 2, 3 = 256 + 32 w + 16 jl
    const int tq0 = (KV_ONLY ? 0 : (upper ? 256 : 0)) + 32 * wave;
    const int to0 = (KV_ONLY ? 256 : 0) + 32 * wave;
#ifdef EEND_RS_TRACE
    unsigned long long ts[32];
    const int tix0 = (item - (int)blockIdx.x) / (int)gridDim.x;
    const int tix = tix0 < 4 ? tix0 : (tix0 >= 12 && tix0 < 16 ? tix0 - 8 : 8);      // items 0 .. 3 (upper) and 12 .. 15 (lower, in the big launch)
#pragma unroll
    for (int k = 0; k < 32; ++k) ts[k] = 0;
#endif
    RS_STAMP(0);
    // the wave's first two token fragments, requested BEFORE the barrier that ends the previous item: a wave that has finished its
    // block has its rows in flight while it waits for the others.  (Requested a phase earlier -- under the previous item's epilogue --
    // their 64 registers spill into the ingest path: measured +20 %.)
    u32x4 xq_r[2][8], xo_r1[2][8];
    request_rows(p.X, seq, f0 + tq0, xq_r);
    if constexpr (KV_ONLY) request_rows(p.X, seq, f0 + to0, xo_r1);      // pass 1: all four fragments (nothing else is live there)

    const size_t sh = (size_t)seq * 4 + h;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.W + (size_t)h * NITEM * WITEM), 0,
                                                                        NITEM * WITEM, 0x00020000);
    auto dma_item = [&](int n_, char* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_char*)(dst + (wave * 2 + i) * 1024), 16, lane * 16, n_ * WITEM + (wave * 2 + i) * 1024, 0, 0);
    };

    // every wave is done with K / V^T / its staging tile of the previous item (not __syncthreads: its fence would wait for the
    // row loads just issued)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    RS_STAMP(1);

    // ---- VMEM issue order of a wave (returns are in order; the waits below count the younger operations):
    //   [xq rows 16] | biases 4 | xlo rows 16 (pass 2 with Xlo) | DMA Q0H Q1H Q0L Q1L V0 V1 (12; pass 1: K0 K1 V0 V1 = 8) | ...
    // the head's biases, one register per tensor: lane l holds bias[t][h*64 + l]; the rows a lane needs are fetched with ds_bpermute
    // (per-lane vector loads at the point of use sat behind the whole weight DMA queue: 1.5 k cycles per item in the s_memtime trace;
    // held per lane and feature fragment they would cost 52 registers)
    float breg[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) breg[t] = p.bias[t * 256 + h * 64 + lane];
    const bool has_lo = !KV_ONLY && p.Xlo != nullptr;
    u32x4 xl_r[2][8];
    if (!KV_ONLY) {
        if (has_lo) request_rows(p.Xlo, seq, f0 + tq0, xl_r);
        else {
#pragma unroll
            for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                for (int i = 0; i < 8; ++i) xl_r[jl][i] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    if (KV_ONLY) {
        dma_item(I_K0, Vs + 0 * WITEM); dma_item(I_K1, Vs + 1 * WITEM);
        dma_item(I_V0, Xs + 0 * WITEM); dma_item(I_V1, Xs + 1 * WITEM);
    } else {
        dma_item(I_Q0H, Vs + 0 * WITEM); dma_item(I_Q1H, Vs + 1 * WITEM); dma_item(I_Q0L, Vs + 2 * WITEM); dma_item(I_Q1L, Vs + 3 * WITEM);
        dma_item(I_V0, Xs + 0 * WITEM); dma_item(I_V1, Xs + 1 * WITEM);
    }

    f16x8 xq[2][8], xo[2][8];
    // the xq rows are older than everything issued in this item
    if (KV_ONLY) __builtin_amdgcn_s_waitcnt(0x0F70 | (28 & 15) | ((28 >> 4) << 14));                     // xo 16 + biases 4 + DMA 8
    else if (has_lo) __builtin_amdgcn_s_waitcnt(0x0F70 | (32 & 15) | ((32 >> 4) << 14));                 // bv 4 + xlo 16 + DMA 12
    else __builtin_amdgcn_s_waitcnt(0x0F70 | (16 & 15) | ((16 >> 4) << 14));                             // bv 4 + DMA 12
    to_frags(xq_r, xq);
    RS_STAMP(2);

    // one weight item over NJ token fragments: acc[hf][j] (+)= W-fragment(ks, hf) x X-fragment(j, ks); VT: rows = token (V^T layout).
    // side(u), u = 0 .. 3: a quarter of the PREVIOUS item's epilogue (clamp / pack / LDS writes / swish), issued behind fragments 1, 5,
    // 9, 13 so that its VALU and LDS instructions run under this item's MFMAs -- both waves of a SIMD are in the same phase
    // (barriers), so an epilogue after the last MFMA is exposed matrix-pipe idle time (s_memtime trace: 1.2 - 1.7 k of a 2.6 k-cycle item).
    auto run_item = [&](const char* wbase, auto NJc, auto VTc, f32x4 (&acc)[2][4], auto getx, auto side) __attribute__((always_inline)) {
        constexpr int NJ = decltype(NJc)::value;
        constexpr bool VT = decltype(VTc)::value;
        relaunder();
        const char* wi = wbase + lane * 16;
        constexpr int PD = NJ == 4 ? 4 : 2, NB = 2 * PD;
        f16x8 wf[NB];
        sfor<PD>([&](auto Q) __attribute__((always_inline)) { wf[decltype(Q)::value] = *(const f16x8*)(wi + decltype(Q)::value * 1024); });
        sfor<16>([&](auto QQ) __attribute__((always_inline)) {
            constexpr int q = decltype(QQ)::value, ks = q >> 1, hf = q & 1;
            const f16x8 w = wf[q % NB];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f16x8 xv = getx(j, ks);
                if constexpr (VT) acc[hf][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xv, w, acc[hf][j], 0, 0, 0);
                else acc[hf][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xv, acc[hf][j], 0, 0, 0);
            }
            if constexpr (q + PD < 16) wf[(q + PD) % NB] = *(const f16x8*)(wi + (q + PD) * 1024);
            if constexpr ((q & 3) == 1) side(IC<(q >> 2)>{});
            if constexpr (q & 1) __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_side = [](auto) __attribute__((always_inline)) {};
    // bias of feature rows ff * 16 + fkg * 4 + r of tensor t (0 q, 1 k, 2 v, 3 g)
    auto bias_rows = [&](int t, int ff) __attribute__((always_inline)) {
        f32x4 b;
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = __shfl(breg[t], ff * 16 + fkg * 4 + r, 64);
        return b;
    };
    // ... of feature column ff * 16 + frow (V^T layout)
    auto bias_col = [&](int t, int ff) __attribute__((always_inline)) { return __shfl(breg[t], ff * 16 + frow, 64); };
    auto wait_barrier = [&](int younger) __attribute__((always_inline)) {
        // this wave's pieces of the item have landed (`younger` operations may stay in flight); then every wave's
        switch (younger) {      // s_waitcnt takes an immediate
            case 0: __builtin_amdgcn_s_waitcnt(0x0F70 | 0); break;
            case 2: __builtin_amdgcn_s_waitcnt(0x0F70 | 2); break;
            case 4: __builtin_amdgcn_s_waitcnt(0x0F70 | 4); break;
            case 6: __builtin_amdgcn_s_waitcnt(0x0F70 | 6); break;
            case 10: __builtin_amdgcn_s_waitcnt(0x0F70 | 10); break;
            case 12: __builtin_amdgcn_s_waitcnt(0x0F70 | 12); break;
            case 20: __builtin_amdgcn_s_waitcnt(0x0F70 | (20 & 15) | ((20 >> 4) << 14)); break;
            case 22: __builtin_amdgcn_s_waitcnt(0x0F70 | (22 & 15) | ((22 >> 4) << 14)); break;
            case 26: __builtin_amdgcn_s_waitcnt(0x0F70 | (26 & 15) | ((26 >> 4) << 14)); break;
            case 28: __builtin_amdgcn_s_waitcnt(0x0F70 | (28 & 15) | ((28 >> 4) << 14)); break;
            default: __builtin_amdgcn_s_waitcnt(0x0F70 | 0); break;
        }
        __builtin_amdgcn_s_barrier();
    };

    u32x2 qh[4][2], ql[4][2], gk[4][2];              // [feature fragment][token fragment]: f16 x 4 of q_hi, q_lo, swish(g)
    if constexpr (!KV_ONLY) {
        // ================================================================== Q (three products), G
        f16x8 xl[2][8];
        __builtin_amdgcn_s_waitcnt(0x0F70 | 12);         // the xlo rows have landed (12 DMA pieces are younger)
        to_frags(xl_r, xl);
        RS_STAMP(3);
        // Two accumulator sets in rotation: while an item accumulates into one, the epilogue of the item before it (pack q / swish g /
        // write K, V^T rows) drains the other in four slices under the MFMAs.
        f32x4 A[2][4], Bc[2][4];
        // (the packed values are pinned where they are computed: left alone, LLVM sinks these side-effect-free epilogues down to their
        // first use in the chunk loop -- keeping 32 f32 accumulators alive instead of 16 packed registers, and undoing the overlap)
        auto pin2 = [](u32x2& v) __attribute__((always_inline)) { unsigned a = v[0], b = v[1]; asm volatile("" : "+v"(a), "+v"(b)); v[0] = a; v[1] = b; };
        auto epi_q = [&](f32x4 (&acc)[2][4], int half, auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, hf = u >> 1, j = u & 1;
            const f32x4 v = acc[hf][j];
            f16x4 hi4;
            f32x4 rem;
#pragma unroll
            for (int r = 0; r < 4; ++r) { hi4[r] = f16_sat(v[r]); rem[r] = v[r] - (float)hi4[r]; }
            qh[half * 2 + hf][j] = __builtin_bit_cast(u32x2, hi4);
            ql[half * 2 + hf][j] = pack_f16x4(rem);
            pin2(qh[half * 2 + hf][j]); pin2(ql[half * 2 + hf][j]);
        };
        auto epi_g = [&](f32x4 (&acc)[2][4], int half, auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, hf = u >> 1, j = u & 1;
            f32x4 v = acc[hf][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-v[r]));
            gk[half * 2 + hf][j] = pack_f16x4(v);
            pin2(gk[half * 2 + hf][j]);
        };
        auto getq = [&](int j, int ks) __attribute__((always_inline)) { return xq[j][ks]; };
        auto getql = [&](int j, int ks) __attribute__((always_inline)) { return j < 2 ? xq[j][ks] : xl[j - 2][ks]; };
        auto seed = [&](f32x4 (&acc)[2][4], int t, int half, bool four) __attribute__((always_inline)) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const f32x4 b4 = bias_rows(t, half * 2 + hf);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[hf][j] = (j < 2 || !four) ? b4 : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto fold = [&](f32x4 (&acc)[2][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[hf][j] += acc[hf][j + 2];
        };
        // one barrier for the four Q items and V0, V1 (all requested together): everything this wave asked for has landed
        wait_barrier(0);
        RS_STAMP(4);
        relaunder();
        dma_item(I_G0, Ks + 2 * WITEM); dma_item(I_G1, Ks + 3 * WITEM);      // the K region's scratch tiles are done (barrier)
        // Q0H, Q1H: W_hi x (x_hi, x_lo) -- four MFMAs per fragment read
        seed(A, 0, 0, true);
        run_item(Vs + 0 * WITEM, IC<4>{}, std::false_type{}, A, getql, no_side);
        fold(A);
        seed(Bc, 0, 1, true);
        RS_STAMP(5);
        run_item(Vs + 1 * WITEM, IC<4>{}, std::false_type{}, Bc, getql, no_side);
        fold(Bc);
        // every wave is done with Q0H, Q1H (their slots take the K items) -- and this wave's pieces of the G items (requested two items
        // ago) have landed, so that G0 needs no barrier of its own (a barrier costs 10 - 15 units of wave skew in the s_memtime trace)
        wait_barrier(0);
        RS_STAMP(6);
        relaunder();
        dma_item(I_K0, Vs + 0 * WITEM); dma_item(I_K1, Vs + 1 * WITEM);
        // Q0L, Q1L: W_lo x x_hi
        run_item(Vs + 2 * WITEM, IC<2>{}, std::false_type{}, A, getq, no_side);
        RS_STAMP(7);
        run_item(Vs + 3 * WITEM, IC<2>{}, std::false_type{}, Bc, getq, [&](auto U) __attribute__((always_inline)) { epi_q(A, 0, U); });
        RS_STAMP(8);
        seed(A, 3, 0, false);
        run_item(Ks + 2 * WITEM, IC<2>{}, std::false_type{}, A, getq, [&](auto U) __attribute__((always_inline)) { epi_q(Bc, 1, U); });
        seed(Bc, 3, 1, false);
        RS_STAMP(9);
        run_item(Ks + 3 * WITEM, IC<2>{}, std::false_type{}, Bc, getq, [&](auto U) __attribute__((always_inline)) { epi_g(A, 0, U); });
        RS_STAMP(10);

        // ================================================================== K, V^T -> LDS (own frames, then -- upper items -- the lower half's)
        auto epi_k = [&](f32x4 (&acc)[2][4], int half, int tok0, auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, hf = u >> 1, j = u & 1;
            const int key = tok0 + 16 * j + frow, d = (half * 2 + hf) * 16 + fkg * 4;
            *(u32x2*)(Ks + (key >> 6) * TILE + swz128(key & 63, d >> 3) + (d & 7) * 2) = pack_f16x4(acc[hf][j]);
        };
        auto epi_v = [&](f32x4 (&acc)[2][4], int half, int tok0, auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, hf = u >> 1, j = u & 1;
            // lane: feature d = column frow, tokens key .. key + 3
            const int d = (half * 2 + hf) * 16 + frow, key = tok0 + 16 * j + fkg * 4;
            f32x4 v = acc[hf][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) if (key + r >= n) v[r] = 0.f;       // frames beyond the chunk do not enter the state
            *(u32x2*)(Vs + (key >> 6) * TILE + swz128(d, (key & 63) >> 3) + (key & 7) * 2) = pack_f16x4(v);
        };
        auto seed_v = [&](f32x4 (&acc)[2][4], int half) __attribute__((always_inline)) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float bc = bias_col(2, half * 2 + hf);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[hf][j] = f32x4{bc, bc, bc, bc};
            }
        };
        // K0: this wave's pieces of K0, K1 have landed; every wave is done with the G weights (where K rows may land).  K1: none.
        wait_barrier(0);
        RS_STAMP(24);
        seed(A, 1, 0, false);
        run_item(Vs + 0 * WITEM, IC<2>{}, std::false_type{}, A, getq, [&](auto U) __attribute__((always_inline)) { epi_g(Bc, 1, U); });
        RS_STAMP(25);
        seed(Bc, 1, 1, false);
        run_item(Vs + 1 * WITEM, IC<2>{}, std::false_type{}, Bc, getq, [&](auto U) __attribute__((always_inline)) { epi_k(A, 0, tq0, U); });
        RS_STAMP(26);
        // V0: every wave is done reading the K weights (lower items: the V^T rows land on them)
        __builtin_amdgcn_s_barrier();
        RS_STAMP(27);
        seed_v(A, 0);
        run_item(Xs + 0 * WITEM, IC<2>{}, std::true_type{}, A, getq, [&](auto U) __attribute__((always_inline)) { epi_k(Bc, 1, tq0, U); });
        RS_STAMP(28);
        seed_v(Bc, 1);
        run_item(Xs + 1 * WITEM, IC<2>{}, std::true_type{}, Bc, getq, [&](auto U) __attribute__((always_inline)) { epi_v(A, 0, tq0, U); });
        RS_STAMP(29);
        relaunder();
        sfor<4>([&](auto U) __attribute__((always_inline)) { epi_v(Bc, 1, tq0, U); });
        RS_STAMP(11);
        if (upper) {
            // the lower half's frames 32 w .. of this wave: requested only now (their 64 registers are not free earlier)
            u32x4 xo_r[2][8];
            request_rows(p.X, seq, f0 + to0, xo_r);
            __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
            RS_STAMP(12);
            to_frags_half(xo_r, xo);                     // wave-private 4-KB tiles in K rows 0 .. 255 (not yet written)
            RS_STAMP(13);
            auto geto = [&](int j, int ks) __attribute__((always_inline)) { return xo[j][ks]; };
            __builtin_amdgcn_s_barrier();                // every wave is done with those tiles
            seed(A, 1, 0, false);
            run_item(Vs + 0 * WITEM, IC<2>{}, std::false_type{}, A, geto, no_side);
            seed(Bc, 1, 1, false);
            run_item(Vs + 1 * WITEM, IC<2>{}, std::false_type{}, Bc, geto, [&](auto U) __attribute__((always_inline)) { epi_k(A, 0, to0, U); });
            __builtin_amdgcn_s_barrier();                // ... with the K weights (the V^T columns 0 .. 255 land on them)
            seed_v(A, 0);
            run_item(Xs + 0 * WITEM, IC<2>{}, std::true_type{}, A, geto, [&](auto U) __attribute__((always_inline)) { epi_k(Bc, 1, to0, U); });
            seed_v(Bc, 1);
            run_item(Xs + 1 * WITEM, IC<2>{}, std::true_type{}, Bc, geto, [&](auto U) __attribute__((always_inline)) { epi_v(A, 0, to0, U); });
            relaunder();
            sfor<4>([&](auto U) __attribute__((always_inline)) { epi_v(Bc, 1, to0, U); });
        }
        RS_STAMP(14);
    } else {
        // pass 1: fragments 2, 3 = the chunk's upper half (biases 4 + DMA 8 are younger)
        __builtin_amdgcn_s_waitcnt(0x0F70 | 12);
        to_frags(xo_r1, xo);
    }

    // ================================================================== K, V^T -> LDS
    // pass 2: two token fragments per pass (x2 = their operand fragments, tok2 = their first local frame); the upper items run a
    // second pass over the lower half's frames, whose rows are only requested then (64 registers that are not free earlier).
    // pass 1: four fragments (xq: frames 32 w .., xo: frames 256 + 32 w ..).
    auto kv_items = [&](auto NJc, const f16x8 (&x2)[2][8], int tok2, int first_wait) __attribute__((always_inline)) {
        constexpr int NJ = decltype(NJc)::value;
        auto tok_of = [&](int j) __attribute__((always_inline)) {
            if (NJ == 2) return tok2 + 16 * j;
            return j < 2 ? tq0 + 16 * j : to0 + 16 * (j - 2);
        };
        auto getx = [&](int j, int ks) __attribute__((always_inline)) {
            if (NJ == 2) return x2[j][ks];
            return j < 2 ? xq[j][ks] : xo[j - 2][ks];
        };
        sfor<4>([&](auto N) __attribute__((always_inline)) {
            constexpr int nn = decltype(N)::value, isv = nn >> 1, half = nn & 1;
            // K0: this wave's pieces of K0 and K1 have landed, every wave is done with what the K rows overwrite.  V0: barrier only --
            // every wave is done reading the K weights where V^T rows may land.  K1, V1: none.
            if (nn == 0) wait_barrier(0);
            else if (nn == 2) __builtin_amdgcn_s_barrier();
            relaunder();
            f32x4 acc[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 b4;
                if constexpr (isv || KV_ONLY) { const float bc = bias_col(isv ? 2 : 1, half * 2 + hf); b4 = f32x4{bc, bc, bc, bc}; }
                else b4 = bias_rows(1, half * 2 + hf);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[hf][j] = b4;
            }
            const char* wb = isv ? Xs + half * WITEM : Vs + half * WITEM;
            if constexpr (isv || KV_ONLY) run_item(wb, IC<NJ>{}, std::true_type{}, acc, getx, no_side);
            else run_item(wb, IC<NJ>{}, std::false_type{}, acc, getx, no_side);
            relaunder();
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (isv || KV_ONLY) {
                        // lane: feature d = column frow, tokens key .. key + 3
                        const int d = (half * 2 + hf) * 16 + frow, key = tok_of(j) + fkg * 4;
                        f32x4 v = acc[hf][j];
                        if constexpr (isv) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (key + r >= n) v[r] = 0.f;       // frames beyond the chunk do not enter the state
                        }
                        *(u32x2*)((isv ? Vs : Ks) + (key >> 6) * TILE + swz128(d, (key & 63) >> 3) + (key & 7) * 2) = pack_f16x4(v);
                    } else {
                        const int key = tok_of(j) + frow, d = (half * 2 + hf) * 16 + fkg * 4;
                        *(u32x2*)(Ks + (key >> 6) * TILE + swz128(key & 63, d >> 3) + (d & 7) * 2) = pack_f16x4(acc[hf][j]);
                    }
                }
        });
    };
    if constexpr (KV_ONLY) {
        // K0, K1 sit in V-region slots 0, 1; everything this wave requested has landed (the xo rows were waited for with vmcnt 0)
        kv_items(IC<4>{}, xq, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");    // K, V^T complete in LDS
    RS_STAMP(15);

    relaunder();
    const int lq = lane & 31, hi = lane >> 5;

    if constexpr (KV_ONLY) {
        // ================================================================== KV_c[kd][hd] = sum_key K^T[kd][key] V^T[hd][key]
        const int ti = (wave >> 1) & 1, tj = wave & 1, kh = wave >> 2;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int s_ = 0; s_ < 16; ++s_) {
            const int key = kh * 256 + s_ * 16 + hi * 8;
            const f16x8 a = *(const f16x8*)(Ks + (key >> 6) * TILE + swz128(ti * 32 + lq, (key & 63) >> 3));
            const f16x8 b = *(const f16x8*)(Vs + (key >> 6) * TILE + swz128(tj * 32 + lq, (key & 63) >> 3));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        float* part = (float*)Ow;                        // 32 x 32 f32 = the wave's 4-KB tile (the V weights there are consumed)
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) part[i * 64 + lane] = acc[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kh == 0) {
            const float* other = (const float*)(Xs + (wave + 4) * OSTG);
            float* __restrict__ KV = p.kv_ws + (sh * p.nc + c) * 4096;
            // C layout: col = hd (tj*32 + lq), rows kd = ti*32 + 8*g + 4*hi + r
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) KV[(ti * 32 + 8 * g + 4 * hi + r) * 64 + tj * 32 + lq] = acc[g * 4 + r] + other[(g * 4 + r) * 64 + lane];
        }
    } else {
        // ================================================================== the chunk loop (retention_full.hip), one 32-query block per wave
        const int qw0 = tq0;
        if (qw0 < n) {
            const int q = qw0 + lq;
            const int qc = q < n ? q : n - 1;
            const int krow = swap23(lq);
            f16x8 qfh[4], qfl[4];
            // Q of the block: registers -> the wave's staging tile ([query][64 d] f16 rows) -> operand layout; hi then lo
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                    for (int ff = 0; ff < 4; ++ff) {
                        const int row = jl * 16 + frow;
                        *(u32x2*)(Ow + row * 128 + (((ff * 2 + (fkg >> 1)) ^ (row & 7)) << 4) + (fkg & 1) * 8) = part == 0 ? qh[ff][jl] : ql[ff][jl];
                    }
                wave_lds_sync();
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f16x8 v = __builtin_bit_cast(f16x8, *(const u32x4*)(Ow + lq * 128 + (((ks * 2 + hi) ^ (lq & 7)) << 4)));
                    if (part == 0) qfh[ks] = v; else qfl[ks] = v;
                }
                wave_lds_sync();
            }
            RS_STAMP(17);
            f32x16 oT[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) { oT[0][i] = 0.f; oT[1][i] = 0.f; }
            float absum = 0.f;
            int wl = qw0 + 31;
            wl = wl < n - 1 ? wl : n - 1;
            const int jend = wl / KB + 1;
            for (int j = 0; j < jend; ++j) {
                const int key0 = j * KB;
                const char* kb_ = Ks + j * TILE;
                const char* vb_ = Vs + j * TILE;
                f32x16 s[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const f16x8 kf = *(const f16x8*)(kb_ + swz128(kb * 32 + krow, ks * 2 + hi));
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qfh[ks], s[kb], 0, 0, 0);
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qfl[ks], s[kb], 0, 0, 0);
                    }
                }
                if (key0 + KB - 1 > qw0) {               // the tile straddles the diagonal for some row of the wave
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int key = key0 + kb * 32 + (i & 7) + 8 * hi + 16 * (i >> 3);
                            if (key > qc) s[kb][i] = 0.f;
                        }
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) absum += __builtin_fabsf(s[kb][i]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        f16x8 pf;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) pf[jj] = f16_sat(s[kb][kk * 8 + jj]);
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const f16x8 vf = *(const f16x8*)(vb_ + swz128(db * 32 + lq, kb * 4 + kk * 2 + hi));
                            oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oT[db], 0, 0, 0);
                        }
                    }
            }
            RS_STAMP(18);
            // ---- cross-chunk term O^T += S_c^T Q^T (hi/lo f16 state x hi/lo q, prescale undone by sexp)
            if (c > 0 || p.has_state_in) {
                // (the state fragments come from L2 / HBM right here: requested before the tile loop their registers spill)
                const _Float16* __restrict__ Sg = (const _Float16*)p.St + (sh * p.nc + c) * 2 * 4096;
                const float sexp = p.sexp[sh * p.nc + c];
                f32x16 x[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) { x[0][i] = 0.f; x[1][i] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const f16x8 sa = *(const f16x8*)(Sg + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                        const f16x8 sb = *(const f16x8*)(Sg + 4096 + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                        x[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa, qfh[ks], x[db], 0, 0, 0);
                        x[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb, qfh[ks], x[db], 0, 0, 0);
                        x[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa, qfl[ks], x[db], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 0; i < 16; ++i) { oT[0][i] = __builtin_fmaf(x[0][i], sexp, oT[0][i]); oT[1][i] = __builtin_fmaf(x[1][i], sexp, oT[1][i]); }
            }
            RS_STAMP(19);
            // ---- scale, per-head LayerNorm, swish gate
            const float cscale = p.cscale[sh * p.nc + c];
            const float ab = absum + __shfl_xor(absum, 32, 64);
            const float rsq = 1.0f / __builtin_sqrtf((float)(qc + 1));
            const float inner_scale = __builtin_fmaxf(1.0f, ab * rsq);
            const float f = rsq / __builtin_fmaxf(inner_scale, cscale);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { oT[0][i] *= f; oT[1][i] *= f; sum += oT[0][i] + oT[1][i]; }
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / 64.0f);
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float a = oT[0][i] - mean, b = oT[1][i] - mean;
                var += a * a + b * b;
            }
            var += __shfl_xor(var, 32, 64);
            const float rstd = 1.0f / __builtin_sqrtf(var * (1.0f / 64.0f) + p.gn_eps);
            // the gate of the block: registers -> staging tile -> this lane's (query, d) positions
#pragma unroll
            for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                for (int ff = 0; ff < 4; ++ff) {
                    const int row = jl * 16 + frow;
                    *(u32x2*)(Ow + row * 128 + (((ff * 2 + (fkg >> 1)) ^ (row & 7)) << 4) + (fkg & 1) * 8) = gk[ff][jl];
                }
            wave_lds_sync();
            f16x4 gg[2][4];
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) gg[db][g] = *(const f16x4*)(Ow + lq * 128 + (((db * 4 + g) ^ (lq & 7)) << 4) + hi * 8);
            wave_lds_sync();
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = to_f16_sat((float)gg[db][g][r] * (oT[db][g * 4 + r] - mean) * rstd);
                    *(f16x4*)(Ow + lq * 128 + (((db * 4 + g) ^ (lq & 7)) << 4) + hi * 8) = o;
                }
            wave_lds_sync();
            _Float16* __restrict__ Og = (_Float16*)p.O + ((size_t)seq * p.Tp + f0 + qw0) * p.ldo + h * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                if (qw0 + row < n) {
                    const u32x4 v = *(const u32x4*)(Ow + row * 128 + ((ch ^ (row & 7)) << 4));
                    *(u32x4*)(Og + (size_t)row * p.ldo + ch * 8) = v;
                }
            }
            wave_lds_sync();
        }
    }
    RS_STAMP(16);
#ifdef EEND_RS_TRACE
    if (!KV_ONLY && tix < 8 && threadIdx.x == 0) {
        ts[20] = (unsigned long long)upper; ts[21] = (unsigned long long)c; ts[22] = (unsigned long long)seq;
#pragma unroll
        for (int k = 0; k < 32; ++k) g_rs_trace[((size_t)blockIdx.x * 8 + tix) * 32 + k] = ts[k];
    }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA / prefetch may outlive the workgroup
}

int n_cu_cached() {
    int n_cu = eend_cu_count() & ~31;                // multiple of 32: a persistent workgroup keeps its head (and its XCD)
    return n_cu > 0 ? n_cu : 32;
}

}  // namespace

#ifdef EEND_RS_TRACE
extern "C" int eend_debug_ret_stream_trace(void* dst, void* stream) {
    return hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(g_rs_trace), sizeof(g_rs_trace), 0, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}
#endif

long eend_ret_stream_packed_nelems() { return 4L * NITEM * WITEM / 2; }

int eend_launch_ret_stream_pack(const float* W, void* out, hipStream_t stream) {
    if (!W || !out) return EEND_EINVAL;
    hipLaunchKernelGGL(ret_stream_pack_kernel, dim3(160), dim3(256), 0, stream, W, (_Float16*)out);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

bool eend_ret_stream_ok(int L, int Tp, int ldx, int ldo) { return L > 0 && L <= 512 && Tp > 0 && (Tp % 64) == 0 && (ldx & 7) == 0 && (ldo & 7) == 0; }

// pass 1 (kv == true): chunk K^T V products of chunks 0 .. nkv-1 into kv_ws; pass 2: the retention rows
int eend_launch_ret_stream(const RetStreamParams& p, bool kv, hipStream_t stream) {
    if (!eend_ret_stream_ok(p.L, p.Tp, p.ldx, p.ldo) || p.nseq <= 0 || p.nc < 1 || p.nc > (p.Tp + p.L - 1) / p.L || !p.X || !p.W || !p.bias ||
        (long)p.nseq * p.nc * 8 > (1L << 30))
        return EEND_EINVAL;
    if (kv) {
        if (!p.kv_ws || p.nkv < 0 || p.nkv > p.nc) return EEND_EINVAL;
        if (p.nkv == 0) return EEND_OK;
    } else if (!p.O || !p.St || !p.cscale || !p.sexp) return EEND_EINVAL;
    static EendOncePerDevice attr_once[2];
    const void* kern = kv ? (const void*)ret_stream_kernel<true> : (const void*)ret_stream_kernel<false>;
    if (!eend_set_dynamic_lds(attr_once[kv], kern, SMEM)) return EEND_ELAUNCH;
    int nunits;
    if (kv) nunits = p.nkv * p.nseq;
    else {
        int ncB = 0;
        for (int c = 0; c < p.nc; ++c) { int n = p.Tp - c * p.L; n = n < p.L ? n : p.L; if (n > 256) ++ncB; }
        nunits = (ncB + p.nc) * p.nseq;
    }
    const int nitems = nunits * 4, n_cu = n_cu_cached();
    if (kv) hipLaunchKernelGGL(ret_stream_kernel<true>, dim3(nitems < n_cu ? nitems : n_cu), dim3(512), SMEM, stream, p);
    else hipLaunchKernelGGL(ret_stream_kernel<false>, dim3(nitems < n_cu ? nitems : n_cu), dim3(512), SMEM, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
