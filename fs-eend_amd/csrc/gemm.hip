// Y = epilogue(X[M,K] * W[N,K]^T): the dense linear layers of the EEND hot path
// on f16 MFMA (v_mfma_f32_16x16x32_f16, fp32 accumulate).
//
// Both operands are K-contiguous, so both MFMA fragments are one ds_read_b128
// per lane out of XOR-swizzled [rows][64-elem] LDS tiles.  Global -> register
// -> LDS staging is double buffered (loads of k-tile i+1 are in flight while
// k-tile i feeds the matrix pipe, one barrier per k-tile).
//
// SWAP=true feeds W as the MFMA "A" operand, so a lane owns 4 *consecutive
// output features* of one token: row-major outputs are written as 8-byte
// (f16x4) / 16-byte (float4) pieces and per-token LayerNorm / L2-norm
// statistics are lane-local + 2 cross-lane steps.  SWAP=false gives a lane 4
// consecutive *tokens* of one feature, which is exactly the transposed V layout
// ([seq][head][d][t]) the attention kernel wants -- the layout change is fused
// into the producer's epilogue instead of being a kernel of its own.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

// Compile-time loop: f(std::integral_constant<int, I>{}) for I in [0, N).  Register arrays
// indexed through it have constant indices at IR-generation time (a `#pragma unroll` loop
// variable is only constant after unrolling, too late for SROA -> the array lands in scratch).
template <class F, int... I>
DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

enum { ALOAD_PLAIN = 0, ALOAD_CONV = 1 };

// Perf-study ablations (tools/gemm_ablate.py): compiled in only with -DEEND_GEMM_ABLATE.
#ifdef EEND_GEMM_ABLATE
#define EEND_DBG_KT(kt) ((p.dbg & 2) ? 0 : (kt))
#define EEND_DBG_NO_MFMA (p.dbg & 4)
#define EEND_DBG_NO_STORE (p.dbg & 1)
#else
#define EEND_DBG_KT(kt) (kt)
#define EEND_DBG_NO_MFMA 0
#define EEND_DBG_NO_STORE 0
#endif

template <class OT> struct OutCvt;
template <> struct OutCvt<_Float16> {
    static DEV f16x4 cvt(float a, float b, float c, float d) {
        f16x4 r; r[0] = to_f16_sat(a); r[1] = to_f16_sat(b); r[2] = to_f16_sat(c); r[3] = to_f16_sat(d); return r;
    }
};
template <> struct OutCvt<__bf16> {
    static DEV bf16x4 cvt(float a, float b, float c, float d) {
        bf16x4 r; r[0] = (__bf16)a; r[1] = (__bf16)b; r[2] = (__bf16)c; r[3] = (__bf16)d; return r;
    }
};

// Operand element type of a GEMM instantiation: f16 for the forward linears, bf16 for the gradient GEMMs of the
// training step (gradients need bf16's exponent range; same MFMA rate, same fragment layout).
template <class ET> struct Mfma16;
template <> struct Mfma16<_Float16> {
    using V = f16x8;
    static DEV f32x4 run(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<__bf16> {
    using V = bf16x8;
    static DEV f32x4 run(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

template <class ET, int BM, int BN, int WGM, int WGN, bool SWAP, int ALOAD, int EPI, int PF>
__global__ __launch_bounds__(WGM * WGN * 64, (EPI == EPI_RES_LNBWD ? 2 : 1))     // (the LayerNorm-backward epilogue must not cost the second workgroup per CU)
void gemm_f16_kernel(const GemmParams p) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int WR = SWAP ? WN : WM;          // wave extent along MFMA rows (register dim)
    constexpr int WL = SWAP ? WM : WN;          // wave extent along MFMA cols (lane dim)
    constexpr int FR = WR / 16, FL = WL / 16;
    constexpr int XCH = BM * 8 / NT;            // 16-B chunks of the X tile per thread
    constexpr int WCH = BN * 8 / NT;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const int ntn = p.N / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int L = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (L / ntn) * BM;
    const int n0 = (L % ntn) * BN;
    const int nk = p.K >> 6;

    const ET* __restrict__ A = (const ET*)p.A;
    const ET* __restrict__ W = (const ET*)p.W;

    // ---- per-thread staging coordinates --------------------------------------------------
    constexpr int NSET = PF > 0 ? PF : 1;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // first-class vector: trivially SROA'd
    u32x4 xr[NSET][XCH], wr[NSET][WCH];     // PF > 0: all PF == nk k-tiles in flight (straight-line code)
    const ET* xsrc[XCH];
    int xseq_t[XCH];            // conv: frame index t inside its slab
    int xilen[XCH];             // conv: valid length of that sequence
    static_for<XCH>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const int q = tid + i * NT;
        int m = m0 + (q >> 3);
        m = m < p.M ? m : p.M - 1;
        if constexpr (ALOAD == ALOAD_CONV) {
            const int seq = m / p.Tp;
            xseq_t[i] = m - seq * p.Tp;
            xilen[i] = p.ilens[seq];
            xsrc[i] = A + (size_t)seq * p.Tp * p.lda + (q & 7) * 8;
        } else {
            xsrc[i] = A + (size_t)m * p.lda + (q & 7) * 8;
        }
    });
    const ET* wsrc[WCH];
    static_for<WCH>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const int q = tid + i * NT;
        wsrc[i] = W + (size_t)(n0 + (q >> 3)) * p.ldw + (q & 7) * 8;
    });

    // staging helpers: SET / i are compile-time constants (see static_for)
    auto gload = [&](auto SET, int kt) __attribute__((always_inline)) {
        static_for<XCH>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if constexpr (ALOAD == ALOAD_CONV) {
                // implicit GEMM for Conv1d: k-tile kt covers input channels cin0..cin0+63 of tap `tap`;
                // source frame t + tap - pad, zero outside [0, ilen) (reference: truncate to ilen,
                // zero re-pad, then the conv's own zero padding).
                const int kpt = p.conv_cin >> 6;
                const int tap = kt / kpt;
                const int cin0 = (kt - tap * kpt) << 6;
                const int ts = xseq_t[i] + tap - p.conv_pad;
                if (ts >= 0 && ts < xilen[i])
                    xr[decltype(SET)::value][i] = *(const u32x4*)(xsrc[i] + (size_t)ts * p.lda + cin0);
                else
                    xr[decltype(SET)::value][i] = u32x4{0u, 0u, 0u, 0u};
            } else {
                xr[decltype(SET)::value][i] = *(const u32x4*)(xsrc[i] + EEND_DBG_KT(kt) * 64);
            }
        });
        static_for<WCH>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            wr[decltype(SET)::value][i] = *(const u32x4*)(wsrc[i] + EEND_DBG_KT(kt) * 64);
        });
    };
    auto lstore = [&](auto SET, int buf) __attribute__((always_inline)) {
        char* xb = smem + buf * TILE_BYTES;
        char* wb = xb + BM * 128;
        static_for<XCH>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int q = tid + i * NT;
            *(u32x4*)(xb + swz128(q >> 3, q & 7)) = xr[decltype(SET)::value][i];
        });
        static_for<WCH>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int q = tid + i * NT;
            *(u32x4*)(wb + swz128(q >> 3, q & 7)) = wr[decltype(SET)::value][i];
        });
    };

    f32x4 acc[FR][FL];
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // tile-local row offsets of this wave's R (register-dim) and L (lane-dim) operands
    const int r_tile_row0 = SWAP ? wn * WN : wm * WM;     // inside W tile if SWAP else X tile
    const int l_tile_row0 = SWAP ? wm * WM : wn * WN;
    const int frow = lane & 15, fkg = lane >> 4;

#define GEMM_COMPUTE(buf_)                                                                       \
    do {                                                                                         \
        const char* xb__ = smem + (buf_) * TILE_BYTES;                                           \
        const char* wb__ = xb__ + BM * 128;                                                      \
        const char* rb__ = SWAP ? wb__ : xb__;                                                   \
        const char* lb__ = SWAP ? xb__ : wb__;                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                       \
            typename Mfma16<ET>::V rf[FR], lf[FL];                                               \
            _Pragma("unroll") for (int i = 0; i < FR; ++i)                                       \
                rf[i] = *(const typename Mfma16<ET>::V*)(rb__ + swz128(r_tile_row0 + i * 16 + frow, ks * 4 + fkg)); \
            _Pragma("unroll") for (int j = 0; j < FL; ++j)                                       \
                lf[j] = *(const typename Mfma16<ET>::V*)(lb__ + swz128(l_tile_row0 + j * 16 + frow, ks * 4 + fkg)); \
            if (!EEND_DBG_NO_MFMA) {                                                               \
            _Pragma("unroll") for (int i = 0; i < FR; ++i)                                       \
                _Pragma("unroll") for (int j = 0; j < FL; ++j)                                   \
                    acc[i][j] = Mfma16<ET>::run(rf[i], lf[j], acc[i][j]);                        \
            } else { _Pragma("unroll") for (int i = 0; i < FR; ++i) acc[i][0][0] += (float)rf[i][0] + (float)lf[i % FL][0]; } \
        }                                                                                        \
    } while (0)

    if constexpr (PF > 0) {
        // nk == PF (K = 64*PF, e.g. K = 256 -> PF = 4): straight-line code, every k-tile's loads are
        // issued up front -- one L2 round trip for the whole K extent instead of one per k-tile
        // (PMC: the looped form spent 65 % of its wave cycles in s_waitcnt at K = 256).  The kernel is
        // LDS-limited to 2 blocks/CU, so the extra staging registers cost no occupancy.  Tile kt goes
        // to LDS buffer kt&1 right before use; a wave passes barrier(kt) only after every wave has
        // finished compute(kt-1), so buffer (kt+1)&1 is free for the next store.
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        if constexpr (PF == 4) {
            using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
            gload(C0{}, 0); gload(C1{}, 1); gload(C2{}, 2); gload(C3{}, 3);
            lstore(C0{}, 0); __syncthreads(); GEMM_COMPUTE(0);
            lstore(C1{}, 1); __syncthreads(); GEMM_COMPUTE(1);
            lstore(C2{}, 0); __syncthreads(); GEMM_COMPUTE(0);
            lstore(C3{}, 1); __syncthreads(); GEMM_COMPUTE(1);
        } else {
            // PF == 2: two k-tiles in flight (64 staging registers: no AGPR round trips), 4 k-tiles
            static_assert(PF == 2, "straight-line variants: PF = 2 or 4, K = 256");
            gload(C0{}, 0); gload(C1{}, 1);
            lstore(C0{}, 0); __syncthreads(); gload(C0{}, 2); GEMM_COMPUTE(0);
            lstore(C1{}, 1); __syncthreads(); gload(C1{}, 3); GEMM_COMPUTE(1);
            lstore(C0{}, 0); __syncthreads(); GEMM_COMPUTE(0);
            lstore(C1{}, 1); __syncthreads(); GEMM_COMPUTE(1);
        }
    } else {
        // generic K: k-tile kt+1 is in flight (registers) while k-tile kt feeds the matrix pipe; the
        // last k-tile is peeled so the steady-state body has no conditionals.
        using Z = std::integral_constant<int, 0>;
        gload(Z{}, 0);
        lstore(Z{}, 0);
        __syncthreads();
        for (int kt = 0; kt < nk - 1; ++kt) {
            const int buf = kt & 1;
            gload(Z{}, kt + 1);
            GEMM_COMPUTE(buf);
            lstore(Z{}, buf ^ 1);
            __syncthreads();
        }
        GEMM_COMPUTE((nk - 1) & 1);
    }
    __syncthreads();

    // ---- epilogues -----------------------------------------------------------------------
    // acc[i][j][r]: R index = R0 + i*16 + (lane>>4)*4 + r ; L index = L0 + j*16 + (lane&15)
    const int R0 = (SWAP ? n0 + wn * WN : m0 + wm * WM) + fkg * 4;
    const int L0 = (SWAP ? m0 + wm * WM : n0 + wn * WN) + frow;

    if constexpr (EPI == EPI_PLAIN_F16 || EPI == EPI_PLAIN_RELU_F16 || EPI == EPI_PLAIN_SWISH_F16 ||
                  EPI == EPI_QK_HEADS || EPI == EPI_QK_HEADS_F16 || EPI == EPI_VT_HEADS || EPI == EPI_KTVT_HEADS_F16 ||
                  EPI == EPI_PLAIN_BF16 || EPI == EPI_MASK_BF16) {
        // 2-byte outputs go through LDS (free after the main loop) so that global stores are full
        // 256-byte row segments, 16 B per lane.  (Measured: the direct form -- 8 B per lane, 32-byte
        // pieces scattered over 16 rows per instruction -- cost as much as the rest of the kernel.)
        // Staged tile = [L index][R index]: SWAP -> rows are tokens, columns features (row-major
        // outputs, Q/K head rows); !SWAP -> rows are features, columns tokens (the transposed
        // K^T / V^T head layout falls out of the same code).
        constexpr bool IS_BF16 = (EPI == EPI_QK_HEADS || EPI == EPI_VT_HEADS || EPI == EPI_PLAIN_BF16 || EPI == EPI_MASK_BF16);
        using OT = typename std::conditional<IS_BF16, __bf16, _Float16>::type;
        constexpr int BL = SWAP ? BM : BN, BR = SWAP ? BN : BM;
        constexpr int SROW = BR + 8;                                   // +16 B pad: bank spread
        static_assert(BL * SROW * 2 <= 2 * TILE_BYTES, "staging tile must fit in the pipeline LDS");
        OT* stage = (OT*)smem;
#pragma unroll
        for (int i = 0; i < FR; ++i) {
            float4 b = make_float4(0, 0, 0, 0);
            if (SWAP && p.bias) b = *(const float4*)(p.bias + R0 + i * 16);
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                if (!SWAP) { const float bb = p.bias ? p.bias[L0 + j * 16] : 0.f; b = make_float4(bb, bb, bb, bb); }
                float v0 = acc[i][j][0] + b.x, v1 = acc[i][j][1] + b.y, v2 = acc[i][j][2] + b.z, v3 = acc[i][j][3] + b.w;
                if (EPI == EPI_MASK_BF16) { const float a_ = p.drop.scale; v0 *= a_; v1 *= a_; v2 *= a_; v3 *= a_; }
                if (EPI == EPI_PLAIN_RELU_F16) {
                    v0 = __builtin_fmaxf(v0, 0.f); v1 = __builtin_fmaxf(v1, 0.f);
                    v2 = __builtin_fmaxf(v2, 0.f); v3 = __builtin_fmaxf(v3, 0.f);
                    if (p.drop.thresh24) {                  // training: dropout after the activation (FFN `self.dropout`)
                        static_assert(EPI != EPI_PLAIN_RELU_F16 || SWAP, "dropout indices assume SWAP");
                        const unsigned m = (unsigned)(m0 + l_tile_row0 + j * 16 + frow), n = (unsigned)(n0 + r_tile_row0 + i * 16 + fkg * 4);
                        v0 = drop_apply(p.drop, v0, m, n); v1 = drop_apply(p.drop, v1, m, n + 1);
                        v2 = drop_apply(p.drop, v2, m, n + 2); v3 = drop_apply(p.drop, v3, m, n + 3);
                    }
                }
                if (EPI == EPI_PLAIN_SWISH_F16) {
                    v0 = v0 / (1.0f + __expf(-v0)); v1 = v1 / (1.0f + __expf(-v1));
                    v2 = v2 / (1.0f + __expf(-v2)); v3 = v3 / (1.0f + __expf(-v3));
                }
                const int lrow = l_tile_row0 + j * 16 + frow, rcol = r_tile_row0 + i * 16 + fkg * 4;
                *(decltype(OutCvt<OT>::cvt(0, 0, 0, 0))*)(stage + lrow * SROW + rcol) = OutCvt<OT>::cvt(v0, v1, v2, v3);
            }
        }
        __syncthreads();
        const int Lbase = SWAP ? m0 : n0, Rbase = SWAP ? n0 : m0;
        constexpr int CPR = BR / 8;                                    // 16-B chunks per staged row
        const int D = p.H * p.dh;
#pragma unroll
        for (int it = 0; it < BL * CPR / NT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / CPR, ch = idx % CPR;
            uint4 v = *(const uint4*)(stage + row * SROW + ch * 8);
            const int li = Lbase + row, ri = Rbase + ch * 8;
            OT* dst;
            if constexpr (EPI == EPI_PLAIN_F16 || EPI == EPI_PLAIN_RELU_F16 || EPI == EPI_PLAIN_SWISH_F16 ||
                          EPI == EPI_PLAIN_BF16 || EPI == EPI_MASK_BF16) {
                if (li >= p.M) continue;
                dst = (OT*)p.out16 + (size_t)li * p.ldo + ri;
                if constexpr (EPI == EPI_MASK_BF16) {
                    // ReLU backward: keep the gradient where the saved forward activation (f16 or bf16, any
                    // 2-byte float: zero <=> no magnitude bits) is non-zero
                    const uint4 mk = *(const uint4*)((const unsigned short*)p.mask + (size_t)li * p.ldmask + ri);
                    auto keep = [](unsigned g, unsigned m) {
                        return g & (((m & 0x7FFFu) ? 0xFFFFu : 0u) | ((m & 0x7FFF0000u) ? 0xFFFF0000u : 0u));
                    };
                    v.x = keep(v.x, mk.x); v.y = keep(v.y, mk.y); v.z = keep(v.z, mk.z); v.w = keep(v.w, mk.w);
                }
            } else if constexpr (EPI == EPI_QK_HEADS || EPI == EPI_QK_HEADS_F16) {
                // li = token, ri = feature in [0, 2D): Q then K, [which][seq][H][Tp][dh]
                if (li >= p.M) continue;
                const int which = ri / D, nn = ri - which * D;
                const int h = nn / p.dh, dd = nn - h * p.dh;
                const int seq = li / p.Tp, t = li - seq * p.Tp;
                dst = (OT*)(which ? p.out16b : p.out16) + (((size_t)seq * p.H + h) * p.Tp + t) * p.dh + dd;
            } else {
                // li = feature (V, or K then V), ri = first of 8 consecutive tokens: [seq][H][dh][Tp]
                if (ri >= p.M) continue;
                const int which = li / D, nn = li - which * D;
                const int h = nn / p.dh, dd = nn - h * p.dh;
                const int seq = ri / p.Tp, t = ri - seq * p.Tp;
                dst = (OT*)(which ? p.out16b : p.out16) + (((size_t)seq * p.H + h) * p.dh + dd) * p.Tp + t;
            }
            if (!EEND_DBG_NO_STORE) *(uint4*)dst = v;
        }
    } else if constexpr (EPI == EPI_GLU_F16) {
        // W rows interleaved (2n = value_n, 2n+1 = gate_n): a lane's 4 consecutive rows are two
        // (value, gate) pairs -> out16[m][n], out16[m][n+1] (GLU over channels, activation.py:39-41)
        static_assert(SWAP, "GLU epilogue expects SWAP");
        _Float16* __restrict__ out = (_Float16*)p.out16;
#pragma unroll
        for (int i = 0; i < FR; ++i) {
            const int r = R0 + i * 16;
            const float4 b = *(const float4*)(p.bias + r);
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                const int m = L0 + j * 16;
                if (m >= p.M) continue;
                const float a0 = acc[i][j][0] + b.x, g0 = acc[i][j][1] + b.y;
                const float a1 = acc[i][j][2] + b.z, g1 = acc[i][j][3] + b.w;
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                f16x2 o;
                o[0] = to_f16_sat(a0 / (1.0f + __expf(-g0)));
                o[1] = to_f16_sat(a1 / (1.0f + __expf(-g1)));
                *(f16x2*)(out + (size_t)m * p.ldo + (r >> 1)) = o;
            }
        }
    } else if constexpr (EPI == EPI_RES_LN || EPI == EPI_L2NORM || EPI == EPI_RES_SCALE || EPI == EPI_RES_SCALE_LN16 ||
                         EPI == EPI_RES_LN_TRAIN || EPI == EPI_L2NORM_TRAIN || EPI == EPI_RES_SCALE_LN16_TRAIN || EPI == EPI_RES_LNBWD) {
        // The block owns complete rows (BN == N, WGM == 1): per-token statistics.
        static_assert(SWAP && WGM == 1, "row-stat epilogues expect SWAP and one wave row");
        constexpr bool LNK = (EPI == EPI_RES_LN || EPI == EPI_RES_LN_TRAIN || EPI == EPI_RES_SCALE_LN16 ||
                              EPI == EPI_RES_SCALE_LN16_TRAIN);                                             // LayerNorm
        constexpr bool PRENORM = (EPI == EPI_RES_SCALE_LN16 || EPI == EPI_RES_SCALE_LN16_TRAIN);  // out32 stays un-normalised
        constexpr bool L2K = (EPI == EPI_L2NORM || EPI == EPI_L2NORM_TRAIN);                               // x / ||x||
        constexpr bool TRAINK = (EPI == EPI_RES_LN_TRAIN || EPI == EPI_L2NORM_TRAIN || EPI == EPI_RES_SCALE_LN16_TRAIN);  // also save the row statistics
        using O16 = decltype(OutCvt<ET>::cvt(0, 0, 0, 0));
        float* red = (float*)smem;                        // [WGN][BM] (main loop is done with LDS)
        float4 bias4[FR];
#pragma unroll
        for (int i = 0; i < FR; ++i)
            bias4[i] = p.bias ? *(const float4*)(p.bias + R0 + i * 16) : make_float4(0, 0, 0, 0);
        // v = acc + bias (+ residual)
#pragma unroll
        for (int j = 0; j < FL; ++j) {
            const int m = L0 + j * 16;
            const bool ok = m < p.M;
#pragma unroll
            for (int i = 0; i < FR; ++i) {
                float4 r = make_float4(0, 0, 0, 0);
                if (!L2K && ok) {
                    if (p.res) r = *(const float4*)(p.res + (size_t)m * p.ldo + R0 + i * 16);
                    else if (p.res16) {
                        const f16x4 h = *(const f16x4*)((const _Float16*)p.res16 + (size_t)m * p.ldo + R0 + i * 16);
                        r = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
                    }
                }
                const float s = L2K ? 1.0f : p.alpha;
                float a0 = acc[i][j][0] + bias4[i].x, a1 = acc[i][j][1] + bias4[i].y, a2 = acc[i][j][2] + bias4[i].z, a3 = acc[i][j][3] + bias4[i].w;
                if constexpr (EPI == EPI_RES_LN_TRAIN || EPI == EPI_RES_SCALE_LN16_TRAIN) {
                    if (p.drop.thresh24) {                  // dropout of the sub-layer output (dropout1 / dropout2 / dropout11 / ...)
                        const unsigned n = (unsigned)(R0 + i * 16);
                        a0 = drop_apply(p.drop, a0, (unsigned)m, n); a1 = drop_apply(p.drop, a1, (unsigned)m, n + 1);
                        a2 = drop_apply(p.drop, a2, (unsigned)m, n + 2); a3 = drop_apply(p.drop, a3, (unsigned)m, n + 3);
                    }
                }
                acc[i][j][0] = a0 * s + r.x;
                acc[i][j][1] = a1 * s + r.y;
                acc[i][j][2] = a2 * s + r.z;
                acc[i][j][3] = a3 * s + r.w;
            }
        }
        float mean[FL], scale[FL];
        auto block_rowsum = [&](float (&part)[FL]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                part[j] = wave_xor_add(part[j], 16);
                part[j] = wave_xor_add(part[j], 32);
            }
            __syncthreads();
            if (fkg == 0) {
#pragma unroll
                for (int j = 0; j < FL; ++j) red[wn * BM + j * 16 + frow] = part[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WGN; ++w) s += red[w * BM + j * 16 + frow];
                part[j] = s;
            }
        };
        // EPI_RES_LNBWD: the LayerNorm backward of the post-norm site in front of the branch, in the epilogue (train_rows.hip ln_bwd_kernel,
        // same algebra).  acc = g, the gradient w.r.t. the LayerNorm output (the data-gradient GEMM's result + the incoming stream):
        //   dz = rstd * (g gamma - mean(g gamma) - x_hat mean(g gamma x_hat))  -> out32 (the residual stream's gradient, unmasked)
        //   out16 = bf16 of dz under the sub-layer output's dropout mask (the branch gradient)
        //   colpart[m-tile][3][256] = sums over the tile's rows of g x_hat | g | the masked dz (d gamma, d beta, d bias partials; this wave's
        //   64 columns, the 16 row lanes by DPP), summed over the tiles in fixed order by wgrad_reduce_multi_kernel
        // (x_hat is read twice -- the second time from L1 / L2 -- and the column sums leave per feature fragment: the version that kept them
        // in registers needed 344 and lost the second workgroup per CU)
        constexpr bool LNB = EPI == EPI_RES_LNBWD;
        float* __restrict__ cp = LNB ? p.colpart + (size_t)(m0 / BM) * 3 * 256 : nullptr;
        if constexpr (LNB) {
            const float invN = 1.0f / (float)p.N;
            float p1[FL], p2[FL];
#pragma unroll
            for (int j = 0; j < FL; ++j) { p1[j] = 0.f; p2[j] = 0.f; }
#pragma unroll
            for (int i = 0; i < FR; ++i) {
                const float4 gam = *(const float4*)(p.gamma + R0 + i * 16);
                const float gk[4] = {gam.x, gam.y, gam.z, gam.w};
                float cgx[4] = {0.f, 0.f, 0.f, 0.f}, cg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < FL; ++j) {
                    const int m = L0 + j * 16;
                    const bool ok = m < p.M;
                    f16x4 h = f16x4{(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
                    if (ok) h = *(const f16x4*)((const _Float16*)p.xhat16 + (size_t)m * p.ldo + R0 + i * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float g_ = ok ? acc[i][j][r] : 0.f, x_ = (float)h[r], d_ = g_ * gk[r];
                        acc[i][j][r] = d_;                     // g gamma
                        p1[j] += d_; p2[j] += d_ * x_;
                        cg[r] += g_; cgx[r] += g_ * x_;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = row16_allreduce_add(cgx[r]), b = row16_allreduce_add(cg[r]);
                    if (frow == 0) { cp[R0 + i * 16 + r - n0] = a; cp[256 + R0 + i * 16 + r - n0] = b; }
                }
            }
            // both row sums in ONE exchange (red: [2][WGN][BM] floats = the 2 KB in front of the staging tile); the rows' 1/sigma travel under it
            float rsv[FL];
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                const int m = L0 + j * 16;
                rsv[j] = m < p.M ? p.rstat[m] : 0.f;
                p1[j] = wave_xor_add(p1[j], 16); p1[j] = wave_xor_add(p1[j], 32);
                p2[j] = wave_xor_add(p2[j], 16); p2[j] = wave_xor_add(p2[j], 32);
            }
            __syncthreads();
            if (fkg == 0) {
#pragma unroll
                for (int j = 0; j < FL; ++j) { red[wn * BM + j * 16 + frow] = p1[j]; red[(WGN + wn) * BM + j * 16 + frow] = p2[j]; }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WGN; ++w) { s1 += red[w * BM + j * 16 + frow]; s2 += red[(WGN + w) * BM + j * 16 + frow]; }
                p1[j] = s1; p2[j] = s2;
            }
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                const int m = L0 + j * 16;
                const bool ok = m < p.M;
                const float rs = rsv[j];
                const float c1 = p1[j] * invN, c2 = p2[j] * invN;
#pragma unroll
                for (int i = 0; i < FR; ++i) {
                    f16x4 h = f16x4{(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
                    if (ok) h = *(const f16x4*)((const _Float16*)p.xhat16 + (size_t)m * p.ldo + R0 + i * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = rs * (acc[i][j][r] - c1 - (float)h[r] * c2);
                }
                mean[j] = 0.f; scale[j] = 1.f;
            }
        } else if constexpr (EPI == EPI_RES_SCALE) {
#pragma unroll
            for (int j = 0; j < FL; ++j) { mean[j] = 0.f; scale[j] = 1.f; }
        } else {
            const float invN = 1.0f / (float)p.N;
            float part[FL];
            if constexpr (LNK) {
#pragma unroll
                for (int j = 0; j < FL; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < FR; ++i) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
                    part[j] = s;
                }
                block_rowsum(part);
#pragma unroll
                for (int j = 0; j < FL; ++j) mean[j] = part[j] * invN;
            } else {
#pragma unroll
                for (int j = 0; j < FL; ++j) mean[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < FR; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[j]; s += d * d; }
                part[j] = s;
            }
            block_rowsum(part);
#pragma unroll
            for (int j = 0; j < FL; ++j)
                scale[j] = L2K ? 1.0f / __builtin_sqrtf(part[j]) : 1.0f / __builtin_sqrtf(part[j] * invN + p.eps);
        }
        float* __restrict__ o32 = (float*)p.out32;
        ET* __restrict__ o16 = (ET*)p.out16;
        if constexpr (TRAINK) {
            // training forward: 1/sigma (LayerNorm) or 1/||x|| (L2 norm) of every row, for the backward kernels
            if (p.rstat && wn == 0 && fkg == 0) {
#pragma unroll
                for (int j = 0; j < FL; ++j)
                    if (L0 + j * 16 < p.M) p.rstat[L0 + j * 16] = scale[j];
            }
        }
        if constexpr (BM == 64 && BN == 256 && WGN == 4 && 2 * TILE_BYTES >= 2048 + BM * BN * 4) {
            // Staged stores (as ffn.hip): the direct form writes 64-byte (fp32) / 32-byte (f16) pieces of 16 rows per
            // instruction -- PMC WRITE_SIZE showed 1.5x the algorithmic bytes for this kernel.  Through LDS every
            // store instruction writes one whole 1 KB row (fp32) or two 512 B rows (f16).
            char* stg = smem + 2048;                          // clear of the row-statistics scratch
            if constexpr (EPI == EPI_RES_SCALE || LNB) __syncthreads();   // (no statistics pass / its scratch was just read: fence before the staging writes)
            O16 h16[LNB ? 1 : FR][LNB ? 1 : FL];
            O16 xh16[TRAINK && LNK ? FR : 1][TRAINK && LNK ? FL : 1];
#pragma unroll
            for (int i = 0; i < FR; ++i) {
                const int n = R0 + i * 16, nl = n - n0;
                float4 g = make_float4(1, 1, 1, 1), be = make_float4(0, 0, 0, 0);
                if (LNK && p.gamma) { g = *(const float4*)(p.gamma + n); be = *(const float4*)(p.beta + n); }
#pragma unroll
                for (int j = 0; j < FL; ++j) {
                    const int row = j * 16 + frow;
                    f32x4 v;
                    v[0] = (acc[i][j][0] - mean[j]) * scale[j] * g.x + be.x;
                    v[1] = (acc[i][j][1] - mean[j]) * scale[j] * g.y + be.y;
                    v[2] = (acc[i][j][2] - mean[j]) * scale[j] * g.z + be.z;
                    v[3] = (acc[i][j][3] - mean[j]) * scale[j] * g.w + be.w;
                    if constexpr (!LNB) h16[i][j] = OutCvt<ET>::cvt(v[0], v[1], v[2], v[3]);      // (LNB: made in the 2-byte staging pass below)
                    if constexpr (TRAINK && LNK)            // normalised, pre-affine row: what LayerNorm backward needs
                        xh16[i][j] = OutCvt<ET>::cvt((acc[i][j][0] - mean[j]) * scale[j], (acc[i][j][1] - mean[j]) * scale[j],
                                                     (acc[i][j][2] - mean[j]) * scale[j], (acc[i][j][3] - mean[j]) * scale[j]);
                    *(f32x4*)(stg + row * 1024 + (((nl >> 2) ^ (row & 7)) << 4)) = PRENORM ? acc[i][j] : v;
                }
            }
            __syncthreads();
            if (o32) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int row = k * 4 + wave;
                    const f32x4 v = *(const f32x4*)(stg + row * 1024 + ((lane ^ (row & 7)) << 4));
                    if (m0 + row < p.M) *(f32x4*)(o32 + (size_t)(m0 + row) * p.ldo + n0 + lane * 4) = v;
                }
            }
            if (o16) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < FR; ++i) {
                    const int nl = R0 + i * 16 - n0;
                    float cds[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < FL; ++j) {
                        const int row = j * 16 + frow;
                        O16 hv;
                        if constexpr (LNB) {                 // the branch gradient carries the forward's (row, column) dropout mask
                            const unsigned mm = (unsigned)(L0 + j * 16), nn = (unsigned)(R0 + i * 16);
                            const float k0 = drop_apply(p.drop, acc[i][j][0], mm, nn), k1 = drop_apply(p.drop, acc[i][j][1], mm, nn + 1);
                            const float k2 = drop_apply(p.drop, acc[i][j][2], mm, nn + 2), k3 = drop_apply(p.drop, acc[i][j][3], mm, nn + 3);
                            hv = OutCvt<ET>::cvt(k0, k1, k2, k3);
                            if (L0 + j * 16 < p.M) { cds[0] += k0; cds[1] += k1; cds[2] += k2; cds[3] += k3; }
                        } else {
                            hv = h16[i][j];
                        }
                        *(O16*)(stg + row * 512 + (((nl >> 3) ^ ((row >> 1) & 7)) << 4) + ((nl >> 2) & 1) * 8) = hv;
                    }
                    if constexpr (LNB) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float c = row16_allreduce_add(cds[r]);
                            if (frow == 0) cp[512 + nl + r] = c;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = (k * 4 + wave) * 2 + (lane >> 5), c = lane & 31;
                    const u32x4 v = *(const u32x4*)(stg + row * 512 + ((c ^ ((row >> 1) & 7)) << 4));
                    if (m0 + row < p.M) *(u32x4*)(o16 + (size_t)(m0 + row) * p.ldo + n0 + c * 8) = v;
                }
            }
            if constexpr (TRAINK && LNK) {
                if (p.xhat16) {                                     // third staged tile: x_hat (same 2-byte staging as out16)
                    ET* __restrict__ xo = (ET*)p.xhat16;
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < FR; ++i) {
                        const int nl = R0 + i * 16 - n0;
#pragma unroll
                        for (int j = 0; j < FL; ++j) {
                            const int row = j * 16 + frow;
                            *(O16*)(stg + row * 512 + (((nl >> 3) ^ ((row >> 1) & 7)) << 4) + ((nl >> 2) & 1) * 8) = xh16[i][j];
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int row = (k * 4 + wave) * 2 + (lane >> 5), c = lane & 31;
                        const u32x4 v = *(const u32x4*)(stg + row * 512 + ((c ^ ((row >> 1) & 7)) << 4));
                        if (m0 + row < p.M) *(u32x4*)(xo + (size_t)(m0 + row) * p.ldo + n0 + c * 8) = v;
                    }
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < FR; ++i) {
            const int n = R0 + i * 16;
            float4 g = make_float4(1, 1, 1, 1), be = make_float4(0, 0, 0, 0);
            if (LNK && p.gamma) { g = *(const float4*)(p.gamma + n); be = *(const float4*)(p.beta + n); }
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                const int m = L0 + j * 16;
                if (m >= p.M) continue;
                const float v0 = (acc[i][j][0] - mean[j]) * scale[j] * g.x + be.x;
                const float v1 = (acc[i][j][1] - mean[j]) * scale[j] * g.y + be.y;
                const float v2 = (acc[i][j][2] - mean[j]) * scale[j] * g.z + be.z;
                const float v3 = (acc[i][j][3] - mean[j]) * scale[j] * g.w + be.w;
                if (PRENORM) {                             // residual stream stays un-normalised
                    if (o32) *(float4*)(o32 + (size_t)m * p.ldo + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                } else if (o32) *(float4*)(o32 + (size_t)m * p.ldo + n) = make_float4(v0, v1, v2, v3);
                if (o16) *(O16*)(o16 + (size_t)m * p.ldo + n) = OutCvt<ET>::cvt(v0, v1, v2, v3);
            }
        }
    } else if constexpr (EPI == EPI_F32_ROWMASK) {
        // Conv1d data gradient (implicit GEMM over (tap, c_out)): f32 rows, zero for frames t >= mask_lens[seq]
        // (the reference truncates the encoder output to ilen before the conv, FS model :38-39, so no gradient
        // reaches frames beyond it).
        static_assert(SWAP, "row-mask epilogue expects SWAP");
        float* __restrict__ o32 = (float*)p.out32;
#pragma unroll
        for (int j = 0; j < FL; ++j) {
            const int m = L0 + j * 16;
            if (m >= p.M) continue;
            const int seq = m / p.Tp, t = m - seq * p.Tp;
            const bool live = t < p.mask_lens[seq];
#pragma unroll
            for (int i = 0; i < FR; ++i) {
                const int n = R0 + i * 16;
                *(float4*)(o32 + (size_t)m * p.ldo + n) = live ? make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3])
                                                             : make_float4(0, 0, 0, 0);
            }
        }
    } else if constexpr (EPI == EPI_CONVERT) {
        // attr0[(b,c), t, :] = acc[(b,t), :] + pc[c, :]   (FS model :113-114 factored:
        // convert([emb; pe_c]) = W[:, :D] emb + (W[:, D:] pe_c + b)); rows fan out to the
        // decoder's (sequence = (b,c)) slab layout.
        static_assert(SWAP, "convert epilogue expects SWAP");
        float* __restrict__ o32 = (float*)p.out32;
        _Float16* __restrict__ o16 = (_Float16*)p.out16;
#pragma unroll
        for (int j = 0; j < FL; ++j) {
            const int m = L0 + j * 16;
            if (m >= p.M) continue;
            const int b = m / p.Tp, t = m - b * p.Tp;
            for (int c = 0; c < p.C; ++c) {
                const size_t row = ((size_t)b * p.C + c) * p.Tp + t;
#pragma unroll
                for (int i = 0; i < FR; ++i) {
                    const int n = R0 + i * 16;
                    const float4 pc = *(const float4*)(p.pc + (size_t)c * p.N + n);
                    const float v0 = acc[i][j][0] + pc.x, v1 = acc[i][j][1] + pc.y, v2 = acc[i][j][2] + pc.z, v3 = acc[i][j][3] + pc.w;
                    if (o32) *(float4*)(o32 + row * p.ldo + n) = make_float4(v0, v1, v2, v3);
                    *(f16x4*)(o16 + row * p.ldo + n) = OutCvt<_Float16>::cvt(v0, v1, v2, v3);
                }
            }
        }
    }
}

template <class ET, int BM, int BN, int WGM, int WGN, bool SWAP, int ALOAD, int EPI, int PF>
int launch_pf(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 2 * (BM + BN) * 128;
    static EendOncePerDevice attr_once;
    auto kern = gemm_f16_kernel<ET, BM, BN, WGM, WGN, SWAP, ALOAD, EPI, PF>;
    if (!eend_set_dynamic_lds(attr_once, (const void*)kern, smem)) return EEND_ELAUNCH;
    const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
    hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(WGM * WGN * 64), smem, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// K == 256 (4 k-tiles) takes the straight-line full-K variant when the tile's staging registers
// fit (128x128: 8 uint4 per k-tile -> 128 VGPRs); everything else the looped variant.
template <class ET, int BM, int BN, int WGM, int WGN, bool SWAP, int ALOAD, int EPI>
int launch(const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return EEND_EINVAL;
    if (p.N % BN != 0 || p.K % 64 != 0 || (p.lda & 7) || (p.ldw & 7)) return EEND_EINVAL;
    if constexpr (BM + BN <= 256 && ALOAD == ALOAD_PLAIN) {
        if (p.K == 256) return launch_pf<ET, BM, BN, WGM, WGN, SWAP, ALOAD, EPI, 2>(p, stream);
    }
    return launch_pf<ET, BM, BN, WGM, WGN, SWAP, ALOAD, EPI, 0>(p, stream);
}

// (128-row tile instantiations for the K >= 1024 whole-row epilogues of the training step were measured slower than the 64-row tiles --
// same box, round 5: [196608, 256, 2048] forward 0.539 vs 0.464 ms, dX 0.428 vs 0.361 ms; 444 registers = one wave per SIMD, no second
// workgroup to hide the k-tile barrier -- and removed: profiles/OPTIMISATION_LOG.md, profiles/r05_c_train_bm{0,1}.json)

}  // namespace

int eend_launch_gemm(const GemmParams& p, int epi, hipStream_t stream) {
    using H = _Float16;
    using B = __bf16;
    if (p.bf16) {                      // gradient GEMMs of the training step: bf16 operands
        switch (epi) {
            case EPI_PLAIN_BF16:  return launch<B, 128, 128, 2, 2, true, ALOAD_PLAIN, EPI_PLAIN_BF16>(p, stream);
            case EPI_MASK_BF16:
                if (!p.mask || (p.ldmask & 7)) return EEND_EINVAL;
                return launch<B, 128, 128, 2, 2, true, ALOAD_PLAIN, EPI_MASK_BF16>(p, stream);
            case EPI_RES_SCALE:
                if (p.N != 256) return EEND_EINVAL;
                return launch<B, 64, 256, 1, 4, true, ALOAD_PLAIN, EPI_RES_SCALE>(p, stream);
            case EPI_RES_LNBWD:
                if (p.N != 256 || p.ldo != 256 || !p.res || !p.out32 || !p.out16 || !p.xhat16 || !p.rstat || !p.gamma || !p.colpart) return EEND_EINVAL;
                return launch<B, 64, 256, 1, 4, true, ALOAD_PLAIN, EPI_RES_LNBWD>(p, stream);
            case EPI_F32_ROWMASK:
                if (p.N != 256 || !p.ilens || !p.mask_lens || !p.out32) return EEND_EINVAL;
                return launch<B, 64, 256, 1, 4, true, ALOAD_CONV, EPI_F32_ROWMASK>(p, stream);
            default: return EEND_EINVAL;
        }
    }
    switch (epi) {
        case EPI_PLAIN_F16:      return launch<H, 128, 128, 2, 2, true, ALOAD_PLAIN, EPI_PLAIN_F16>(p, stream);
        case EPI_PLAIN_RELU_F16: return launch<H, 128, 128, 2, 2, true, ALOAD_PLAIN, EPI_PLAIN_RELU_F16>(p, stream);
        case EPI_PLAIN_SWISH_F16:return launch<H, 128, 128, 2, 2, true, ALOAD_PLAIN, EPI_PLAIN_SWISH_F16>(p, stream);
        case EPI_GLU_F16:        return launch<H, 128, 128, 2, 2, true, ALOAD_PLAIN, EPI_GLU_F16>(p, stream);
        case EPI_QK_HEADS_F16:   return launch<H, 128, 128, 2, 2, true, ALOAD_PLAIN, EPI_QK_HEADS_F16>(p, stream);
        case EPI_KTVT_HEADS_F16: return launch<H, 128, 128, 2, 2, false, ALOAD_PLAIN, EPI_KTVT_HEADS_F16>(p, stream);
        case EPI_QK_HEADS:       return launch<H, 128, 128, 2, 2, true, ALOAD_PLAIN, EPI_QK_HEADS>(p, stream);
        case EPI_VT_HEADS:       return launch<H, 128, 128, 2, 2, false, ALOAD_PLAIN, EPI_VT_HEADS>(p, stream);
        case EPI_RES_LN:
            if (p.N != 256) return EEND_EINVAL;
            return launch<H, 64, 256, 1, 4, true, ALOAD_PLAIN, EPI_RES_LN>(p, stream);
        case EPI_RES_LN_TRAIN:
            if (p.N != 256) return EEND_EINVAL;
            return launch<H, 64, 256, 1, 4, true, ALOAD_PLAIN, EPI_RES_LN_TRAIN>(p, stream);
        case EPI_RES_SCALE_LN16_TRAIN:
            if (p.N != 256 || p.ldo != 256) return EEND_EINVAL;
            return launch<H, 64, 256, 1, 4, true, ALOAD_PLAIN, EPI_RES_SCALE_LN16_TRAIN>(p, stream);
        case EPI_RES_SCALE_LN16:
            if (p.N != 256) return EEND_EINVAL;
            return launch<H, 64, 256, 1, 4, true, ALOAD_PLAIN, EPI_RES_SCALE_LN16>(p, stream);
        case EPI_RES_SCALE:
            if (p.N != 256) return EEND_EINVAL;
            return launch<H, 64, 256, 1, 4, true, ALOAD_PLAIN, EPI_RES_SCALE>(p, stream);
        case EPI_L2NORM:
            if (p.N != 256) return EEND_EINVAL;
            return launch<H, 64, 256, 1, 4, true, ALOAD_CONV, EPI_L2NORM>(p, stream);
        case EPI_L2NORM_TRAIN:
            if (p.N != 256) return EEND_EINVAL;
            return launch<H, 64, 256, 1, 4, true, ALOAD_CONV, EPI_L2NORM_TRAIN>(p, stream);
        case EPI_CONVERT:
            if (p.N != 256) return EEND_EINVAL;
            return launch<H, 64, 256, 1, 4, true, ALOAD_PLAIN, EPI_CONVERT>(p, stream);
        default: return EEND_EINVAL;
    }
}
