// Speaker-axis self-attention, input side, in ONE launch:
//     qkv = x W_in^T + b_in  (N = 768)   and   o = softmax(q k^T / sqrt(64)) v  over the C slots of a frame
// i.e. `self_attn2`'s in-projection + attention core of the fusion layers (_sa_block2: FS
// merge_tfm_encoder.py:388-394, LS merge_retnet_layer.py:301-306).  As two launches the [M][768] f16 qkv
// tensor is written and read back through HBM (2 x 302 MB per layer at B=64, C=6, T=500) by two kernels
// that are both bound by exactly that traffic; here it only ever exists as three 16 KB LDS tiles.
//
// A tile is "all C slots of G = 128/C consecutive frames of one utterance" (local row = c*G + t'): rows of
// the (b,c)-major slab are gathered with stride Tp, so the speaker mix is tile-local.  Per tile
// (persistent, 1 block/CU, 512 threads, 128 KB LDS):
//   * X rows -> swizzled staging tile -> each wave keeps its 32-token x 256 fragments in registers;
//   * 12 weight slices of 64 output features (per head: q, k, v) stream by LDS-DMA into a double buffer
//     (swz128 image via permuted source addresses; buffer_load...lds, see ffn.hip on why not
//     global_load_lds); each slice is one 128 x 64 x 256 MFMA pass whose f16 result lands in LDS;
//   * after a head's v slice: 4 threads per (frame, query slot) compute the C scores (fp32, quad DPP
//     reduce), softmax, and P.V for their 16 of the 64 head dims, and store 32 B of the output row.
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;

template <class F, int... I>
DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int BM = 128;
constexpr int KD = 256;
constexpr int FC = 64;                    // output features per weight slice (= one head's q, k or v)
constexpr int NT = 512;
constexpr int WS_BYTES = 4 * FC * 128;    // 32 KB: 4 k-tiles of [64 rows][128 B]
constexpr int L_W = 0;                    // 2 x 32 KB weight double buffer
constexpr int L_X = 2 * WS_BYTES;         // 64 KB: X staging, afterwards the q / k / v tiles (3 x 16 KB)
constexpr int S_BYTES = BM * 128;
constexpr int SMEM_BYTES = L_X + 4 * BM * 128;   // 131072

DEV float quad_allreduce_add(float v) {
    int t;
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, t);
    return v;
}

// Perf-study build (-DEEND_SPK_TRACE, tools/spk_trace.py): s_memtime stamps of the tile phases of thread 0 of every workgroup,
// read back through eend_debug_spk_trace; never defined in the shipped library.
#ifdef EEND_SPK_TRACE
__device__ unsigned long long g_spk_trace[256 * 16 * 12];
#define SPK_STAMP(k)                                                                                              \
    do {                                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 256 && (tile - (int)blockIdx.x) / (int)gridDim.x < 16)              \
            g_spk_trace[((size_t)blockIdx.x * 16 + (tile - (int)blockIdx.x) / (int)gridDim.x) * 12 + (k)] =      \
                __builtin_amdgcn_s_memtime();                                                                     \
    } while (0)
#else
#define SPK_STAMP(k) do {} while (0)
#endif

template <int C>
__global__ __launch_bounds__(NT)
void spk_qkv_attn_kernel(const SpkFusedParams p) {
    constexpr int G = BM / C;                 // frames per tile
    constexpr int ROWS = G * C;               // tile rows in use (local row = c * G + t')
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_per_b = (p.Tv + G - 1) / G;          // only the Tv valid frames of a slab: at T = 500, Tp = 512, C = 6 that is
    const int ntiles = p.B * tiles_per_b;                 // 24 instead of 25 tiles per utterance -- 1536 = 6 x 256 CUs, no 7th round
    const _Float16* __restrict__ X = (const _Float16*)p.X;
    _Float16* __restrict__ O = (_Float16*)p.O;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_b, t0 = (tile - b * tiles_per_b) * G;
        if (tile != (int)blockIdx.x)          // the previous tile's LDS reads are retired before anything lands
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        SPK_STAMP(0);
        int tid = threadIdx.x;                // laundered per tile: see ffn.hip
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int frow = lane & 15, fkg = lane >> 4;
        const int gm = (wave >> 1) * 32, gn = (wave & 1) * 32;     // wave tile: 32 tokens x 32 features

        // ---- weight slices by LDS-DMA: slice ch = 3*head + {0:q, 1:k, 2:v} = rows s*256 + head*64 .. +64 of W_in
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 3 * KD * KD * 2, 0x00020000);
        const int drow = lane >> 3, dslot = lane & 7;
        int vo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;                       // 32 pieces of 8 rows x 128 B
            const int kt = piece >> 3, row = (piece & 7) * 8 + drow;
            vo[i] = (row * KD + kt * 64 + (dslot ^ ((row >> 1) & 7)) * 8) * 2;
        }
        auto dma_w = [&](int ch, int buf) __attribute__((always_inline)) {
            const int wrow = (ch % 3) * KD + (ch / 3) * FC;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_char*)(smem + L_W + buf * WS_BYTES + (wave * 4 + i) * 1024), 16, vo[i],
                                                         wrow * KD * 2, 0, 0);
        };
        dma_w(0, 0);
        dma_w(1, 1);

        // ---- gather the tile's X rows (stride Tp between slots) into the swizzled staging tile
        char* Xst = smem + L_X;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = tid + i * NT;
            const int row = q >> 5, c32 = q & 31;
            const int rr = row < ROWS ? row : ROWS - 1;
            const int cs = rr / G, tt = rr - cs * G;
            int t = t0 + tt;
            t = t < p.Tv ? t : p.Tv - 1;
            const size_t grow = ((size_t)b * C + cs) * p.Tp + t;
            const u32x4 v = *(const u32x4*)(X + grow * p.ldx + c32 * 8);
            *(u32x4*)(Xst + (c32 >> 3) * (BM * 128) + swz128(row, c32 & 7)) = v;
        }
        __syncthreads();
        SPK_STAMP(1);
        f16x8 xf[4][2][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    xf[kt][ks][j] = *(const f16x8*)(Xst + kt * (BM * 128) + swz128(gm + j * 16 + frow, ks * 4 + fkg));
        __syncthreads();                      // staging tile is free: it becomes the q / k / v tiles
        SPK_STAMP(2);

        // attention of one head on the three staged tiles
        const int item = tid >> 2, part = tid & 3;                 // 4 threads per (slot, frame): 16 head dims each
        const int cq = item / G, tq = item - cq * G;
        auto attention = [&](int head) __attribute__((always_inline)) {
            if (item >= ROWS) return;
            const char* Sq = smem + L_X;
            const char* Sk = Sq + S_BYTES;
            const char* Sv = Sk + S_BYTES;
            const f16x8 q0 = *(const f16x8*)(Sq + swz128(item, 2 * part));
            const f16x8 q1 = *(const f16x8*)(Sq + swz128(item, 2 * part + 1));
            float s[C];
            float mx = -INFINITY;
#pragma unroll
            for (int c2 = 0; c2 < C; ++c2) {
                const int row = c2 * G + tq;
                const f16x8 k0 = *(const f16x8*)(Sk + swz128(row, 2 * part));
                const f16x8 k1 = *(const f16x8*)(Sk + swz128(row, 2 * part + 1));
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = __builtin_fmaf((float)q0[e], (float)k0[e], d);
#pragma unroll
                for (int e = 0; e < 8; ++e) d = __builtin_fmaf((float)q1[e], (float)k1[e], d);
                s[c2] = quad_allreduce_add(d) * p.scale;
                mx = __builtin_fmaxf(mx, s[c2]);
            }
            float den = 0.f;
#pragma unroll
            for (int c2 = 0; c2 < C; ++c2) { s[c2] = __expf(s[c2] - mx); den += s[c2]; }
            const float inv = 1.0f / den;
            float o[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
            for (int c2 = 0; c2 < C; ++c2) {
                const int row = c2 * G + tq;
                const f16x8 v0 = *(const f16x8*)(Sv + swz128(row, 2 * part));
                const f16x8 v1 = *(const f16x8*)(Sv + swz128(row, 2 * part + 1));
                const float pw = s[c2] * inv;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(pw, (float)v0[e], o[e]);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[8 + e] = __builtin_fmaf(pw, (float)v1[e], o[8 + e]);
            }
            const int t = t0 + tq;
            if (t < p.Tv) {
                f16x8 o0, o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o0[e] = to_f16_sat(o[e]); o1[e] = to_f16_sat(o[8 + e]); }
                _Float16* dst = O + (((size_t)b * C + cq) * p.Tp + t) * KD + head * 64 + part * 16;
                *(f16x8*)dst = o0;
                *(f16x8*)(dst + 8) = o1;
            }
        };

        for (int head = 0; head < 4; ++head) {
            static_for<3>([&](auto SS) __attribute__((always_inline)) {
                constexpr int sidx = decltype(SS)::value;          // 0 q, 1 k, 2 v
                const int ch = head * 3 + sidx;
                const char* Ws = smem + L_W + (ch & 1) * WS_BYTES;
                char* S = smem + L_X + sidx * S_BYTES;
                const int wrow = sidx * KD + head * FC;
                f32x4 h[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {                      // accumulators start at the bias
                    const float4 bb = *(const float4*)(p.bias + wrow + gn + i * 16 + fkg * 4);
#pragma unroll
                    for (int j = 0; j < 2; ++j) h[i][j] = f32x4{bb.x, bb.y, bb.z, bb.w};
                }
                f16x8 wa[3][2];
                auto ld = [&](auto K) __attribute__((always_inline)) {
                    constexpr int k = decltype(K)::value;
                    if constexpr (k < 8) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            wa[k % 3][i] = *(const f16x8*)(Ws + (k >> 1) * (FC * 128) + swz128(gn + i * 16 + frow, (k & 1) * 4 + fkg));
                    }
                };
                ld(std::integral_constant<int, 0>{});
                ld(std::integral_constant<int, 1>{});
                static_for<8>([&](auto K) __attribute__((always_inline)) {
                    constexpr int k = decltype(K)::value;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            h[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[k % 3][i], xf[k >> 1][k & 1][j], h[i][j], 0, 0, 0);
                    ld(std::integral_constant<int, k + 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int fl = gn + i * 16 + fkg * 4;          // feature inside the slice
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f16x4 o;
                        o[0] = to_f16_sat(h[i][j][0]); o[1] = to_f16_sat(h[i][j][1]);
                        o[2] = to_f16_sat(h[i][j][2]); o[3] = to_f16_sat(h[i][j][3]);
                        *(f16x4*)(S + swz128(gm + j * 16 + frow, fl >> 3) + ((fl >> 2) & 1) * 8) = o;
                    }
                }
                __syncthreads();              // tile visible; this slice's weight buffer is free; slice ch+1 has landed
                if (ch + 2 < 12) dma_w(ch + 2, ch & 1);
                if (head == 0) SPK_STAMP(3 + sidx);
            });
            attention(head);
            if (head < 3)                     // every thread is done with the q / k / v tiles before they are rewritten
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            SPK_STAMP(6 + head);
        }
    }
}

template <int C>
int launch(const SpkFusedParams& p, hipStream_t stream) {
    static EendOncePerDevice attr_once;
    auto kern = spk_qkv_attn_kernel<C>;
    if (!eend_set_dynamic_lds(attr_once, (const void*)kern, SMEM_BYTES)) return EEND_ELAUNCH;
    const int ncu = eend_cu_count();
    constexpr int G = BM / C;
    const int ntiles = p.B * ((p.Tv + G - 1) / G);
    hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(NT), SMEM_BYTES, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

}  // namespace

#ifdef EEND_SPK_TRACE
extern "C" int eend_debug_spk_trace(void* dst, void* stream) {
    return hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(g_spk_trace), sizeof(g_spk_trace), 0, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}
#endif

int eend_launch_spk_qkv_attn(const SpkFusedParams& p, hipStream_t stream) {
    if (!p.X || !p.W || !p.bias || !p.O || p.B <= 0 || p.Tp <= 0 || p.Tv <= 0 || p.Tv > p.Tp || (p.ldx & 7)) return EEND_EINVAL;
    switch (p.C) {
#define SPK_CASE(n) case n: return launch<n>(p, stream);
        SPK_CASE(1) SPK_CASE(2) SPK_CASE(3) SPK_CASE(4) SPK_CASE(5) SPK_CASE(6) SPK_CASE(7) SPK_CASE(8)
        SPK_CASE(9) SPK_CASE(10) SPK_CASE(11) SPK_CASE(12)
#undef SPK_CASE
        default: return EEND_EINVAL;
    }
}
