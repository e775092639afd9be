// g[M][256] += A[M][K] Wt^T on a packed weight stream (round 6): the data gradient of a K -> 256 linear layer joining the f32
// residual-gradient stream (autograd of nn.MultiheadAttention's in_proj, FS model :147 / merge_tfm_encoder.py:379-385: K = 768; of
// MultiScaleRetention's q / k / v / g projections, LS retention.py:146-160: K = 1024).  It is the second GEMM of ffn_train_stream.hip's
// data-gradient form with the operand rows read from HBM instead of produced on chip: one 256-thread workgroup per CU, one wave per SIMD
// owning 16 NJ rows end to end, the 256 output features of its rows in accumulators (seeded with the gradient stream's rows, written
// back once through the wave's staging tile as whole rows), the weights as 16-KB items [256 features x 32 k] through the 8-slot LDS-DMA
// ring with one barrier per item, the operand rows as one 16-byte load per lane, token fragment and item, four items ahead.
// gemm.hip's 64 x 256 tiles (two 4-wave workgroups per CU, a barrier pair per 32 k) run this shape at 0.16 of the MFMA peak:
// [196608, 256, 768] 187 us, [393216, 256, 1024] 417 us.
// (A form with the LayerNorm backward of the post-norm site in this kernel's epilogue -- row sums in the accumulator layout, column sums in
// the row-major view of the staging passes -- was built and matched the tiled kernel's fused epilogue bit for bit in its masks, but cost
// 146 us on top of the 172-us GEMM at [196608, 256, 768] against 95 us for eend_layernorm_bwd_f32 as its own pass: two extra staging
// passes per 16 rows with their loads exposed, and the accumulators spilling.  Not kept; the caller runs the two launches.)
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

constexpr int SLOT = 16384;            // one stream item: 16 fragments of 1 KB = 256 output features x 32 k
constexpr int NSLOT = 8;
constexpr int STAGE = NSLOT * SLOT;    // 4 x 4 KB wave-private output staging (8 rows x 512 B)
constexpr int SMEM = STAGE + 4 * 4096; // 147456
constexpr int NB = 8;                  // weight-fragment registers in rotation
constexpr int PD = 6;                  // fragment prefetch distance
constexpr int INFL = 4 * (NSLOT - 3);  // this wave's DMA pieces younger than the ones a barrier needs
constexpr int PA = 4;                  // operand-row prefetch distance (items); 8 measured the same (and doubles the fragment registers)
constexpr int MAXK = 2048;

// item k, fragment i : lane (f, g) <- Wt[(f>>2)*64 + 4 i + (f&3)][32 k + 8 g + e]     (Wt = the [256][K] transposed weight, bf16)
__global__ void gemm_acc_stream_pack_kernel(const unsigned short* __restrict__ Wt, int ldw, unsigned short* __restrict__ out, int K) {
    const long total = (long)(K / 32) * (SLOT / 16);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int k = (int)(t >> 10), w = (int)(t & 1023);
        const int i = w >> 6, l = w & 63, f = l & 15, g = l >> 4;
        *(uint4*)(out + t * 8) = *(const uint4*)(Wt + (size_t)((f >> 2) * 64 + 4 * i + (f & 3)) * ldw + k * 32 + g * 8);
    }
}

template <int NJ>
__global__ __launch_bounds__(256, 1)
void gemm_acc_stream_kernel(const GemmAccStreamParams p) {
    constexpr int TM = 64 * NJ, WM = 16 * NJ;
    constexpr int STEADY = INFL + 5 * NJ;          // + the operand-row loads of the last five items (every item issues NJ of them)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = p.K >> 5;
    const int ntiles = (p.M + TM - 1) / TM;

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, g = lane >> 4;
    int fo = g * 64;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, S * SLOT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (p.M - 1) * p.lda * 2 + p.K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.M * 1024, 0x00020000);
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

    const char* wl = smem + lane * 16;
    bf16x8 wf[NB];
    f32x4 acc[16][NJ];
    bf16x8 hq[PA][NJ];                                   // operand-row fragments of the next PA items
    int aoff[NJ];                                        // byte offset of this lane's 16 bytes of item 0, per token fragment

    __builtin_amdgcn_s_waitcnt(0x0070 | ((4 * (NSLOT - 2)) & 15) | (((4 * (NSLOT - 2)) >> 4) << 14));   // item 0 of this wave has landed; lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    sfor<PD>([&](auto Q) __attribute__((always_inline)) { wf[decltype(Q)::value % NB] = *(const bf16x8*)(wl + decltype(Q)::value * 1024); });

    auto load_a = [&](auto SL, int k) __attribute__((always_inline)) {
        constexpr int sl = decltype(SL)::value;
#pragma unroll
        for (int j = 0; j < NJ; ++j) hq[sl][j] = __builtin_bit_cast(bf16x8, bload(rsA, aoff[j] + k * 64));
    };
    auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i][j]));
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; frow = lane & 15; g = lane >> 4; fo = g * 64;
        dvo = lane * 16 + wave * 4096;
        wl = smem + lane * 16;
        // ---- accumulators: the gradient stream's rows (rows beyond M read as zeros and are dropped at the end).  (Starting from zero and
        //      adding the rows in the epilogue, their loads under the last item, measured the same and costs 64 registers + VALU work on
        //      the accumulators, which then leave the AGPRs.)
        sfor<NJ>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const int row = tile * TM + wave * WM + j * 16 + frow;
            aoff[j] = row * (p.lda * 2) + g * 16;
            const int off = row * 1024 + fo * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i][j] = __builtin_bit_cast(f32x4, bload(rsG, off + i * 16));
        });
        sfor<PA>([&](auto D) __attribute__((always_inline)) { load_a(D, decltype(D)::value); });
        pin_acc();

        for (int k0 = 0; k0 < S; k0 += PA) {
            sfor<PA>([&](auto U) __attribute__((always_inline)) {
                constexpr int u = decltype(U)::value;
                const int k = k0 + u;
                // the item's barrier: this wave's pieces of the NEXT item have landed.  Younger accesses of this wave: INFL DMA pieces and,
                // from the sixth item of a tile on, the NJ operand-row loads of each of the last five items (counting fewer is the safe side)
                if (k < 5) __builtin_amdgcn_s_waitcnt(0x0F70 | (INFL & 15) | ((INFL >> 4) << 14));
                else __builtin_amdgcn_s_waitcnt(0x0F70 | (STEADY & 15) | ((STEADY >> 4) << 14));
                __builtin_amdgcn_s_barrier();
                const char* wc = wl + slot * SLOT;
                const char* wn = wl + ((slot + 1) & (NSLOT - 1)) * SLOT;
                const int sd = (slot + NSLOT - 1) & (NSLOT - 1);
                sfor<8>([&](auto P2) __attribute__((always_inline)) {
                    sfor<2>([&](auto PH) __attribute__((always_inline)) {
                        constexpr int pi = decltype(P2)::value * 2 + decltype(PH)::value;
                        const bf16x8 w = wf[pi % NB];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[pi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, hq[u][j], acc[pi][j], 0, 0, 0);
                        if constexpr (pi + PD < 16) wf[(pi + PD) % NB] = *(const bf16x8*)(wc + (pi + PD) * 1024);
                        else wf[(pi + PD) % NB] = *(const bf16x8*)(wn + (pi + PD - 16) * 1024);
                        if constexpr (pi < 4) dma_piece(sd, IC<pi>{});
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                dma_advance();
                slot = (slot + 1) & (NSLOT - 1);
                // the rows of item k + PA take the registers this item has just used (past the last item: a repeat of it, so that every
                // item issues the same number of accesses)
                load_a(U, k + PA < S ? k + PA : S - 1);
            });
        }
        pin_acc();

        // ---- the rows leave through the wave's 4-KB staging tile as whole rows
        asm volatile("" : "+v"(tid));
        lane = tid & 63; frow = lane & 15; g = lane >> 4; fo = g * 64;
        char* st = smem + STAGE + wave * 4096;
        sfor<NJ>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const int rbase = tile * TM + wave * WM + j * 16;
#pragma unroll
            for (int fh = 0; fh < 2; ++fh)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if ((frow >> 3) == half) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) *(f32x4*)(st + (frow & 7) * 512 + (((g * 8 + e) ^ (frow & 7)) << 4)) = acc[fh * 8 + e][j];
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int rr = 2 * q4 + (lane >> 5), cc = lane & 31;
                        const f32x4 v = *(const f32x4*)(st + rr * 512 + ((cc ^ rr) << 4));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsG,
                                                               (rbase + half * 8 + rr) * 1024 + (cc >> 3) * 256 + fh * 128 + (cc & 7) * 16, 0, 0);
                    }
                    wave_lds_sync();
                }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the workgroup
}

template <int NJ>
int launch_nj(const GemmAccStreamParams& p, int ncu, hipStream_t stream) {
    static EendOncePerDevice attr_once;
    auto kern = gemm_acc_stream_kernel<NJ>;
    if (!eend_set_dynamic_lds(attr_once, (const void*)kern, SMEM)) return EEND_ELAUNCH;
    const int ntiles = (p.M + 64 * NJ - 1) / (64 * NJ);
    hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), SMEM, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

}  // namespace

long eend_gemm_acc_stream_nelems(int K) { return (K < 256 || (K % 128) != 0 || K > MAXK) ? 0 : (long)(K / 32) * (SLOT / 2); }

int eend_launch_gemm_acc_stream_pack(const void* Wt, int ldw, void* out, int K, hipStream_t stream) {
    if (!Wt || !out || eend_gemm_acc_stream_nelems(K) == 0 || ldw < K || (ldw & 7) || (((size_t)Wt | (size_t)out) & 15)) return EEND_EINVAL;
    const long total = eend_gemm_acc_stream_nelems(K) / 8;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(gemm_acc_stream_pack_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, stream, (const unsigned short*)Wt, ldw,
                       (unsigned short*)out, K);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// rows one launch takes: 32-bit buffer offsets into A and the f32 rows
bool eend_gemm_acc_stream_fits(int M, int K, int lda) {
    if (M <= 0 || eend_gemm_acc_stream_nelems(K) == 0 || lda < K || (lda & 7)) return false;
    return ((long)M + 256) * lda * 2 < (1L << 31) && ((long)M + 256) * 1024 < (1L << 31);
}

int eend_launch_gemm_acc_stream(const GemmAccStreamParams& p, hipStream_t stream) {
    if (!eend_gemm_acc_stream_fits(p.M, p.K, p.lda) || !p.A || !p.wstream || !p.g || (((size_t)p.A | (size_t)p.wstream | (size_t)p.g) & 15)) return EEND_EINVAL;
    const int ncu = eend_cu_count();
    const long t3 = (p.M + 191) / 192, t2 = (p.M + 127) / 128;
    const long c3 = ((t3 + ncu - 1) / ncu) * (3 * 10 + 9), c2 = ((t2 + ncu - 1) / ncu) * (2 * 10 + 9);     // rounds x (rows + fixed part), as ffn_stream.hip
    return c2 < c3 ? launch_nj<2>(p, ncu, stream) : launch_nj<3>(p, ncu, stream);
}
