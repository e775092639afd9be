// K = 256 input projections on a packed weight stream (round 6), the decomposition of ffn_train_stream.hip's first GEMM without the second:
//   one 256-thread workgroup per CU, one wave per SIMD OWNING 64 token rows (their 256 input features live in registers as MFMA operand
//   fragments, read from HBM once per tile and prefetched one tile ahead), the weights flowing through an 8-slot LDS-DMA ring in items of
//   32 output features (16 KB, one barrier each), every weight fragment feeding four (or eight) MFMAs, results leaving straight from the
//   accumulators -- no staging tile, no per-chunk barrier pair.
// Serves, per 256-feature output group, up to three destinations from ONE pass over the rows (ProjStreamParams):
//   rows   : [M][ld] row-major or [seq][H][Tp][64] head rows, f16 or bf16
//   rows2  : a second copy as bf16 head rows (the training forward keeps bf16 Q / K / V for the hand-written backward next to the f16
//            operands of its own attention / retention kernel: one projection instead of two)
//   heads_t: [seq][H][64][Tp] transposed head rows (K^T, V^T of the retention) -- the MFMA operands swapped, a lane owns 4 consecutive
//            tokens of one feature
// Reference sites: nn.MultiheadAttention in_proj (FS model :147, merge_tfm_encoder.py:379-385), MultiScaleRetention q / k / v / g
// projections (LS retention.py:146-160).  proj.hip (X tile resident in LDS, two workgroup barriers per 64 features and destination)
// stays as the general form: [196608, 768]: 189 us there.
// Study switches (timing only, tools/build_variant.sh; several are not valid kernels): PS_NOSTORE no destination stores, PS_WAIT60 the item
// barrier waits for (almost) nothing, PS_NOBARRIER no item barrier, PS_NODMA no weight DMA after the prologue.
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
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int SLOT = 16384;            // one stream item: 16 fragments of 1 KB = 32 output features x 256 inputs
constexpr int NSLOT = 8;
constexpr int BIASL = NSLOT * SLOT;    // bias table, up to 1024 features
constexpr int MAXN = 1024;
constexpr int SMEM = BIASL + MAXN * 4; // 135168
constexpr int NB = 8;                  // weight-fragment registers in rotation
constexpr int PD = 6;                  // fragment prefetch distance
constexpr int INFL = 4 * (NSLOT - 3);  // this wave's DMA pieces younger than the ones a barrier needs
constexpr int NJ = 4, TM = 64 * NJ, WM = 16 * NJ;

// item q, fragment p = s*2 + hf : lane (f, g) <- W[32 q + (f>>2)*8 + hf*4 + (f&3)][32 s + 8 g + e]
// (a lane of the result then holds 8 CONSECUTIVE features of the item: g*8 + hf*4 + r)
__global__ void proj_stream_pack_kernel(const unsigned short* __restrict__ W, unsigned short* __restrict__ out, int N) {
    const long total = (long)(N / 32) * (SLOT / 16);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int q = (int)(t >> 10), w = (int)(t & 1023);
        const int pfrag = w >> 6, l = w & 63, f = l & 15, g = l >> 4, s_ = pfrag >> 1, hf = pfrag & 1;
        *(uint4*)(out + t * 8) = *(const uint4*)(W + (size_t)(q * 32 + (f >> 2) * 8 + hf * 4 + (f & 3)) * 256 + s_ * 32 + g * 8);
    }
}

DEV u32x4 pack8(const f32x4 a, const f32x4 b, bool bf) {
    if (bf) {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r] = (__bf16)a[r]; o[4 + r] = (__bf16)b[r]; }
        return __builtin_bit_cast(u32x4, o);
    }
    f16x8 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o[r] = to_f16_sat(a[r]); o[4 + r] = to_f16_sat(b[r]); }
    return __builtin_bit_cast(u32x4, o);
}
DEV u32x2 pack4(const f32x4 a, bool bf) {
    if (bf) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (__bf16)a[r];
        return __builtin_bit_cast(u32x2, o);
    }
    f16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = to_f16_sat(a[r]);
    return __builtin_bit_cast(u32x2, o);
}

// vmcnt is a 6-bit field split over bits 3:0 and 15:14; lgkmcnt untouched (0xF at 11:8), expcnt 7
DEV void wait_vm(int n) {
#define PS_W(v) case v: __builtin_amdgcn_s_waitcnt(0x0F70 | ((v) & 15) | (((v) >> 4) << 14)); break;
    switch (n >> 2) {
        PS_W(5) PS_W(6) PS_W(7) PS_W(8) PS_W(9) PS_W(10) PS_W(11) PS_W(12) PS_W(13) PS_W(14)
        default: __builtin_amdgcn_s_waitcnt(0x0F70 | (60 & 15) | ((60 >> 4) << 14)); break;
    }
#undef PS_W
}
// (the cases above wait for 4 * (n >> 2) <= n outstanding accesses: never more than asked for)
static_assert(INFL == 20, "wait_vm's first case");

__global__ __launch_bounds__(256, 1)
void proj_stream_kernel(const ProjStreamParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = p.N >> 5;
    const int ntiles = (p.M + TM - 1) / TM;

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, g = lane >> 4;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, S * SLOT, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (p.M - 1) * p.ldx * 2 + 512, 0x00020000);
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
    float* bl = (float*)(smem + BIASL);
    for (int i = tid; i < p.N; i += 256) bl[i] = p.bias[i];

    const char* wl = smem + lane * 16;
    f16x8 wf[NB];
    f16x8 xf[8][NJ], xn[8][NJ];
    auto load_rows = [&](int tile, f16x8 (&x)[8][NJ]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int off = (tile * TM + wave * WM + j * 16 + frow) * (p.ldx * 2) + g * 16;
#pragma unroll
            for (int s = 0; s < 8; ++s) x[s][j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsA, off + s * 64, 0, 0));
        }
    };

    __builtin_amdgcn_s_waitcnt(0x0070 | ((4 * (NSLOT - 2)) & 15) | (((4 * (NSLOT - 2)) >> 4) << 14));   // item 0 of this wave has landed; lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    if ((int)blockIdx.x < ntiles) load_rows(blockIdx.x, xf);
    sfor<PD>([&](auto Q) __attribute__((always_inline)) { wf[decltype(Q)::value % NB] = *(const f16x8*)(wl + decltype(Q)::value * 1024); });

    // stores this wave issued in the last five items: all younger than the pieces the next barrier waits for (requested six items ago).
    // Counting fewer is safe but makes the barrier wait for stores that have nothing to do with it (two items counted: 149 us instead
    // of ... for [196608, 768]: a store is acknowledged some microseconds after its issue)
    int e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; frow = lane & 15; g = lane >> 4;
        dvo = lane * 16 + wave * 4096;
        wl = smem + lane * 16;
        const int row0 = tile * TM + wave * WM;                       // the wave's 64 rows: inside one sequence (Tp is a multiple of 64)
        const int seq = row0 / p.Tp, t0 = row0 - seq * p.Tp;

        for (int q = 0; q < S; ++q) {
            const int n0 = q << 5, grp = n0 >> 8, nn = n0 & 255, head = nn >> 6, half = (nn >> 5) & 1;
            const int kind_a = p.kind_a[grp];
            const bool has_b = p.out_b[grp] != nullptr, has_t = p.out_t[grp] != nullptr;
            const bool has_n = kind_a != 0 || has_b;
            // ---- the item's barrier: this wave's pieces of the NEXT item have landed (its first fragments are prefetched below)
            int extra = (q == 3 || q == 4) ? 8 * NJ : 0;             // the next tile's rows, requested in front of item 3
            int allow = INFL + e1 + e2 + e3 + e4 + e5 + extra;
#ifdef PS_WAIT60
            allow = 63;
#endif
            wait_vm(allow < 63 ? allow : 63);
#ifndef PS_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
            if (q == 2) load_rows(tile + (int)gridDim.x, xn);         // rows beyond M read as zeros
            const char* wc = wl + slot * SLOT;
            const char* wn = wl + ((slot + 1) & (NSLOT - 1)) * SLOT;
            const int sd = (slot + NSLOT - 1) & (NSLOT - 1);

            f32x4 h[2][NJ], ht[2][NJ];
            auto run = [&](auto HN, auto HT) __attribute__((always_inline)) {
                constexpr bool hn = decltype(HN)::value, htr = decltype(HT)::value;
                f32x4 bn[2], bt[2];
                if constexpr (hn) {
                    bn[0] = *(const f32x4*)(bl + n0 + g * 8);
                    bn[1] = *(const f32x4*)(bl + n0 + g * 8 + 4);
                }
                if constexpr (htr) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const float b = bl[n0 + (frow >> 2) * 8 + hf * 4 + (frow & 3)];
                        bt[hf] = f32x4{b, b, b, b};
                    }
                }
                sfor<8>([&](auto P2) __attribute__((always_inline)) {
                    sfor<2>([&](auto PH) __attribute__((always_inline)) {
                        constexpr int pi = decltype(P2)::value * 2 + decltype(PH)::value, s_ = pi >> 1, hf = pi & 1;
                        const f16x8 w = wf[pi % NB];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            if constexpr (hn) h[hf][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xf[s_][j], s_ == 0 ? bn[hf] : h[hf][j], 0, 0, 0);
                            if constexpr (htr) ht[hf][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[s_][j], w, s_ == 0 ? bt[hf] : ht[hf][j], 0, 0, 0);
                        }
                        if constexpr (pi + PD < 16) wf[(pi + PD) % NB] = *(const f16x8*)(wc + (pi + PD) * 1024);
                        else wf[(pi + PD) % NB] = *(const f16x8*)(wn + (pi + PD - 16) * 1024);
#ifndef PS_NODMA
                        if constexpr (pi < 4) dma_piece(sd, IC<pi>{});
#endif
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            if (has_n && has_t) run(std::true_type{}, std::true_type{});
            else if (has_t) run(std::false_type{}, std::true_type{});
            else run(std::true_type{}, std::false_type{});
            dma_advance();
            slot = (slot + 1) & (NSLOT - 1);

            // ---- the item's 64 rows x 32 features leave from the accumulators
            int issued = 0;
#ifdef PS_NOSTORE
            if (row0 < 0) {
#else
            if (row0 < p.M) {
#endif
                if (kind_a != 0) {
                    const bool bf = p.bf_a[grp] != 0;
                    if (kind_a == 1) {                                // row-major [M][ld]: 16 bytes per lane and token
                        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out_a[grp], 0, ((p.M - 1) * p.ld_a[grp] + 256) * 2, 0x00020000);
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            __builtin_amdgcn_raw_buffer_store_b128(pack8(h[0][j], h[1][j], bf), ro, ((row0 + j * 16 + frow) * p.ld_a[grp] + nn + g * 8) * 2, 0, 0);
                    } else {                                          // head rows [seq][H][Tp][64]
                        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out_a[grp], 0, p.M * 512, 0x00020000);
                        const int base = ((seq * p.H + head) * p.Tp + t0) * 128 + half * 64;
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            __builtin_amdgcn_raw_buffer_store_b128(pack8(h[0][j], h[1][j], bf), ro, base + (j * 16 + frow) * 128 + g * 16, 0, 0);
                    }
                    issued += NJ;
                }
                if (has_b) {                                          // bf16 head rows
                    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out_b[grp], 0, p.M * 512, 0x00020000);
                    const int base = ((seq * p.H + head) * p.Tp + t0) * 128 + half * 64;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(pack8(h[0][j], h[1][j], true), ro, base + (j * 16 + frow) * 128 + g * 16, 0, 0);
                    issued += NJ;
                }
                if (has_t) {                                          // [seq][H][64][Tp]: 4 tokens of one feature per lane
                    const bool bf = p.bf_t[grp] != 0;
                    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out_t[grp], 0, p.M * 512, 0x00020000);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int d = half * 32 + (frow >> 2) * 8 + hf * 4 + (frow & 3);
                        const int base = (((seq * p.H + head) * 64 + d) * p.Tp + t0 + g * 4) * 2;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) __builtin_amdgcn_raw_buffer_store_b64(pack4(ht[hf][j], bf), ro, base + j * 32, 0, 0);
                    }
                    issued += 2 * NJ;
                }
            }
            e5 = e4; e4 = e3; e3 = e2; e2 = e1; e1 = issued;
        }
        // the next tile's rows become the current ones
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < NJ; ++j) xf[s][j] = xn[s][j];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the workgroup
}

}  // namespace

long eend_proj_stream_nelems(int N) { return (N <= 0 || (N % 256) != 0 || N > MAXN) ? 0 : (long)(N / 32) * (SLOT / 2); }

int eend_launch_proj_stream_pack(const void* W, void* out, int N, hipStream_t stream) {
    if (!W || !out || eend_proj_stream_nelems(N) == 0 || (((size_t)W | (size_t)out) & 15)) return EEND_EINVAL;
    const long total = eend_proj_stream_nelems(N) / 8;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(proj_stream_pack_kernel, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, stream, (const unsigned short*)W, (unsigned short*)out, N);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// shapes one launch takes: 32-bit buffer offsets with the row prefetch running one grid of tiles past the end; head / transposed
// destinations need whole 64-row pieces inside a sequence
bool eend_proj_stream_fits(const ProjStreamParams& p) {
    if (p.M <= 0 || eend_proj_stream_nelems(p.N) == 0 || p.ldx < 256 || (p.ldx & 7) || !p.X || !p.wstream || !p.bias) return false;
    if (((long)p.M + 65536 + TM) * p.ldx * 2 >= (1L << 31)) return false;
    for (int gq = 0; gq < p.N / 256; ++gq) {
        const bool heads = p.kind_a[gq] == 2 || p.out_b[gq] || p.out_t[gq];
        if (p.kind_a[gq] == 0 && !p.out_b[gq] && !p.out_t[gq]) return false;
        if (p.kind_a[gq] < 0 || p.kind_a[gq] > 2 || (p.kind_a[gq] != 0 && !p.out_a[gq])) return false;
        if (heads && (p.H != 4 || p.Tp <= 0 || (p.Tp % 64) != 0 || (p.M % p.Tp) != 0 || ((long)p.M + TM) * 512 >= (1L << 31))) return false;
        if (p.kind_a[gq] == 1 && ((p.ld_a[gq] & 7) || p.ld_a[gq] < 256 || ((long)p.M + TM) * p.ld_a[gq] * 2 >= (1L << 31))) return false;
        if ((((size_t)p.out_a[gq] | (size_t)p.out_b[gq] | (size_t)p.out_t[gq]) & 15)) return false;
    }
    return (((size_t)p.X | (size_t)p.wstream) & 15) == 0;
}

int eend_launch_proj_stream(const ProjStreamParams& p, hipStream_t stream) {
    if (!eend_proj_stream_fits(p)) return EEND_EINVAL;
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)proj_stream_kernel, SMEM)) return EEND_ELAUNCH;
    const int ncu = eend_cu_count();
    const int ntiles = (p.M + TM - 1) / TM;
    hipLaunchKernelGGL(proj_stream_kernel, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), SMEM, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
