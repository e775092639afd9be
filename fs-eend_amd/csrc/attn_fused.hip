// In-projection + causal multi-head attention of one (sequence, head) in ONE workgroup, for chunks that fit on chip
// (Tp <= 512: the T = 500 chunks FS-EEND is trained and benchmarked on).  Replaces the pair
//     proj_xres_kernel (packed QKV in-projection, head-scatter layouts)  ->  attn_causal_full_kernel
// of every time-axis attention (nn.TransformerEncoderLayer.self_attn, FS model :147; self_attn1 of the fusion layers,
// merge_tfm_encoder.py:379-385), so that K and V NEVER reach HBM: the in-projection used to write Q | K | V^T
// (3 x 2 B x 256 per token: 302 MB per decoder layer at B = 64, C = 6) and the attention kernel read it straight
// back -- together 19 % of the step, both HBM-bound.  Here a workgroup streams its sequence's X rows ([Tp][256]
// f16, 32 rows at a time, register-staged into a double-buffered LDS tile) past the head's W_k / W_v / W_q slices held in registers
// (MFMA f16 16x16x32, the A- and B-operand fragments of one 16 x 32 block have the same lane layout, so K -- wanted
// key-major -- and V^T -- wanted d-major -- come from the SAME X fragments with the operand order swapped), drops
// the bf16 K / V^T tiles into the XOR-swizzled LDS images the flash loop reads, and then runs that loop
// (attn_full.hip: transposed formulation on v_mfma_f32_32x32x16_bf16, lazy softmax reference, causally balanced
// query-block pairs, no barriers).  Q (64 KB per head) makes a round trip through an L2-resident scratch buffer:
// it is produced feature-split across the waves and consumed query-split, and LDS is full (64 + 64 + 32 KB).
//
// Round 3: the kernel is PERSISTENT -- one workgroup per CU walks the (sequence, head) items (item L, L + gridDim.x, ...).
// The Q scratch is then a per-WORKGROUP 64 KB slot that is rewritten for every item, i.e. it stays in that XCD's L2
// instead of making a 2 x 100 MB round trip through HBM per decoder launch (r02 PMC: 422 MB per launch against 200 MB
// algorithmic); the dispatch of a fresh 160 KB-LDS workgroup per item (5.7 us per round in the s_memtime traces) is paid
// once; and the first X tile of the next item is requested while the current item's flash loop runs.
//
// The q rows of W / bias must be pre-multiplied by 1/sqrt(dh) * log2(e) (ops.QSCALE_LOG2), as for the LAZY path of
// attn_causal_full_kernel.  Block index -> (sequence, head) is XCD-aware: the four heads of a sequence run on the
// same XCD, so three of the four reads of its X rows are L2 hits.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace {

#ifndef EEND_AF_NT
#define EEND_AF_NT 0            // 1: X loads and O stores carry the non-temporal hint, so that the streaming traffic does not
#endif                          //    evict the workgroup's Q slot from L2 between its write (phase 1) and its read (phase 2)
#ifndef EEND_AF_REGSTAGE
#define EEND_AF_REGSTAGE 1      // X staging through registers (1) or by LDS-DMA (0: the first form, kept for the A/B)
#endif

constexpr int KB = 64;
constexpr int TILE = KB * 128;            // one [64][64] bf16 tile
constexpr int NW = 8;
constexpr int OSTG = 32 * 128;            // per-wave O staging: 32 rows x 128 B
constexpr int XR = 32;                    // X rows per projection step
constexpr int XBUF = 4 * XR * 128;        // [4 k-tiles][32 rows][128 B] f16

typedef __attribute__((address_space(3))) char lds_char;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

DEV int swap23(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

DEV u32x2 pack_bf16x4(const f32x4 v) {
    bf16x4 o;
    o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
    return __builtin_bit_cast(u32x2, o);
}

__global__ __launch_bounds__(512)
void inproj_attn_kernel(const InprojAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntiles = p.Tp / KB;                    // <= 8
    char* Ks = smem;                                 // [ntiles][64 keys][128 B]
    char* Vs = smem + ntiles * TILE;                 // [ntiles][64 d][128 B]
    char* Xs = smem + 2 * ntiles * TILE;             // 2 x 16 KB X staging; afterwards the 8 x 4 KB O staging

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fkg = lane >> 4;
    const int nitems = p.nseq * 4;
    // item -> (sequence, head).  XCD = workgroup id % 8 (hardware round-robin): with nseq % 8 == 0 and a grid that is a
    // multiple of 8, item L runs on XCD L % 8 and the four heads of a sequence are four consecutive slots of one XCD --
    // concurrent workgroups sharing an L2, so three of the four reads of the sequence's X rows are L2 hits.
    const bool xcd_map = (p.nseq & 7) == 0 && ((gridDim.x & 7) == 0 || (int)gridDim.x >= nitems);
    auto item_of = [&](int L, int& seq_, int& h_) __attribute__((always_inline)) {
        if (xcd_map) {
            const int xcd = L & 7, slot = L >> 3;
            seq_ = (slot >> 2) * 8 + xcd; h_ = slot & 3;
        } else {
            seq_ = L >> 2; h_ = L & 3;
        }
    };
    __bf16* __restrict__ Qs = (__bf16*)p.Qs + (size_t)blockIdx.x * p.Tp * 64;      // this workgroup's L2-resident Q slot
    u32x4 pre[2];                                    // first X tile of the NEXT item, requested during the flash loop
    bool have_pre = false;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    int seq, h;
    item_of(item, seq, h);
    int seq_n = 0, h_n = 0;
    const bool has_next = item + (int)gridDim.x < nitems;
    if (has_next) item_of(item + gridDim.x, seq_n, h_n);
    (void)h_n;

    // ================================================================== phase 1: K, V^T -> LDS, Q -> scratch
    {
        const _Float16* __restrict__ W = (const _Float16*)p.W;
        const int kvsel = wave >> 2;                 // waves 0-3: K features, 4-7: V features (16 each); all: 16 Q features
        const int f0 = (wave & 3) * 16;
        // X tile xt (32 rows) -> staging buffer: 16 pieces of 8 rows x 128 B (k-tile kt = piece >> 2), 2 per wave;
        // swz128 image via the per-lane source address
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)p.X + (size_t)seq * p.Tp * p.ldx), 0,
                                                                            p.Tp * p.ldx * 2, 0x00020000);
        int vox[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave * 2 + i, kt = pc >> 2, r = (pc & 3) * 8 + (lane >> 3);
            vox[i] = r * p.ldx * 2 + kt * 128 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        }
        auto dma_x = [&](int xt) __attribute__((always_inline)) {
            char* dst = Xs + (xt & 1) * XBUF;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_char*)(dst + (wave * 2 + i) * 1024), 16, vox[i], xt * XR * p.ldx * 2, 0, 0);
        };
        const int nxt = p.Tp / XR;                   // <= 16
#if !EEND_AF_REGSTAGE
        dma_x(0);
        if (nxt > 1) dma_x(1);
#endif
        f16x8 wkv[8], wq[8];
        {
            const _Float16* wr = W + (size_t)((1 + kvsel) * 256 + h * 64 + f0 + frow) * 256 + fkg * 8;
            const _Float16* qr = W + (size_t)(h * 64 + f0 + frow) * 256 + fkg * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) { wkv[ks] = *(const f16x8*)(wr + ks * 32); wq[ks] = *(const f16x8*)(qr + ks * 32); }
        }
        f32x4 bkv, bq;
        {
            const float4 b4 = *(const float4*)(p.bias + h * 64 + f0 + fkg * 4);
            bq = f32x4{b4.x, b4.y, b4.z, b4.w};
            if (kvsel == 0) {
                const float4 k4 = *(const float4*)(p.bias + 256 + h * 64 + f0 + fkg * 4);
                bkv = f32x4{k4.x, k4.y, k4.z, k4.w};
            } else {
                const float bv = p.bias[512 + h * 64 + f0 + frow];
                bkv = f32x4{bv, bv, bv, bv};
            }
        }
        // pinned: their wait (which, VMEM being in order, also covers the first two X tiles) happens once, here, instead
        // of being re-inserted by the compiler in front of the uses inside the loop, behind the younger requests
        asm volatile("" : "+v"(wkv[0]), "+v"(wkv[1]), "+v"(wkv[2]), "+v"(wkv[3]), "+v"(wkv[4]), "+v"(wkv[5]), "+v"(wkv[6]), "+v"(wkv[7]));
        asm volatile("" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(wq[4]), "+v"(wq[5]), "+v"(wq[6]), "+v"(wq[7]),
                          "+v"(bkv), "+v"(bq));
        const int jq = wave >> 2;                    // the token fragment this wave projects Q for
        // one projection step on the 32 X rows in buffer xb: K / V^T fragments -> LDS, Q fragment -> scratch
        auto proj_step = [&](int xt, const char* xb) __attribute__((always_inline)) {
            f32x4 akv[2] = {bkv, bkv};
            f32x4 aq = bq;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {         // (hoisting all 16 fragment reads in front of the MFMAs measured 3 % slower)
                f16x8 x[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) x[j] = *(const f16x8*)(xb + (ks >> 1) * (XR * 128) + swz128(j * 16 + frow, (ks & 1) * 4 + fkg));
                if (kvsel == 0) {                    // K[key][d]: rows = d (A = W_k), columns = key
                    akv[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wkv[ks], x[0], akv[0], 0, 0, 0);
                    akv[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wkv[ks], x[1], akv[1], 0, 0, 0);
                } else {                             // V^T[d][key]: rows = key (A = X), columns = d
                    akv[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[0], wkv[ks], akv[0], 0, 0, 0);
                    akv[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[1], wkv[ks], akv[1], 0, 0, 0);
                }
                aq = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[ks], jq ? x[1] : x[0], aq, 0, 0, 0);
            }
            // lane holds rows fkg*4 .. +3 of column frow of each 16 x 16 block
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (kvsel == 0) {
                    const int key = xt * XR + j * 16 + frow, d = f0 + fkg * 4;
                    *(u32x2*)(Ks + (key >> 6) * TILE + swz128(key & 63, d >> 3) + (d & 7) * 2) = pack_bf16x4(akv[j]);
                } else {
                    const int d = f0 + frow, key = xt * XR + j * 16 + fkg * 4;
                    *(u32x2*)(Vs + (key >> 6) * TILE + swz128(d, (key & 63) >> 3) + (key & 7) * 2) = pack_bf16x4(akv[j]);
                }
            }
            {
                const int tok = xt * XR + jq * 16 + frow, d = f0 + fkg * 4;
                *(u32x2*)(Qs + (size_t)tok * 64 + d) = pack_bf16x4(aq);
            }
        };
#if EEND_AF_REGSTAGE
        // X through REGISTERS: plain global loads ingest faster than LDS-DMA here (the s_memtime traces put the DMA stream
        // at ~1 KB per 130 cycles per CU however it is scheduled; register-staged loads of the stand-alone attention kernel
        // ran at more than twice that).  Thread q of the 512 moves chunks q and q + 512 of a tile's 1024 16-byte chunks
        // (row = chunk >> 5, 16-byte column = chunk & 31); two register sets, so a tile is requested two steps before it is
        // written to the staging buffer the step before it is read.  One barrier per step.
        const _Float16* __restrict__ Xg = (const _Float16*)p.X + (size_t)seq * p.Tp * p.ldx;
        int xsrc[2], xdst[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 512, row = c >> 5, col = c & 31;
            xsrc[i] = row * p.ldx + col * 8;
            xdst[i] = (col >> 3) * (XR * 128) + swz128(row, col & 7);
        }
        u32x4 ra[2], rb[2];
        auto gload = [&](int xt, u32x4 (&r)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#if EEND_AF_NT
                r[i] = __builtin_nontemporal_load((const u32x4*)(Xg + (size_t)xt * XR * p.ldx + xsrc[i]));
#else
                r[i] = *(const u32x4*)(Xg + (size_t)xt * XR * p.ldx + xsrc[i]);
#endif
            }
        };
        auto lstore = [&](int xt, const u32x4 (&r)[2]) __attribute__((always_inline)) {
            char* dst = Xs + (xt & 1) * XBUF;
#pragma unroll
            for (int i = 0; i < 2; ++i) *(u32x4*)(dst + xdst[i]) = r[i];
        };
        if (have_pre) { ra[0] = pre[0]; ra[1] = pre[1]; }
        else gload(0, ra);
        lstore(0, ra);
        if (nxt > 1) gload(1, ra);
        __syncthreads();
        for (int xt = 0; xt < nxt; xt += 2) {
            if (xt + 2 < nxt) gload(xt + 2, rb);
            proj_step(xt, Xs);
            if (xt + 1 < nxt) lstore(xt + 1, ra);
            __syncthreads();
            if (xt + 1 >= nxt) break;
            if (xt + 3 < nxt) gload(xt + 3, ra);
            proj_step(xt + 1, Xs + XBUF);
            if (xt + 2 < nxt) lstore(xt + 2, rb);
            __syncthreads();
        }
#else
        for (int xt = 0; xt < nxt; ++xt) {
            // tile xt has landed: VMEM returns in order; younger than its pieces are at most this wave's Q store of tile
            // xt-1 and the 2 pieces of tile xt+1
            if (xt == 0 && nxt > 1) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            else if (xt + 1 < nxt) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            proj_step(xt, Xs + (xt & 1) * XBUF);
            if (xt + 2 < nxt) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is done reading this buffer
                dma_x(xt + 2);
            }
        }
#endif
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");    // K, V^T complete in LDS; Q complete in L2
    }

    // ================================================================== phase 2: the flash loop (attn_full.hip, LAZY)
    const int lq = lane & 31, hi = lane >> 5;
    const __bf16* __restrict__ Qg = Qs;
    const int nq = p.Tp / 32;
    const int krow = swap23(lq);
    char* Ow = Xs + wave * OSTG;
    const int qb_big = nq - 1 - wave, qb_small = wave;
    const bool has_big = qb_big >= qb_small;
    const bool has_small = qb_small < qb_big;

    bf16x8 qf[4];
    f32x16 oT[2];
    f32x16 mneg;
    float l_run;
    int qw0, q;

    auto begin_pass = [&](int qb) __attribute__((always_inline)) {
        qw0 = qb * 32;
        q = qw0 + lq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Qg + (size_t)q * 64 + ks * 16 + hi * 8);
#pragma unroll
        for (int i = 0; i < 16; ++i) { oT[0][i] = 0.f; oT[1][i] = 0.f; mneg[i] = 0.f; }
        l_run = 0.f;
    };
    auto tile = [&](int j) __attribute__((always_inline)) {
        const int key0 = j * KB;
        const char* kb_ = Ks + j * TILE;
        const char* vb_ = Vs + j * TILE;
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kb_ + swz128(kb * 32 + krow, ks * 2 + hi));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? mneg : s[kb], 0, 0, 0);
            }
        const int wlim = qw0 + p.mask_delay < p.kv_len - 1 ? qw0 + p.mask_delay : p.kv_len - 1;
        if (key0 + KB - 1 > wlim) {
            const int lim = q + p.mask_delay < p.kv_len - 1 ? q + p.mask_delay : p.kv_len - 1;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = key0 + kb * 32 + (i & 7) + 8 * hi + 16 * (i >> 3);
                    if (key > lim) s[kb][i] = -INFINITY;
                }
        }
        float tmax = s[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = __builtin_fmaxf(tmax, s[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = __builtin_fmaxf(tmax, s[1][i]);
        tmax = wave_xor_max(tmax, 32);
        // the reference only moves when a row outgrows it by 2^8 (or, on the first tile, sits far below it)
        const bool move = tmax > 8.0f || (j == 0 && tmax < -8.0f);
        if (__builtin_amdgcn_ballot_w64(move) != 0) {
            float d = j == 0 ? tmax : __builtin_fmaxf(tmax, 0.f);
            d = d == -INFINITY ? 0.f : d;
            const float alpha = __builtin_amdgcn_exp2f(-d);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                oT[0][i] *= alpha; oT[1][i] *= alpha;
                s[0][i] -= d; s[1][i] -= d;
                mneg[i] -= d;
            }
        }
        float lsum0 = 0.f, lsum1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[0][i] = __builtin_amdgcn_exp2f(s[0][i]);
            s[1][i] = __builtin_amdgcn_exp2f(s[1][i]);
            lsum0 += s[0][i];
            lsum1 += s[1][i];
        }
        l_run += lsum0 + lsum1;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 pf;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pf[jj] = (__bf16)s[kb][kk * 8 + jj];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = *(const bf16x8*)(vb_ + swz128(db * 32 + lq, kb * 4 + kk * 2 + hi));
                    oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oT[db], 0, 0, 0);
                }
            }
    };
    bool pre_issued = false;
    auto run_pass = [&](int qb) __attribute__((always_inline)) {
        begin_pass(qb);
#if EEND_AF_REGSTAGE
        if (has_next && !pre_issued) {
            // behind this pass's Q fragments in the (in-order) VMEM queue, so nothing in this pass waits for it; the
            // second pass's Q loads do, ~15 us later
            const _Float16* __restrict__ Xn = (const _Float16*)p.X + (size_t)seq_n * p.Tp * p.ldx;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + i * 512, row = c >> 5, col = c & 31;
#if EEND_AF_NT
                pre[i] = __builtin_nontemporal_load((const u32x4*)(Xn + (size_t)row * p.ldx + col * 8));
#else
                pre[i] = *(const u32x4*)(Xn + (size_t)row * p.ldx + col * 8);
#endif
            }
            pre_issued = true;
        }
#endif
        int last_key = qw0 + 31 + p.mask_delay;
        last_key = last_key < p.kv_len - 1 ? last_key : p.kv_len - 1;
        const int jend = last_key < 0 ? 0 : last_key / KB + 1;
        for (int j = 0; j < jend; ++j) tile(j);
        // O[q][d] = O^T / l: stage the wave's 32 x 64 f16 tile, then 128-byte rows to HBM
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 o;
                o[0] = to_f16_sat(oT[db][g * 4 + 0] * inv);
                o[1] = to_f16_sat(oT[db][g * 4 + 1] * inv);
                o[2] = to_f16_sat(oT[db][g * 4 + 2] * inv);
                o[3] = to_f16_sat(oT[db][g * 4 + 3] * inv);
                *(f16x4*)(Ow + lq * 128 + (((db * 4 + g) ^ (lq & 7)) << 4) + hi * 8) = o;
            }
        __builtin_amdgcn_wave_barrier();
        _Float16* __restrict__ Og = (_Float16*)p.O + ((size_t)seq * p.Tp + qw0) * p.ldo + h * 64;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            const u32x4 v = *(const u32x4*)(Ow + row * 128 + ((ch ^ (row & 7)) << 4));
#if EEND_AF_NT
            __builtin_nontemporal_store(v, (u32x4*)(Og + (size_t)row * p.ldo + ch * 8));
#else
            *(u32x4*)(Og + (size_t)row * p.ldo + ch * 8) = v;
#endif
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (has_big) run_pass(qb_big);
    if (has_small) run_pass(qb_small);
    have_pre = pre_issued;
    __syncthreads();                                 // every wave is done with K / V^T / the Q slot before the next item
    }
}

}  // namespace

int eend_launch_inproj_attn(const InprojAttnParams& p, hipStream_t stream) {
    if (p.Tp <= 0 || p.Tp > 512 || (p.Tp % 64) != 0 || (p.ldo & 7) || (p.ldx & 7) || p.H != 4 || p.nseq <= 0) return EEND_EINVAL;
    const int smem = 2 * (p.Tp / KB) * TILE + NW * OSTG;
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)inproj_attn_kernel, 2 * 8 * TILE + NW * OSTG)) return EEND_ELAUNCH;
    int n_cu = eend_cu_count() & ~31;                // multiple of 32: a persistent workgroup keeps its head (and its XCD)
    if (n_cu <= 0) n_cu = 32;
    if (const char* e = getenv("EEND_AF_PERSIST")) { if (atoi(e) == 0) n_cu = 1 << 30; }     // A/B: one workgroup per item
    const int nitems = p.nseq * 4;
    hipLaunchKernelGGL(inproj_attn_kernel, dim3(nitems < n_cu ? nitems : n_cu), dim3(512), smem, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
