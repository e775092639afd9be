// Backward of the causal time-axis attention (attn.hip / attn_full.hip), flash style: the (T, T) probabilities
// are recomputed from Q, K and the saved per-row log-sum-exp; nothing quadratic is stored.
//
//   S2 = (Q K^T) * scale_log2        P = 2^(S2 - L2)  (masked: j - i <= mask_delay, j < kv_len)
//   dP = dO V^T       D_i = <dO_i, O_i>      dS = P o (dP - D)
//   dQ = sq * dS K        dK = sk * dS^T Q        dV = P^T dO
//
// Two kernels, each deterministic (no atomics):
//   attn_bwd_dq_kernel : one workgroup per 128 queries, loops over key tiles -- the forward kernel's transposed
//                        formulation (lane = query, S^T = K Q^T), with dP^T = V dO^T beside it and
//                        dQ^T += K^T dS^T where the forward has O^T += V^T P^T.
//   attn_bwd_dkv_kernel: one workgroup per 128 keys (a wave owns 32 keys, so no cross-wave reduction), loops over
//                        32-query blocks staged in LDS and shared by the 4 waves; lane = key (S = Q K^T), and
//                        dV^T += dO^T P, dK^T += Q^T dS.
// Every MFMA operand is contraction-contiguous in memory: Q, K, V come in [t][d] head layout and Q^T, K^T in
// [d][t] (both written by the in-projection, proj.hip PROJ_HEADS_BOTH); dO is row-major and dO^T is produced by
// heads_transpose_kernel below (8x8 register transposes).  Query rows of the A operands are fed with index
// bits 2<->3 swapped (swap23), which makes the 8 contraction elements a lane holds after the first MFMA
// contiguous for the second one (same trick as the forward kernel).
#include "train_common.h"
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) char lds_char;

constexpr int KB = 64;
constexpr int TILE = KB * 128;       // one [64][64] bf16 tile

DEV int swap23(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

// RET: the same two kernels for the LS-EEND retention core (retention.py:146-194 with the detached scales folded into
// dO = o~ = c_t * d out_t): no softmax -- "dS" is the masked A = o~ V^T itself and "P" the masked S = Q K^T -- the
// mask is block-diagonal causal over chunks of p.L frames, and the cross-chunk terms come from the 64x64 states of
// ret_bwd_scan_kernel (retention_bwd.hip):  dQ += o~ Spre^T,  dK += R v,  dV += R^T k.
template <bool RET>
__global__ __launch_bounds__(256, 2)
void attn_bwd_dq_kernel(const AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem[6 * TILE];   // K[2], V[2], K^T[2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqt = (p.Tp + 127) / 128;
    const int qt = nqt - 1 - (int)blockIdx.x;            // heavy (late) query tiles first
    const int h = blockIdx.y, seq = blockIdx.z;
    const int q0 = qt * 128, qw0 = q0 + wave * 32;
    const int lq = lane & 31, hi = lane >> 5;
    const int q = qw0 + lq;
    const int qc = q < p.Tp ? q : p.Tp - 1;
    const size_t sh = (size_t)seq * p.H + h;
    const __bf16* __restrict__ Qg = (const __bf16*)p.Q + sh * p.Tp * 64;
    const __bf16* __restrict__ Kg = (const __bf16*)p.K + sh * p.Tp * 64;
    const __bf16* __restrict__ Vg = (const __bf16*)p.V + sh * p.Tp * 64;
    const __bf16* __restrict__ Ktg = (const __bf16*)p.Kt + sh * 64 * p.Tp;
    const __bf16* __restrict__ dOg = (const __bf16*)p.dO + (size_t)seq * p.Tp * p.ldo + h * 64;

    int last_key = q0 + 127 + p.mask_delay;
    last_key = last_key < p.kv_len - 1 ? last_key : p.kv_len - 1;
    const int ntiles = last_key < 0 ? 0 : last_key / KB + 1;
    // RET: keys start at the chunk of the block's first query; per lane at its own chunk start cs_q
    const int jt0 = RET ? ((q0 / p.L) * p.L) / KB : 0;
    const int cs_q = RET ? (qc / p.L) * p.L : 0;
    const int qw0c = qw0 < p.Tp - 1 ? qw0 : p.Tp - 1;
    const int w_first_key = RET ? (qw0c / p.L) * p.L : 0;

    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *(const bf16x8*)(Qg + (size_t)qc * 64 + ks * 16 + hi * 8);
        dof[ks] = *(const bf16x8*)(dOg + (size_t)qc * p.ldo + ks * 16 + hi * 8);
    }
    const float L2 = RET ? 0.f : p.Lse[sh * p.Tp + qc], Dq = RET ? 0.f : p.Dh[sh * p.Tp + qc];

    // staging by LDS-DMA (no staging registers): per 64-key tile 8 pieces each of K, V ([64 keys][128 B]) and K^T
    // ([64 d][128 B]), swz128 images via the per-lane source address; 6 pieces per wave
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, p.Tp * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, p.Tp * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)Ktg, 0, p.Tp * 128, 0x00020000);
    const int r8 = lane >> 3, c8 = lane & 7;
#define DQ_DMA(j, buf)                                                                                         \
    do {                                                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                                     \
            const int pc = wave * 6 + i_, kind = pc >> 3, sp = pc & 7;                                         \
            const int row = sp * 8 + r8, ch = c8 ^ ((row >> 1) & 7);                                           \
            char* dst = smem + (2 * kind + (buf)) * TILE + sp * 1024;                                          \
            if (RET && kind == 0) continue;                                                                    \
            if (kind == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_char*)dst, 16, row * 128 + ch * 16, (j) * KB * 128, 0, 0); \
            else if (kind == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_char*)dst, 16, row * 128 + ch * 16, (j) * KB * 128, 0, 0); \
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (lds_char*)dst, 16, row * p.Tp * 2 + ch * 16, (j) * KB * 2, 0, 0); \
        }                                                                                                      \
    } while (0)

    f32x16 dqT[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dqT[0][i] = 0.f; dqT[1][i] = 0.f; }

    if (ntiles > jt0) DQ_DMA(jt0, 0);
    const int krow = swap23(lq);
    for (int j = jt0; j < ntiles; ++j) {
        const int buf = (j - jt0) & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile j landed; every wave is past the other buffer
        if (j + 1 < ntiles) DQ_DMA(j + 1, buf ^ 1);
        const int key0 = j * KB;
        if (key0 <= qw0 + 31 + p.mask_delay && (!RET || key0 + KB - 1 >= w_first_key)) {
            const char* kb_ = smem + buf * TILE;
            const char* vb_ = smem + (2 + buf) * TILE;
            const char* tb_ = smem + (4 + buf) * TILE;
            f32x16 s[2], dp[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { s[kb][i] = 0.f; dp[kb][i] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if constexpr (!RET) {
                        const bf16x8 kf = *(const bf16x8*)(kb_ + swz128(kb * 32 + krow, ks * 2 + hi));
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
                    }
                    const bf16x8 vf = *(const bf16x8*)(vb_ + swz128(kb * 32 + krow, ks * 2 + hi));
                    dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp[kb], 0, 0, 0);
                }
            }
            // reg i of s[kb] in lane (q, hi) <-> key = key0 + kb*32 + (i&7) + 8*hi + 16*(i>>3)
            const int lim = q + p.mask_delay < p.kv_len - 1 ? q + p.mask_delay : p.kv_len - 1;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = key0 + kb * 32 + (i & 7) + 8 * hi + 16 * (i >> 3);
                    if constexpr (RET) {
                        s[kb][i] = (key <= lim && key >= cs_q) ? dp[kb][i] : 0.f;       // A^T = V o~^T, block-diagonal causal
                    } else {
                        const float pv = key <= lim ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], p.scale_log2, -L2)) : 0.f;
                        // forward: O = (P o keep * scale) V  ->  dP = (dO V^T) o keep * scale
                        const float dpe = drop_apply(p.drop, dp[kb][i], (unsigned)(sh * p.Tp + qc), (unsigned)key);
                        s[kb][i] = pv * (dpe - Dq);
                    }
                }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 pf;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) pf[jj] = (__bf16)s[kb][kk * 8 + jj];
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const bf16x8 tf = *(const bf16x8*)(tb_ + swz128(db * 32 + lq, kb * 4 + kk * 2 + hi));
                        dqT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, pf, dqT[db], 0, 0, 0);
                    }
                }
        }
    }
#undef DQ_DMA
    if constexpr (RET) {
        // cross-chunk term dQ^T += Spre_c o~^T for every chunk c the wave's rows belong to (prefix state, hi/lo bf16)
        const int c_lo = qw0c / p.L;
        int wq_last = qw0 + 31;
        wq_last = wq_last < p.Tp - 1 ? wq_last : p.Tp - 1;
        const int c_hi = wq_last / p.L, c_q = qc / p.L;
        for (int c = c_lo; c <= c_hi; ++c) {
            if (c == 0 || c >= p.nc) continue;
            const __bf16* __restrict__ Sg = (const __bf16*)p.St + ((sh * p.nc + c) * 6) * 4096;
            const bool mine = (c_q == c);
            f32x16 x[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) { x[0][i] = 0.f; x[1][i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 om = dof[ks];
                if (!mine) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) om[jj] = (__bf16)0.f;
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 sa = *(const bf16x8*)(Sg + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                    const bf16x8 sb = *(const bf16x8*)(Sg + 4096 + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                    x[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, om, x[db], 0, 0, 0);
                    x[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb, om, x[db], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { dqT[0][i] += x[0][i]; dqT[1][i] += x[1][i]; }
        }
    }
    // dQ[q][h*64 + d] = sq * dQ^T[d][q]; reg i of dqT[db] <-> d = db*32 + 8*(i>>2) + 4*hi + (i&3)
    if (q < p.Tp) {
        __bf16* __restrict__ out = (__bf16*)p.dQKV + ((size_t)seq * p.Tp + q) * p.ldg + h * 64;
        const float sc = q < p.q_len ? p.sq : 0.f;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16(dqT[db][g * 4 + 0] * sc, dqT[db][g * 4 + 1] * sc);
                pk.y = pack_bf16(dqT[db][g * 4 + 2] * sc, dqT[db][g * 4 + 3] * sc);
                *(uint2*)(out + db * 32 + g * 8 + hi * 4) = pk;
            }
    }
}

// LDS image of a [64 d][32 q] transposed query block: 64-byte rows, 16-byte chunks XOR-swizzled with (row>>2)&3 so
// the 16-lane groups of a ds_read_b128 (rows 0-3,12-15,20-27 / 4-11,16-19,28-31) hit 16 distinct bank quads.
DEV int swz64(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

// Staging of attn_bwd_dkv_kernel: 64 queries (two 32-query sub-blocks) per stage, by LDS-DMA (no staging registers);
// per sub-block u the four tiles of the round-1 layout -- Q[32][64], dO[32][64] (swz128 images), Q^T[64][32],
// dO^T[64][32] (swz64 images) -- then one 1 KB slot each for the stage's 64 log-sum-exps and 64 D values.
constexpr int DKV_SUB = 16384;
constexpr int DKV_STAGE = 2 * DKV_SUB + 2048;

#ifndef EEND_DKV_WAVES
#define EEND_DKV_WAVES 4          // waves (32 keys each) per workgroup: 4 -> 128 keys, two workgroups per CU; 8 -> 256 keys, one
#endif
constexpr int DKV_NW = EEND_DKV_WAVES;

template <bool RET>
__global__ __launch_bounds__(DKV_NW * 64, DKV_NW == 4 ? 2 : 1)
void attn_bwd_dkv_kernel(const AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];         // 2 stages x 34 KB: two workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, seq = blockIdx.z;
    const int k0 = blockIdx.x * (DKV_NW * 32), kw0 = k0 + wave * 32;
    const int lq = lane & 31, hi = lane >> 5;
    const size_t sh = (size_t)seq * p.H + h;
    const __bf16* __restrict__ Kg = (const __bf16*)p.K + sh * p.Tp * 64;
    const __bf16* __restrict__ Vg = (const __bf16*)p.V + sh * p.Tp * 64;

    const bool active = kw0 < p.Tp;
    const int key = kw0 + lq;                          // this lane's key (B-operand column)
    const int keyc = key < p.Tp ? key : p.Tp - 1;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *(const bf16x8*)(Kg + (size_t)keyc * 64 + ks * 16 + hi * 8);
        vf[ks] = *(const bf16x8*)(Vg + (size_t)keyc * 64 + ks * 16 + hi * 8);
    }
    asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]));

    int qlim = p.q_len < p.Tp ? p.q_len : p.Tp;
    int qstart = k0 - p.mask_delay;
    if (qstart < 0) qstart = 0;
    int qend = qlim;
    int ce_key = p.Tp, w_ce = p.Tp;                    // RET: first frame after the chunk of this lane's key / of the wave's last key
    if constexpr (RET) {
        int k_last = k0 + DKV_NW * 32 - 1;
        k_last = k_last < p.Tp - 1 ? k_last : p.Tp - 1;
        const int be = (k_last / p.L + 1) * p.L;       // queries beyond the chunk of the block's last key see none of its keys
        qend = be < qlim ? be : qlim;
        ce_key = (keyc / p.L + 1) * p.L;
        int kwl = kw0 + 31;
        kwl = kwl < p.Tp - 1 ? kwl : p.Tp - 1;
        w_ce = (kwl / p.L + 1) * p.L;
    }
    const int nqb = (qend + 31) / 32;
    const int st0 = qstart / 64, nst = (nqb + 1) / 2;  // stages of 64 queries

    // 34 DMA pieces of 1 KB per stage: per sub-block 4 (Q) + 4 (dO) + 4 (Q^T) + 4 (dO^T), then L and D; 8 per wave (+2 on
    // wave 0).  The swizzled images are produced by permuting the per-lane SOURCE address.
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)((const __bf16*)p.Q + sh * p.Tp * 64), 0, p.Tp * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t rqt = __builtin_amdgcn_make_buffer_rsrc((void*)((const __bf16*)p.Qt + sh * 64 * p.Tp), 0, p.Tp * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)((const __bf16*)p.dO + (size_t)seq * p.Tp * p.ldo + h * 64), 0,
                                                                         (p.Tp - 1) * p.ldo * 2 + 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdot = __builtin_amdgcn_make_buffer_rsrc((void*)((const __bf16*)p.dOt + sh * 64 * p.Tp), 0, p.Tp * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Lse + sh * p.Tp), 0, p.Tp * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Dh + sh * p.Tp), 0, p.Tp * 4, 0x00020000);
    // [32][64] tiles: a piece = 8 rows x 128 B; [64][32] tiles: a piece = 16 rows x 64 B
    const int r8 = lane >> 3, c8 = lane & 7, r4 = lane >> 2, c4 = lane & 3;
    auto dma_stage = [&](int st, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * DKV_STAGE;
#pragma unroll
        for (int i = 0; i < 32 / DKV_NW; ++i) {
            const int pc = wave * (32 / DKV_NW) + i;   // 0..31
            const int u = pc >> 4, kind = (pc >> 2) & 3, sp = pc & 3;
            char* dst = base + u * DKV_SUB + kind * 4096 + sp * 1024;
            const int q0 = st * 64 + u * 32;
            if (kind < 2) {                            // Q / dO rows q0 + sp*8 + r8, chunk c8 ^ ((row >> 1) & 7)
                const int row = sp * 8 + r8, ch = c8 ^ ((row >> 1) & 7);
                if (kind == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_char*)dst, 16, row * 128 + ch * 16, q0 * 128, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rdo, (lds_char*)dst, 16, row * p.ldo * 2 + ch * 16, q0 * p.ldo * 2, 0, 0);
            } else {                                   // Q^T / dO^T rows d = sp*16 + r4, chunk c4 ^ ((d >> 2) & 3)
                const int d = sp * 16 + r4, ch = c4 ^ ((d >> 2) & 3);
                if (kind == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rqt, (lds_char*)dst, 16, d * p.Tp * 2 + ch * 16, q0 * 2, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rdot, (lds_char*)dst, 16, d * p.Tp * 2 + ch * 16, q0 * 2, 0, 0);
            }
        }
        if (!RET && wave == 0) {                       // 64 floats each: lanes 0..15 carry them, the rest re-read in range
            const int off = (lane & 15) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_char*)(base + 2 * DKV_SUB), 16, off, st * 256, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_char*)(base + 2 * DKV_SUB + 1024), 16, off, st * 256, 0, 0);
        }
    };

    f32x16 dkT[2], dvT[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dkT[0][i] = 0.f; dkT[1][i] = 0.f; dvT[0][i] = 0.f; dvT[1][i] = 0.f; }

    if (st0 < nst) dma_stage(st0, 0);
    const int qrow = swap23(lq);
    for (int st = st0; st < nst; ++st) {
        const int buf = (st - st0) & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage landed; every wave is past the other buffer
        if (st + 1 < nst) dma_stage(st + 1, buf ^ 1);
        const char* sb = smem + buf * DKV_STAGE;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int qw0 = st * 64 + u * 32;
            if (!(active && qw0 < qlim && kw0 <= qw0 + 31 + p.mask_delay && (!RET || qw0 < w_ce))) continue;
            const char* b_ = sb + u * DKV_SUB;
            f32x16 s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 qa = *(const bf16x8*)(b_ + swz128(qrow, ks * 2 + hi));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
                const bf16x8 da = *(const bf16x8*)(b_ + 4096 + swz128(qrow, ks * 2 + hi));
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
            }
            // reg r in lane (key, hi) <-> query = qw0 + (r&7) + 8*hi + 16*(r>>3): two runs of 8 consecutive queries
            bf16x8 pf[2], sf[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float* lp = (const float*)(sb + 2 * DKV_SUB) + u * 32 + 16 * g + 8 * hi;
                const float* dpt = (const float*)(sb + 2 * DKV_SUB + 1024) + u * 32 + 16 * g + 8 * hi;
                float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, d0 = a0, d1 = a0;
                if constexpr (!RET) { a0 = *(const float4*)lp; a1 = *(const float4*)(lp + 4); d0 = *(const float4*)dpt; d1 = *(const float4*)(dpt + 4); }
                const float l2v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = g * 8 + e;
                    const int qi = qw0 + e + 8 * hi + 16 * g;
                    const bool ok = key <= qi + p.mask_delay && key < p.kv_len && qi < qlim && (!RET || qi < ce_key);
                    if constexpr (RET) {                           // P := masked S = Q K^T, dS := masked A = o~ V^T
                        pf[g][e] = (__bf16)(ok ? s[r] : 0.f);
                        sf[g][e] = (__bf16)(ok ? dp[r] : 0.f);
                        continue;
                    }
                    const float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.scale_log2, -l2v[e])) : 0.f;
                    float kfac = 1.0f;                             // the forward's dropout factor of this (query, key) pair
                    if (p.drop.thresh24) kfac = drop_keep(p.drop, (unsigned)(sh * p.Tp + qi), (unsigned)key) ? p.drop.scale : 0.f;
                    pf[g][e] = (__bf16)(pv * kfac);
                    sf[g][e] = (__bf16)(pv * (dp[r] * kfac - dv[e]));
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 dot = *(const bf16x8*)(b_ + 12288 + swz64(db * 32 + lq, kk * 2 + hi));
                    dvT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf[kk], dvT[db], 0, 0, 0);
                    const bf16x8 qt = *(const bf16x8*)(b_ + 8192 + swz64(db * 32 + lq, kk * 2 + hi));
                    dkT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, sf[kk], dkT[db], 0, 0, 0);
                }
        }
    }
    if constexpr (RET) {
        // cross-chunk terms from the queries of later chunks: dK^T += R_c V^T, dV^T += R_c^T K^T (suffix state, hi/lo bf16)
        if (active) {
            int kwl = kw0 + 31;
            kwl = kwl < p.Tp - 1 ? kwl : p.Tp - 1;
            const int c_lo = kw0 / p.L, c_hi = kwl / p.L, c_k = keyc / p.L;
            for (int c = c_lo; c <= c_hi; ++c) {
                if (c >= p.nc - 1) continue;                   // no later chunk (or slab padding beyond the last chunk)
                const __bf16* __restrict__ Rg = (const __bf16*)p.St + ((sh * p.nc + c) * 6 + 2) * 4096;
                const bool mine = (c_k == c);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bf16x8 vm = vf[ks], km = kf[ks];
                    if (!mine) {
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) { vm[jj] = (__bf16)0.f; km[jj] = (__bf16)0.f; }
                    }
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const int off = (db * 32 + lq) * 64 + ks * 16 + hi * 8;
                        dkT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Rg + off), vm, dkT[db], 0, 0, 0);
                        dkT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Rg + 4096 + off), vm, dkT[db], 0, 0, 0);
                        dvT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Rg + 2 * 4096 + off), km, dvT[db], 0, 0, 0);
                        dvT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Rg + 3 * 4096 + off), km, dvT[db], 0, 0, 0);
                    }
                }
            }
        }
    }
    // dK[key][h*64 + d], dV likewise; reg i <-> d = db*32 + 8*(i>>2) + 4*hi + (i&3)
    if (active && key < p.Tp) {
        __bf16* __restrict__ out = (__bf16*)p.dQKV + ((size_t)seq * p.Tp + key) * p.ldg + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16(dkT[db][g * 4 + 0] * p.sk, dkT[db][g * 4 + 1] * p.sk);
                pk.y = pack_bf16(dkT[db][g * 4 + 2] * p.sk, dkT[db][g * 4 + 3] * p.sk);
                *(uint2*)(out + 256 + db * 32 + g * 8 + hi * 4) = pk;
                pk.x = pack_bf16(dvT[db][g * 4 + 0], dvT[db][g * 4 + 1]);
                pk.y = pack_bf16(dvT[db][g * 4 + 2], dvT[db][g * 4 + 3]);
                *(uint2*)(out + 512 + db * 32 + g * 8 + hi * 4) = pk;
            }
    }
}

// bf16 [nseq*Tp][ld] (head h at columns h*64..) -> [nseq][H][64][Tp]: 8x8 register transposes, one unit per thread.
__global__ __launch_bounds__(256)
void heads_transpose_kernel(const unsigned short* __restrict__ in, int ld, unsigned short* __restrict__ out, int nseq, int H, int Tp) {
    const long u = (long)blockIdx.x * 256 + threadIdx.x;
    const int tg8 = Tp >> 3;
    const long total = (long)nseq * H * tg8 * 8;
    if (u >= total) return;
    const int dch = (int)(u & 7);
    const long v = u >> 3;
    const int tg = (int)(v % tg8);
    const long sh = v / tg8;
    const int seq = (int)(sh / H), h = (int)(sh - (long)seq * H);
    u32x4 a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = *(const u32x4*)(in + ((size_t)seq * Tp + tg * 8 + r) * ld + h * 64 + dch * 8);
    transpose8x8_b16(a, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) *(u32x4*)(out + ((size_t)sh * 64 + dch * 8 + e) * Tp + tg * 8) = b[e];
}

}  // namespace

int eend_launch_attn_bwd(const AttnBwdParams& p, hipStream_t stream) {
    if (!p.Q || !p.K || !p.V || !p.dO || !p.dOt || !p.Lse || !p.Dh || !p.dQKV) return EEND_EINVAL;
    if (p.nseq <= 0 || p.nseq > 65535 || p.H <= 0 || p.Tp <= 0 || (p.Tp % 64) || (p.ldo & 7) || (p.ldg & 3) || p.kv_len <= 0 ||
        p.kv_len > p.Tp || p.q_len <= 0)
        return EEND_EINVAL;
#ifndef EEND_ATTN_BWD_TWO_KERNELS                       // (study build: the round-2 two-kernel form at every size)
    if (eend_attn_bwd_fused_ok(p, false)) return eend_launch_attn_bwd_fused(p, false, stream);
#endif
    if (!p.Qt || !p.Kt) return EEND_EINVAL;             // the two-kernel form reads the [d][t] copies
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3((p.Tp + 127) / 128, p.H, p.nseq), dim3(256), 0, stream, p);
    if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)attn_bwd_dkv_kernel<false>, 2 * DKV_STAGE)) return EEND_ELAUNCH;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, dim3((p.Tp + DKV_NW * 32 - 1) / (DKV_NW * 32), p.H, p.nseq), dim3(DKV_NW * 64), 2 * DKV_STAGE, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// Retention core backward (see the RET note above): p.dO = o~ (bf16 rows), p.dOt its head-transposed copy, p.St the
// states of ret_bwd_scan_kernel, p.L / p.nc the chunking; only query / key tiles inside the nc * L valid frames run.
int eend_launch_ret_bwd(const AttnBwdParams& p, hipStream_t stream) {
    if (!p.Q || !p.K || !p.V || !p.dO || !p.St || !p.dQKV) return EEND_EINVAL;
    if (p.nseq <= 0 || p.nseq > 65535 || p.H <= 0 || p.Tp <= 0 || (p.Tp % 64) || (p.ldo & 7) || (p.ldg & 3) || p.L <= 0 || p.nc <= 0 ||
        (long)p.nc * p.L > p.Tp || p.mask_delay != 0 || p.kv_len != p.nc * p.L || p.q_len != p.nc * p.L)
        return EEND_EINVAL;
#ifndef EEND_ATTN_BWD_TWO_KERNELS
    if (eend_attn_bwd_fused_ok(p, true)) return eend_launch_attn_bwd_fused(p, true, stream);
#endif
    if (!p.Qt || !p.Kt || !p.dOt) return EEND_EINVAL;   // the two-kernel form reads the [d][t] copies
    // full-slab grids: the blocks beyond the nc * L valid frames only write the zero rows of dQKV
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3((p.Tp + 127) / 128, p.H, p.nseq), dim3(256), 0, stream, p);
    if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)attn_bwd_dkv_kernel<true>, 2 * DKV_STAGE)) return EEND_ELAUNCH;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, dim3((p.Tp + DKV_NW * 32 - 1) / (DKV_NW * 32), p.H, p.nseq), dim3(DKV_NW * 64), 2 * DKV_STAGE, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_heads_transpose(const void* in, int ld, void* out, int nseq, int H, int Tp, hipStream_t stream) {
    if (!in || !out || nseq <= 0 || H <= 0 || Tp <= 0 || (Tp & 7) || (ld & 7)) return EEND_EINVAL;
    const long total = (long)nseq * H * (Tp >> 3) * 8;
    hipLaunchKernelGGL(heads_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const unsigned short*)in, ld,
                       (unsigned short*)out, nseq, H, Tp);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
