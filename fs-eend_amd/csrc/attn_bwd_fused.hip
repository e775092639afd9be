// Backward of the causal time-axis attention / of the retention core for windows of up to 512 frames, ONE launch, one 8-wave
// workgroup per (sequence, head[, chunk]) (round 6; attn_bwd.hip stays as the form for longer windows).
//
//   S2 = (Q K^T) * scale_log2        P = 2^(S2 - L2)  (visible: key <= query + mask_delay, key < kv_len)
//   dP = dO V^T (o keep * scale)     D_i = <dO_i, O_i>      dS = P o (dP - D)
//   dQ = sq * dS K        dK = sk * dS^T Q        dV = (P o keep * scale)^T dO
//
// Why one workgroup per (sequence, head): every operand is read from HBM / L2 exactly once -- K and V of the whole window sit in
// LDS (2 x 64 KB) for the dQ phase, then Q and dO take their place for the dK / dV phase -- instead of once per 128-query /
// 128-key tile (attn_bwd.hip: 34 KB of LDS-DMA behind a vmcnt(0) + barrier per 64 rows, 0.06 of the MFMA peak).  There is no
// barrier inside a phase: a wave only reads the resident images.
//   * phase 1, lane = query (S^T = K Q^T as in the forward kernel): a wave owns two 32-query blocks (b, nblk-1-b: equal causal
//     work for every wave), keeps Q / dO of the block in registers and walks the visible 32-key blocks:
//     S^T, dP^T (8 MFMAs, A = K / V rows from LDS), the elementwise part in registers, dQ^T += K^T dS^T (4 MFMAs).
//   * phase 2, lane = key (S = Q K^T): two 32-key blocks per wave, K / V of the block in registers, walks the 32-query blocks that
//     see it: S, dP (8 MFMAs, A = Q / dO rows from LDS), dV^T += dO^T P, dK^T += Q^T dS (8 MFMAs).
//   * the transposed operands (K^T, Q^T, dO^T) are ds_read_b64_tr_b16 reads of the SAME row-major images: a 16-lane group reads
//     4 rows x 16 features and every lane receives the 4 rows of its feature.  The accumulator of the first product holds, per
//     lane, rows {0-3, 8-11} / {4-7, 12-15} (+16): exactly two such reads -- no row permutation of the first operand, and no
//     [d][t] copies of Q, K, dO in HBM (Qt / Kt / dOt of AttnBwdParams are not read here).
//   * LDS image [row][128 B], 16-byte chunk c of row r at c ^ g(r), g(r) = ((r>>1)&3) | ((((r>>1)^(r>>3))&1)<<2): conflict-free
//     for the MFMA row fragments (ds_read_b128) AND for the transposing reads (found by exhaustive search over the GF(2)-linear
//     swizzles, tools in the round-6 log); written by LDS-DMA with the permutation applied to the per-lane source address.
//   * both phases recompute S and dP (7 products instead of 5): the kernel is bound by the elementwise VALU work (exp2, the
//     dropout hash, the visibility mask on the diagonal blocks only), not by the MFMA pipe.
// RET: the same for the LS-EEND retention core (retention.py:146-194; no softmax: "dS" is the masked A = o~ V^T and "P" the masked
// S = Q K^T), one workgroup per chunk of p.L <= 512 frames, cross-chunk terms from the states of ret_bwd_scan_kernel.
// Deterministic: no atomics, fixed summation order.
#include "train_common.h"
#include "kernels.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) char lds_char;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int FW = 512;                       // window rows resident in LDS
constexpr int IMG = FW * 128;                 // one [512][64] bf16 image
constexpr int L_A = 0, L_B = IMG, L_LSE = 2 * IMG, L_DH = 2 * IMG + FW * 4;
constexpr int FUSED_SMEM = 2 * IMG + 2 * FW * 4;          // 135168

DEV int gsw(int r) { return ((r >> 1) & 3) | ((((r >> 1) ^ (r >> 3)) & 1) << 2); }

template <int OFF>
DEV u32x2 tr_read(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// (the compiler does not see the reads above as LDS operations: their registers are released by hand)
DEV void tr_wait0(u32x2 (&a)[2][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0][0]), "+v"(a[0][0][1]), "+v"(a[0][1][0]), "+v"(a[0][1][1]), "+v"(a[1][0][0]),
                 "+v"(a[1][0][1]), "+v"(a[1][1][0]), "+v"(a[1][1][1]));
}
DEV bf16x8 frag_of(const u32x2 (&r)[2]) {
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    const u32x4v v = u32x4v{r[0][0], r[0][1], r[1][0], r[1][1]};
    return __builtin_bit_cast(bf16x8, v);
}

template <bool RET, bool DROP>
__global__ __launch_bounds__(512, 2)
void attn_bwd_fused_kernel(const AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x;
    const int item = blockIdx.y;
    const int seq = RET ? item / p.nc : item, c = RET ? item - seq * p.nc : 0;
    const int t0 = RET ? c * p.L : 0;                     // first frame of the window
    const int wl = RET ? p.L : p.Tp;                      // frames of the window
    const int nblk = (wl + 31) >> 5;
    const int md = RET ? 0 : p.mask_delay;
    const int kvl = RET ? p.L : (p.kv_len < p.Tp ? p.kv_len : p.Tp);          // keys of the window that exist
    const int qlim = RET ? p.L : (p.q_len < p.Tp ? p.q_len : p.Tp);           // queries that receive a gradient
    const size_t sh = (size_t)seq * p.H + h;
    const int lq = lane & 31, hi = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;

    // ---- LDS-DMA of a [rows][64] bf16 image: piece = 8 rows x 128 B, lane -> (row, position), source chunk = position ^ g(row)
    const int r8 = lane >> 3, c8 = lane & 7;
    auto dma_image = [&](const __amdgpu_buffer_rsrc_t& rs, int row_bytes, int dst) __attribute__((always_inline)) {
        for (int pc = wave; pc < nblk * 4; pc += 8) {
            const int row = pc * 8 + r8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_char*)(smem + dst + pc * 1024), 16, row * row_bytes + ((c8 ^ gsw(row)) << 4), 0, 0, 0);
        }
    };
    const char* Qg = (const char*)p.Q + (sh * p.Tp + t0) * 128;
    const char* Kg = (const char*)p.K + (sh * p.Tp + t0) * 128;
    const char* Vg = (const char*)p.V + (sh * p.Tp + t0) * 128;
    const char* dOg = (const char*)p.dO + (((size_t)seq * p.Tp + t0) * p.ldo + h * 64) * 2;
    // rows beyond the window's keys / queries read as zeros (RET: they belong to the next chunk)
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, kvl * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, kvl * 128, 0x00020000);
    dma_image(rk, 128, L_A);
    dma_image(rv, 128, L_B);
    if constexpr (!RET) {
        float* lse = (float*)(smem + L_LSE);
        float* dh = (float*)(smem + L_DH);
        if (tid < wl) { lse[tid] = p.Lse[sh * p.Tp + tid]; dh[tid] = p.Dh[sh * p.Tp + tid]; }
    }

    // per-lane offsets of the row fragments (lane = row lq, chunk ks * 2 + hi) and of the transposing reads
    // (16-lane group g: features db*32 + (g&1)*16 + ..., rows 16*m + 8*rd + 4*(g>>1) + (i16>>2); rd = 0 / 1 differ in chunk bit 2)
    unsigned offA[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) offA[ks] = lds0 + lq * 128 + (((ks * 2 + hi) ^ gsw(lq)) << 4);
    const int g16 = lane >> 4, i16 = lane & 15;
    unsigned offT[2][2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int row = 8 * rd + 4 * (g16 >> 1) + (i16 >> 2);
            const int kc = db * 4 + (g16 & 1) * 2 + ((i16 & 3) >> 1);
            offT[db][rd] = lds0 + row * 128 + ((kc ^ gsw(row)) << 4) + (i16 & 1) * 8;
        }
    // dropout hash: h = drop_mix((a * G + b) ^ seed), a = (seq*H + h)*Tp + query, b = key (common.h drop_keep)
    const unsigned aG0 = (unsigned)(sh * p.Tp + t0) * 0x9E3779B1u;
    const unsigned thr8 = p.drop.thresh24 << 8;            // (h >> 8) >= thresh24  <=>  h >= thresh24 << 8
    auto hash_keep = [&](unsigned x) __attribute__((always_inline)) {          // x = a * G + b
        return drop_mix(x ^ p.drop.seed) >= thr8;
    };

    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // K, V images and the row statistics are in LDS

    // ================= phase 1: dQ (lane = query) =================
    for (int half = 0; half < 2; ++half) {
        const int qb = half == 0 ? wave : nblk - 1 - wave;                       // blocks (w, nblk-1-w): equal causal work per wave
        if (wave > nblk - 1 - wave || (half == 1 && qb <= wave)) continue;       // (wave-uniform)
        const int q0 = qb * 32, q = q0 + lq;                                    // window-relative
        const int qc = q < wl ? q : wl - 1;
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (!RET) qf[ks] = *(const bf16x8*)(Qg + (size_t)qc * 128 + ks * 32 + hi * 16);
            dof[ks] = *(const bf16x8*)(dOg + (size_t)qc * p.ldo * 2 + ks * 32 + hi * 16);
        }
        float L2 = 0.f, Dq = 0.f;
        if constexpr (!RET) { L2 = p.Lse[sh * p.Tp + qc]; Dq = p.Dh[sh * p.Tp + qc]; }
        int kb_last = (q0 + 31 + md) >> 5;
        if (kb_last > nblk - 1) kb_last = nblk - 1;
        if (kb_last > (kvl - 1) >> 5) kb_last = (kvl - 1) >> 5;
        const int lim = q + md < kvl - 1 ? q + md : kvl - 1;                     // last key this lane's query sees
        const unsigned aGq = aG0 + (unsigned)q * 0x9E3779B1u;
        f32x16 dqT[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { dqT[0][i] = 0.f; dqT[1][i] = 0.f; }
        for (int kb = 0; kb <= kb_last; ++kb) {
            const int key0 = kb * 32;
            const unsigned rb = (unsigned)key0 * 128;
            // transposed K fragments of this key block (requested first: they are needed last)
            u32x2 kt[2][2][2];                                                   // [db][m][rd]
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                kt[db][0][0] = tr_read<0>(offT[db][0] + rb);         kt[db][0][1] = tr_read<0>(offT[db][1] + rb);            // m = 0: rows 0-15
                kt[db][1][0] = tr_read<16 * 128>(offT[db][0] + rb);  kt[db][1][1] = tr_read<16 * 128>(offT[db][1] + rb);     // m = 1: rows 16-31
            }
            f32x16 s = zero16, dp = zero16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if constexpr (!RET) {
                    const bf16x8 kf = *(const bf16x8*)(smem + L_A + (offA[ks] - lds0) + rb);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
                }
                const bf16x8 vf = *(const bf16x8*)(smem + L_B + (offA[ks] - lds0) + rb);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);
            }
            // reg r of s / dp in lane (query, hi) <-> key = key0 + 8*(r>>2) + 4*hi + (r&3)
            // blocks every query of the wave sees completely take the path without the visibility compares (two code paths: left
            // as one select the compiler evaluates the 16 compares for every block)
            const bool full = key0 + 31 <= q0 + md && key0 + 31 < kvl && q0 + 31 < wl;       // (wave-uniform)
            float ds[16];
            auto elementwise = [&](auto FULL) __attribute__((always_inline)) {
                constexpr bool F = decltype(FULL)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + 8 * (r >> 2) + 4 * hi + (r & 3);
                    if constexpr (RET) {
                        ds[r] = (F || key <= lim) ? dp[r] : 0.f;                 // A^T = V o~^T, causal inside the chunk
                    } else {
                        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.scale_log2, -L2));
                        if constexpr (!F) pv = key <= lim ? pv : 0.f;
                        float t;
                        if constexpr (DROP) {
                            const bool keep = hash_keep(aGq + (unsigned)(t0 + key));
                            t = keep ? __builtin_fmaf(dp[r], p.drop.scale, -Dq) : -Dq;
                        } else {
                            t = dp[r] - Dq;
                        }
                        ds[r] = pv * t;
                    }
                }
            };
            // (the two asm comments keep the paths apart: identical up to the selects, the optimiser would merge them again)
            if (full) { asm volatile("; fully visible block"); elementwise(std::true_type{}); }
            else { asm volatile("; block on the mask boundary"); elementwise(std::false_type{}); }
            bf16x8 pf[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[m][e] = (__bf16)ds[m * 8 + e];
            tr_wait0(kt);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dqT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(kt[db][m]), pf[m], dqT[db], 0, 0, 0);
        }
        if constexpr (RET) {
            // cross-chunk term dQ^T += Spre_c o~^T (prefix state of the earlier chunks, hi/lo bf16)
            if (c > 0) {
                const __bf16* __restrict__ Sg = (const __bf16*)p.St + ((sh * p.nc + c) * 6) * 4096;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const bf16x8 sa = *(const bf16x8*)(Sg + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                        const bf16x8 sb = *(const bf16x8*)(Sg + 4096 + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                        dqT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, dof[ks], dqT[db], 0, 0, 0);
                        dqT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb, dof[ks], dqT[db], 0, 0, 0);
                    }
            }
        }
        // dQ[q][h*64 + d] = sq * dQ^T[d][q]; reg i of dqT[db] <-> d = db*32 + 8*(i>>2) + 4*hi + (i&3)
        if (q < wl) {
            __bf16* __restrict__ out = (__bf16*)p.dQKV + ((size_t)seq * p.Tp + t0 + q) * p.ldg + h * 64;
            const float sc = q < qlim ? p.sq : 0.f;
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

    // ================= phase 2: dK, dV (lane = key); Q and dO take the place of K and V =================
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done with the K / V images
    {
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)Qg, 0, qlim * 128, 0x00020000);
        const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)dOg, 0, (qlim - 1) * p.ldo * 2 + 128, 0x00020000);
        dma_image(rq, 128, L_A);
        dma_image(rdo, p.ldo * 2, L_B);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const int nqb = (qlim + 31) >> 5;
    for (int half = 0; half < 2; ++half) {
        const int kb = half == 0 ? wave : nblk - 1 - wave;
        if (wave > nblk - 1 - wave || (half == 1 && kb <= wave)) continue;
        const int key0 = kb * 32, key = key0 + lq;
        const int keyc = key < wl ? key : wl - 1;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = *(const bf16x8*)(Kg + (size_t)keyc * 128 + ks * 32 + hi * 16);
            vf[ks] = *(const bf16x8*)(Vg + (size_t)keyc * 128 + ks * 32 + hi * 16);
        }
        const bool key_ok = key < kvl;
        int qb0 = key0 - md;
        qb0 = qb0 < 0 ? 0 : qb0 >> 5;
        f32x16 dkT[2], dvT[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { dkT[0][i] = 0.f; dkT[1][i] = 0.f; dvT[0][i] = 0.f; dvT[1][i] = 0.f; }
        const unsigned bkey = aG0 + (unsigned)(t0 + key);                        // + query * G per element
        for (int qb = qb0; qb < nqb; ++qb) {
            const int q0 = qb * 32;
            const unsigned rb = (unsigned)q0 * 128;
            u32x2 qt[2][2][2], dot[2][2][2];                                     // [db][m][rd]
            const unsigned rbB = rb + L_B;                                       // (65536 does not fit the 16-bit offset field)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                qt[db][0][0] = tr_read<0>(offT[db][0] + rb);                qt[db][0][1] = tr_read<0>(offT[db][1] + rb);
                qt[db][1][0] = tr_read<16 * 128>(offT[db][0] + rb);         qt[db][1][1] = tr_read<16 * 128>(offT[db][1] + rb);
                dot[db][0][0] = tr_read<0>(offT[db][0] + rbB);              dot[db][0][1] = tr_read<0>(offT[db][1] + rbB);
                dot[db][1][0] = tr_read<16 * 128>(offT[db][0] + rbB);       dot[db][1][1] = tr_read<16 * 128>(offT[db][1] + rbB);
            }
            f32x16 s = zero16, dp = zero16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 qa = *(const bf16x8*)(smem + L_A + (offA[ks] - lds0) + rb);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
                const bf16x8 da = *(const bf16x8*)(smem + L_B + (offA[ks] - lds0) + rb);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[ks], dp, 0, 0, 0);
            }
            // reg r in lane (key, hi) <-> query = q0 + 8*(r>>2) + 4*hi + (r&3)
            const bool full = key0 + 31 <= q0 + md && key0 + 31 < kvl && q0 + 31 < qlim;     // (wave-uniform)
            unsigned pw[8], sw[8];                                               // packed bf16 pairs: word (r >> 1) of P / dS
            auto elementwise = [&](auto FULL) __attribute__((always_inline)) {
                constexpr bool F = decltype(FULL)::value;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float pe[4], se[4];
                    f32x4 l4 = f32x4{0.f, 0.f, 0.f, 0.f}, d4 = l4;
                    if constexpr (!RET) {
                        l4 = *(const f32x4*)(smem + L_LSE + (q0 + 8 * g + 4 * hi) * 4);
                        d4 = *(const f32x4*)(smem + L_DH + (q0 + 8 * g + 4 * hi) * 4);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = g * 4 + e;
                        const int qi = q0 + 8 * g + 4 * hi + e;
                        const bool ok = F || (key_ok && key <= qi + md && qi < qlim);
                        if constexpr (RET) {                                     // P := masked S = Q K^T, dS := masked A = o~ V^T
                            pe[e] = ok ? s[r] : 0.f;
                            se[e] = ok ? dp[r] : 0.f;
                        } else {
                            float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.scale_log2, -l4[e]));
                            if constexpr (!F) pv = ok ? pv : 0.f;
                            if constexpr (DROP) {
                                const bool keep = hash_keep(bkey + (unsigned)qi * 0x9E3779B1u);
                                pe[e] = keep ? pv * p.drop.scale : 0.f;
                                se[e] = pv * (keep ? __builtin_fmaf(dp[r], p.drop.scale, -d4[e]) : -d4[e]);
                            } else {
                                pe[e] = pv;
                                se[e] = pv * (dp[r] - d4[e]);
                            }
                        }
                    }
                    pw[g * 2] = pack_bf16(pe[0], pe[1]); pw[g * 2 + 1] = pack_bf16(pe[2], pe[3]);
                    sw[g * 2] = pack_bf16(se[0], se[1]); sw[g * 2 + 1] = pack_bf16(se[2], se[3]);
                }
            };
            // (the two asm comments keep the paths apart: identical up to the selects, the optimiser would merge them again)
            if (full) { asm volatile("; fully visible block"); elementwise(std::true_type{}); }
            else { asm volatile("; block on the mask boundary"); elementwise(std::false_type{}); }
            bf16x8 pf[2], sf[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                pf[m] = __builtin_bit_cast(bf16x8, u32x4{pw[m * 4], pw[m * 4 + 1], pw[m * 4 + 2], pw[m * 4 + 3]});
                sf[m] = __builtin_bit_cast(bf16x8, u32x4{sw[m * 4], sw[m * 4 + 1], sw[m * 4 + 2], sw[m * 4 + 3]});
            }
            tr_wait0(qt);
            tr_wait0(dot);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dvT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(dot[db][m]), pf[m], dvT[db], 0, 0, 0);
                    dkT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(qt[db][m]), sf[m], dkT[db], 0, 0, 0);
                }
        }
        if constexpr (RET) {
            // cross-chunk terms from the queries of later chunks: dK^T += R_c V^T, dV^T += R_c^T K^T (suffix state, hi/lo bf16)
            if (c < p.nc - 1) {
                const __bf16* __restrict__ Rg = (const __bf16*)p.St + ((sh * p.nc + c) * 6 + 2) * 4096;
                const bool mine = key < wl;
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
        // dK[key][256 + h*64 + d], dV[key][512 + h*64 + d]; reg i <-> d = db*32 + 8*(i>>2) + 4*hi + (i&3)
        if (key < wl) {
            __bf16* __restrict__ out = (__bf16*)p.dQKV + ((size_t)seq * p.Tp + t0 + key) * p.ldg + h * 64;
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
}

// RET: rows of the slab beyond the nc * L valid frames receive zero gradients (the per-chunk workgroups do not cover them)
__global__ __launch_bounds__(256)
void zero_tail_rows_kernel(__bf16* __restrict__ dQKV, int ldg, int nseq, int Tp, int first) {
    const int rows = Tp - first;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // one 16-byte piece of the 768 gradient columns of a row
    if (idx >= (long)nseq * rows * 96) return;
    const int pc = (int)(idx % 96);
    const long rr = idx / 96;
    const int seq = (int)(rr / rows), r = (int)(rr - (long)seq * rows);
    *(uint4*)(dQKV + ((size_t)seq * Tp + first + r) * ldg + pc * 8) = make_uint4(0, 0, 0, 0);
}

}  // namespace

bool eend_attn_bwd_fused_ok(const AttnBwdParams& p, bool ret) {
    if (p.H <= 0 || p.nseq <= 0 || (p.ldg & 7) || (p.ldo & 7)) return false;
    if (ret) return p.L > 0 && p.L <= FW && p.nc > 0 && (long)p.nc * p.L <= p.Tp && p.nseq * p.nc <= 65535 && (p.L & 3) == 0;
    return p.Tp <= FW && p.nseq <= 65535;
}

int eend_launch_attn_bwd_fused(const AttnBwdParams& p, bool ret, hipStream_t stream) {
    if (!eend_attn_bwd_fused_ok(p, ret)) return EEND_EINVAL;
    const dim3 grid(p.H, ret ? p.nseq * p.nc : p.nseq);
#define FUSED_LAUNCH(R, D)                                                                                              \
    do {                                                                                                                \
        static EendOncePerDevice attr_once;                                                                             \
        if (!eend_set_dynamic_lds(attr_once, (const void*)attn_bwd_fused_kernel<R, D>, FUSED_SMEM)) return EEND_ELAUNCH; \
        hipLaunchKernelGGL((attn_bwd_fused_kernel<R, D>), grid, dim3(512), FUSED_SMEM, stream, p);                      \
    } while (0)
    if (ret) {
        FUSED_LAUNCH(true, false);
        const int first = p.nc * p.L;
        if (first < p.Tp) {
            const long n = (long)p.nseq * (p.Tp - first) * 96;
            hipLaunchKernelGGL(zero_tail_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (__bf16*)p.dQKV, p.ldg, p.nseq, p.Tp, first);
        }
    } else if (p.drop.thresh24) {
        FUSED_LAUNCH(false, true);
    } else {
        FUSED_LAUNCH(false, false);
    }
#undef FUSED_LAUNCH
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
