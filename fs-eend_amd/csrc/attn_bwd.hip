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

constexpr int KB = 64;
constexpr int TILE = KB * 128;       // one [64][64] bf16 tile

DEV int swap23(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

__global__ __launch_bounds__(256)
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

    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *(const bf16x8*)(Qg + (size_t)qc * 64 + ks * 16 + hi * 8);
        dof[ks] = *(const bf16x8*)(dOg + (size_t)qc * p.ldo + ks * 16 + hi * 8);
    }
    const float L2 = p.Lse[sh * p.Tp + qc], Dq = p.Dh[sh * p.Tp + qc];

    uint4 kr0, kr1, vr0, vr1, tr0, tr1;
    const int c0row = tid >> 3, c0ch = tid & 7, c1row = c0row + 32;
#define DQ_GLOAD(j)                                                                   \
    do {                                                                              \
        kr0 = *(const uint4*)(Kg + (size_t)((j) * KB + c0row) * 64 + c0ch * 8);       \
        kr1 = *(const uint4*)(Kg + (size_t)((j) * KB + c1row) * 64 + c0ch * 8);       \
        vr0 = *(const uint4*)(Vg + (size_t)((j) * KB + c0row) * 64 + c0ch * 8);       \
        vr1 = *(const uint4*)(Vg + (size_t)((j) * KB + c1row) * 64 + c0ch * 8);       \
        tr0 = *(const uint4*)(Ktg + (size_t)c0row * p.Tp + (j) * KB + c0ch * 8);      \
        tr1 = *(const uint4*)(Ktg + (size_t)c1row * p.Tp + (j) * KB + c0ch * 8);      \
    } while (0)
#define DQ_LSTORE(buf)                                                                \
    do {                                                                              \
        *(uint4*)(smem + (buf) * TILE + swz128(c0row, c0ch)) = kr0;                   \
        *(uint4*)(smem + (buf) * TILE + swz128(c1row, c0ch)) = kr1;                   \
        *(uint4*)(smem + (2 + (buf)) * TILE + swz128(c0row, c0ch)) = vr0;             \
        *(uint4*)(smem + (2 + (buf)) * TILE + swz128(c1row, c0ch)) = vr1;             \
        *(uint4*)(smem + (4 + (buf)) * TILE + swz128(c0row, c0ch)) = tr0;             \
        *(uint4*)(smem + (4 + (buf)) * TILE + swz128(c1row, c0ch)) = tr1;             \
    } while (0)

    f32x16 dqT[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dqT[0][i] = 0.f; dqT[1][i] = 0.f; }

    if (ntiles > 0) {
        DQ_GLOAD(0);
        DQ_LSTORE(0);
    }
    __syncthreads();
    const int krow = swap23(lq);
    for (int j = 0; j < ntiles; ++j) {
        const int buf = j & 1;
        if (j + 1 < ntiles) DQ_GLOAD(j + 1);
        const int key0 = j * KB;
        if (key0 <= qw0 + 31 + p.mask_delay) {
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
                    const bf16x8 kf = *(const bf16x8*)(kb_ + swz128(kb * 32 + krow, ks * 2 + hi));
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
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
                    const float pv = key <= lim ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], p.scale_log2, -L2)) : 0.f;
                    // forward: O = (P o keep * scale) V  ->  dP = (dO V^T) o keep * scale
                    const float dpe = drop_apply(p.drop, dp[kb][i], (unsigned)(sh * p.Tp + qc), (unsigned)key);
                    s[kb][i] = pv * (dpe - Dq);
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
        if (j + 1 < ntiles) DQ_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef DQ_GLOAD
#undef DQ_LSTORE
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

__global__ __launch_bounds__(256)
void attn_bwd_dkv_kernel(const AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];     // per stage: Q[32][64], dO[32][64], Q^T[64][32], dO^T[64][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, seq = blockIdx.z;
    const int k0 = blockIdx.x * 128, kw0 = k0 + wave * 32;
    const int lq = lane & 31, hi = lane >> 5;
    const size_t sh = (size_t)seq * p.H + h;
    const __bf16* __restrict__ Qg = (const __bf16*)p.Q + sh * p.Tp * 64;
    const __bf16* __restrict__ Qtg = (const __bf16*)p.Qt + sh * 64 * p.Tp;
    const __bf16* __restrict__ Kg = (const __bf16*)p.K + sh * p.Tp * 64;
    const __bf16* __restrict__ Vg = (const __bf16*)p.V + sh * p.Tp * 64;
    const __bf16* __restrict__ dOg = (const __bf16*)p.dO + (size_t)seq * p.Tp * p.ldo + h * 64;
    const __bf16* __restrict__ dOtg = (const __bf16*)p.dOt + sh * 64 * p.Tp;
    const float* __restrict__ Lg = p.Lse + sh * p.Tp;
    const float* __restrict__ Dg = p.Dh + sh * p.Tp;

    const bool active = kw0 < p.Tp;
    const int key = kw0 + lq;                          // this lane's key (B-operand column)
    const int keyc = key < p.Tp ? key : p.Tp - 1;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *(const bf16x8*)(Kg + (size_t)keyc * 64 + ks * 16 + hi * 8);
        vf[ks] = *(const bf16x8*)(Vg + (size_t)keyc * 64 + ks * 16 + hi * 8);
    }

    int qlim = p.q_len < p.Tp ? p.q_len : p.Tp;
    const int nqb = (qlim + 31) / 32;
    int qstart = k0 - p.mask_delay;
    if (qstart < 0) qstart = 0;
    const int qb0 = qstart / 32;

    // staging: 4 tiles x 256 chunks of 16 B, one chunk of each per thread
    const int r8 = tid >> 3, c8 = tid & 7;             // [32][64] tiles
    const int r4 = tid >> 2, c4 = tid & 3;             // [64][32] tiles
    uint4 g0, g1, g2, g3;
#define DKV_GLOAD(qb)                                                                    \
    do {                                                                                 \
        const int qr = (qb) * 32 + r8 < p.Tp ? (qb) * 32 + r8 : p.Tp - 1;                \
        g0 = *(const uint4*)(Qg + (size_t)qr * 64 + c8 * 8);                             \
        g1 = *(const uint4*)(dOg + (size_t)qr * p.ldo + c8 * 8);                         \
        g2 = *(const uint4*)(Qtg + (size_t)r4 * p.Tp + (qb) * 32 + c4 * 8);              \
        g3 = *(const uint4*)(dOtg + (size_t)r4 * p.Tp + (qb) * 32 + c4 * 8);             \
    } while (0)
#define DKV_LSTORE(buf)                                                                  \
    do {                                                                                 \
        char* b_ = smem + (buf) * 16384;                                                 \
        *(uint4*)(b_ + swz128(r8, c8)) = g0;                                             \
        *(uint4*)(b_ + 4096 + swz128(r8, c8)) = g1;                                      \
        *(uint4*)(b_ + 8192 + swz64(r4, c4)) = g2;                                       \
        *(uint4*)(b_ + 12288 + swz64(r4, c4)) = g3;                                      \
    } while (0)

    f32x16 dkT[2], dvT[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dkT[0][i] = 0.f; dkT[1][i] = 0.f; dvT[0][i] = 0.f; dvT[1][i] = 0.f; }

    if (qb0 < nqb) {
        DKV_GLOAD(qb0);
        DKV_LSTORE(0);
    }
    __syncthreads();
    const int qrow = swap23(lq);
    for (int qb = qb0; qb < nqb; ++qb) {
        const int buf = (qb - qb0) & 1;
        if (qb + 1 < nqb) DKV_GLOAD(qb + 1);
        const int qw0 = qb * 32;
        if (active && kw0 <= qw0 + 31 + p.mask_delay) {
            const char* b_ = smem + buf * 16384;
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
            float l2v[16], dv[16];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int qb_ = qw0 + 16 * g + 8 * hi;         // multiple of 8 and < Tp (Tp % 64 == 0)
                const float4 a0 = *(const float4*)(Lg + qb_), a1 = *(const float4*)(Lg + qb_ + 4);
                const float4 d0 = *(const float4*)(Dg + qb_), d1 = *(const float4*)(Dg + qb_ + 4);
                l2v[g * 8 + 0] = a0.x; l2v[g * 8 + 1] = a0.y; l2v[g * 8 + 2] = a0.z; l2v[g * 8 + 3] = a0.w;
                l2v[g * 8 + 4] = a1.x; l2v[g * 8 + 5] = a1.y; l2v[g * 8 + 6] = a1.z; l2v[g * 8 + 7] = a1.w;
                dv[g * 8 + 0] = d0.x; dv[g * 8 + 1] = d0.y; dv[g * 8 + 2] = d0.z; dv[g * 8 + 3] = d0.w;
                dv[g * 8 + 4] = d1.x; dv[g * 8 + 5] = d1.y; dv[g * 8 + 6] = d1.z; dv[g * 8 + 7] = d1.w;
            }
            bf16x8 pf[2], sf[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = qw0 + (r & 7) + 8 * hi + 16 * (r >> 3);
                const bool ok = key <= qi + p.mask_delay && key < p.kv_len && qi < qlim;
                const float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], p.scale_log2, -l2v[r])) : 0.f;
                float kf = 1.0f;                               // the forward's dropout factor of this (query, key) pair
                if (p.drop.thresh24) kf = drop_keep(p.drop, (unsigned)(sh * p.Tp + qi), (unsigned)key) ? p.drop.scale : 0.f;
                pf[r >> 3][r & 7] = (__bf16)(pv * kf);
                sf[r >> 3][r & 7] = (__bf16)(pv * (dp[r] * kf - dv[r]));
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
        if (qb + 1 < nqb) DKV_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef DKV_GLOAD
#undef DKV_LSTORE
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
    if (!p.Q || !p.Qt || !p.K || !p.Kt || !p.V || !p.dO || !p.dOt || !p.Lse || !p.Dh || !p.dQKV) return EEND_EINVAL;
    if (p.nseq <= 0 || p.nseq > 65535 || p.H <= 0 || p.Tp <= 0 || (p.Tp % 64) || (p.ldo & 7) || (p.ldg & 3) || p.kv_len <= 0 ||
        p.kv_len > p.Tp || p.q_len <= 0)
        return EEND_EINVAL;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((p.Tp + 127) / 128, p.H, p.nseq), dim3(256), 0, stream, p);
    if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((p.Tp + 127) / 128, p.H, p.nseq), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_heads_transpose(const void* in, int ld, void* out, int nseq, int H, int Tp, hipStream_t stream) {
    if (!in || !out || nseq <= 0 || H <= 0 || Tp <= 0 || (Tp & 7) || (ld & 7)) return EEND_EINVAL;
    const long total = (long)nseq * H * (Tp >> 3) * 8;
    hipLaunchKernelGGL(heads_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const unsigned short*)in, ld,
                       (unsigned short*)out, nseq, H, Tp);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
