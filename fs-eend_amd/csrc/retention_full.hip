// Chunk-resident retention core (LS-EEND, recurrent_chunk_size <= 512 -- 500 in every shipped
// config): ONE workgroup per (sequence, head, chunk) keeps the chunk's K ([L][64]) and V^T
// ([64][L]) in LDS, loaded once, and its 8 waves run the masked linear-attention loop with no
// further barriers; the 32-row query blocks of the chunk are dealt to the waves in causally
// balanced pairs (w, nq-1-w).  Because a workgroup is exactly one chunk, the cross-chunk term uses a
// single state S_c for all rows and the lower index bound of the mask disappears.
//
// Arithmetic, scaling, per-head LayerNorm and swish gate are those of ret_chunk_kernel
// (retention.hip; reference LS-EEND/nnet/modules/retention.py:146-194,222-224); the chunk states
// and cross_scale come from ret_state_scan_kernel.  Chunk starts are multiples of L = 500, i.e. not
// 16-byte aligned in the V^T rows: V^T is fetched with 8-byte loads.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int LMAX = 512;
constexpr int KB = 64;
constexpr int TILE = KB * 128;
constexpr int NW = 8;
constexpr int OSTG = 32 * 128;

DEV int swap23(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512)
void ret_chunk_full_kernel(const RetParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x, h = blockIdx.y, seq = blockIdx.z;
    const int lq = lane & 31, hi = lane >> 5;
    const int f0 = c * p.L;
    int f1 = f0 + p.L;
    f1 = f1 < p.Tp ? f1 : p.Tp;
    const int n = f1 - f0;                              // frames in this chunk (<= 512)
    const int ntl = (n + KB - 1) / KB;
    char* Ks = smem;
    char* Vs = smem + ntl * TILE;
    char* Os = smem + 2 * ntl * TILE;

    const size_t sh = (size_t)seq * p.H + h;
    const _Float16* __restrict__ Qg = (const _Float16*)p.Q + (sh * p.Tp + f0) * 64;
    const _Float16* __restrict__ Kg = (const _Float16*)p.K + (sh * p.Tp + f0) * 64;
    const _Float16* __restrict__ Vg = (const _Float16*)p.Vt + sh * 64 * p.Tp;

    // ---- chunk K (16-B loads) and V^T (8-B loads: f0 is only 8-byte aligned), clamped to valid frames.
    // Fixed trip counts (<= 512 frames): every load of the phase is in flight before the first LDS store.
    {
        const int nrow = ntl * KB;                      // padded local frames (multiple of 64)
        u32x4 kr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cidx = tid + i * 512;
            const int row = cidx >> 3, ch = cidx & 7;
            if (row < nrow) {
                const int r = row < n ? row : n - 1;    // rows >= n are never unmasked; keep them finite
                kr[i] = *(const u32x4*)(Kg + (size_t)r * 64 + ch * 8);
            }
        }
        u32x2 vr[16];
        const int sh8 = __builtin_ctz(nrow >> 2) ;      // nrow/4 pieces per V^T row; nrow is 64 * {1..8}
        const int nc8 = nrow >> 2;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cidx = tid + i * 512;
            const int d = cidx / nc8, pc = cidx - d * nc8;
            if (d < 64) {
                int fr = f0 + pc * 4;
                fr = fr + 4 <= p.Tp ? fr : p.Tp - 4;    // stay inside the row (clamped frames are masked)
                vr[i] = *(const u32x2*)(Vg + (size_t)d * p.Tp + fr);
            }
        }
        (void)sh8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cidx = tid + i * 512;
            const int row = cidx >> 3, ch = cidx & 7;
            if (row < nrow) *(u32x4*)(Ks + (row >> 6) * TILE + swz128(row & 63, ch)) = kr[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cidx = tid + i * 512;
            const int d = cidx / nc8, pc = cidx - d * nc8;
            if (d < 64) {
                const int key = pc * 4;                 // local key of the first of the 4 frames
                *(u32x2*)(Vs + (key >> 6) * TILE + swz128(d, (key & 63) >> 3) + ((key >> 2) & 1) * 8) = vr[i];
            }
        }
    }
    __syncthreads();

    const int nq = (n + 31) / 32;
    const int krow = swap23(lq);
    char* Ow = Os + wave * OSTG;
    const _Float16* __restrict__ Sg = (const _Float16*)p.St + (sh * p.nc + c) * 2 * 4096;
    const float cscale = p.cscale[sh * p.nc + c];
    const float sexp = p.sexp[sh * p.nc + c];

    for (int pass = 0; pass < 2; ++pass) {
        const int qb = pass == 0 ? wave : nq - 1 - wave;
        if (qb < 0 || qb >= nq) continue;
        if (pass == 1 && qb <= wave) continue;
        if (pass == 0 && wave > nq - 1 - wave) continue;
        const int qw0 = qb * 32;
        const int q = qw0 + lq;                         // local frame
        const int qc = q < n ? q : n - 1;

        f16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(Qg + (size_t)qc * 64 + ks * 16 + hi * 8);

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
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kb], 0, 0, 0);
                }
            }
            if (key0 + KB - 1 > qw0) {                  // tile straddles the diagonal for some row of the wave
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
                    for (int jj = 0; jj < 8; ++jj) pf[jj] = to_f16_sat(s[kb][kk * 8 + jj]);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const f16x8 vf = *(const f16x8*)(vb_ + swz128(db * 32 + lq, kb * 4 + kk * 2 + hi));
                        oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oT[db], 0, 0, 0);
                    }
                }
        }
        // ---- cross-chunk term O^T += S_c^T Q^T (hi/lo f16 state, prescale undone by sexp)
        if (c > 0 || p.state_in) {                      // chunk 0 has a predecessor state only when one is carried in
            f32x16 x[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) { x[0][i] = 0.f; x[1][i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const f16x8 sa = *(const f16x8*)(Sg + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                    const f16x8 sb = *(const f16x8*)(Sg + 4096 + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                    x[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa, qf[ks], x[db], 0, 0, 0);
                    x[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb, qf[ks], x[db], 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 16; ++i) { oT[0][i] = __builtin_fmaf(x[0][i], sexp, oT[0][i]); oT[1][i] = __builtin_fmaf(x[1][i], sexp, oT[1][i]); }
        }
        // ---- scale, per-head LayerNorm, swish gate
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
        const size_t grow = (size_t)seq * p.Tp + f0 + qc;
        const _Float16* __restrict__ Gg = (const _Float16*)p.G + grow * p.ldg + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + g * 8 + hi * 4;
                const f16x4 gg = *(const f16x4*)(Gg + d);
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gv = (float)gg[r];
                    o[r] = to_f16_sat(gv / (1.0f + __expf(-gv)) * (oT[db][g * 4 + r] - mean) * rstd);
                }
                *(f16x4*)(Ow + lq * 128 + (((db * 4 + g) ^ (lq & 7)) << 4) + hi * 8) = o;
            }
        __builtin_amdgcn_wave_barrier();
        _Float16* __restrict__ Og = (_Float16*)p.O + ((size_t)seq * p.Tp + f0 + qw0) * p.ldo + h * 64;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            if (qw0 + row < n) {
                const uint4 v = *(const uint4*)(Ow + row * 128 + ((ch ^ (row & 7)) << 4));
                *(uint4*)(Og + (size_t)row * p.ldo + ch * 8) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (p.Rhat) {
            // training forward: the normalised rows (input of the gate) and, per (row, head), 1/sigma times the detached
            // row scale f -- d out_t / d (q_t . prefix) = f exactly because the reference detaches inner_scale / kv_scale
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = to_f16_sat((oT[db][g * 4 + r] - mean) * rstd);
                    *(f16x4*)(Ow + lq * 128 + (((db * 4 + g) ^ (lq & 7)) << 4) + hi * 8) = o;
                }
            __builtin_amdgcn_wave_barrier();
            _Float16* __restrict__ Rg = (_Float16*)p.Rhat + ((size_t)seq * p.Tp + f0 + qw0) * p.ldo + h * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                if (qw0 + row < n) {
                    const uint4 v = *(const uint4*)(Ow + row * 128 + ((ch ^ (row & 7)) << 4));
                    *(uint4*)(Rg + (size_t)row * p.ldo + ch * 8) = v;
                }
            }
            if (p.Rc && hi == 0 && q < n) p.Rc[((size_t)seq * p.Tp + f0 + q) * p.H + h] = rstd * f;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace

int eend_launch_ret_chunk_full(const RetParams& p, hipStream_t stream) {
    if (p.L <= 0 || p.L > LMAX || p.nseq <= 0 || p.nseq > 65535 || p.Tp <= 0 || (p.Tp % 64) != 0 || (p.ldo & 7) || (p.ldg & 3) ||
        p.nc < 1 || p.nc > (p.Tp + p.L - 1) / p.L || (p.L & 3))
        return EEND_EINVAL;
    const int ntl = ((p.L < p.Tp ? p.L : p.Tp) + KB - 1) / KB;
    const int smem = 2 * ntl * TILE + NW * OSTG;
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)ret_chunk_full_kernel, 2 * (LMAX / KB) * TILE + NW * OSTG)) return EEND_ELAUNCH;
    hipLaunchKernelGGL(ret_chunk_full_kernel, dim3(p.nc, p.H, p.nseq), dim3(512), smem, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
