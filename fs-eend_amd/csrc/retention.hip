// LS-EEND multi-scale retention with decay == 1 (LS-EEND/nnet/modules/retention.py), batch
// ("chunk-recurrent", :146-194) form, fused with the per-head LayerNorm (:222) and the
// swish gate (:224).
//
// With decay = log(1) the reference's chunk-wise tables collapse to (retention.py:36-46)
//   mask[i,j] = 1/sqrt(i+1) (j <= i), inner_decay[i] = sqrt(L)/sqrt(i+1), cross_decay = 1,
// and its output for frame t (chunk c = t / L, i = t % L) is algebraically
//   out_t = ( sum_{cL <= j <= t} (q_t.k_j) v_j  +  q_t . S_c ) / ( sqrt(i+1) * all_t )
//   S_c   = sum_{j < cL} k_j (x) v_j                                   (unscaled chunk state)
//   all_t = max( max(1, sum_j |q_t.k_j| / sqrt(i+1)),  max(1, max_d sum_k |S_c[k][d]| / sqrt(L)) )
// The positive scalar only matters through the eps = 1e-6 of the following per-head
// LayerNorm, so it is reproduced exactly as the reference defines it.
//
// Two kernels:
//   ret_state_scan : per (sequence, head) walks the chunks once, S += K_c^T V_c on f16 MFMA
//                    (fp32 accumulate), and emits for every chunk the state *before* it as a
//                    hi/lo f16 pair (power-of-two prescaled: 22 significant bits, no overflow)
//                    plus the reference's cross_scale.
//   ret_chunk      : flash-style tile loop (same transposed formulation as attn.hip, no
//                    softmax): S^T = K Q^T, |S| row sums, O^T += V^T S^T, then the cross term
//                    O^T += S_c^T Q^T, the scale, the per-head LayerNorm and the gate.
// All MFMA operands are f16 (bf16's 8 significand bits cost > 1e-3 on the logits here; the
// retention output is not a convex combination, and the eps = 1e-6 LayerNorm amplifies
// relative error -- measured in the precision study).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int QB = 128;
constexpr int KB = 64;
constexpr int TILE = KB * 128;

DEV int swap23(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

// zero the elements of an 8 x f16 fragment whose frame index j0 + e lies outside [lo, hi)
DEV uint4 mask_frames(uint4 v, int j0, int lo, int hi) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ja = j0 + 2 * e, jb = ja + 1;
        unsigned m = 0;
        if (ja >= lo && ja < hi) m |= 0x0000FFFFu;
        if (jb >= lo && jb < hi) m |= 0xFFFF0000u;
        w[e] &= m;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Phase 1 (chunk-parallel): KV_c = K_c^T V_c, one workgroup per (chunk, head, sequence), 4 waves each
// owning a 32x32 tile of the 64x64 result, fp32, written to the KV workspace.  The frame loop is
// unrolled 4x so 8 independent 16-byte loads are in flight per lane.
__global__ __launch_bounds__(256)
void ret_kv_chunk_kernel(const RetParams p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ti = wave >> 1, tj = wave & 1;
    const int c = blockIdx.x, h = blockIdx.y, seq = blockIdx.z;
    const int lq = lane & 31, hi = lane >> 5;
    const size_t sh = (size_t)seq * p.H + h;
    const _Float16* __restrict__ Kt = (const _Float16*)p.Kt + sh * 64 * p.Tp + (size_t)(ti * 32 + lq) * p.Tp;
    const _Float16* __restrict__ Vt = (const _Float16*)p.Vt + sh * 64 * p.Tp + (size_t)(tj * 32 + lq) * p.Tp;
    const int f0 = c * p.L;
    int f1 = f0 + p.L;
    f1 = f1 < p.Tp ? f1 : p.Tp;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int jbeg = f0 & ~15;
    for (int j0 = jbeg; j0 < f1; j0 += 64) {
        uint4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int jj = j0 + u * 16 + hi * 8;
            jj = jj + 8 <= p.Tp ? jj : p.Tp - 8;                       // stay in the row; masked below
            a[u] = *(const uint4*)(Kt + jj);
            b[u] = *(const uint4*)(Vt + jj);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int js = j0 + u * 16;
            if (js >= f1) break;
            uint4 am = a[u];
            if (js < f0 || js + 16 > f1) am = mask_frames(am, js + hi * 8, f0, f1);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, am), __builtin_bit_cast(f16x8, b[u]), acc, 0, 0, 0);
        }
    }
    // C layout: col = hd (tj*32 + lq), rows kd = ti*32 + 8*g + 4*hi + r  ->  KV[kd][hd] fp32
    float* __restrict__ KV = p.kv_ws + (sh * p.nc + c) * 4096;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) KV[(ti * 32 + 8 * g + 4 * hi + r) * 64 + tj * 32 + lq] = acc[g * 4 + r];
}

// Phase 2: per (sequence, head) exclusive prefix sum of the chunk KVs; emits for every chunk the state
// *before* it as a power-of-two-prescaled hi/lo f16 pair ([hd][kd], the A operand of the cross term)
// plus the reference's cross_scale = max(1, max_hd sum_kd |S| / sqrt(L)) (retention.py:176-180).
// 256 threads: thread t owns state elements (kd = t>>2 (+0), hd = (t&3)*16 .. +15).
__global__ __launch_bounds__(256)
void ret_state_scan_kernel(const RetParams p) {
    __shared__ float colsum[4][64];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, seq = blockIdx.y;
    const size_t sh = (size_t)seq * p.H + h;
    const int kd = tid >> 2, hd0 = (tid & 3) * 16;
    const float inv_sqrtL = 1.0f / __builtin_sqrtf((float)p.L);
    float st[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f;
    if (p.state_in) {                                    // state carried in from the previous call (long-form, chunk at a time)
        const float* __restrict__ S0 = p.state_in + sh * 4096 + kd * 64 + hd0;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = S0[i];
    }
    _Float16* __restrict__ St = (_Float16*)p.St + sh * p.nc * 2 * 4096;
    for (int c = 0; c < p.nc; ++c) {
        // |S| column sums over kd and the global max
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = __builtin_fmaxf(mx, __builtin_fabsf(st[i]));
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) mx = wave_xor_max(mx, m);
        // lanes of a wave: kd = wave*16 + (lane>>2), hd group = lane&3  -> sum over the 16 kd of the wave
        float cs[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = __builtin_fabsf(st[i]);
            v = wave_xor_add(v, 4); v = wave_xor_add(v, 8); v = wave_xor_add(v, 16); v = wave_xor_add(v, 32);
            cs[i] = v;
        }
        __syncthreads();
        if (lane < 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) colsum[wave][lane * 16 + i] = cs[i];
        }
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        const float M = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
        float colmax = colsum[0][lane] + colsum[1][lane] + colsum[2][lane] + colsum[3][lane];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) colmax = wave_xor_max(colmax, m);
        int e = 0;
        if (M > 0.f) e = ilogbf(M) - 9;                   // M * 2^-e in [512, 1024)
        const float down = ldexpf(1.0f, -e);
        _Float16* hi_m = St + (size_t)c * 2 * 4096;
        _Float16* lo_m = hi_m + 4096;
#pragma unroll
        for (int i = 0; i < 16; ++i) {                    // St[hd][kd]
            const float v = st[i] * down;
            const _Float16 hh = (_Float16)v;
            hi_m[(hd0 + i) * 64 + kd] = hh;
            lo_m[(hd0 + i) * 64 + kd] = (_Float16)(v - (float)hh);
        }
        if (tid == 0) {
            p.cscale[sh * p.nc + c] = __builtin_fmaxf(1.0f, colmax * inv_sqrtL);
            p.sexp[sh * p.nc + c] = ldexpf(1.0f, e);
        }
        if (c == p.nc - 1 && !p.state_out) break;
        const float* __restrict__ KV = p.kv_ws + (sh * p.nc + c) * 4096 + kd * 64 + hd0;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const float4 v = *(const float4*)(KV + i);
            st[i] += v.x; st[i + 1] += v.y; st[i + 2] += v.z; st[i + 3] += v.w;
        }
    }
    if (p.state_out) {                                   // state after the last chunk of this call: [kd][hd] f32
        float* __restrict__ S1 = p.state_out + sh * 4096 + kd * 64 + hd0;
#pragma unroll
        for (int i = 0; i < 16; ++i) S1[i] = st[i];
    }
}

__global__ __launch_bounds__(256)
void ret_chunk_kernel(const RetParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // K[2], Vt[2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = blockIdx.x, h = blockIdx.y, seq = blockIdx.z;
    const int q0 = qt * QB;
    const int qw0 = q0 + wave * 32;
    const int lq = lane & 31, hi = lane >> 5;
    const int q = qw0 + lq;
    const int qc = q < p.Tp ? q : p.Tp - 1;
    const int L = p.L;
    const int c_q = qc / L;                     // this lane's chunk
    const int cs_q = c_q * L;                   // first frame of its chunk
    const int i_loc = qc - cs_q;

    const size_t sh = (size_t)seq * p.H + h;
    const _Float16* __restrict__ Qg = (const _Float16*)p.Q + sh * p.Tp * 64;
    const _Float16* __restrict__ Kg = (const _Float16*)p.K + sh * p.Tp * 64;
    const _Float16* __restrict__ Vg = (const _Float16*)p.Vt + sh * 64 * p.Tp;

    const int blk_first_key = (q0 / L) * L;
    int blk_last_key = q0 + QB - 1;
    blk_last_key = blk_last_key < p.Tp - 1 ? blk_last_key : p.Tp - 1;
    const int jt0 = blk_first_key / KB, jt1 = blk_last_key / KB;      // inclusive tile range
    int wq_last = qw0 + 31;
    wq_last = wq_last < p.Tp - 1 ? wq_last : p.Tp - 1;
    const int wq_first = qw0 < p.Tp - 1 ? qw0 : p.Tp - 1;
    const int w_first_key = (wq_first / L) * L;                         // lowest chunk start among the wave's rows
    const int w_hi_start = (wq_last / L) * L;                           // highest chunk start among them

    f16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(Qg + (size_t)qc * 64 + ks * 16 + hi * 8);

    uint4 kr0, kr1, vr0, vr1;
    const int c0row = tid >> 3, c0ch = tid & 7, c1row = (tid + 256) >> 3;
#define RET_GLOAD(j)                                                                  \
    do {                                                                              \
        kr0 = *(const uint4*)(Kg + (size_t)((j) * KB + c0row) * 64 + c0ch * 8);       \
        kr1 = *(const uint4*)(Kg + (size_t)((j) * KB + c1row) * 64 + c0ch * 8);       \
        vr0 = *(const uint4*)(Vg + (size_t)c0row * p.Tp + (j) * KB + c0ch * 8);       \
        vr1 = *(const uint4*)(Vg + (size_t)c1row * p.Tp + (j) * KB + c0ch * 8);       \
    } while (0)
#define RET_LSTORE(buf)                                                               \
    do {                                                                              \
        *(uint4*)(smem + (buf) * TILE + swz128(c0row, c0ch)) = kr0;                   \
        *(uint4*)(smem + (buf) * TILE + swz128(c1row, c0ch)) = kr1;                   \
        *(uint4*)(smem + (2 + (buf)) * TILE + swz128(c0row, c0ch)) = vr0;             \
        *(uint4*)(smem + (2 + (buf)) * TILE + swz128(c1row, c0ch)) = vr1;             \
    } while (0)

    f32x16 oT[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oT[0][i] = 0.f; oT[1][i] = 0.f; }
    float absum = 0.f;

    RET_GLOAD(jt0);
    RET_LSTORE(0);
    __syncthreads();

    const int krow = swap23(lq);
    for (int j = jt0; j <= jt1; ++j) {
        const int buf = (j - jt0) & 1;
        if (j < jt1) RET_GLOAD(j + 1);
        const int key0 = j * KB;
        if (key0 <= wq_last && key0 + KB - 1 >= w_first_key) {          // wave-uniform
            const char* kb_ = smem + buf * TILE;
            const char* vb_ = smem + (2 + buf) * TILE;
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
            // reg i of s[kb] in lane (q, hi) <-> key = key0 + kb*32 + (i&7) + 8*hi + 16*(i>>3)
            const bool edge = (key0 + KB - 1 > wq_first) || (key0 < w_hi_start);
            if (edge) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = key0 + kb * 32 + (i & 7) + 8 * hi + 16 * (i >> 3);
                        if (key > qc || key < cs_q) s[kb][i] = 0.f;
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
        if (j < jt1) RET_LSTORE(buf ^ 1);
        __syncthreads();
    }

    // ---- cross-chunk term: O^T += S_c^T Q^T for every chunk c the wave's rows belong to
    const int c_lo = wq_first / L, c_hi = wq_last / L;
    const _Float16* __restrict__ Sg = (const _Float16*)p.St + sh * p.nc * 2 * 4096;
    for (int c = c_lo; c <= c_hi; ++c) {
        if (c == 0 && !p.state_in) continue;                  // state before the first chunk is zero unless carried in
        f32x16 x[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { x[0][i] = 0.f; x[1][i] = 0.f; }
        const _Float16* hi_m = Sg + (size_t)c * 2 * 4096;
        const _Float16* lo_m = hi_m + 4096;
        const bool mine = (c_q == c);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 qm = qf[ks];
            if (!mine) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) qm[jj] = (_Float16)0.f;
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const f16x8 sa = *(const f16x8*)(hi_m + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                const f16x8 sb = *(const f16x8*)(lo_m + (db * 32 + lq) * 64 + ks * 16 + hi * 8);
                x[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa, qm, x[db], 0, 0, 0);
                x[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sb, qm, x[db], 0, 0, 0);
            }
        }
        const float up = p.sexp[sh * p.nc + c];
#pragma unroll
        for (int i = 0; i < 16; ++i) { oT[0][i] = __builtin_fmaf(x[0][i], up, oT[0][i]); oT[1][i] = __builtin_fmaf(x[1][i], up, oT[1][i]); }
    }

    // ---- scale, per-head LayerNorm (eps 1e-6, no affine), swish gate, f16 store
    const float ab = absum + __shfl_xor(absum, 32, 64);
    const float rsq = 1.0f / __builtin_sqrtf((float)(i_loc + 1));
    const float inner_scale = __builtin_fmaxf(1.0f, ab * rsq);
    const float all = __builtin_fmaxf(inner_scale, p.cscale[sh * p.nc + c_q]);
    const float f = rsq / all;
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
    if (q < p.Tp) {
        const size_t row = (size_t)seq * p.Tp + q;
        const _Float16* __restrict__ Gg = (const _Float16*)p.G + row * p.ldg + h * 64;
        _Float16* __restrict__ Og = (_Float16*)p.O + row * p.ldo + h * 64;
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
                    const float sw = gv / (1.0f + __expf(-gv));
                    o[r] = to_f16_sat(sw * (oT[db][g * 4 + r] - mean) * rstd);
                }
                *(f16x4*)(Og + d) = o;
            }
    }
}

}  // namespace

int eend_launch_ret_state_scan(const RetParams& p, hipStream_t stream) {
    if (p.nseq <= 0 || p.nseq > 65535 || p.H <= 0 || p.Tp <= 0 || (p.Tp % 64) != 0 || p.L <= 0 || p.nc < 1 || p.nc > (p.Tp + p.L - 1) / p.L ||
        !p.kv_ws)
        return EEND_EINVAL;
    const int nkv = p.state_out ? p.nc : p.nc - 1;       // the last chunk's KV only matters when the state is carried out
    if (nkv > 0) {
        hipLaunchKernelGGL(ret_kv_chunk_kernel, dim3(nkv, p.H, p.nseq), dim3(256), 0, stream, p);
        if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    }
    hipLaunchKernelGGL(ret_state_scan_kernel, dim3(p.H, p.nseq), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// the prefix scan alone (ret_stream.hip's pass 1 has filled kv_ws)
int eend_launch_ret_state_scan_only(const RetParams& p, hipStream_t stream) {
    if (p.nseq <= 0 || p.nseq > 65535 || p.H <= 0 || p.Tp <= 0 || (p.Tp % 64) != 0 || p.L <= 0 || p.nc < 1 || p.nc > (p.Tp + p.L - 1) / p.L ||
        !p.kv_ws || !p.St || !p.cscale || !p.sexp)
        return EEND_EINVAL;
    hipLaunchKernelGGL(ret_state_scan_kernel, dim3(p.H, p.nseq), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_ret_chunk(const RetParams& p, hipStream_t stream) {
    if (p.nseq <= 0 || p.nseq > 65535 || p.H <= 0 || p.Tp <= 0 || (p.Tp % 64) != 0 || p.L <= 0 || p.nc < 1 || p.nc > (p.Tp + p.L - 1) / p.L ||
        (p.ldo & 3) || (p.ldg & 3))
        return EEND_EINVAL;
    hipLaunchKernelGGL(ret_chunk_kernel, dim3((p.Tp + QB - 1) / QB, p.H, p.nseq), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
