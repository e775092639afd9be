// Gradient of the embedding-consistency loss (FS model :46-57; forward value: embloss.hip) w.r.t. the unit
// embeddings e (B, T, 256):
//     loss = c * sum_{b,i,j} (A_ij - L_ij)^2,   A_ij = <e_i, e_j> / (|e_i||e_j| + 1e-6),   L_ij = label cosine map
//     dL/de_i = 4 c * sum_j (A_ij - L_ij) / (|e_i||e_j| + 1e-6) * e_j     (+ a component along e_i)
// c = 1/(B T T).  The component along e_i (from d|e_i|/de_i and from the diagonal) is dropped: e = y/|y| comes out
// of an L2 normalisation whose backward (l2norm_bwd_kernel) projects every gradient onto the complement of e_i, so
// it cannot reach any parameter.  |e_i| = 1 to fp32 rounding, so the denominators are taken as 1 + 1e-6.
//
// Structure: a flash-attention-shaped double product without a softmax.  A block owns 64 rows i of one utterance
// and walks the 64-row tiles j: S^T = E_j E_i^T (f16 MFMA, K = 256), E' = (S/(1+eps) - L)/(1+eps) on the VALU (labels,
// C <= 16: one more small MFMA), G^T += E_j^T E'^T (f16 MFMA).  E_j^T is produced while staging (8x8 register transposes), and
// the rows of E_j are fed to the first product in an order that makes the 8 j's a lane holds afterwards contiguous,
// so E' is the B operand of the second product straight from registers.  The (T, T) maps never exist in memory.
// Output: de[b, i, :] += 4 c * G (f32, each element owned by one lane -- no atomics).
#include "train_common.h"
#include "kernels.h"

namespace {

constexpr int TS = 64, CMAX = 16, DM = 256;

// [64][256] f16 tile, 512-byte rows: 16-byte chunk index (0..31) XOR (row & 15) in its low 4 bits -> the 16 rows of a
// fragment read land in 16 distinct slots of the 256-byte bank row
DEV int swzRow(int row, int c) { return row * 512 + ((c ^ (row & 15)) << 4); }

__global__ __launch_bounds__(256)
void emb_consistency_bwd_kernel(const _Float16* __restrict__ emb16, const float* __restrict__ tgt, const int* __restrict__ lens,
                                float coef, float* __restrict__ de, int T, int Tp, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ei = smem;                       // [64][256] f16
    char* Ej = smem + 32768;
    char* EjT = smem + 65536;              // [256 d][64 j] f16 (128-byte rows, swzT)
    _Float16* Yi = (_Float16*)(smem + 98304);   // [64][16] f16 (labels are 0 / 1: exact)
    _Float16* Yj = Yi + TS * CMAX;
    float* nyi = (float*)(Yj + TS * CMAX);      // [64]
    float* nyj = nyi + TS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, i0 = blockIdx.x * TS;
    const _Float16* __restrict__ E = emb16 + (size_t)b * Tp * DM;
    const float* __restrict__ Y = tgt + (size_t)b * T * C;
    const int elen = lens ? (lens[b] < T ? lens[b] : T) : T;
    const int dch = tid & 31, rg = tid >> 5;                     // staging unit: 8 rows rg*8.., 16-byte chunk dch
    const int frow = lane & 15, g = lane >> 4;
    const float inv1 = 1.0f / (1.0f + 1e-6f);

    auto stage_rows = [&](char* dst, int r0) __attribute__((always_inline)) {   // row-major copy only (E_i)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = rg * 8 + r, t = r0 + row;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (t < elen) v = *(const u32x4*)(E + (size_t)t * DM + dch * 8);
            *(u32x4*)(dst + swzRow(row, dch)) = v;
        }
    };
    auto stage_labels = [&](_Float16* Yd, float* nyd, int r0) __attribute__((always_inline)) {
        if (tid < TS) {
            const int t = r0 + tid;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                const float v = (t < T && c < C) ? Y[(size_t)t * C + c] : 0.f;
                Yd[tid * CMAX + c] = (_Float16)v;
                s = __builtin_fmaf(v, v, s);
            }
            nyd[tid] = __builtin_sqrtf(s);
        }
    };

    stage_rows(Ei, i0);
    stage_labels(Yi, nyi, i0);
    __syncthreads();
    const int il = wave * 16 + frow;                              // this lane's row i (B-operand column)
    const f16x4 yi = *(const f16x4*)(Yi + il * CMAX + g * 4);   // B operand of the label product (column i, labels 4g .. 4g+3)
    const float nyi_l = nyi[il];

    f32x4 acc[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntj = (T + TS - 1) / TS;
    for (int jt = 0; jt < ntj; ++jt) {
        const int j0 = jt * TS;
        __syncthreads();                                          // previous tile's readers are done
        {   // E_j rows rg*8 .. +7, chunk dch: row-major copy + transposed copy
            u32x4 in[8], out[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int row = rg * 8 + r, t = j0 + row;
                in[r] = u32x4{0u, 0u, 0u, 0u};
                if (t < elen) in[r] = *(const u32x4*)(E + (size_t)t * DM + dch * 8);
                *(u32x4*)(Ej + swzRow(row, dch)) = in[r];
            }
            transpose8x8_b16(in, out);
#pragma unroll
            for (int e = 0; e < 8; ++e) *(u32x4*)(EjT + swzT(dch * 8 + e, rg)) = out[e];
        }
        stage_labels(Yj, nyj, j0);
        __syncthreads();

#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // S^T for the j's 32kk .. 32kk+31: hardware row rho of 16-row block jb <-> j = 32kk + 8*(rho>>2) + 4*jb + (rho&3)
            f32x4 s[2];
            s[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            s[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const f16x8 bi = *(const f16x8*)(Ei + swzRow(il, ks * 4 + g));
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const int jr = 32 * kk + 8 * (frow >> 2) + 4 * jb + (frow & 3);
                    const f16x8 aj = *(const f16x8*)(Ej + swzRow(jr, ks * 4 + g));
                    s[jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aj, bi, s[jb], 0, 0, 0);
                }
            }
            // the label map of the same (j, i) pairs on the MFMA (16x16x16 f16, exact: 0 / 1 labels), rows in the same order
            f32x4 ls[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int jr = 32 * kk + 8 * (frow >> 2) + 4 * jb + (frow & 3);
                const f16x4 yj = *(const f16x4*)(Yj + jr * CMAX + g * 4);
                ls[jb] = __builtin_amdgcn_mfma_f32_16x16x16f16(yj, yi, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            // lane (i = il, g): reg r of s[jb] <-> j = 32kk + 8g + 4jb + r
            f16x8 ef;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jl = 32 * kk + 8 * g + 4 * jb + r;
                    const float lm = ls[jb][r] / (nyi_l * nyj[jl] + 1e-6f);
                    float e = (s[jb][r] * inv1 - lm) * inv1;
                    if (j0 + jl >= T || i0 + il >= T) e = 0.f;
                    ef[jb * 4 + r] = (_Float16)e;
                }
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                const f16x8 at = *(const f16x8*)(EjT + swzT(d * 16 + frow, 4 * kk + g));
                acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(at, ef, acc[d], 0, 0, 0);
            }
        }
    }
    // acc[d][r]: feature = d*16 + 4g + r, row i = i0 + il
    if (i0 + il < T) {
        float* __restrict__ o = de + ((size_t)b * Tp + i0 + il) * DM;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            float4 v = *(const float4*)(o + d * 16 + 4 * g);
            v.x += coef * acc[d][0]; v.y += coef * acc[d][1]; v.z += coef * acc[d][2]; v.w += coef * acc[d][3];
            *(float4*)(o + d * 16 + 4 * g) = v;
        }
    }
}

}  // namespace

int eend_launch_emb_consistency_bwd(const void* emb16, const float* tgt, const int* lens, float inv_count, float* de,
                                    int B, int T, int Tp, int D, int C, hipStream_t stream) {
    if (!emb16 || !tgt || !de || B <= 0 || B > 65535 || T <= 0 || Tp < T || D != DM || C < 1 || C > CMAX) return EEND_EINVAL;
    const int smem = 98304 + 2 * TS * CMAX * 2 + 2 * TS * 4;
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)emb_consistency_bwd_kernel, smem)) return EEND_ELAUNCH;
    const float inv = inv_count > 0.f ? inv_count : 1.0f / ((float)B * (float)T * (float)T);
    hipLaunchKernelGGL(emb_consistency_bwd_kernel, dim3((T + TS - 1) / TS, B), dim3(256), smem, stream, (const _Float16*)emb16, tgt,
                       lens, 4.0f * inv, de, T, Tp, C);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
