// Cross-chunk states of the retention backward (LS-EEND training step; forward: retention.hip / retention_full.hip,
// reference LS-EEND/nnet/modules/retention.py:146-194).  With the reference's detached scales folded into
// o~_t = c_t * d out_t, the retention core is the linear attention  out_t = c_t * q_t . sum_{s <= t} k_s (x) v_s, so
//   dq_t = sum_{s<=t} (o~_t . v_s) k_s ,  dk_s = sum_{t>=s} (o~_t . v_s) q_t ,  dv_s = sum_{t>=s} (q_t . k_s) o~_t .
// The intra-chunk part of those sums runs in attn_bwd.hip (RET kernels); the parts that cross chunk boundaries only
// need two 64 x 64 states per (sequence, head, chunk c):
//   Spre_c = sum_{chunks < c} K^T V   [kd][hd]     ->  dq_t += Spre_c o~_t
//   R_c    = sum_{chunks > c} Q^T O~  [kd][hd]     ->  dk_s += R_c v_s ,  dv_s += R_c^T k_s
// ret_bwd_outer_kernel: per-chunk outer products A_c^T B_c on bf16 MFMA (fp32 accumulate), one workgroup per
// (chunk, head, sequence) -- run for (K, V) and for (Q, O~).  ret_bwd_scan_kernel: exclusive prefix / suffix sums
// in fp32, emitted as bf16 hi/lo pairs (16 significand bits) in the operand layouts of the RET kernels.
#include "train_common.h"
#include "kernels.h"

namespace {

// zero the elements of an 8 x 16-bit fragment whose frame index j0 + e lies outside [lo, hi)
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

// out[(seq*H + h)*nc + c][a][b] = sum_{t in chunk c} At[a][t] * Bt[b][t];  At, Bt bf16 [nseq][H][64][Tp].
// 4 waves, each a 32x32 tile of the 64x64 result (as ret_kv_chunk_kernel, bf16 operands).
__global__ __launch_bounds__(256)
void ret_bwd_outer_kernel(const __bf16* __restrict__ At, const __bf16* __restrict__ Bt, float* __restrict__ out, int H, int Tp, int L,
                          int nc) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ti = wave >> 1, tj = wave & 1;
    const int c = blockIdx.x, h = blockIdx.y, seq = blockIdx.z;
    const int lq = lane & 31, hi = lane >> 5;
    const size_t sh = (size_t)seq * H + h;
    const __bf16* __restrict__ Ar = At + sh * 64 * Tp + (size_t)(ti * 32 + lq) * Tp;
    const __bf16* __restrict__ Br = Bt + sh * 64 * Tp + (size_t)(tj * 32 + lq) * Tp;
    const int f0 = c * L;
    int f1 = f0 + L;
    f1 = f1 < Tp ? f1 : Tp;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int jbeg = f0 & ~15;
    for (int j0 = jbeg; j0 < f1; j0 += 64) {
        uint4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int jj = j0 + u * 16 + hi * 8;
            jj = jj + 8 <= Tp ? jj : Tp - 8;                           // stay in the row; masked below
            a[u] = *(const uint4*)(Ar + jj);
            b[u] = *(const uint4*)(Br + jj);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int js = j0 + u * 16;
            if (js >= f1) break;
            uint4 am = a[u];
            if (js < f0 || js + 16 > f1) am = mask_frames(am, js + hi * 8, f0, f1);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, am), __builtin_bit_cast(bf16x8, b[u]), acc, 0, 0, 0);
        }
    }
    // C layout: col = b index (tj*32 + lq), rows a index = ti*32 + 8*g + 4*hi + r
    float* __restrict__ O = out + (sh * nc + c) * 4096;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) O[(ti * 32 + 8 * g + 4 * hi + r) * 64 + tj * 32 + lq] = acc[g * 4 + r];
}

// Round 6: the same per-chunk outer products from ROW-MAJOR operands (token rows of 64 features, any row stride): the token rows of a
// 64-frame stage go to LDS as they are and the MFMA fragments are fetched transposed by ds_read_b64_tr_b16 (a 16-lane group reads a
// [4 tokens][16 features] block, every lane receives the 4 tokens of ITS feature; two reads = the 8-token k-group of
// v_mfma_f32_32x32x16_bf16 -- the idiom of wgrad.hip).  With it the [d][t] copies of Q, K, V (three of the training in-projection's six
// outputs) and the head-transposed copy of o~ need not exist for chunk lengths up to 512 (the one-launch backward, attn_bwd_fused.hip).
//   LDS image per operand: [64 tokens][128 B], the two 64-byte halves of row r swapped when r is odd.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
DEV u32x2 tr_read_b64(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__global__ __launch_bounds__(256)
void ret_bwd_outer_rm_kernel(const __bf16* __restrict__ A, long a_seq, long a_head, int lda, const __bf16* __restrict__ B, long b_seq, long b_head,
                             int ldb, float* __restrict__ out, int H, int Tp, int L, int nc) {
    __shared__ __attribute__((aligned(16))) char sm[2 * 8192];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ti = wave >> 1, tj = wave & 1;
    const int c = blockIdx.x, h = blockIdx.y, seq = blockIdx.z;
    const __bf16* __restrict__ Ab = A + seq * a_seq + h * a_head;      // element offsets; row t at + t * ld
    const __bf16* __restrict__ Bb = B + seq * b_seq + h * b_head;
    const int f0 = c * L;
    int f1 = f0 + L;
    f1 = f1 < Tp ? f1 : Tp;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // fragment address of this lane inside a stage (k-step 0): token (lane>>5)*8 + x2 (+4 for the second read), features of the wave's
    // 32-feature half: (g&1)*16 + (i16&3)*4 .. +3
    const int g = lane >> 4, i16 = lane & 15, x2 = i16 >> 2;
    const int lrow = ((g >> 1) * 8 + x2) * 128 + (g & 1) * 32 + (i16 & 3) * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
    const unsigned addrA = lds0 + lrow + ((ti ^ (x2 & 1)) << 6);
    const unsigned addrB = lds0 + 8192 + lrow + ((tj ^ (x2 & 1)) << 6);
    for (int j0 = f0; j0 < f1; j0 += 64) {
        __syncthreads();                                              // the previous stage's fragments are in registers
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = tid + it * 256, r = q >> 3, c16 = q & 7;    // token row of the stage, 16-byte piece of its 128 bytes
            const int t = j0 + r;
            uint4 va = make_uint4(0, 0, 0, 0), vb = va;
            if (t < f1) {
                va = *(const uint4*)(Ab + (size_t)t * lda + c16 * 8);
                vb = *(const uint4*)(Bb + (size_t)t * ldb + c16 * 8);
            }
            const int dst = r * 128 + ((((c16 >> 2) ^ (r & 1)) << 6) | ((c16 & 3) << 4));
            *(uint4*)(sm + dst) = va;
            *(uint4*)(sm + 8192 + dst) = vb;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (j0 + ks * 16 >= f1) break;
            u32x2 a0 = tr_read_b64(addrA + ks * 16 * 128), a1 = tr_read_b64(addrA + (ks * 16 + 4) * 128);
            u32x2 b0 = tr_read_b64(addrB + ks * 16 * 128), b1 = tr_read_b64(addrB + (ks * 16 + 4) * 128);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
            const bf16x8 af = __builtin_shufflevector(__builtin_bit_cast(bf16x4, a0), __builtin_bit_cast(bf16x4, a1), 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 bf = __builtin_shufflevector(__builtin_bit_cast(bf16x4, b0), __builtin_bit_cast(bf16x4, b1), 0, 1, 2, 3, 4, 5, 6, 7);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
        }
    }
    const size_t sh = (size_t)seq * H + h;
    const int lq = lane & 31, hi = lane >> 5;
    float* __restrict__ O = out + (sh * nc + c) * 4096;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)
#pragma unroll
        for (int r = 0; r < 4; ++r) O[(ti * 32 + 8 * gg + 4 * hi + r) * 64 + tj * 32 + lq] = acc[gg * 4 + r];
}

DEV void put_hilo(__bf16* hi_m, __bf16* lo_m, int idx, float v) {
    const __bf16 hh = (__bf16)v;
    hi_m[idx] = hh;
    lo_m[idx] = (__bf16)(v - (float)hh);
}

// per (sequence, head): St[c] = {Spre hi, lo [kd][hd]; R hi, lo [kd][hd]; R^T hi, lo [hd][kd]}.
// 256 threads: thread t owns kd = t >> 2, hd = (t & 3) * 16 .. +15.
__global__ __launch_bounds__(256)
void ret_bwd_scan_kernel(const float* __restrict__ kv, const float* __restrict__ g, __bf16* __restrict__ St, int H, int nc) {
    const int tid = threadIdx.x;
    const int h = blockIdx.x, seq = blockIdx.y;
    const size_t sh = (size_t)seq * H + h;
    const int kd = tid >> 2, hd0 = (tid & 3) * 16;
    float st[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f;
    for (int c = 0; c < nc; ++c) {
        __bf16* base = St + ((sh * nc + c) * 6) * 4096;
#pragma unroll
        for (int i = 0; i < 16; ++i) put_hilo(base, base + 4096, kd * 64 + hd0 + i, st[i]);
        if (c == nc - 1) break;
        const float* __restrict__ KV = kv + (sh * nc + c) * 4096 + kd * 64 + hd0;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const float4 v = *(const float4*)(KV + i);
            st[i] += v.x; st[i + 1] += v.y; st[i + 2] += v.z; st[i + 3] += v.w;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f;
    for (int c = nc - 1; c >= 0; --c) {
        __bf16* base = St + ((sh * nc + c) * 6) * 4096;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            put_hilo(base + 2 * 4096, base + 3 * 4096, kd * 64 + hd0 + i, st[i]);
            put_hilo(base + 4 * 4096, base + 5 * 4096, (hd0 + i) * 64 + kd, st[i]);
        }
        if (c == 0) break;
        const float* __restrict__ G = g + (sh * nc + c) * 4096 + kd * 64 + hd0;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const float4 v = *(const float4*)(G + i);
            st[i] += v.x; st[i + 1] += v.y; st[i + 2] += v.z; st[i + 3] += v.w;
        }
    }
}

}  // namespace

// The same states from row-major operands: K, V, Q bf16 [nseq][H][Tp][64] (head layout), dO = o~ bf16 rows [nseq*Tp][ldo] (head h in
// columns 64 h ..).
int eend_launch_ret_bwd_states_rm(const void* K, const void* V, const void* Q, const void* dO, int ldo, float* kv_ws, float* g_ws, void* St,
                                  int nseq, int H, int Tp, int L, int nc, hipStream_t stream) {
    if (!K || !V || !Q || !dO || !kv_ws || !g_ws || !St || nseq <= 0 || nseq > 65535 || H <= 0 || Tp <= 0 || (Tp % 64) || L <= 0 ||
        nc <= 0 || (long)nc * L > Tp || ldo < 64 * H || (ldo & 7))
        return EEND_EINVAL;
    if (nc > 1) {
        const long hs = (long)Tp * 64, ss = hs * H;
        hipLaunchKernelGGL(ret_bwd_outer_rm_kernel, dim3(nc, H, nseq), dim3(256), 0, stream, (const __bf16*)K, ss, hs, 64, (const __bf16*)V, ss, hs, 64,
                           kv_ws, H, Tp, L, nc);
        hipLaunchKernelGGL(ret_bwd_outer_rm_kernel, dim3(nc, H, nseq), dim3(256), 0, stream, (const __bf16*)Q, ss, hs, 64, (const __bf16*)dO,
                           (long)Tp * ldo, 64L, ldo, g_ws, H, Tp, L, nc);
        if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    }
    hipLaunchKernelGGL(ret_bwd_scan_kernel, dim3(H, nseq), dim3(256), 0, stream, kv_ws, g_ws, (__bf16*)St, H, nc);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_ret_bwd_states(const void* Kt, const void* Vt, const void* Qt, const void* dOt, float* kv_ws, float* g_ws, void* St,
                               int nseq, int H, int Tp, int L, int nc, hipStream_t stream) {
    if (!Kt || !Vt || !Qt || !dOt || !kv_ws || !g_ws || !St || nseq <= 0 || nseq > 65535 || H <= 0 || Tp <= 0 || (Tp % 64) || L <= 0 ||
        nc <= 0 || (long)nc * L > Tp)
        return EEND_EINVAL;
    if (nc > 1) {
        hipLaunchKernelGGL(ret_bwd_outer_kernel, dim3(nc, H, nseq), dim3(256), 0, stream, (const __bf16*)Kt, (const __bf16*)Vt, kv_ws, H, Tp, L, nc);
        hipLaunchKernelGGL(ret_bwd_outer_kernel, dim3(nc, H, nseq), dim3(256), 0, stream, (const __bf16*)Qt, (const __bf16*)dOt, g_ws, H, Tp, L, nc);
        if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    }
    hipLaunchKernelGGL(ret_bwd_scan_kernel, dim3(H, nseq), dim3(256), 0, stream, kv_ws, g_ws, (__bf16*)St, H, nc);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
