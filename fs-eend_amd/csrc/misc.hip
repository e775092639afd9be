// Bandwidth-bound edge kernels of the EEND hot path.
#include "common.h"
#include "kernels.h"

namespace {

// x f32 [B][T][Fin] -> f16 [B][Tp][Fpad]: eval-mode BatchNorm1d over the feature axis
// (FS model :166, running statistics) fused with the cast to the MFMA operand type and the
// zero padding to the frame slab (Tp % 64 == 0) and to K % 64 == 0.  One wave per frame row.
__global__ __launch_bounds__(256)
void bn_cast_pad_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                        const float* __restrict__ mean, const float* __restrict__ var, float eps,
                        _Float16* __restrict__ out, int B, int T, int Tp, int Fin, int Fpad, int apply_bn) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);       // b*Tp + t
    if (row >= (long)B * Tp) return;
    const int bb = (int)(row / Tp), t = (int)(row - (long)bb * Tp);
    _Float16* o = out + row * Fpad;
    if (t >= T) {
        for (int k = lane; k < Fpad; k += 64) o[k] = (_Float16)0.f;
        return;
    }
    const float* xi = x + ((long)bb * T + t) * Fin;
    for (int k = lane; k < Fpad; k += 64) {
        float v = 0.f;
        if (k < Fin) {
            v = xi[k];
            if (apply_bn) v = (v - mean[k]) / __builtin_sqrtf(var[k] + eps) * w[k] + b[k];
        }
        o[k] = to_f16_sat(v);
    }
}

// Same as bn_cast_pad_kernel, but reads every utterance through a device pointer table
// (x_ptrs[b] -> f32 [len[b]][Fin]) and supplies the reference's pad value for t >= len[b]
// (-1 for FS-EEND, model :165; 0 for LS-EEND, model :280): pad_sequence + BatchNorm + cast +
// slab padding in one launch instead of B small device-to-device copies.
constexpr int GB_ROWS = 8;      // slab rows per wave: the per-feature BatchNorm scale / shift (a sqrt and a divide each) are
                                // computed once per wave and reused; a lane owns feature pairs (2l, 2l+1) + 128 j -> 4-byte stores
__global__ __launch_bounds__(256)
void gather_bn_cast_pad_kernel(const float* const* __restrict__ x_ptrs, const int* __restrict__ lens, float pad_value,
                               const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, _Float16* __restrict__ out, int B, int T, int Tp,
                               int Fin, int Fpad, int apply_bn) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * GB_ROWS;
    const long nrows = (long)B * Tp;
    if (row0 >= nrows) return;
    constexpr int NJ = 4;                               // feature pairs per lane: covers Fpad <= 512
    float sc[NJ][2], sh[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = 2 * lane + 128 * j + e;
            sc[j][e] = 1.0f; sh[j][e] = 0.0f;
            if (apply_bn && k < Fin) {
                sc[j][e] = w[k] / __builtin_sqrtf(var[k] + eps);
                sh[j][e] = b[k] - mean[k] * sc[j][e];
            }
        }
    for (int r = 0; r < GB_ROWS; ++r) {
        const long row = row0 + r;
        if (row >= nrows) return;
        const int bb = (int)(row / Tp), t = (int)(row - (long)bb * Tp);
        unsigned* o = (unsigned*)(out + row * Fpad);
        const float* xi = (t < T && t < lens[bb]) ? x_ptrs[bb] + (long)t * Fin : nullptr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = 2 * lane + 128 * j;
            if (k >= Fpad) break;
            float v0 = 0.f, v1 = 0.f;
            if (t < T) {
                if (k < Fin) v0 = (xi ? xi[k] : pad_value) * sc[j][0] + sh[j][0];
                if (k + 1 < Fin) v1 = (xi ? xi[k + 1] : pad_value) * sc[j][1] + sh[j][1];
            }
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 pk;
            pk[0] = to_f16_sat(v0); pk[1] = to_f16_sat(v1);
            o[k >> 1] = __builtin_bit_cast(unsigned, pk);
        }
    }
}

// attractors /= ||attractors||_2 (FS model :43/:76, no eps) and
// logits[b,t,c] = <emb[b,t,:], attractors[b,t,c,:]> (FS model :60/:79), one wave per (b,t,c).
// attr slab rows are ((b*C + c)*Tp + t); outputs are dense (B,T,C,D) / (B,T,C).
// AT = float, or _Float16 when the decoder's last LayerNorm output is only kept as f16 (FS-EEND f16 residual stream).
template <class AT>
DEV float4 load4f(const AT* p) {
    if constexpr (sizeof(AT) == 4) return *(const float4*)p;
    else { const f16x4 h = *(const f16x4*)p; return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); }
}

template <class AT>
__global__ __launch_bounds__(256)
void head_kernel(const float* __restrict__ emb, const AT* __restrict__ attr, float* __restrict__ attr_out,
                 float* __restrict__ logits, int B, int T, int Tp, int C, int D) {
    const int lane = threadIdx.x & 63;
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b*T + t)*C + c
    if (idx >= (long)B * T * C) return;
    const int c = (int)(idx % C);
    const long bt = idx / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const AT* a = attr + (((long)b * C + c) * Tp + t) * D;
    const float* e = emb + ((long)b * Tp + t) * D;
    float ss = 0.f, dot = 0.f;
    float4 av[2];
    int nv = 0;
    for (int k = lane * 4; k < D; k += 256, ++nv) {
        const float4 x = load4f(a + k);
        const float4 y = *(const float4*)(e + k);
        if (nv < 2) av[nv] = x;
        ss += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        dot += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { ss = wave_xor_add(ss, m); dot = wave_xor_add(dot, m); }
    const float inv = 1.0f / __builtin_sqrtf(ss);
    float* ao = attr_out + idx * D;
    nv = 0;
    for (int k = lane * 4; k < D; k += 256, ++nv) {
        float4 x = (nv < 2) ? av[nv] : load4f(a + k);
        x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv;
        *(float4*)(ao + k) = x;
    }
    if (lane == 0) logits[idx] = dot * inv;
}

// D = 256 (every model here): a wave takes R rows and has all their loads in flight before the first reduction (one dependent load
// chain per wave left the memory system at 3.9 TB/s on the 280 MB of the FS head: 71 us).
template <class AT, int R>
__global__ __launch_bounds__(256)
void head_rows_kernel(const float* __restrict__ emb, const AT* __restrict__ attr, float* __restrict__ attr_out,
                      float* __restrict__ logits, int B, int T, int Tp, int C) {
    constexpr int D = 256;
    const int lane = threadIdx.x & 63;
    const long n = (long)B * T * C;
    const long base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;      // (b*T + t)*C + c
    if (base >= n) return;
    float4 x[R], y[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long idx = base + r < n ? base + r : n - 1;
        const int c = (int)(idx % C);
        const long bt = idx / C;
        const int t = (int)(bt % T), b = (int)(bt / T);
        x[r] = load4f(attr + (((long)b * C + c) * Tp + t) * D + lane * 4);
        y[r] = *(const float4*)(emb + ((long)b * Tp + t) * D + lane * 4);
    }
    float ss[R], dot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ss[r] = x[r].x * x[r].x + x[r].y * x[r].y + x[r].z * x[r].z + x[r].w * x[r].w;
        dot[r] = x[r].x * y[r].x + x[r].y * y[r].y + x[r].z * y[r].z + x[r].w * y[r].w;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) { ss[r] = wave_xor_add(ss[r], m); dot[r] = wave_xor_add(dot[r], m); }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (base + r >= n) break;
        const float inv = 1.0f / __builtin_sqrtf(ss[r]);
        float4 v = x[r];
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        *(float4*)(attr_out + (base + r) * D + lane * 4) = v;
        if (lane == 0) logits[base + r] = dot[r] * inv;
    }
}

// Stand-alone LayerNorm f32 [M][D] -> f16 (D <= 1024, D % 4 == 0), one wave per row.  Used where
// LS-EEND applies two LayerNorms back to back (block-final LN followed by the next module's
// pre-norm: conformer/encoder.py:104-110 then feed_forward.py:48).
__global__ __launch_bounds__(256)
void layernorm_f16_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                          float eps, _Float16* __restrict__ out, long M, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * D;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + i * 256;
        v[i] = (k < D) ? *(const float4*)(xr + k) : make_float4(0, 0, 0, 0);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) s = wave_xor_add(s, m);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + i * 256;
        if (k < D) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) q = wave_xor_add(q, m);
    const float rstd = 1.0f / __builtin_sqrtf(q / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane * 4 + i * 256;
        if (k < D) {
            const float4 g = *(const float4*)(gamma + k), b = *(const float4*)(beta + k);
            f16x4 o;
            o[0] = to_f16_sat((v[i].x - mean) * rstd * g.x + b.x);
            o[1] = to_f16_sat((v[i].y - mean) * rstd * g.y + b.y);
            o[2] = to_f16_sat((v[i].z - mean) * rstd * g.z + b.z);
            o[3] = to_f16_sat((v[i].w - mean) * rstd * g.w + b.w);
            *(f16x4*)(out + row * D + k) = o;
        }
    }
}

// Causal depthwise Conv1d(k taps, left context k-1, no bias) -> BatchNorm1d(eval) -> Swish
// (LS-EEND/nnet/conformer/convolution.py:65-68,143-147).  x,out f16 [nseq][Tp][D]; w f32 [D][k].
// One thread per channel, a 64-frame strip per block.  The k-tap window slides through registers:
// every input element is read exactly once (2*D-byte coalesced rows, 8 frames per load batch so
// 8 independent loads are in flight), k FMAs per output.  K is a template parameter for the
// shipped kernel sizes; other sizes take the generic (re-reading) kernel below.
template <int K>
__global__ __launch_bounds__(256)
void dwconv_bn_swish_win_kernel(const _Float16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bw,
                                const float* __restrict__ bb, const float* __restrict__ bm, const float* __restrict__ bv,
                                float eps, _Float16* __restrict__ out, int Tp, int D, const _Float16* __restrict__ halo) {
    const int seq = blockIdx.y, t0 = blockIdx.x * 64;
    const int c = blockIdx.z * 256 + threadIdx.x;
    if (c >= D) return;
    const float sc = bw[c] / __builtin_sqrtf(bv[c] + eps);
    const float sh = bb[c] - bm[c] * sc;
    const _Float16* xs = x + (size_t)seq * Tp * D + c;
    _Float16* os = out + (size_t)seq * Tp * D + c;
    float wk[K], win[K];
#pragma unroll
    for (int j = 0; j < K; ++j) wk[j] = w[(size_t)c * K + j];
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        const int ts = t0 - (K - 1) + j;
        // frames before the slab: zero (start of the recording) or the K-1 frames carried over from the previous call
        win[j] = ts >= 0 ? (float)xs[(size_t)ts * D] : (halo ? (float)halo[((size_t)seq * (K - 1) + (ts + K - 1)) * D + c] : 0.f);
    }
    for (int tb = t0; tb < t0 + 64 && tb < Tp; tb += 8) {
        float xn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xn[u] = (tb + u < Tp) ? (float)xs[(size_t)(tb + u) * D] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            win[K - 1] = xn[u];
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) y = __builtin_fmaf(wk[j], win[j], y);
#pragma unroll
            for (int j = 0; j < K - 1; ++j) win[j] = win[j + 1];
            y = y * sc + sh;
            if (tb + u < Tp) os[(size_t)(tb + u) * D] = to_f16_sat(y / (1.0f + __expf(-y)));
        }
    }
}

__global__ __launch_bounds__(256)
void dwconv_bn_swish_kernel(const _Float16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bw,
                            const float* __restrict__ bb, const float* __restrict__ bm, const float* __restrict__ bv,
                            float eps, _Float16* __restrict__ out, int Tp, int D, int k, const _Float16* __restrict__ halo) {
    const int seq = blockIdx.y, t0 = blockIdx.x * 64;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float sc = bw[c] / __builtin_sqrtf(bv[c] + eps);
        const float sh = bb[c] - bm[c] * sc;
        const _Float16* xs = x + (size_t)seq * Tp * D + c;
        const float* wc = w + (size_t)c * k;
        for (int t = t0; t < t0 + 64 && t < Tp; ++t) {
            float y = 0.f;
            for (int j = 0; j < k; ++j) {
                const int ts = t - (k - 1) + j;
                if (ts >= 0) y = __builtin_fmaf(wc[j], (float)xs[(size_t)ts * D], y);
                else if (halo) y = __builtin_fmaf(wc[j], (float)halo[((size_t)seq * (k - 1) + (ts + k - 1)) * D + c], y);
            }
            y = y * sc + sh;
            out[((size_t)seq * Tp + t) * D + c] = to_f16_sat(y / (1.0f + __expf(-y)));
        }
    }
}

}  // namespace

int eend_launch_layernorm_f16(const float* x, const float* gamma, const float* beta, float eps, void* out16,
                              long M, int D, hipStream_t stream) {
    if (M <= 0 || D <= 0 || D > 1024 || (D & 3)) return EEND_EINVAL;
    hipLaunchKernelGGL(layernorm_f16_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, x, gamma, beta, eps,
                       (_Float16*)out16, M, D);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_dwconv_bn_swish(const void* x16, const float* w, const float* bn_w, const float* bn_b,
                                const float* bn_mean, const float* bn_var, float eps, void* out16, int nseq,
                                int Tp, int D, int k, const void* halo16, hipStream_t stream) {
    if (nseq <= 0 || nseq > 65535 || Tp <= 0 || D <= 0 || k <= 0) return EEND_EINVAL;
    const dim3 grid((Tp + 63) / 64, nseq, (D + 255) / 256);
#define DW_CASE(KK)                                                                                                    \
    case KK:                                                                                                           \
        hipLaunchKernelGGL(dwconv_bn_swish_win_kernel<KK>, grid, dim3(256), 0, stream, (const _Float16*)x16, w, bn_w,   \
                           bn_b, bn_mean, bn_var, eps, (_Float16*)out16, Tp, D, (const _Float16*)halo16);              \
        break;
    switch (k) {
        DW_CASE(16) DW_CASE(7) DW_CASE(15) DW_CASE(31) DW_CASE(32)
        default:
            hipLaunchKernelGGL(dwconv_bn_swish_kernel, dim3((Tp + 63) / 64, nseq), dim3(256), 0, stream, (const _Float16*)x16,
                               w, bn_w, bn_b, bn_mean, bn_var, eps, (_Float16*)out16, Tp, D, k, (const _Float16*)halo16);
    }
#undef DW_CASE
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_cast_pad(const float* x, const float* bn_w, const float* bn_b, const float* bn_mean,
                            const float* bn_var, float eps, void* out16, int B, int T, int Tp, int Fin,
                            int Fpad, int apply_bn, hipStream_t stream) {
    if (B <= 0 || T <= 0 || Tp < T || Fin <= 0 || Fpad < Fin) return EEND_EINVAL;
    const long rows = (long)B * Tp;
    hipLaunchKernelGGL(bn_cast_pad_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, bn_w,
                       bn_b, bn_mean, bn_var, eps, (_Float16*)out16, B, T, Tp, Fin, Fpad, apply_bn);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_gather_bn_cast_pad(const float* const* x_ptrs, const int* lens, float pad_value, const float* bn_w,
                                   const float* bn_b, const float* bn_mean, const float* bn_var, float eps, void* out16,
                                   int B, int T, int Tp, int Fin, int Fpad, int apply_bn, hipStream_t stream) {
    if (B <= 0 || T <= 0 || Tp < T || Fin <= 0 || Fpad < Fin || (Fpad & 1) || Fpad > 512) return EEND_EINVAL;
    const long rows = (long)B * Tp;
    hipLaunchKernelGGL(gather_bn_cast_pad_kernel, dim3((unsigned)((rows + 4 * GB_ROWS - 1) / (4 * GB_ROWS))), dim3(256), 0, stream, x_ptrs, lens,
                       pad_value, bn_w, bn_b, bn_mean, bn_var, eps, (_Float16*)out16, B, T, Tp, Fin, Fpad, apply_bn);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_head(const float* emb, const void* attr, int attr_is_f16, float* attr_out, float* logits, int B, int T,
                     int Tp, int C, int D, hipStream_t stream) {
    if (B <= 0 || T <= 0 || Tp < T || C <= 0 || D <= 0 || (D & 3)) return EEND_EINVAL;
    const long n = (long)B * T * C;
    if (D == 256) {
        constexpr int R = 4;
        const unsigned blocks = (unsigned)((n + 4 * R - 1) / (4 * R));
        if (attr_is_f16)
            hipLaunchKernelGGL((head_rows_kernel<_Float16, R>), dim3(blocks), dim3(256), 0, stream, emb, (const _Float16*)attr, attr_out, logits, B, T, Tp, C);
        else
            hipLaunchKernelGGL((head_rows_kernel<float, R>), dim3(blocks), dim3(256), 0, stream, emb, (const float*)attr, attr_out, logits, B, T, Tp, C);
        return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
    }
    if (attr_is_f16)
        hipLaunchKernelGGL(head_kernel<_Float16>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, emb, (const _Float16*)attr,
                           attr_out, logits, B, T, Tp, C, D);
    else
        hipLaunchKernelGGL(head_kernel<float>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, emb, (const float*)attr, attr_out,
                           logits, B, T, Tp, C, D);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
