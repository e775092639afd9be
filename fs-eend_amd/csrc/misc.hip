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

// attractors /= ||attractors||_2 (FS model :43/:76, no eps) and
// logits[b,t,c] = <emb[b,t,:], attractors[b,t,c,:]> (FS model :60/:79), one wave per (b,t,c).
// attr slab rows are ((b*C + c)*Tp + t); outputs are dense (B,T,C,D) / (B,T,C).
__global__ __launch_bounds__(256)
void head_kernel(const float* __restrict__ emb, const float* __restrict__ attr, float* __restrict__ attr_out,
                 float* __restrict__ logits, int B, int T, int Tp, int C, int D) {
    const int lane = threadIdx.x & 63;
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b*T + t)*C + c
    if (idx >= (long)B * T * C) return;
    const int c = (int)(idx % C);
    const long bt = idx / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const float* a = attr + (((long)b * C + c) * Tp + t) * D;
    const float* e = emb + ((long)b * Tp + t) * D;
    float ss = 0.f, dot = 0.f;
    float4 av[2];
    int nv = 0;
    for (int k = lane * 4; k < D; k += 256, ++nv) {
        const float4 x = *(const float4*)(a + k);
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
        float4 x = (nv < 2) ? av[nv] : *(const float4*)(a + k);
        x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv;
        *(float4*)(ao + k) = x;
    }
    if (lane == 0) logits[idx] = dot * inv;
}

}  // namespace

int eend_launch_bn_cast_pad(const float* x, const float* bn_w, const float* bn_b, const float* bn_mean,
                            const float* bn_var, float eps, void* out16, int B, int T, int Tp, int Fin,
                            int Fpad, int apply_bn, hipStream_t stream) {
    if (B <= 0 || T <= 0 || Tp < T || Fin <= 0 || Fpad < Fin) return EEND_EINVAL;
    const long rows = (long)B * Tp;
    hipLaunchKernelGGL(bn_cast_pad_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, bn_w,
                       bn_b, bn_mean, bn_var, eps, (_Float16*)out16, B, T, Tp, Fin, Fpad, apply_bn);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_head(const float* emb, const float* attr, float* attr_out, float* logits, int B, int T,
                     int Tp, int C, int D, hipStream_t stream) {
    if (B <= 0 || T <= 0 || Tp < T || C <= 0 || D <= 0 || (D & 3)) return EEND_EINVAL;
    const long n = (long)B * T * C;
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, emb, attr, attr_out,
                       logits, B, T, Tp, C, D);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
