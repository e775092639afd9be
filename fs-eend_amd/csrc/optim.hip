// Optimiser side of the training step, on flat f32 buffers (parameters, gradients, Adam moments are each ONE
// contiguous allocation; the 8 never-graded tensors of the model simply keep a zero gradient, for which Adam's
// update is exactly zero -- the reference's optimiser skips them because their .grad is None):
//   grad_sumsq        deterministic two-stage sum of squares  -> global gradient norm for clipping
//   adam_step         clip_grad_norm_ (Lightning gradient_clip_val, FS-EEND/train_dia.py:153) folded into
//                     torch.optim.Adam(betas, eps) (train_dia.py:83-88); lr / bias corrections come from a small
//                     device array so the launch is graph-capturable
//   prep_weights      table-driven re-layout + cast of the updated f32 parameters into the MFMA operand copies the
//                     forward (f16) and backward (bf16, transposed) kernels read -- one launch for all tensors
//   scalar_sum        fixed-order sum of a partial vector (loss)
#include "train_common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256)
void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = ((const float4*)g)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = scale * sum(partial[0..n))  (single block, fixed order)
__global__ __launch_bounds__(256)
void scalar_sum_kernel(const float* __restrict__ partial, long n, float scale, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// hp = {lr, 1 - beta1^t, 1 - beta2^t, max_norm (<= 0: no clipping)}; gsumsq = sum of squares of ALL gradients.
__global__ __launch_bounds__(256)
void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                 const float* __restrict__ hp, const float* __restrict__ gsumsq, float b1, float b2, float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lr = hp[0], bc1 = hp[1], bc2 = hp[2], max_norm = hp[3];
    float coef = 1.0f;
    if (max_norm > 0.f) {
        coef = max_norm / (__builtin_sqrtf(gsumsq[0]) + 1e-6f);
        coef = coef < 1.0f ? coef : 1.0f;
    }
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = __builtin_sqrtf(vi) / __builtin_sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

// acc = (first ? 0 : acc) + scale * g  (gradient accumulation over micro-batches, FS-EEND/train_dia.py:151
// accumulate_grad_batches: Lightning divides each micro-batch loss by the number of accumulated batches)
__global__ __launch_bounds__(256)
void grad_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ g, float scale, int first, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    acc[i] = (first ? 0.f : acc[i]) + scale * g[i];
}

__global__ __launch_bounds__(256)
void prep_weights_kernel(const PrepEntry* __restrict__ tab) {
    const PrepEntry e = tab[blockIdx.y];
    const long total = (long)e.A * e.B * e.Cpad;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % e.Cpad);
        const long ab = idx / e.Cpad;
        const int b = (int)(ab % e.B), a = (int)(ab / e.B);
        float x = 0.f;
        if (c < e.C) {
            x = e.src[e.off + (long)a * e.sa + (long)b * e.sb + (long)c * e.sc];
            if (a < e.nscale) x *= e.scale;
        }
        if (e.dtype == 0) ((_Float16*)e.dst)[idx] = to_f16_sat(x);
        else if (e.dtype == 1) ((__bf16*)e.dst)[idx] = (__bf16)x;
        else ((float*)e.dst)[idx] = x;
    }
}

}  // namespace

int eend_launch_grad_sumsq(const float* g, long n, float* partial_ws, float* out, hipStream_t stream) {
    if (!g || !partial_ws || !out || n <= 0) return EEND_EINVAL;
    long nbl = (n / 4 + 255) / 256;
    const int nb = (int)(nbl < 1 ? 1 : nbl > 1024 ? 1024 : nbl);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, stream, g, n, partial_ws);
    if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    hipLaunchKernelGGL(scalar_sum_kernel, dim3(1), dim3(256), 0, stream, partial_ws, (long)nb, 1.0f, out);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_scalar_sum(const float* partial, long n, float scale, float* out, hipStream_t stream) {
    if (!partial || !out || n <= 0) return EEND_EINVAL;
    hipLaunchKernelGGL(scalar_sum_kernel, dim3(1), dim3(256), 0, stream, partial, n, scale, out);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_adam(float* p, const float* g, float* m, float* v, long n, const float* hp, const float* gsumsq, float b1, float b2,
                     float eps, hipStream_t stream) {
    if (!p || !g || !m || !v || !hp || !gsumsq || n <= 0) return EEND_EINVAL;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, m, v, n, hp, gsumsq, b1, b2, eps);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_grad_accumulate(float* acc, const float* g, float scale, int first, long n, hipStream_t stream) {
    if (!acc || !g || n <= 0) return EEND_EINVAL;
    hipLaunchKernelGGL(grad_accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, acc, g, scale, first, n);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_prep_weights(const PrepEntry* tab, int n_entries, hipStream_t stream) {
    if (!tab || n_entries <= 0 || n_entries > 65535) return EEND_EINVAL;
    hipLaunchKernelGGL(prep_weights_kernel, dim3(64, n_entries), dim3(256), 0, stream, tab);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
