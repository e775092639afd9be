// Embedding-consistency loss (FS model :46-57, LS model :92-113):
//     A[b,i,j] = <e_i, e_j> / (|e_i| |e_j| + 1e-6)        e = frame embeddings (B, T, 256)
//     L[b,i,j] = <y_i, y_j> / (|y_i| |y_j| + 1e-6)        y = zero-padded speaker labels (B, T, C)
//     loss     = mean over (b, i, j) of (A - L)^2
// The reference materialises both (B,T,T) maps; here a block owns one 64 x 64 tile of one utterance,
// forms it with the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: the reference is an fp32 matmul and this is a
// loss value, so no reduced-precision operands), adds the label map on the VALU (C <= 16) and reduces the
// squared difference to one partial per block.  A second, single-block launch sums the partials in a fixed
// order (deterministic, no atomics) and applies the normalisation (1 / (B T T) for FS-EEND's mse_loss; the LS
// model zeroes the embeddings of frames beyond each utterance's length and divides by sum(len^2)).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TS = 64;            // tile side
constexpr int KC = 32;            // k chunk staged per step
constexpr int LDE = KC + 1;       // padded row stride (floats)
constexpr int CMAX = 16;

__global__ __launch_bounds__(256)
void emb_consistency_tile_kernel(const float* __restrict__ emb, const float* __restrict__ tgt, const int* __restrict__ lens,
                                 float* __restrict__ partial, int T, int Tp, int D, int C) {
    __shared__ float Ei[TS * LDE], Ej[TS * LDE];
    __shared__ float Yi[TS * CMAX], Yj[TS * CMAX];
    __shared__ float n2i[TS], n2j[TS], ny2i[TS], ny2j[TS];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
    const float* __restrict__ E = emb + (size_t)b * Tp * D;
    const float* __restrict__ Y = tgt + (size_t)b * T * C;
    const int elen = lens ? (lens[b] < T ? lens[b] : T) : T;     // LS variant: embeddings of frames >= len count as zero (model :100)

    // labels of the two row sets (zero beyond T) and their squared norms
    for (int q = tid; q < 2 * TS * CMAX; q += 256) {
        const int which = q / (TS * CMAX), r = (q % (TS * CMAX)) / CMAX, c = q % CMAX;
        const int t = (which ? j0 : i0) + r;
        const float v = (t < T && c < C) ? Y[(size_t)t * C + c] : 0.f;
        (which ? Yj : Yi)[r * CMAX + c] = v;
    }
    if (tid < TS) { n2i[tid] = 0.f; n2j[tid] = 0.f; }
    __syncthreads();
    if (tid < 2 * TS) {
        const float* y = (tid < TS ? Yi : Yj) + (tid & (TS - 1)) * CMAX;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) sum += y[c] * y[c];
        (tid < TS ? ny2i : ny2j)[tid & (TS - 1)] = sum;
    }

    // A tile: wave w owns rows i0 + 16 w .. +16, all 64 columns (4 MFMA column tiles)
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;
    for (int k0 = 0; k0 < D; k0 += KC) {
        __syncthreads();
        for (int q = tid; q < 2 * TS * KC; q += 256) {          // stage [64][32] chunks of both row sets
            const int which = q / (TS * KC), r = (q % (TS * KC)) / KC, k = q % KC;
            const int t = (which ? j0 : i0) + r;
            const float v = t < elen ? E[(size_t)t * D + k0 + k] : 0.f;
            (which ? Ej : Ei)[r * LDE + k] = v;
        }
        __syncthreads();
        if (tid < 2 * TS) {                                      // running squared norms of the rows
            const float* e = (tid < TS ? Ei : Ej) + (tid & (TS - 1)) * LDE;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < KC; ++k) sum += e[k] * e[k];
            (tid < TS ? n2i : n2j)[tid & (TS - 1)] += sum;
        }
#pragma unroll
        for (int ks = 0; ks < KC; ks += 4) {
            const float a = Ei[(wave * 16 + frow) * LDE + ks + fk];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const float bv = Ej[(ct * 16 + frow) * LDE + ks + fk];
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[ct], 0, 0, 0);
            }
        }
    }
    __syncthreads();

    // D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
    float sq = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int jl = ct * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int il = wave * 16 + (lane >> 4) * 4 + r;
            if (i0 + il < T && j0 + jl < T) {
                const float am = acc[ct][r] / (__builtin_sqrtf(n2i[il]) * __builtin_sqrtf(n2j[jl]) + 1e-6f);
                float dot = 0.f;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) dot = __builtin_fmaf(Yi[il * CMAX + c], Yj[jl * CMAX + c], dot);
                const float lm = dot / (__builtin_sqrtf(ny2i[il]) * __builtin_sqrtf(ny2j[jl]) + 1e-6f);
                const float d = am - lm;
                sq = __builtin_fmaf(d, d, sq);
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) sq = wave_xor_add(sq, m);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0)
        partial[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256)
void emb_consistency_sum_kernel(const float* __restrict__ partial, float* __restrict__ out, int n, float inv_count) {
    __shared__ float red[256];
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) sum += partial[i];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * inv_count;
}

}  // namespace

int eend_launch_emb_consistency(const float* emb, const float* tgt, const int* lens, float inv_count, float* partial_ws, float* out,
                                int B, int T, int Tp, int D, int C, hipStream_t stream) {
    if (!emb || !tgt || !partial_ws || !out || B <= 0 || T <= 0 || Tp < T || D <= 0 || (D % KC) != 0 || C < 1 || C > CMAX)
        return EEND_EINVAL;
    const int nt = (T + TS - 1) / TS;
    hipLaunchKernelGGL(emb_consistency_tile_kernel, dim3(nt, nt, B), dim3(256), 0, stream, emb, tgt, lens, partial_ws, T, Tp, D, C);
    if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    const float inv = inv_count > 0.f ? inv_count : 1.0f / ((float)B * (float)T * (float)T);
    hipLaunchKernelGGL(emb_consistency_sum_kernel, dim3(1), dim3(256), 0, stream, partial_ws, out, nt * nt * B, inv);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
