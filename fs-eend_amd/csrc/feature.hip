// Feature front-end: 8 kHz waveform -> STFT (Hann 200 zero-padded to n_fft 256,
// hop 80, centred) -> |.|^2 -> 23 Slaney mel bands -> log10(max(., 1e-10)) -> optional (cumulative) mean
// normalisation -> +-c frame splice -> subsample, i.e. {LS,FS}-EEND/datasets/feature.py: stft :166-191,
// transform :43-131 (logmel23 / logmel23_mn / logmel23_cummn), splice :141-163, subsample :133-138,
// extract_fbank :324-336.  One hour of audio is 115 MB in and 50 MB out: the stage is bandwidth-trivial next
// to the model, so the kernels are written for exactness (fp32 end to end, exact-fp32 MFMA) and simplicity.
//
// stft_logmel_kernel: a block owns 64 frames.  The 129-bin DFT of the 200 non-zero window taps is a
// [64 x 200] x [200 x 288] product on v_mfma_f32_16x16x4_f32 (exact fp32 multiply, fp32 accumulate): the
// table already carries the window, columns 0..128 are the cosine (real) and 144..272 the -sine (imaginary)
// parts, so real and imaginary parts of a bin sit in the same lane/register of accumulator tiles t and t+9 and
// the power needs no data movement.  The table streams through LDS in 20-tap slices; the block's samples are
// staged once (frames overlap: 5240 samples for 64 frames).  Power tiles go through LDS to become the A operand
// of the [16 x 132] x [132 x 32] mel product; log10 in the epilogue.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int WIN = 200, HOP = 80, NCOL = 288, IMOFF = 144, NBIN = 129, KMEL = 132, NMEL = 23, MELP = 32;
constexpr int FB = 64;                               // frames per block
constexpr int XSPAN = (FB - 1) * HOP + WIN;          // 5240 samples
constexpr int KC = 20;                               // window taps per staged table slice
constexpr int PSTR = KMEL + 1;                       // padded power row
constexpr int SM_X = 0;
constexpr int SM_B = SM_X + XSPAN * 4;               // 20960
constexpr int SM_P = SM_B + KC * NCOL * 4;           // + 23040
constexpr int SM_M = SM_P + 4 * 16 * PSTR * 4;       // + 34048
constexpr int SM_BYTES = SM_M + KMEL * MELP * 4;     // + 16896 = 94944

__global__ __launch_bounds__(256)
void stft_logmel_kernel(const float* __restrict__ y, long len, long first, int n_frames, const float* __restrict__ dft,
                        const float* __restrict__ melT, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)(smem + SM_X);
    float* Bs = (float*)(smem + SM_B);
    float* Pw = (float*)(smem + SM_P);
    float* Ms = (float*)(smem + SM_M);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fk = lane >> 4;
    const int f0 = blockIdx.x * FB;
    const long base = first + (long)f0 * HOP;            // sample index of tap 0 of the block's first frame

    for (int i = tid; i < XSPAN; i += 256) {
        const long idx = base + i;
        xs[i] = (idx >= 0 && idx < len) ? y[idx] : 0.f;
    }
    for (int i = tid; i < KMEL * MELP; i += 256) Ms[i] = melT[i];

    f32x4 acc[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < WIN; k0 += KC) {
        __syncthreads();
        for (int i = tid; i < KC * NCOL; i += 256) Bs[i] = dft[(long)k0 * NCOL + i];
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KC; ks += 4) {
            const float a = xs[(wave * 16 + frow) * HOP + k0 + ks + fk];
#pragma unroll
            for (int t = 0; t < 18; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[(ks + fk) * NCOL + t * 16 + frow], acc[t], 0, 0, 0);
        }
    }
    // power spectrum of the wave's 16 frames -> LDS [frame][bin]  (D layout: column = lane & 15, row = (lane >> 4) * 4 + reg)
    float* P = Pw + wave * 16 * PSTR;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int bin = t * 16 + frow;
        if (bin < KMEL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) P[(fk * 4 + r) * PSTR + bin] = acc[t][r] * acc[t][r] + acc[t + 9][r] * acc[t + 9][r];
        }
    }
    __syncthreads();
    f32x4 am[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 3
    for (int k0 = 0; k0 < KMEL; k0 += 4) {
        const float a = P[frow * PSTR + k0 + fk];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            am[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Ms[(k0 + fk) * MELP + mt * 16 + frow], am[mt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = mt * 16 + frow;
        if (m < NMEL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = f0 + wave * 16 + fk * 4 + r;
                if (f < n_frames) out[(long)f * NMEL + m] = log10f(__builtin_fmaxf(am[mt][r], 1e-10f));
            }
        }
    }
}

// Mean normalisations of a (T, F) map, one block per column, the running column sum carried in fp64:
//   mode 1 (logmel23_mn)    out = Y - mean over all frames
//   mode 2 (logmel23_cummn) out[t] = Y[t] - (Y[0] + .. + Y[t]) / (t + 1)
constexpr int NT2 = 1024, PER = 4;
__global__ __launch_bounds__(NT2)
void colnorm_kernel(const float* __restrict__ Y, float* __restrict__ out, int T, int F, int mode) {
    __shared__ double wsum[NT2 / 64];
    __shared__ double carry_s;
    const int col = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0.0;
    __syncthreads();
    if (mode == 1) {
        double s = 0.0;
        for (int t = tid; t < T; t += NT2) s += (double)Y[(long)t * F + col];
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) wsum[wave] = s;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < NT2 / 64; ++w) tot += wsum[w];
            carry_s = tot / (double)T;
        }
        __syncthreads();
        const float mean = (float)carry_s;
        for (int t = tid; t < T; t += NT2) out[(long)t * F + col] = Y[(long)t * F + col] - mean;
        return;
    }
    for (int t0 = 0; t0 < T; t0 += NT2 * PER) {
        float v[PER];
        double loc[PER], s = 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int t = t0 + tid * PER + i;
            v[i] = t < T ? Y[(long)t * F + col] : 0.f;
            s += (double)v[i];
            loc[i] = s;
        }
        double incl = s;                                         // inclusive scan of the per-thread sums inside the wave
        for (int d = 1; d < 64; d <<= 1) {
            const double o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        double pre = carry_s;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        pre += incl - s;                                         // exclusive prefix of this thread
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int t = t0 + tid * PER + i;
            if (t < T) out[(long)t * F + col] = v[i] - (float)((pre + loc[i]) / (double)(t + 1));
        }
        __syncthreads();
        if (tid == NT2 - 1) carry_s = pre + s;
        __syncthreads();
    }
}

// out[j][c * F + m] = Y[j * sub + c - ctx][m] (zero outside [0, T)): splice of 2 ctx + 1 frames, every sub-th row
__global__ __launch_bounds__(256)
void splice_subsample_kernel(const float* __restrict__ Y, int T, int F, int ctx, int sub, int To, float* __restrict__ out) {
    const int W = F * (2 * ctx + 1);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)To * W) return;
    const int j = (int)(idx / W), e = (int)(idx - (long)j * W);
    const int c = e / F, m = e - c * F;
    const long t = (long)j * sub + c - ctx;
    out[idx] = (t >= 0 && t < T) ? Y[t * F + m] : 0.f;
}

}  // namespace

int eend_launch_stft_logmel(const float* y, long len, long first, int n_frames, const float* dft, const float* melT, float* out,
                            hipStream_t stream) {
    if (!y || !dft || !melT || !out || len <= 0 || n_frames <= 0) return EEND_EINVAL;
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)stft_logmel_kernel, SM_BYTES)) return EEND_ELAUNCH;
    hipLaunchKernelGGL(stft_logmel_kernel, dim3((n_frames + FB - 1) / FB), dim3(256), SM_BYTES, stream, y, len, first, n_frames, dft, melT, out);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_colnorm(const float* Y, float* out, int T, int F, int mode, hipStream_t stream) {
    if (!Y || !out || T <= 0 || F <= 0 || mode < 1 || mode > 2) return EEND_EINVAL;
    hipLaunchKernelGGL(colnorm_kernel, dim3(F), dim3(NT2), 0, stream, Y, out, T, F, mode);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_splice_subsample(const float* Y, int T, int F, int ctx, int sub, float* out, hipStream_t stream) {
    if (!Y || !out || T <= 0 || F <= 0 || ctx < 0 || sub < 1) return EEND_EINVAL;
    const int To = (T + sub - 1) / sub;
    const long n = (long)To * F * (2 * ctx + 1);
    hipLaunchKernelGGL(splice_subsample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, Y, T, F, ctx, sub, To, out);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
