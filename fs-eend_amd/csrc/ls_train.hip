// Row-local / channel-local (HBM-bound) kernels of the LS-EEND training step: everything of the Conformer-retention
// encoder block and the retention module that is not a GEMM, an attention-shaped product or already in train_rows.hip.
//
//   reference forward                                             kernels here
//   FeedForwardModule        feed_forward.py:47-57                swish_drop_fwd / swish_bwd (activation + its dropout)
//   pre-norm residual blocks modules.py:32-33, encoder.py:76-113  layernorm_train, ln_bwd2 (bf16 / f32 in, accumulate),
//                                                                 resgrad_cast (residual-stream gradient -> branch gradient)
//   ConformerConvModule      convolution.py:138-149               glu_dwconv_fwd, bn_colstats16 / bn_merge (train-mode
//                                                                 BatchNorm1d statistics, SyncBatchNorm-mergeable),
//                                                                 bn_swish_fwd, bn_swish_bwd_stats / _apply, dwconv_glu_bwd
//   MultiScaleRetention      retention.py:222-224                 ret_gate_gn_bwd (swish gate + per-head LayerNorm backward,
//                                                                 emits o~ = c_t * d out_t for the RET kernels of attn_bwd.hip)
//
// d_model = 256.  Row kernels: one wave per row, a lane owns 4 consecutive features (16-byte f32 / 8-byte 2-byte
// accesses).  Channel kernels (depthwise conv, BatchNorm): one thread per channel walking a strip of frames, rows
// are read as coalesced 512-byte lines.  Parameter-gradient and statistics sums are written as per-block partials
// and summed in fixed order by wgrad_reduce_kernel -- deterministic, no atomics.  Frames t >= Tv of a sequence slab
// (Tv = the reference's chunk-padded length, Tp = Tv rounded up to 64) are not part of the reference's tensors: they
// are excluded from every statistic and receive zero gradients.
#include "train_common.h"
#include "kernels.h"

namespace {

constexpr int D = 256;

DEV float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }
DEV float swish_grad(float z) { const float s = sigm(z); return s * (1.0f + z * (1.0f - s)); }

// ---------------------------------------------------------------------------------------------------------------
// a = dropout(swish(z)); z f16 [M][F] (pre-activation, saved), a f16.  8 elements per thread.
__global__ __launch_bounds__(256)
void swish_drop_fwd_kernel(const unsigned short* __restrict__ z, unsigned short* __restrict__ a, long M, int F, const DropSpec drop) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int f8 = F >> 3;
    if (idx >= M * f8) return;
    const long row = idx / f8;
    const int col = (int)(idx - row * f8) * 8;
    const u32x4 v = *(const u32x4*)(z + row * F + col);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned x = v[j];
        float z0 = f16_lo(x), z1 = f16_hi(x);
        float a0 = z0 * sigm(z0), a1 = z1 * sigm(z1);
        a0 = drop_apply(drop, a0, (unsigned)row, (unsigned)(col + 2 * j));
        a1 = drop_apply(drop, a1, (unsigned)row, (unsigned)(col + 2 * j + 1));
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 p;
        p[0] = to_f16_sat(a0);
        p[1] = to_f16_sat(a1);
        o[j] = __builtin_bit_cast(unsigned, p);
    }
    *(u32x4*)(a + row * F + col) = o;
}

// dz = da * keep * scale * swish'(z), in place on the bf16 gradient.
__global__ __launch_bounds__(256)
void swish_bwd_kernel(unsigned short* __restrict__ dz, const unsigned short* __restrict__ z, long M, int F, const DropSpec drop) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int f8 = F >> 3;
    if (idx >= M * f8) return;
    const long row = idx / f8;
    const int col = (int)(idx - row * f8) * 8;
    const u32x4 g = *(const u32x4*)(dz + row * F + col);
    const u32x4 v = *(const u32x4*)(z + row * F + col);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned gx = g[j], zx = v[j];
        float g0 = bf16_lo(gx), g1 = bf16_hi(gx);
        g0 = drop_apply(drop, g0, (unsigned)row, (unsigned)(col + 2 * j)) * swish_grad(f16_lo(zx));
        g1 = drop_apply(drop, g1, (unsigned)row, (unsigned)(col + 2 * j + 1)) * swish_grad(f16_hi(zx));
        o[j] = pack_bf16(g0, g1);
    }
    *(u32x4*)(dz + row * F + col) = o;
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm forward that saves what its backward needs: y16 = x_hat * gamma + beta, x_hat16, 1/sigma.
__global__ __launch_bounds__(256)
void layernorm_train_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                            _Float16* __restrict__ y16, _Float16* __restrict__ xhat16, float* __restrict__ rstd_out, long M) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float4 v = *(const float4*)(x + row * D + lane * 4);
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / D);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    const float rs = 1.0f / __builtin_sqrtf(wave_sum(a * a + b * b + c * c + d * d) * (1.0f / D) + eps);
    const float4 g = *(const float4*)(gamma + lane * 4), be = *(const float4*)(beta + lane * 4);
    f16x4 xh, y;
    xh[0] = to_f16_sat(a * rs); xh[1] = to_f16_sat(b * rs); xh[2] = to_f16_sat(c * rs); xh[3] = to_f16_sat(d * rs);
    y[0] = to_f16_sat(a * rs * g.x + be.x); y[1] = to_f16_sat(b * rs * g.y + be.y);
    y[2] = to_f16_sat(c * rs * g.z + be.z); y[3] = to_f16_sat(d * rs * g.w + be.w);
    *(f16x4*)(xhat16 + row * D + lane * 4) = xh;
    *(f16x4*)(y16 + row * D + lane * 4) = y;
    if (lane == 0) rstd_out[row] = rs;
}

// LayerNorm backward, generalised (train_rows.hip ln_bwd_kernel is the f32-in / overwrite special case):
//   G16: the gradient w.r.t. the LayerNorm output arrives as bf16 (from a data-gradient GEMM) instead of f32;
//   ACC: ds32 += ds (pre-norm blocks: the LayerNorm sits on a branch, its input gradient joins the residual stream).
// Optional ds16 = bf16(alpha16 * dropout(ds)) with its column sums (bias gradient of the linear in front of a
// post-norm LayerNorm, scaled like the branch).  partial: [gridDim.x][3][256] = dgamma, dbeta, colsum(ds16).
template <bool G16, bool ACC>
__global__ __launch_bounds__(256)
void ln_bwd2_kernel(const void* __restrict__ gin, const _Float16* __restrict__ xhat, const float* __restrict__ rstd,
                    const float* __restrict__ gamma, float* ds32, __bf16* __restrict__ ds16, float alpha16,
                    float* __restrict__ partial, long M, const DropSpec drop) {
    __shared__ float red[4][3][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 gm = *(const float4*)(gamma + lane * 4);
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, dbr[4] = {0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        float4 gy;
        if constexpr (G16) {
            const uint2 pk = *(const uint2*)((const __bf16*)gin + row * D + lane * 4);
            gy = make_float4(bf16_lo(pk.x), bf16_hi(pk.x), bf16_lo(pk.y), bf16_hi(pk.y));
        } else {
            gy = *(const float4*)((const float*)gin + row * D + lane * 4);
        }
        const f16x4 xh = *(const f16x4*)(xhat + row * D + lane * 4);
        const float rs = rstd[row];
        const float x0 = (float)xh[0], x1 = (float)xh[1], x2 = (float)xh[2], x3 = (float)xh[3];
        const float d0 = gy.x * gm.x, d1 = gy.y * gm.y, d2 = gy.z * gm.z, d3 = gy.w * gm.w;
        const float c1 = wave_sum(d0 + d1 + d2 + d3) * (1.0f / D);
        const float c2 = wave_sum(d0 * x0 + d1 * x1 + d2 * x2 + d3 * x3) * (1.0f / D);
        const float o0 = rs * (d0 - c1 - x0 * c2), o1 = rs * (d1 - c1 - x1 * c2);
        const float o2 = rs * (d2 - c1 - x2 * c2), o3 = rs * (d3 - c1 - x3 * c2);
        if (ds32) {
            float4 r = make_float4(o0, o1, o2, o3);
            if constexpr (ACC) {
                const float4 p = *(const float4*)(ds32 + row * D + lane * 4);
                r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w;
            }
            *(float4*)(ds32 + row * D + lane * 4) = r;
        }
        if (ds16) {
            const unsigned n = (unsigned)(lane * 4);
            const float m0 = alpha16 * drop_apply(drop, o0, (unsigned)row, n), m1 = alpha16 * drop_apply(drop, o1, (unsigned)row, n + 1);
            const float m2 = alpha16 * drop_apply(drop, o2, (unsigned)row, n + 2), m3 = alpha16 * drop_apply(drop, o3, (unsigned)row, n + 3);
            uint2 pk;
            pk.x = pack_bf16(m0, m1);
            pk.y = pack_bf16(m2, m3);
            *(uint2*)(ds16 + row * D + lane * 4) = pk;
            dbr[0] += m0; dbr[1] += m1; dbr[2] += m2; dbr[3] += m3;
        }
        dg[0] += gy.x * x0; dg[1] += gy.y * x1; dg[2] += gy.z * x2; dg[3] += gy.w * x3;
        db[0] += gy.x; db[1] += gy.y; db[2] += gy.z; db[3] += gy.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wave][0][lane * 4 + e] = dg[e]; red[wave][1][lane * 4 + e] = db[e]; red[wave][2][lane * 4 + e] = dbr[e]; }
    __syncthreads();
    const int c = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        partial[((size_t)blockIdx.x * 3 + k) * D + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
}

// Residual-stream gradient -> gradient of a pre-norm branch output: ds16 = bf16(alpha * dropout(g32)) and its column
// sums (the bias gradient of the branch's last linear).  partial: [gridDim.x][256].
__global__ __launch_bounds__(256)
void resgrad_cast_kernel(const float* __restrict__ g, __bf16* __restrict__ ds16, float alpha, float* __restrict__ partial, long M,
                         const DropSpec drop) {
    __shared__ float red[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        const float4 v = *(const float4*)(g + row * D + lane * 4);
        const unsigned n = (unsigned)(lane * 4);
        const float m0 = alpha * drop_apply(drop, v.x, (unsigned)row, n), m1 = alpha * drop_apply(drop, v.y, (unsigned)row, n + 1);
        const float m2 = alpha * drop_apply(drop, v.z, (unsigned)row, n + 2), m3 = alpha * drop_apply(drop, v.w, (unsigned)row, n + 3);
        uint2 pk;
        pk.x = pack_bf16(m0, m1);
        pk.y = pack_bf16(m2, m3);
        *(uint2*)(ds16 + row * D + lane * 4) = pk;
        acc[0] += m0; acc[1] += m1; acc[2] += m2; acc[3] += m3;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = acc[e];
    __syncthreads();
    const int c = threadIdx.x;
    partial[(size_t)blockIdx.x * D + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ---------------------------------------------------------------------------------------------------------------
// Conformer conv module, train mode.  P f16 [nseq*Tp][512] = pointwise-conv-1 output (value | gate halves);
// u = value * sigmoid(gate) (GLU over channels, activation.py:39-41); c[t] = sum_j w[ch][j] u[t - (K-1) + j] (causal
// depthwise conv, zero left context, convolution.py:65-68).  c16 f16 [nseq*Tp][256]; rows t >= Tv are written as zero.
template <int K>
__global__ __launch_bounds__(256)
void glu_dwconv_fwd_kernel(const _Float16* __restrict__ P, const float* __restrict__ w, _Float16* __restrict__ c16, int Tp, int Tv) {
    const int seq = blockIdx.y, t0 = blockIdx.x * 64, ch = threadIdx.x;
    const _Float16* ps = P + (size_t)seq * Tp * 2 * D + ch;
    _Float16* os = c16 + (size_t)seq * Tp * D + ch;
    float wk[K], win[K];
#pragma unroll
    for (int j = 0; j < K; ++j) wk[j] = w[(size_t)ch * K + j];
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        const int ts = t0 - (K - 1) + j;
        win[j] = ts >= 0 ? (float)ps[(size_t)ts * 2 * D] * sigm((float)ps[(size_t)ts * 2 * D + D]) : 0.f;
    }
    for (int tb = t0; tb < t0 + 64 && tb < Tp; tb += 8) {
        float xn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = tb + u;
            xn[u] = t < Tv ? (float)ps[(size_t)t * 2 * D] * sigm((float)ps[(size_t)t * 2 * D + D]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            win[K - 1] = xn[u];
            float y = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) y = __builtin_fmaf(wk[j], win[j], y);
#pragma unroll
            for (int j = 0; j < K - 1; ++j) win[j] = win[j + 1];
            if (tb + u < Tp) os[(size_t)(tb + u) * D] = tb + u < Tv ? to_f16_sat(y) : (_Float16)0.f;
        }
    }
}

// Column statistics of c16 over the valid frames (t < Tv of every sequence): shift == nullptr -> sums; else
// sums of (c - shift)^2.  One thread per channel, a block per contiguous range of valid rows; partial [gridDim.x][256].
__global__ __launch_bounds__(256)
void bn_colstats16_kernel(const _Float16* __restrict__ c16, const float* __restrict__ shift, float* __restrict__ partial, int nseq,
                          int Tp, int Tv, long rows_per_block) {
    const int ch = threadIdx.x;
    const long nrows = (long)nseq * Tv;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    r1 = r1 < nrows ? r1 : nrows;
    const float mu = shift ? shift[ch] : 0.f;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    long seq = r0 / Tv;
    int t = (int)(r0 - seq * Tv);
    for (long r = r0; r < r1; r += 4) {                       // four rows in flight, (sequence, frame) advanced incrementally (round 6)
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = shift ? mu : 0.f;                          // (rows beyond the range contribute nothing)
            if (r + k < r1) v[k] = (float)c16[((size_t)seq * Tp + t) * D + ch];
            if (++t == Tv) { t = 0; ++seq; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += shift ? (v[k] - mu) * (v[k] - mu) : v[k];
    }
    partial[(size_t)blockIdx.x * D + ch] = (a[0] + a[1]) + (a[2] + a[3]);
}

// Merge the (mean, M2, n) triples of R ranks (R = 1: this rank alone) -- Chan's parallel variance -- into the batch
// statistics the normalisation uses, and update the running statistics like torch.nn.BatchNorm1d / SyncBatchNorm in
// train mode (momentum, unbiased variance).  stats: [R][2*256 + 1] = mean[256], M2[256], n.
__global__ __launch_bounds__(256)
void bn_merge_kernel(const float* __restrict__ stats, int R, float* __restrict__ mean_out, float* __restrict__ var_out,
                     float* __restrict__ n_out, float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
    const int ch = threadIdx.x;
    const int stride = 2 * D + 1;
    double n = 0.0, mean = 0.0;
    for (int r = 0; r < R; ++r) { const double nr = stats[(size_t)r * stride + 2 * D]; n += nr; mean += nr * stats[(size_t)r * stride + ch]; }
    mean /= n;
    double m2 = 0.0;
    for (int r = 0; r < R; ++r) {
        const double nr = stats[(size_t)r * stride + 2 * D], d = stats[(size_t)r * stride + ch] - mean;
        m2 += stats[(size_t)r * stride + D + ch] + nr * d * d;
    }
    mean_out[ch] = (float)mean;
    var_out[ch] = (float)(m2 / n);
    if (ch == 0 && n_out) n_out[0] = (float)n;
    if (run_mean) {
        run_mean[ch] = (1.0f - momentum) * run_mean[ch] + momentum * (float)mean;
        run_var[ch] = (1.0f - momentum) * run_var[ch] + momentum * (float)(m2 / (n - 1.0));
    }
}

// local (sum -> mean) finalisation between the two statistics passes: stats[0..255] = sum / n; stats[512] = n
__global__ __launch_bounds__(256)
void bn_local_mean_kernel(const float* __restrict__ sum, float n, float* __restrict__ stats) {
    stats[threadIdx.x] = sum[threadIdx.x] / n;
    if (threadIdx.x == 0) stats[2 * D] = n;
}

// s = swish(gamma * (c - mean) * rstd + beta), f16; 8 elements per thread.
__global__ __launch_bounds__(256)
void bn_swish_fwd_kernel(const unsigned short* __restrict__ c16, const float* __restrict__ mean, const float* __restrict__ var, float eps,
                         const float* __restrict__ gamma, const float* __restrict__ beta, unsigned short* __restrict__ s16, long M) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * (D / 8)) return;
    const long row = idx / (D / 8);
    const int col = (int)(idx - row * (D / 8)) * 8;
    const u32x4 v = *(const u32x4*)(c16 + row * D + col);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned x = v[j];
        const int c0 = col + 2 * j, c1 = c0 + 1;
        const float sc0 = gamma[c0] / __builtin_sqrtf(var[c0] + eps), sc1 = gamma[c1] / __builtin_sqrtf(var[c1] + eps);
        const float y0 = (f16_lo(x) - mean[c0]) * sc0 + beta[c0], y1 = (f16_hi(x) - mean[c1]) * sc1 + beta[c1];
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 p;
        p[0] = to_f16_sat(y0 * sigm(y0));
        p[1] = to_f16_sat(y1 * sigm(y1));
        o[j] = __builtin_bit_cast(unsigned, p);
    }
    *(u32x4*)(s16 + row * D + col) = o;
}

// BatchNorm + Swish backward, pass 1: per channel S1 = sum d_y, S2 = sum d_y * c_hat over the valid frames, with
// d_y = d_s * swish'(gamma * c_hat + beta).  partial [gridDim.x][2][256].
__global__ __launch_bounds__(256)
void bn_swish_bwd_stats_kernel(const __bf16* __restrict__ ds, const _Float16* __restrict__ c16, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                               float* __restrict__ partial, int nseq, int Tp, int Tv, long rows_per_block) {
    const int ch = threadIdx.x;
    const long nrows = (long)nseq * Tv;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    r1 = r1 < nrows ? r1 : nrows;
    const float mu = mean[ch], rs = 1.0f / __builtin_sqrtf(var[ch] + eps), g = gamma[ch], b = beta[ch];
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    long seq = r0 / Tv;
    int t = (int)(r0 - seq * Tv);
    for (long r = r0; r < r1; r += 4) {                       // four rows in flight (see bn_colstats16_kernel)
        float cv[4], dv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cv[k] = mu; dv[k] = 0.f;
            if (r + k < r1) {
                const size_t off = ((size_t)seq * Tp + t) * D + ch;
                cv[k] = (float)c16[off];
                dv[k] = (float)ds[off];
            }
            if (++t == Tv) { t = 0; ++seq; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ch_ = (cv[k] - mu) * rs;
            const float dy = dv[k] * swish_grad(g * ch_ + b);
            a1[k] += dy;
            a2[k] += dy * ch_;
        }
    }
    partial[((size_t)blockIdx.x * 2) * D + ch] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    partial[((size_t)blockIdx.x * 2 + 1) * D + ch] = (a2[0] + a2[1]) + (a2[2] + a2[3]);
}

// pass 2: d_c = gamma * rstd * (d_y - S1/n - c_hat * S2/n) for the valid frames, zero elsewhere; in place over ds (bf16).
// sums: [2][256] (global over all ranks under SyncBatchNorm), n_dev: the matching frame count.
__global__ __launch_bounds__(256)
void bn_swish_bwd_apply_kernel(__bf16* __restrict__ ds, const _Float16* __restrict__ c16, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ sums, const float* __restrict__ n_dev, int Tp, int Tv) {
    const int seq = blockIdx.y, t0 = blockIdx.x * 16, ch = threadIdx.x;
    const float inv_n = 1.0f / n_dev[0];
    const float mu = mean[ch], rs = 1.0f / __builtin_sqrtf(var[ch] + eps), g = gamma[ch], b = beta[ch];
    const float m1 = sums[ch] * inv_n, m2 = sums[D + ch] * inv_n;
#pragma unroll 4
    for (int t = t0; t < t0 + 16 && t < Tp; ++t) {
        const size_t off = ((size_t)seq * Tp + t) * D + ch;
        float out = 0.f;
        if (t < Tv) {
            const float ch_ = ((float)c16[off] - mu) * rs;
            const float dy = (float)ds[off] * swish_grad(g * ch_ + b);
            out = g * rs * (dy - m1 - ch_ * m2);
        }
        ds[off] = (__bf16)out;
    }
}

// Depthwise conv + GLU backward.  d_u[t] = sum_m w[K-1-m] d_c[t+m];  d_value = d_u * sigmoid(gate),
// d_gate = d_u * value * sigmoid(gate) * (1 - sigmoid(gate));  dw[ch][K-1-m] += u[t] * d_c[t+m].
// dP bf16 [nseq*Tp][512]; dw partial [gridDim.y * gridDim.x][256][K].  One thread per channel, 64-frame strips.
template <int K>
__global__ __launch_bounds__(256)
void dwconv_glu_bwd_kernel(const __bf16* __restrict__ dc, const _Float16* __restrict__ P, const float* __restrict__ w,
                           __bf16* __restrict__ dP, float* __restrict__ partial, int Tp, int Tv) {
    const int seq = blockIdx.y, t0 = blockIdx.x * 64, ch = threadIdx.x;
    const __bf16* ds = dc + (size_t)seq * Tp * D + ch;
    const _Float16* ps = P + (size_t)seq * Tp * 2 * D + ch;
    __bf16* os = dP + (size_t)seq * Tp * 2 * D + ch;
    float wk[K], win[K], dw[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { wk[j] = w[(size_t)ch * K + j]; dw[j] = 0.f; }
#pragma unroll
    for (int m = 0; m < K - 1; ++m) {
        const int t = t0 + m;
        win[m + 1] = t < Tv ? (float)ds[(size_t)t * D] : 0.f;           // win[m+1] = d_c[t0 + m]; shifted down at each step
    }
    for (int t = t0; t < t0 + 64 && t < Tp; ++t) {
#pragma unroll
        for (int m = 0; m < K - 1; ++m) win[m] = win[m + 1];            // win[m] = d_c[t + m], m < K-1
        win[K - 1] = t + K - 1 < Tv ? (float)ds[(size_t)(t + K - 1) * D] : 0.f;
        float du = 0.f, u = 0.f, a = 0.f, s = 0.f;
        if (t < Tv) {
            a = (float)ps[(size_t)t * 2 * D];
            s = sigm((float)ps[(size_t)t * 2 * D + D]);
            u = a * s;
#pragma unroll
            for (int m = 0; m < K; ++m) {
                du = __builtin_fmaf(wk[K - 1 - m], win[m], du);
                dw[K - 1 - m] = __builtin_fmaf(u, win[m], dw[K - 1 - m]);
            }
        }
        os[(size_t)t * 2 * D] = (__bf16)(du * s);
        os[(size_t)t * 2 * D + D] = (__bf16)(du * a * s * (1.0f - s));
    }
    float* pp = partial + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * D + ch) * K;
#pragma unroll
    for (int j = 0; j < K; ++j) pp[j] = dw[j];
}

// ---------------------------------------------------------------------------------------------------------------
// Retention gate + per-head LayerNorm backward (retention.py:222-224: out = swish(g) * LN_head(r), eps 1e-6, no affine):
//   d_rhat = d_out * swish(g);  d_g = d_out * rhat * swish'(g);
//   d_r = rstd * (d_rhat - mean_head(d_rhat) - rhat * mean_head(d_rhat * rhat));  o~ = c_t * d_r  (rc = rstd * c_t)
// dctx f32 [M][256] (the out-projection's data gradient, kept in f32: the LayerNorm backward below subtracts its two
// dominant components, so rounding it to bf16 first would be amplified); g16 f16 [M][ldg]; rhat16 f16 [M][256];
// rc f32 [M][4]; dg bf16 [M][ldq] (caller offsets the column); ot bf16 [M][256].  Rows t >= Tv: zeros.  One wave per
// row, 16 lanes per head.
__global__ __launch_bounds__(256)
void ret_gate_gn_bwd_kernel(const float* __restrict__ dctx, const _Float16* __restrict__ g16, int ldg, const _Float16* __restrict__ rhat16,
                            const float* __restrict__ rc, __bf16* __restrict__ dg, int ldq, __bf16* __restrict__ ot, long M, int Tp,
                            int Tv) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int t = (int)(row % Tp);
    uint2 pg = make_uint2(0u, 0u), po = make_uint2(0u, 0u);
    if (t < Tv) {
        const float4 dk = *(const float4*)(dctx + row * D + lane * 4);
        const float d[4] = {dk.x, dk.y, dk.z, dk.w};
        const f16x4 gv = *(const f16x4*)(g16 + row * ldg + lane * 4);
        const f16x4 rv = *(const f16x4*)(rhat16 + row * D + lane * 4);
        float dr[4], dgv[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = (float)gv[e], rh = (float)rv[e], sg = sigm(gg);
            dgv[e] = d[e] * rh * sg * (1.0f + gg * (1.0f - sg));
            dr[e] = d[e] * gg * sg;
            s1 += dr[e];
            s2 += dr[e] * rh;
        }
        s1 = row16_allreduce_add(s1) * (1.0f / 64.0f);
        s2 = row16_allreduce_add(s2) * (1.0f / 64.0f);
        const float k = rc[row * 4 + (lane >> 4)];
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = k * (dr[e] - s1 - (float)rv[e] * s2);
        pg.x = pack_bf16(dgv[0], dgv[1]); pg.y = pack_bf16(dgv[2], dgv[3]);
        po.x = pack_bf16(o[0], o[1]); po.y = pack_bf16(o[2], o[3]);
    }
    *(uint2*)(dg + row * ldq + lane * 4) = pg;
    *(uint2*)(ot + row * D + lane * 4) = po;
}

int grid_rows(long M) {
    long nb = (M + 3) / 4;
    return (int)(nb < 1024 ? nb : 1024);
}

}  // namespace

int eend_launch_swish_drop_fwd(const void* z16, void* a16, long M, int F, DropSpec drop, hipStream_t stream) {
    if (!z16 || !a16 || M <= 0 || F <= 0 || (F & 7)) return EEND_EINVAL;
    const long n = M * (F >> 3);
    hipLaunchKernelGGL(swish_drop_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const unsigned short*)z16,
                       (unsigned short*)a16, M, F, drop);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_swish_bwd(void* dz16, const void* z16, long M, int F, DropSpec drop, hipStream_t stream) {
    if (!z16 || !dz16 || M <= 0 || F <= 0 || (F & 7)) return EEND_EINVAL;
    const long n = M * (F >> 3);
    hipLaunchKernelGGL(swish_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (unsigned short*)dz16,
                       (const unsigned short*)z16, M, F, drop);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_layernorm_train(const float* x, const float* gamma, const float* beta, float eps, void* y16, void* xhat16, float* rstd,
                                long M, hipStream_t stream) {
    if (!x || !gamma || !beta || !y16 || !xhat16 || !rstd || M <= 0) return EEND_EINVAL;
    hipLaunchKernelGGL(layernorm_train_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, x, gamma, beta, eps, (_Float16*)y16,
                       (_Float16*)xhat16, rstd, M);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_ln_bwd2(const void* g, int g_is_bf16, const void* xhat16, const float* rstd, const float* gamma, float* ds32,
                        int accumulate, void* ds16, float alpha16, float* partial, int* nblocks_out, long M, DropSpec drop,
                        hipStream_t stream) {
    if (!g || !xhat16 || !rstd || !gamma || !partial || M <= 0 || (accumulate && !ds32)) return EEND_EINVAL;
    const int nb = grid_rows(M);
    *nblocks_out = nb;
#define LN2(G16, ACC) hipLaunchKernelGGL((ln_bwd2_kernel<G16, ACC>), dim3(nb), dim3(256), 0, stream, g, (const _Float16*)xhat16, rstd, gamma, \
                                         ds32, (__bf16*)ds16, alpha16, partial, M, drop)
    if (g_is_bf16) { if (accumulate) LN2(true, true); else LN2(true, false); }
    else { if (accumulate) LN2(false, true); else LN2(false, false); }
#undef LN2
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_resgrad_cast(const float* g, void* ds16, float alpha, float* partial, int* nblocks_out, long M, DropSpec drop,
                             hipStream_t stream) {
    if (!g || !ds16 || !partial || M <= 0) return EEND_EINVAL;
    const int nb = grid_rows(M);
    *nblocks_out = nb;
    hipLaunchKernelGGL(resgrad_cast_kernel, dim3(nb), dim3(256), 0, stream, g, (__bf16*)ds16, alpha, partial, M, drop);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_glu_dwconv_fwd(const void* P16, const float* w, void* c16, int nseq, int Tp, int Tv, int k, hipStream_t stream) {
    if (!P16 || !w || !c16 || nseq <= 0 || nseq > 65535 || Tp <= 0 || Tv <= 0 || Tv > Tp) return EEND_EINVAL;
    const dim3 grid((Tp + 63) / 64, nseq);
    switch (k) {
        case 16: hipLaunchKernelGGL(glu_dwconv_fwd_kernel<16>, grid, dim3(256), 0, stream, (const _Float16*)P16, w, (_Float16*)c16, Tp, Tv); break;
        case 7: hipLaunchKernelGGL(glu_dwconv_fwd_kernel<7>, grid, dim3(256), 0, stream, (const _Float16*)P16, w, (_Float16*)c16, Tp, Tv); break;
        case 15: hipLaunchKernelGGL(glu_dwconv_fwd_kernel<15>, grid, dim3(256), 0, stream, (const _Float16*)P16, w, (_Float16*)c16, Tp, Tv); break;
        case 31: hipLaunchKernelGGL(glu_dwconv_fwd_kernel<31>, grid, dim3(256), 0, stream, (const _Float16*)P16, w, (_Float16*)c16, Tp, Tv); break;
        default: return EEND_EINVAL;
    }
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_colstats16(const void* c16, const float* shift, float* partial, int nseq, int Tp, int Tv, int nblocks,
                              hipStream_t stream) {
    if (!c16 || !partial || nseq <= 0 || Tp <= 0 || Tv <= 0 || Tv > Tp || nblocks <= 0) return EEND_EINVAL;
    const long nrows = (long)nseq * Tv;
    const long rpb = (nrows + nblocks - 1) / nblocks;
    hipLaunchKernelGGL(bn_colstats16_kernel, dim3(nblocks), dim3(256), 0, stream, (const _Float16*)c16, shift, partial, nseq, Tp, Tv, rpb);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_local_mean(const float* sum, float n, float* stats, hipStream_t stream) {
    hipLaunchKernelGGL(bn_local_mean_kernel, dim3(1), dim3(256), 0, stream, sum, n, stats);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_merge(const float* stats, int R, float* mean, float* var, float* n_out, float* run_mean, float* run_var, float momentum,
                         hipStream_t stream) {
    if (!stats || R <= 0 || !mean || !var) return EEND_EINVAL;
    hipLaunchKernelGGL(bn_merge_kernel, dim3(1), dim3(256), 0, stream, stats, R, mean, var, n_out, run_mean, run_var, momentum);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_swish_fwd(const void* c16, const float* mean, const float* var, float eps, const float* gamma, const float* beta,
                             void* s16, long M, hipStream_t stream) {
    if (!c16 || !mean || !var || !gamma || !beta || !s16 || M <= 0) return EEND_EINVAL;
    const long n = M * (D / 8);
    hipLaunchKernelGGL(bn_swish_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const unsigned short*)c16, mean, var,
                       eps, gamma, beta, (unsigned short*)s16, M);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_swish_bwd_stats(const void* ds16, const void* c16, const float* mean, const float* var, float eps, const float* gamma,
                                   const float* beta, float* partial, int nseq, int Tp, int Tv, int nblocks, hipStream_t stream) {
    if (!ds16 || !c16 || !partial || nseq <= 0 || Tp <= 0 || Tv <= 0 || Tv > Tp || nblocks <= 0) return EEND_EINVAL;
    const long nrows = (long)nseq * Tv;
    const long rpb = (nrows + nblocks - 1) / nblocks;
    hipLaunchKernelGGL(bn_swish_bwd_stats_kernel, dim3(nblocks), dim3(256), 0, stream, (const __bf16*)ds16, (const _Float16*)c16, mean, var,
                       eps, gamma, beta, partial, nseq, Tp, Tv, rpb);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_swish_bwd_apply(void* ds16, const void* c16, const float* mean, const float* var, float eps, const float* gamma,
                                   const float* beta, const float* sums, const float* n_dev, int nseq, int Tp, int Tv, hipStream_t stream) {
    if (!ds16 || !c16 || !sums || !n_dev || nseq <= 0 || nseq > 65535 || Tp <= 0) return EEND_EINVAL;
    hipLaunchKernelGGL(bn_swish_bwd_apply_kernel, dim3((Tp + 15) / 16, nseq), dim3(256), 0, stream, (__bf16*)ds16, (const _Float16*)c16, mean,
                       var, eps, gamma, beta, sums, n_dev, Tp, Tv);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_dwconv_glu_bwd(const void* dc16, const void* P16, const float* w, void* dP16, float* partial, int nseq, int Tp, int Tv,
                               int k, hipStream_t stream) {
    if (!dc16 || !P16 || !w || !dP16 || !partial || nseq <= 0 || nseq > 65535 || Tp <= 0 || Tv <= 0 || Tv > Tp) return EEND_EINVAL;
    const dim3 grid((Tp + 63) / 64, nseq);
#define DWB(KK) hipLaunchKernelGGL(dwconv_glu_bwd_kernel<KK>, grid, dim3(256), 0, stream, (const __bf16*)dc16, (const _Float16*)P16, w, \
                                   (__bf16*)dP16, partial, Tp, Tv)
    switch (k) {
        case 16: DWB(16); break;
        case 7: DWB(7); break;
        case 15: DWB(15); break;
        case 31: DWB(31); break;
        default: return EEND_EINVAL;
    }
#undef DWB
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_ret_gate_gn_bwd(const float* dctx32, const void* g16, int ldg, const void* rhat16, const float* rc, void* dg16, int ldq,
                                void* ot16, int nseq, int Tp, int Tv, hipStream_t stream) {
    if (!dctx32 || !g16 || !rhat16 || !rc || !dg16 || !ot16 || nseq <= 0 || Tp <= 0 || Tv <= 0 || Tv > Tp || (ldg & 3) || (ldq & 3))
        return EEND_EINVAL;
    const long M = (long)nseq * Tp;
    hipLaunchKernelGGL(ret_gate_gn_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, dctx32, (const _Float16*)g16,
                       ldg, (const _Float16*)rhat16, rc, (__bf16*)dg16, ldq, (__bf16*)ot16, M, Tp, Tv);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
