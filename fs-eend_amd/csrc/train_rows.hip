// Row-local (HBM-bound) kernels of the training step: LayerNorm backward, the embedding.attractor head with its
// BCE loss and gradient, L2-norm backward, the speaker-slot reductions of `convert`, speaker-axis attention
// backward, train-mode BatchNorm statistics and its parameter gradients.  d_model = 256: one wave owns one row,
// a lane 4 consecutive features (16-byte accesses), reductions are wave shuffles.  Parameter-gradient sums are
// accumulated per wave over a persistent row loop, reduced per block through LDS and written as partials that
// wgrad_reduce_kernel (wgrad.hip) sums in a fixed order -- deterministic, no atomics.
#include "train_common.h"
#include "kernels.h"

namespace {

constexpr int D = 256;

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  y = x_hat * gamma + beta, x_hat = (s - mean) * rstd:
//   ds = rstd * (dy*gamma - mean_j(dy*gamma) - x_hat * mean_j(dy*gamma*x_hat));  dgamma = sum_rows dy*x_hat;  dbeta = sum_rows dy
// g: gradient w.r.t. the LayerNorm output (f32); ds32 may alias g.  partial: [gridDim.x][3][256]: dgamma, dbeta and the
// column sums of the (dropout-masked) branch gradient = the bias gradient of the linear in front of the LayerNorm.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void ln_bwd_kernel(const float* g, const _Float16* __restrict__ xhat, const float* __restrict__ rstd,
                   const float* __restrict__ gamma, float* ds32, __bf16* __restrict__ ds16, float* __restrict__ partial, long M,
                   const DropSpec drop) {
    __shared__ float red[4][3][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 gm = *(const float4*)(gamma + lane * 4);
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, dbr[4] = {0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        const float4 gy = *(const float4*)(g + row * D + lane * 4);
        const f16x4 xh = *(const f16x4*)(xhat + row * D + lane * 4);
        const float rs = rstd[row];
        const float x0 = (float)xh[0], x1 = (float)xh[1], x2 = (float)xh[2], x3 = (float)xh[3];
        const float d0 = gy.x * gm.x, d1 = gy.y * gm.y, d2 = gy.z * gm.z, d3 = gy.w * gm.w;
        const float c1 = wave_sum(d0 + d1 + d2 + d3) * (1.0f / D);
        const float c2 = wave_sum(d0 * x0 + d1 * x1 + d2 * x2 + d3 * x3) * (1.0f / D);
        const float o0 = rs * (d0 - c1 - x0 * c2), o1 = rs * (d1 - c1 - x1 * c2);
        const float o2 = rs * (d2 - c1 - x2 * c2), o3 = rs * (d3 - c1 - x3 * c2);
        if (ds32) *(float4*)(ds32 + row * D + lane * 4) = make_float4(o0, o1, o2, o3);
        if (ds16) {
            // ds16 feeds the sub-layer branch (the linear in front of this LayerNorm): the forward dropped its output
            // with the same (row, column) mask; the residual path (ds32) is not masked
            const unsigned n = (unsigned)(lane * 4);
            const float m0 = drop_apply(drop, o0, (unsigned)row, n), m1 = drop_apply(drop, o1, (unsigned)row, n + 1);
            const float m2 = drop_apply(drop, o2, (unsigned)row, n + 2), m3 = drop_apply(drop, o3, (unsigned)row, n + 3);
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

// ---------------------------------------------------------------------------------------------------------------
// Head forward + BCE loss + gradient, one wave per frame (b, t), t < Tp:
//   a_hat_c = a_c / |a_c|;  logit_c = <e, a_hat_c>                                       (FS model :43,:60)
//   loss   += w_b * BCEwithLogits(logit_c, label_c),  w_b = 1 / (ncols_b * n_frames)     (train/utils/loss.py:119-125)
//   dlogit  = w_b * (sigmoid(logit) - label)         for t < ilen_b, c < ncols_b, else 0
//   de     += dlogit * a_hat_c ;  da_c = dlogit * (e - a_hat_c * logit_c) / |a_c|
// emb e f32 [B][Tp][256] (unit rows), a f32 slab rows (b*C + c)*Tp + t.  Rows t >= T get zero gradients.
// ---------------------------------------------------------------------------------------------------------------
// With `dlogits_in` (f32 [B][T][C], the caller's d loss / d logits, e.g. from torch autograd over its own loss) the BCE
// part is skipped and that gradient is propagated instead.
__global__ __launch_bounds__(256)
void head_bce_kernel(const float* __restrict__ emb, const float* __restrict__ attr, const float* __restrict__ labels,
                     const int* __restrict__ ilens, const int* __restrict__ ncols, float inv_frames,
                     const float* __restrict__ dlogits_in, float* __restrict__ logits, float* __restrict__ da,
                     float* __restrict__ de, float* __restrict__ loss_partial, int B, int T, int Tp, int C) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long frame = (long)blockIdx.x * 4 + wave;
    float loss = 0.f;
    if (frame < (long)B * Tp) {
        const int b = (int)(frame / Tp), t = (int)(frame - (long)b * Tp);
        float4 dev = make_float4(0, 0, 0, 0);
        if (t >= T) {
            for (int c = 0; c < C; ++c) *(float4*)(da + (((size_t)b * C + c) * Tp + t) * D + lane * 4) = dev;
        } else {
            const float4 e = *(const float4*)(emb + frame * D + lane * 4);
            const int il = dlogits_in ? T : ilens[b], nc = dlogits_in ? C : ncols[b];
            const float w = inv_frames / (float)nc;
            for (int c = 0; c < C; ++c) {
                const size_t row = ((size_t)b * C + c) * Tp + t;
                const float4 a = *(const float4*)(attr + row * D + lane * 4);
                const float ss = wave_sum(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
                const float dot = wave_sum(a.x * e.x + a.y * e.y + a.z * e.z + a.w * e.w);
                const float inv = 1.0f / __builtin_sqrtf(ss);
                const float y = dot * inv;
                if (logits && lane == 0) logits[((size_t)b * T + t) * C + c] = y;
                float dl = 0.f;
                if (dlogits_in) {
                    dl = dlogits_in[((size_t)b * T + t) * C + c];
                } else if (t < il && c < nc) {
                    const float lab = labels[((size_t)b * T + t) * C + c];
                    const float ay = __builtin_fabsf(y);
                    const float sp = log1pf(__expf(-ay));          // log(1 + exp(-|y|))
                    loss += w * (__builtin_fmaxf(y, 0.f) - y * lab + sp);
                    const float sg = 1.0f / (1.0f + __expf(-y));
                    dl = w * (sg - lab);
                }
                const float k1 = dl * inv, k2 = dl * inv * y * inv;          // da = dl*inv*e - dl*y*inv*inv*a
                *(float4*)(da + row * D + lane * 4) =
                    make_float4(k1 * e.x - k2 * a.x, k1 * e.y - k2 * a.y, k1 * e.z - k2 * a.z, k1 * e.w - k2 * a.w);
                dev.x += k1 * a.x; dev.y += k1 * a.y; dev.z += k1 * a.z; dev.w += k1 * a.w;
            }
        }
        *(float4*)(de + frame * D + lane * 4) = dev;
    }
    if (lane == 0) red[wave] = loss;           // identical in every lane (all terms come from wave sums)
    __syncthreads();
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// y = x / |x| backward: dx = (dy - y <y, dy>) * inv_norm -> bf16.  y f32 unit rows, dy f32, rows t >= T zero.
__global__ __launch_bounds__(256)
void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, const float* __restrict__ inv_norm,
                       __bf16* __restrict__ dx16, int B, int T, int Tp) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * Tp) return;
    const int t = (int)(row % Tp);
    uint2 pk = make_uint2(0u, 0u);
    if (t < T) {
        const float4 yv = *(const float4*)(y + row * D + lane * 4);
        const float4 g = *(const float4*)(dy + row * D + lane * 4);
        const float dot = wave_sum(yv.x * g.x + yv.y * g.y + yv.z * g.z + yv.w * g.w);
        const float inv = inv_norm[row];
        pk.x = pack_bf16((g.x - yv.x * dot) * inv, (g.y - yv.y * dot) * inv);
        pk.y = pack_bf16((g.z - yv.z * dot) * inv, (g.w - yv.w * dot) * inv);
    }
    *(uint2*)(dx16 + row * D + lane * 4) = pk;
}

// `convert` backward reductions (FS model :113-114, factored form attr0[(b,c),t] = W1 e[b,t] + pc[c]):
//   gsum[b,t,:] = sum_c g0[(b,c),t,:]  (bf16: operand of the W1 data / weight gradient GEMMs)
//   dpc[c,:]    = sum_{b,t} g0[(b,c),t,:]  -> partial[gridDim.x][C][256]
template <int C>
__global__ __launch_bounds__(256)
void slot_sum_kernel(const float* __restrict__ g0, __bf16* __restrict__ gsum, float* __restrict__ partial, int B, int Tp) {
    __shared__ float red[4][C][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c) { acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f; }
    const long nframes = (long)B * Tp;
    for (long frame = (long)blockIdx.x * 4 + wave; frame < nframes; frame += (long)gridDim.x * 4) {
        const int b = (int)(frame / Tp), t = (int)(frame - (long)b * Tp);
        float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 v = *(const float4*)(g0 + (((size_t)b * C + c) * Tp + t) * D + lane * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            acc[c][0] += v.x; acc[c][1] += v.y; acc[c][2] += v.z; acc[c][3] += v.w;
        }
        uint2 pk;
        pk.x = pack_bf16(s.x, s.y);
        pk.y = pack_bf16(s.z, s.w);
        *(uint2*)(gsum + frame * D + lane * 4) = pk;
    }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][c][lane * 4 + e] = acc[c][e];
    __syncthreads();
    for (int c = 0; c < C; ++c)
        partial[((size_t)blockIdx.x * C + c) * D + threadIdx.x] =
            (red[0][c][threadIdx.x] + red[1][c][threadIdx.x]) + (red[2][c][threadIdx.x] + red[3][c][threadIdx.x]);
}

// convert.weight[:, D:] / convert.bias gradients and the forward constant, all (C, 256)-sized:
//   mode 0: pc[c][n] = sum_k W[n][D + k] pe[c][k] + bias[n]                         (forward constant)
//   mode 1: dW[n][D + k] = sum_c dpc[c][n] pe[c][k];  dbias[n] = sum_c dpc[c][n]    (gradients)
// W / dW are the full (256, 512) convert weight (row stride 512).  One block per output row n.
__global__ __launch_bounds__(256)
void convert_const_kernel(int mode, const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ pe,
                          float* __restrict__ pc, const float* __restrict__ dpc, float* __restrict__ dW,
                          float* __restrict__ dbias, int C) {
    const int n = blockIdx.x, k = threadIdx.x;
    if (mode == 0) {
        __shared__ float red[4];
        const float w = W[(size_t)n * 2 * D + D + k];
        for (int c = 0; c < C; ++c) {
            float v = wave_sum(w * pe[(size_t)c * D + k]);
            if ((k & 63) == 0) red[k >> 6] = v;
            __syncthreads();
            if (k == 0) pc[(size_t)c * D + n] = (red[0] + red[1]) + (red[2] + red[3]) + bias[n];
            __syncthreads();
        }
    } else {
        float s = 0.f, sb = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = dpc[(size_t)c * D + n];
            s += d * pe[(size_t)c * D + k];
            sb += d;
        }
        dW[(size_t)n * 2 * D + D + k] = s;
        if (k == 0) dbias[n] = sb;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Speaker-axis attention backward (_sa_block2, merge_tfm_encoder.py:388-394): one wave per frame, lane = (head,
// 4-wide d slice) as in the forward kernel (attn.hip), every slot's q, k, v, dO in registers, probabilities
// recomputed.  qkv f16 [rows][768]; dO bf16 [rows][256]; dqkv bf16 [rows][768]; row = (b*C + c)*Tp + t.
// ---------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256)
void spk_attn_bwd_kernel(const _Float16* __restrict__ qkv, const __bf16* __restrict__ dO, __bf16* __restrict__ dqkv,
                         int B, int Tp, float scale, const DropSpec drop) {
    const int lane = threadIdx.x & 63;
    const long frame = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= (long)B * Tp) return;
    const int b = (int)(frame / Tp), t = (int)(frame - (long)b * Tp);
    const int col = lane * 4;
    float q[C][4], k[C][4], v[C][4], go[C][4], dk[C][4], dv[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t row = ((size_t)b * C + c) * Tp + t;
        const f16x4 q4 = *(const f16x4*)(qkv + row * 3 * D + col);
        const f16x4 k4 = *(const f16x4*)(qkv + row * 3 * D + D + col);
        const f16x4 v4 = *(const f16x4*)(qkv + row * 3 * D + 2 * D + col);
        const uint2 g2 = *(const uint2*)(dO + row * D + col);
        go[c][0] = bf16_lo(g2.x); go[c][1] = bf16_hi(g2.x); go[c][2] = bf16_lo(g2.y); go[c][3] = bf16_hi(g2.y);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            q[c][e] = (float)q4[e]; k[c][e] = (float)k4[e]; v[c][e] = (float)v4[e];
            dk[c][e] = 0.f; dv[c][e] = 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float s[C], dp[C];
        float mx = -INFINITY;
#pragma unroll
        for (int c2 = 0; c2 < C; ++c2) {
            float d = q[c][0] * k[c2][0];
            d = __builtin_fmaf(q[c][1], k[c2][1], d);
            d = __builtin_fmaf(q[c][2], k[c2][2], d);
            d = __builtin_fmaf(q[c][3], k[c2][3], d);
            s[c2] = row16_allreduce_add(d) * scale;
            mx = __builtin_fmaxf(mx, s[c2]);
            float e = go[c][0] * v[c2][0];
            e = __builtin_fmaf(go[c][1], v[c2][1], e);
            e = __builtin_fmaf(go[c][2], v[c2][2], e);
            e = __builtin_fmaf(go[c][3], v[c2][3], e);
            dp[c2] = row16_allreduce_add(e);
        }
        float den = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < C; ++c2) { s[c2] = __expf(s[c2] - mx); den += s[c2]; }
        const float inv = 1.0f / den;
        float dsum = 0.f;
        float pd[C];                                               // dropped probabilities (what multiplied V in the forward)
        const unsigned da = ((unsigned)frame * 4u + (unsigned)(lane >> 4)) * 16u + (unsigned)c;
#pragma unroll
        for (int c2 = 0; c2 < C; ++c2) {
            s[c2] *= inv;
            const float kf = drop.thresh24 == 0 ? 1.0f : (drop_keep(drop, da, (unsigned)c2) ? drop.scale : 0.f);
            pd[c2] = s[c2] * kf;
            dp[c2] *= kf;
            dsum = __builtin_fmaf(s[c2], dp[c2], dsum);
        }
        float dq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c2 = 0; c2 < C; ++c2) {
            const float ds = s[c2] * (dp[c2] - dsum) * scale;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dq[e] = __builtin_fmaf(ds, k[c2][e], dq[e]);
                dk[c2][e] = __builtin_fmaf(ds, q[c][e], dk[c2][e]);
                dv[c2][e] = __builtin_fmaf(pd[c2], go[c][e], dv[c2][e]);
            }
        }
        const size_t row = ((size_t)b * C + c) * Tp + t;
        uint2 pk;
        pk.x = pack_bf16(dq[0], dq[1]);
        pk.y = pack_bf16(dq[2], dq[3]);
        *(uint2*)(dqkv + row * 3 * D + col) = pk;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t row = ((size_t)b * C + c) * Tp + t;
        uint2 pk;
        pk.x = pack_bf16(dk[c][0], dk[c][1]);
        pk.y = pack_bf16(dk[c][2], dk[c][3]);
        *(uint2*)(dqkv + row * 3 * D + D + col) = pk;
        pk.x = pack_bf16(dv[c][0], dv[c][1]);
        pk.y = pack_bf16(dv[c][2], dv[c][3]);
        *(uint2*)(dqkv + row * 3 * D + 2 * D + col) = pk;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Train-mode BatchNorm1d over the padded (B, T, F) input (FS model :165-166: pad_sequence(-1) then self.bn).
// Pass A: partial[s][0][c] = sum (x - shift_c), partial[s][1][c] = sum (x - shift_c)^2 over the split's rows.
// Two passes (shift = 0, then shift = mean) give a cancellation-free variance.  Thread = feature column.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void bn_colstats_kernel(const float* const* __restrict__ x_ptrs, const int* __restrict__ lens, float pad_value,
                        const float* __restrict__ shift, float* __restrict__ partial, int B, int T, int F, long rows_per_split) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= F) return;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > (long)B * T) r1 = (long)B * T;
    const float sh = shift ? shift[c] : 0.f;
    // sixteen rows in flight per iteration, (utterance, frame) advanced incrementally, summed IN ROW ORDER (round 6: the one-row loop with a
    // division, a length and a pointer load per row was latency-bound -- 123 us per pass over 44 MB; same sums bit for bit)
    float s1 = 0.f, s2 = 0.f;
    int b = (int)(r0 / T), t = (int)(r0 - (long)b * T);
    for (long r = r0; r < r1; r += 16) {
        float x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            // branch-free: a row beyond the split or the utterance loads from a harmless address (this thread's partial slot) and is replaced
            const bool in = r + k < r1;
            const int bb = in ? b : 0;
            const bool real = in && t < lens[bb];
            const float* src = real ? x_ptrs[bb] + (size_t)t * F + c : partial + c;
            const float v = *src;
            x[k] = real ? v : pad_value;
            if (++t == T) { t = 0; ++b; }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (r + k < r1) { const float d = x[k] - sh; s1 += d; s2 = __builtin_fmaf(d, d, s2); }
        }
    }
    partial[((size_t)blockIdx.y * 2 + 0) * F + c] = s1;
    partial[((size_t)blockIdx.y * 2 + 1) * F + c] = s2;
}

// pass 0: mean = sums[0] / n.   pass 1 (sums taken about `mean`): mean += d, var = sums[1]/n - d^2 with d = sums[0]/n,
// then the running statistics: rm = (1-mom) rm + mom mean; rv = (1-mom) rv + mom var n/(n-1)   (torch BatchNorm1d).
__global__ __launch_bounds__(256)
void bn_finalize_kernel(int pass, const float* __restrict__ sums, float n, float* __restrict__ mean, float* __restrict__ var,
                        float* __restrict__ run_mean, float* __restrict__ run_var, float momentum, int F) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= F) return;
    if (pass == 0) {
        mean[c] = sums[c] / n;
    } else {
        const float d = sums[c] / n;
        const float m = mean[c] + d;
        const float v = sums[F + c] / n - d * d;
        mean[c] = m;
        var[c] = v;
        if (run_mean) {
            run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * m;
            run_var[c] = (1.0f - momentum) * run_var[c] + momentum * v * (n / (n - 1.0f));
        }
    }
}

// BatchNorm parameter gradients: dgamma_c = sum dy x_hat, dbeta_c = sum dy over the B*T padded frames;
// dy bf16 [B*Tp][ld] (slab rows), x through the pointer table.  partial[s][2][F].
__global__ __launch_bounds__(256)
void bn_bwd_kernel(const float* const* __restrict__ x_ptrs, const int* __restrict__ lens, float pad_value,
                   const float* __restrict__ mean, const float* __restrict__ var, float eps, const __bf16* __restrict__ dy, int ld,
                   float* __restrict__ partial, int B, int T, int Tp, int F, long rows_per_split) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= F) return;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > (long)B * T) r1 = (long)B * T;
    const float mu = mean[c], rs = 1.0f / __builtin_sqrtf(var[c] + eps);
    float s1 = 0.f, s2 = 0.f;
    int b = (int)(r0 / T), t = (int)(r0 - (long)b * T);
    for (long r = r0; r < r1; r += 16) {                      // sixteen rows in flight, summed in row order (see bn_colstats_kernel)
        float x[16], g[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool in = r + k < r1;                       // branch-free, as in bn_colstats_kernel
            const int bb = in ? b : 0, tt = in ? t : 0;
            const bool real = in && t < lens[bb];
            const float* src = real ? x_ptrs[bb] + (size_t)t * F + c : mean + c;
            const float v = *src;
            x[k] = real ? v : pad_value;
            g[k] = (float)dy[((size_t)bb * Tp + tt) * ld + c];
            if (++t == T) { t = 0; ++b; }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (r + k < r1) { s1 = __builtin_fmaf(g[k], (x[k] - mu) * rs, s1); s2 += g[k]; }
        }
    }
    partial[((size_t)blockIdx.y * 2 + 0) * F + c] = s1;
    partial[((size_t)blockIdx.y * 2 + 1) * F + c] = s2;
}

// D_i = sum_d dO[i][h*64 + d] * O[i][h*64 + d] per (row, head): the softmax-backward row constant of flash attention.
// dO bf16 [nseq*Tp][256], O f16 [nseq*Tp][256] -> Dh f32 [nseq][4][Tp].  One wave per row, 16 lanes per head.
__global__ __launch_bounds__(256)
void attn_rowdot_kernel(const __bf16* __restrict__ dO, const _Float16* __restrict__ O, float* __restrict__ Dh, int nseq, int Tp) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)nseq * Tp) return;
    const uint2 g2 = *(const uint2*)(dO + row * D + lane * 4);
    const f16x4 o4 = *(const f16x4*)(O + row * D + lane * 4);
    float d = bf16_lo(g2.x) * (float)o4[0] + bf16_hi(g2.x) * (float)o4[1] + bf16_lo(g2.y) * (float)o4[2] + bf16_hi(g2.y) * (float)o4[3];
    d = row16_allreduce_add(d);
    const long seq = row / Tp;
    if ((lane & 15) == 0) Dh[(seq * 4 + (lane >> 4)) * Tp + (row - seq * Tp)] = d;
}

}  // namespace

static int persistent_blocks(long rows) {
    long nb = (rows + 3) / 4;
    return (int)(nb < 1024 ? nb : 1024);
}

int eend_launch_ln_bwd(const float* g, const void* xhat16, const float* rstd, const float* gamma, float* ds32, void* ds16,
                       float* partial, int* nblocks_out, long M, DropSpec drop, hipStream_t stream) {
    if (!g || !xhat16 || !rstd || !gamma || !partial || M <= 0) return EEND_EINVAL;
    const int nb = persistent_blocks(M);
    if (nblocks_out) *nblocks_out = nb;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(256), 0, stream, g, (const _Float16*)xhat16, rstd, gamma, ds32, (__bf16*)ds16, partial, M, drop);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_head_bce(const float* emb, const float* attr, const float* labels, const int* ilens, const int* ncols,
                         float inv_frames, const float* dlogits_in, float* logits, float* da, float* de, float* loss_partial, int B,
                         int T, int Tp, int C, hipStream_t stream) {
    if (!emb || !attr || !da || !de || !loss_partial || B <= 0 || T <= 0 || Tp < T || C <= 0) return EEND_EINVAL;
    if (!dlogits_in && (!labels || !ilens || !ncols)) return EEND_EINVAL;
    const long nb = ((long)B * Tp + 3) / 4;
    hipLaunchKernelGGL(head_bce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, emb, attr, labels, ilens, ncols, inv_frames, dlogits_in,
                       logits, da, de, loss_partial, B, T, Tp, C);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_l2norm_bwd(const float* y, const float* dy, const float* inv_norm, void* dx16, int B, int T, int Tp, hipStream_t stream) {
    if (!y || !dy || !inv_norm || !dx16 || B <= 0 || T <= 0 || Tp < T) return EEND_EINVAL;
    const long nb = ((long)B * Tp + 3) / 4;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, y, dy, inv_norm, (__bf16*)dx16, B, T, Tp);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_slot_sum(const float* g0, void* gsum16, float* partial, int* nblocks_out, int B, int Tp, int C, hipStream_t stream) {
    if (!g0 || !gsum16 || !partial || B <= 0 || Tp <= 0) return EEND_EINVAL;
    long nbl = ((long)B * Tp + 3) / 4;
    const int nb = (int)(nbl < 256 ? nbl : 256);
    if (nblocks_out) *nblocks_out = nb;
    switch (C) {
#define SS_CASE(n) case n: hipLaunchKernelGGL(slot_sum_kernel<n>, dim3(nb), dim3(256), 0, stream, g0, (__bf16*)gsum16, partial, B, Tp); break;
        SS_CASE(1) SS_CASE(2) SS_CASE(3) SS_CASE(4) SS_CASE(5) SS_CASE(6) SS_CASE(7) SS_CASE(8) SS_CASE(9) SS_CASE(10) SS_CASE(11) SS_CASE(12)
#undef SS_CASE
        default: return EEND_EINVAL;
    }
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_convert_const(int mode, const float* W, const float* bias, const float* pe, float* pc, const float* dpc, float* dW,
                              float* dbias, int C, hipStream_t stream) {
    if (!pe || C <= 0 || (mode == 0 && (!W || !bias || !pc)) || (mode == 1 && (!dpc || !dW || !dbias)) || mode < 0 || mode > 1)
        return EEND_EINVAL;
    hipLaunchKernelGGL(convert_const_kernel, dim3(D), dim3(256), 0, stream, mode, W, bias, pe, pc, dpc, dW, dbias, C);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_spk_attn_bwd(const void* qkv16, const void* dO16, void* dqkv16, int B, int C, int Tp, float scale, DropSpec drop,
                             hipStream_t stream) {
    if (!qkv16 || !dO16 || !dqkv16 || B <= 0 || Tp <= 0) return EEND_EINVAL;
    const long nb = ((long)B * Tp + 3) / 4;
    switch (C) {
#define SB_CASE(n) case n: hipLaunchKernelGGL(spk_attn_bwd_kernel<n>, dim3((unsigned)nb), dim3(256), 0, stream, (const _Float16*)qkv16, (const __bf16*)dO16, (__bf16*)dqkv16, B, Tp, scale, drop); break;
        SB_CASE(1) SB_CASE(2) SB_CASE(3) SB_CASE(4) SB_CASE(5) SB_CASE(6) SB_CASE(7) SB_CASE(8) SB_CASE(9) SB_CASE(10) SB_CASE(11) SB_CASE(12)
#undef SB_CASE
        default: return EEND_EINVAL;
    }
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_colstats(const float* const* x_ptrs, const int* lens, float pad_value, const float* shift, float* partial,
                            int B, int T, int F, int nsplit, hipStream_t stream) {
    if (!x_ptrs || !lens || !partial || B <= 0 || T <= 0 || F <= 0 || nsplit <= 0) return EEND_EINVAL;
    const long rps = ((long)B * T + nsplit - 1) / nsplit;
    hipLaunchKernelGGL(bn_colstats_kernel, dim3((F + 255) / 256, nsplit), dim3(256), 0, stream, x_ptrs, lens, pad_value, shift, partial,
                       B, T, F, rps);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_finalize(int pass, const float* sums, float n, float* mean, float* var, float* run_mean, float* run_var,
                            float momentum, int F, hipStream_t stream) {
    if (!sums || !mean || (pass == 1 && !var) || F <= 0 || n <= 1.0f) return EEND_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((F + 255) / 256), dim3(256), 0, stream, pass, sums, n, mean, var, run_mean, run_var,
                       momentum, F);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_bn_bwd(const float* const* x_ptrs, const int* lens, float pad_value, const float* mean, const float* var, float eps,
                       const void* dy16, int ld, float* partial, int B, int T, int Tp, int F, int nsplit, hipStream_t stream) {
    if (!x_ptrs || !lens || !mean || !var || !dy16 || !partial || B <= 0 || T <= 0 || Tp < T || F <= 0 || ld < F || nsplit <= 0)
        return EEND_EINVAL;
    const long rps = ((long)B * T + nsplit - 1) / nsplit;
    hipLaunchKernelGGL(bn_bwd_kernel, dim3((F + 255) / 256, nsplit), dim3(256), 0, stream, x_ptrs, lens, pad_value, mean, var, eps,
                       (const __bf16*)dy16, ld, partial, B, T, Tp, F, rps);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_attn_rowdot(const void* dO16, const void* O16, float* Dh, int nseq, int H, int Tp, hipStream_t stream) {
    if (!dO16 || !O16 || !Dh || nseq <= 0 || Tp <= 0 || H != 4) return EEND_EINVAL;
    const long M = (long)nseq * Tp;
    hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, (const __bf16*)dO16, (const _Float16*)O16, Dh, nseq, Tp);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
