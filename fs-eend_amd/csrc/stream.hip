// Frame-by-frame (streaming) state kernels of LS-EEND.  One frame of one stream is far too
// little work for the matrix pipe: these are latency-bound VALU kernels whose job is to keep
// the recurrent state resident in HBM/L2 in the layout the reference's driver owns
// (LS-EEND/streaming_infer_dia.py:37-49) and to touch it exactly once per frame.
#include "common.h"
#include "kernels.h"

namespace {

// MultiScaleRetention.recurrent_forward (LS-EEND/nnet/modules/retention.py:126-144) with
// decay == 1, fused with the per-head LayerNorm (:222, eps, no affine) and the swish gate (:224).
//   scale_t = scale_{t-1} + 1
//   kv_t[a][b] = kv_{t-1}[a][b] * sqrt(scale_{t-1} / scale_t) + v[a] k[b] / sqrt(scale_t)
//   o[a] = sum_b q[b] kv_t[a][b]
// qkvg f16 [N][4*D] rows = [q | k (already * dk^-0.5) | v | g]; kv f32 [N][H][64 (a: value dim)][64 (b: key dim)]
// updated in place; scale_in/scale_out f32 [H] (first frame: scale_in = 0, kv = 0).
// One wave per (n, h): lane a owns row kv[a][:].
// QT = _Float16: the batch path's operand precision; QT = float: the projections of eend_retention_proj_step_f32 (frame-by-
// frame sessions: see there).
template <class QT>
__global__ __launch_bounds__(256)
void ret_step_kernel(const QT* __restrict__ qkvg, float* __restrict__ kv, const float* __restrict__ scale_in,
                     float* __restrict__ scale_out, _Float16* __restrict__ out, float* __restrict__ out32, int N, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);       // n*H + h
    if (idx >= N * H) return;
    const int n = idx / H, h = idx - n * H;
    const int D = H * 64;
    const QT* row = qkvg + (size_t)n * 4 * D;
    const float ps = scale_in[h];
    const float ns = ps + 1.0f;
    // the decay of the old state is applied 36 000 times over an hour of audio: evaluate it in double and round once, so
    // that the running product of the factors follows sqrt(s/t) to fp32 rounding noise instead of accumulating the bias
    // of the device's fast f32 sqrt / divide sequences
    const float keep = (float)__builtin_sqrt((double)ps / (double)ns);
    const float add = (float)(1.0 / __builtin_sqrt((double)ns));
    const float va = (float)row[2 * D + h * 64 + lane] * add;
    float* st = kv + ((size_t)idx * 64 + lane) * 64;
    float o = 0.f;
#pragma unroll
    for (int b = 0; b < 64; b += 4) {
        float4 s = *(const float4*)(st + b);
        typedef QT qt4 __attribute__((ext_vector_type(4)));
        const qt4 kk = *(const qt4*)(row + D + h * 64 + b);
        const qt4 qq = *(const qt4*)(row + h * 64 + b);
        s.x = __builtin_fmaf(s.x, keep, va * (float)kk[0]);
        s.y = __builtin_fmaf(s.y, keep, va * (float)kk[1]);
        s.z = __builtin_fmaf(s.z, keep, va * (float)kk[2]);
        s.w = __builtin_fmaf(s.w, keep, va * (float)kk[3]);
        *(float4*)(st + b) = s;
        o += (float)qq[0] * s.x + (float)qq[1] * s.y + (float)qq[2] * s.z + (float)qq[3] * s.w;
    }
    float sum = o;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) sum = wave_xor_add(sum, m);
    const float mean = sum * (1.0f / 64.0f);
    float var = (o - mean) * (o - mean);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) var = wave_xor_add(var, m);
    const float y = (o - mean) / __builtin_sqrtf(var * (1.0f / 64.0f) + eps);
    const float g = (float)row[3 * D + h * 64 + lane];
    const float r = g / (1.0f + __expf(-g)) * y;
    if (out) out[(size_t)n * D + h * 64 + lane] = to_f16_sat(r);
    if (out32) out32[(size_t)n * D + h * 64 + lane] = r;
    if (n == 0 && lane == 0) scale_out[h] = ns;
}

// Retention projections of ONE frame in full f32 (frame-by-frame sessions): qkvg[n][f] = <LN(x[n]), W[f]> + b[f] for the
// packed (4*256, 256) f32 weight [q; k * dk^-0.5; v; g], optional LayerNorm in front (the Conformer's pre-norm; the
// decoder feeds its post-norm residual stream directly).  Why not the f16 skinny GEMM the other frame-step linears use:
// in the recurrent form the state kv_t is, after tens of thousands of frames, dominated by the slowly growing mean of
// k (x) v (~sqrt(t)), and the per-head LayerNorm that follows keeps only the O(1) part around it -- f16 rounding of q, k,
// v (2^-11 relative) is then amplified into > 1e-3 on the logits late in a one-hour stream (tests/test_long_horizon.py;
// emulated on the oracle: f16 retention projections alone give 1.1e-3 within the first 6000 frames, every other f16 choice
// of the step together stays below).  One frame has <= 16 rows, so f32 costs nothing: the 1 MB weight is the traffic.
// Block = 4 waves = 16 output features; the (LayerNorm-ed) rows sit in LDS; a wave reduces one feature over the 256 inputs.
__global__ __launch_bounds__(256)
void ret_proj_step_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                          const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out, int N) {
    __shared__ __attribute__((aligned(16))) float xs[16][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    x += (size_t)blockIdx.y * 16 * 256;                  // rows 16*blockIdx.y .. +15 of the frame's N rows
    out += (size_t)blockIdx.y * 16 * 1024;
    N = N - (int)blockIdx.y * 16 < 16 ? N - (int)blockIdx.y * 16 : 16;
    for (int r = wave; r < N; r += 4) {
        const float4 v = *(const float4*)(x + (size_t)r * 256 + lane * 4);
        float4 y = v;
        if (gamma) {
            float s = v.x + v.y + v.z + v.w;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) s = wave_xor_add(s, m);
            const float mean = s * (1.0f / 256.0f);
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            float q = a * a + b * b + c * c + d * d;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) q = wave_xor_add(q, m);
            const float rs = 1.0f / __builtin_sqrtf(q * (1.0f / 256.0f) + eps);
            const float4 g = *(const float4*)(gamma + lane * 4), be = *(const float4*)(beta + lane * 4);
            y = make_float4(a * rs * g.x + be.x, b * rs * g.y + be.y, c * rs * g.z + be.z, d * rs * g.w + be.w);
        }
        *(float4*)(&xs[r][lane * 4]) = y;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = blockIdx.x * 16 + wave * 4 + i;
        const float4 w = *(const float4*)(W + (size_t)f * 256 + lane * 4);
        const float bf = bias[f];
        for (int r = 0; r < N; ++r) {
            const float4 v = *(const float4*)(&xs[r][lane * 4]);
            float p = w.x * v.x + w.y * v.y + w.z * v.z + w.w * v.w;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) p = wave_xor_add(p, m);
            if (lane == 0) out[(size_t)r * 1024 + f] = p + bf;
        }
    }
}

// Decoder input of one frame in f32 (LS model :229-233, `convert(cat(emb, pe))`, in the split form of DESIGN 3:
// out[b*C + c] = W[:, :256] emb[b] + pc[c]).  The decoder's retention normalises by a near-zero-mean statistic with
// eps 1e-6, so the f16 rounding of emb / W here is amplified ~30x at some frames of a long stream (emulated on the
// oracle: 7.8e-4 in the logits by frame 2000 from this linear alone).  W f32 [256][ldw] (the parameter itself);
// emb f32 [B][256]; pc f32 [C][256]; out32 f32 / out16 f16 [B*C][256].  A wave owns one output feature.
__global__ __launch_bounds__(256)
void convert_step_f32_kernel(const float* __restrict__ emb, const float* __restrict__ W, int ldw, const float* __restrict__ pc,
                             float* __restrict__ out32, _Float16* __restrict__ out16, int B, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.x * 4 + wave;
    const float4 w = *(const float4*)(W + (size_t)f * ldw + lane * 4);
    for (int b = 0; b < B; ++b) {
        const float4 v = *(const float4*)(emb + (size_t)b * 256 + lane * 4);
        float p = w.x * v.x + w.y * v.y + w.z * v.z + w.w * v.w;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) p = wave_xor_add(p, m);
        if (lane < C) {
            const float y = p + pc[(size_t)lane * 256 + f];
            out32[((size_t)b * C + lane) * 256 + f] = y;
            out16[((size_t)b * C + lane) * 256 + f] = to_f16_sat(y);
        }
    }
}

// Speaker-axis attention of one frame in f32 (self_attn2 of the LS decoder layer, merge_retnet_layer.py:300-307, inside
// the all-f32 decoder frame step): qkv f32 [B*C][768] = [q | k | v] (head h at column h*64), out f32 [B*C][256].
// One wave per (row, head); lane = head dimension; C <= 16 scores per wave.
__global__ __launch_bounds__(64)
void spk_attn_step_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int C, float scale) {
    const int lane = threadIdx.x, row = blockIdx.x, h = blockIdx.y;
    const int b0 = (row / C) * C;
    const float q = qkv[(size_t)row * 768 + h * 64 + lane] * scale;
    float sc[16];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float d = 0.f;
        if (c < C) {
            d = q * qkv[(size_t)(b0 + c) * 768 + 256 + h * 64 + lane];
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) d = wave_xor_add(d, m);
            mx = __builtin_fmaxf(mx, d);
        }
        sc[c] = d;
    }
    float den = 0.f, o = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < C) {
            const float pr = __expf(sc[c] - mx);
            den += pr;
            o = __builtin_fmaf(pr, qkv[(size_t)(b0 + c) * 768 + 512 + h * 64 + lane], o);
        }
    out[(size_t)row * 256 + h * 64 + lane] = o / den;
}

// y[r] = x[r] / ||x[r]||_2 over 256 features, f32 rows (the embedding normalisation of the f32 frame step; LS model :87, no eps).
__global__ __launch_bounds__(64)
void l2norm_rows_f32_kernel(const float* __restrict__ x, float* __restrict__ y) {
    const int lane = threadIdx.x;
    const float4 v = *(const float4*)(x + (size_t)blockIdx.x * 256 + lane * 4);
    float q = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) q = wave_xor_add(q, m);
    const float r = 1.0f / __builtin_sqrtf(q);
    *(float4*)(y + (size_t)blockIdx.x * 256 + lane * 4) = make_float4(v.x * r, v.y * r, v.z * r, v.w * r);
}

// ConformerConvModule.forward_one_step, depthwise part (conformer/convolution.py:157-163):
// window = [cache (k-1 frames) | x_t]; y = sum_j w[c][j] window[c][j]; BatchNorm(eval); Swish;
// new cache = window[:, 1:].  x f16 [B][D]; cache f32 [B][D][k-1] (the driver's layout), in place.
__global__ __launch_bounds__(256)
void dwconv_step_kernel(const _Float16* __restrict__ x, float* __restrict__ cache, const float* __restrict__ w,
                        const float* __restrict__ bw, const float* __restrict__ bb, const float* __restrict__ bm,
                        const float* __restrict__ bv, float eps, _Float16* __restrict__ out, int B, int D, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // b*D + c
    if (i >= B * D) return;
    const int c = i % D;
    float* cc = cache + (size_t)i * (k - 1);
    const float* wc = w + (size_t)c * k;
    const float xn = (float)x[i];
    float y = wc[k - 1] * xn;
    float prev = xn;
    for (int j = k - 2; j >= 0; --j) {           // walk backwards so the shift can be done in place
        const float cur = cc[j];
        y = __builtin_fmaf(wc[j], cur, y);
        cc[j] = prev;                            // new_cache[j] = window[j+1]
        prev = cur;
    }
    const float sc = bw[c] / __builtin_sqrtf(bv[c] + eps);
    y = (y - bm[c]) * sc + bb[c];
    out[i] = to_f16_sat(y / (1.0f + __expf(-y)));
}

// FS-EEND incremental self-attention (FS-EEND/nnet/modules/streaming_tfm.py:15-37): one new token
// per sequence attends over its growing key/value history.  The reference caches the layer inputs
// and re-projects all t keys every frame; caching the projected K/V is the same arithmetic at
// O(t) instead of O(t * D^2) per frame.  qkv f16 [N][3*D] (packed in-proj of the new token);
// caches f16 [N][H][cap][64]; `t` = tokens already cached.  One wave per (n, h): appends the new
// k/v row, then an online softmax over 64-key chunks (lane = key for the scores, lane = d for PV).
// t_dev (optional): the token count lives in device memory, so that a captured hipGraph of the frame step stays
// valid while the history grows (the host only re-captures when the cache capacity changes); a count that has
// reached the capacity turns the launch into a no-op instead of an out-of-bounds append.
__global__ __launch_bounds__(256)
void attn_decode_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ Kc, _Float16* __restrict__ Vc,
                        _Float16* __restrict__ out, int N, int H, int cap, int t, const int* __restrict__ t_dev, float scale) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= N * H) return;
    if (t_dev) t = __builtin_amdgcn_readfirstlane(*t_dev);
    if (t >= cap) return;
    const int n = idx / H, h = idx - n * H;
    const int D = H * 64;
    const _Float16* row = qkv + (size_t)n * 3 * D + h * 64;
    _Float16* Kh = Kc + (size_t)idx * cap * 64;
    _Float16* Vh = Vc + (size_t)idx * cap * 64;
    const _Float16 kn = row[D + lane], vn = row[2 * D + lane];
    Kh[(size_t)t * 64 + lane] = kn;
    Vh[(size_t)t * 64 + lane] = vn;
    float qf[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f16x8 q8 = *(const f16x8*)(row + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[i * 8 + e] = (float)q8[e] * scale;
    }
    // the new token's own score and value come from registers (its cache rows are not read back)
    float s_new = qf[0] * 0.f;
    {
        float part = (float)row[lane] * scale * (float)kn;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) part = wave_xor_add(part, m);
        s_new = part;
    }
    float m_run = s_new, l_run = 1.0f, o = (float)vn;        // softmax state seeded with the new token
    for (int c0 = 0; c0 < t; c0 += 64) {
        const int key = c0 + lane;
        float s = -INFINITY;
        if (key < t) {
            const _Float16* kr = Kh + (size_t)key * 64;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f16x8 k8 = *(const f16x8*)(kr + i * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qf[i * 8 + e], (float)k8[e], acc);
            }
            s = acc;
        }
        float cm = s;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) cm = wave_xor_max(cm, m);
        const float m_new = __builtin_fmaxf(m_run, cm);
        const float alpha = __expf(m_run - m_new);
        const float p = __expf(s - m_new);                    // 0 for key >= t
        float ps = p;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) ps = wave_xor_add(ps, m);
        l_run = l_run * alpha + ps;
        o *= alpha;
        const int nk = (t - c0) < 64 ? (t - c0) : 64;
        for (int j = 0; j < nk; ++j) {
            const float pj = __shfl(p, j, 64);
            o = __builtin_fmaf(pj, (float)Vh[(size_t)(c0 + j) * 64 + lane], o);
        }
        m_run = m_new;
    }
    out[(size_t)n * D + h * 64 + lane] = to_f16_sat(o / l_run);
}


// ---- long histories: the same decode step split over the key axis ("flash decoding").  One wave per (n, h) walks its
// history serially -- fine for the t <= 1000 the model is trained on, but at the one-hour position of BASELINE config 5
// (t = 36 000) a frame then costs 86 ms: 64 waves on a 256-CU part, each chasing 36 000 dependent rows.  Here a
// workgroup owns DEC_R consecutive keys of one (n, h) (its 4 waves take the 64-key chunks round-robin, online softmax
// per wave, merged through LDS) and writes (max, sum, o[64]) partials; a second tiny kernel merges the partials of a
// (n, h) and adds the new token.  Grid = (N*H, cap / DEC_R): fixed per cache capacity, so a captured hipGraph stays valid
// while t (read from device memory) grows; blocks whose key range starts at or beyond t write an empty partial.
constexpr int DEC_R = 512;

__global__ __launch_bounds__(256)
void attn_decode_split_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ Kc, _Float16* __restrict__ Vc,
                              float* __restrict__ part, int N, int H, int cap, int nsplit, const int* __restrict__ t_dev, float scale) {
    __shared__ float red[4][66];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x, sp = blockIdx.y;
    const int t = __builtin_amdgcn_readfirstlane(*t_dev);
    const int n = idx / H, h = idx - n * H;
    const int D = H * 64;
    const _Float16* row = qkv + (size_t)n * 3 * D + h * 64;
    _Float16* Kh = Kc + (size_t)idx * cap * 64;
    _Float16* Vh = Vc + (size_t)idx * cap * 64;
    if (sp == 0 && wave == 0 && t < cap) {                    // append the new token's k / v (never read back this step)
        Kh[(size_t)t * 64 + lane] = row[D + lane];
        Vh[(size_t)t * 64 + lane] = row[2 * D + lane];
    }
    const int k0 = sp * DEC_R;
    int k1 = k0 + DEC_R;
    k1 = k1 < t ? k1 : t;
    float m_run = -INFINITY, l_run = 0.f, o = 0.f;
    if (k0 < t && t < cap) {
        float qf[64];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f16x8 q8 = *(const f16x8*)(row + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[i * 8 + e] = (float)q8[e] * scale;
        }
        for (int c0 = k0 + wave * 64; c0 < k1; c0 += 256) {
            const int key = c0 + lane;
            float s = -INFINITY;
            if (key < k1) {
                const _Float16* kr = Kh + (size_t)key * 64;
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f16x8 k8 = *(const f16x8*)(kr + i * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qf[i * 8 + e], (float)k8[e], acc);
                }
                s = acc;
            }
            float cm = s;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) cm = wave_xor_max(cm, m);
            const float m_new = __builtin_fmaxf(m_run, cm);
            const float alpha = __expf(m_run - m_new);         // exp(-inf) = 0 on the first chunk
            const float p = __expf(s - m_new);                 // 0 for key >= k1
            float ps = p;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) ps = wave_xor_add(ps, m);
            l_run = l_run * alpha + ps;
            o *= alpha;
            const int nk = (k1 - c0) < 64 ? (k1 - c0) : 64;
            for (int j8 = 0; j8 < nk; j8 += 8) {               // 8 independent V rows in flight per step
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = (j8 + jj < nk) ? (float)Vh[(size_t)(c0 + j8 + jj) * 64 + lane] : 0.f;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) o = __builtin_fmaf(__shfl(p, j8 + jj, 64), v[jj], o);
            }
            m_run = m_new;
        }
    }
    red[wave][lane] = o;
    if (lane == 0) { red[wave][64] = m_run; red[wave][65] = l_run; }
    __syncthreads();
    if (wave == 0) {
        const float m0 = red[0][64], m1 = red[1][64], m2 = red[2][64], m3 = red[3][64];
        const float M = __builtin_fmaxf(__builtin_fmaxf(m0, m1), __builtin_fmaxf(m2, m3));
        float L = 0.f, O = 0.f;
        if (M > -INFINITY) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float mw = red[w][64];
                const float f = mw > -INFINITY ? __expf(mw - M) : 0.f;
                L += red[w][65] * f;
                O += red[w][lane] * f;
            }
        }
        float* pp = part + ((size_t)idx * nsplit + sp) * 66;
        pp[lane] = O;
        if (lane == 0) { pp[64] = M; pp[65] = L; }
    }
}

__global__ __launch_bounds__(64)
void attn_decode_merge_kernel(const _Float16* __restrict__ qkv, const float* __restrict__ part, _Float16* __restrict__ out, int N, int H,
                              int cap, int nsplit, const int* __restrict__ t_dev, float scale) {
    const int lane = threadIdx.x, idx = blockIdx.x;
    const int t = __builtin_amdgcn_readfirstlane(*t_dev);
    if (t >= cap) return;
    const int n = idx / H, h = idx - n * H;
    const int D = H * 64;
    const _Float16* row = qkv + (size_t)n * 3 * D + h * 64;
    float sn = (float)row[lane] * scale * (float)row[D + lane];            // the new token's own score q . k_new
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) sn = wave_xor_add(sn, m);
    float M = sn, L = 1.0f, O = (float)row[2 * D + lane];
    const int ns = (t + DEC_R - 1) / DEC_R;
    for (int s = 0; s < ns && s < nsplit; ++s) {
        const float* pp = part + ((size_t)idx * nsplit + s) * 66;
        const float ms = pp[64], ls = pp[65];
        if (!(ls > 0.f)) continue;
        const float Mn = __builtin_fmaxf(M, ms);
        const float a = __expf(M - Mn), b = __expf(ms - Mn);
        L = L * a + ls * b;
        O = O * a + pp[lane] * b;
        M = Mn;
    }
    out[(size_t)n * D + h * 64 + lane] = to_f16_sat(O / L);
}

__global__ void counter_add_kernel(int* c, int inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}

}  // namespace

int eend_launch_attn_decode(const void* qkv, void* Kc, void* Vc, void* out16, int N, int H, int cap, int t, const int* t_dev,
                            float scale, hipStream_t stream) {
    if (N <= 0 || H <= 0 || cap <= 0 || (!t_dev && (t < 0 || t >= cap))) return EEND_EINVAL;
    hipLaunchKernelGGL(attn_decode_kernel, dim3((N * H + 3) / 4), dim3(256), 0, stream, (const _Float16*)qkv, (_Float16*)Kc,
                       (_Float16*)Vc, (_Float16*)out16, N, H, cap, t, t_dev, scale);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_attn_decode_split(const void* qkv, void* Kc, void* Vc, void* out16, float* part, long part_floats, int N, int H, int cap,
                                  const int* t_dev, float scale, hipStream_t stream) {
    if (N <= 0 || H <= 0 || cap <= 0 || !t_dev || !part) return EEND_EINVAL;
    const int nsplit = (cap + DEC_R - 1) / DEC_R;
    if (nsplit > 65535 || part_floats < (long)N * H * nsplit * 66) return EEND_EINVAL;
    hipLaunchKernelGGL(attn_decode_split_kernel, dim3(N * H, nsplit), dim3(256), 0, stream, (const _Float16*)qkv, (_Float16*)Kc, (_Float16*)Vc,
                       part, N, H, cap, nsplit, t_dev, scale);
    if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    hipLaunchKernelGGL(attn_decode_merge_kernel, dim3(N * H), dim3(64), 0, stream, (const _Float16*)qkv, part, (_Float16*)out16, N, H, cap,
                       nsplit, t_dev, scale);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_counter_add(int* c, int inc, hipStream_t stream) {
    if (!c) return EEND_EINVAL;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, stream, c, inc);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_ret_step(const void* qkvg, float* kv, const float* scale_in, float* scale_out, void* out16, int N,
                         int H, float eps, hipStream_t stream) {
    if (N <= 0 || H <= 0) return EEND_EINVAL;
    hipLaunchKernelGGL(ret_step_kernel<_Float16>, dim3((N * H + 3) / 4), dim3(256), 0, stream, (const _Float16*)qkvg, kv, scale_in,
                       scale_out, (_Float16*)out16, (float*)nullptr, N, H, eps);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_ret_step_f32in(const float* qkvg, float* kv, const float* scale_in, float* scale_out, void* out16, float* out32, int N,
                               int H, float eps, hipStream_t stream) {
    if (N <= 0 || H <= 0 || (!out16 && !out32)) return EEND_EINVAL;
    hipLaunchKernelGGL(ret_step_kernel<float>, dim3((N * H + 3) / 4), dim3(256), 0, stream, qkvg, kv, scale_in, scale_out, (_Float16*)out16,
                       out32, N, H, eps);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_ret_proj_step(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias,
                              float* out, int N, hipStream_t stream) {
    if (!x || !W || !bias || !out || N <= 0 || N > 16 * 65535 || (gamma && !beta)) return EEND_EINVAL;
    hipLaunchKernelGGL(ret_proj_step_kernel, dim3(64, (N + 15) / 16), dim3(256), 0, stream, x, gamma, beta, eps, W, bias, out, N);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_convert_step_f32(const float* emb, const float* W, int ldw, const float* pc, float* out32, void* out16, int B, int C,
                                 hipStream_t stream) {
    if (!emb || !W || !pc || !out32 || !out16 || B <= 0 || C <= 0 || C > 64 || ldw < 256 || (ldw & 3)) return EEND_EINVAL;
    hipLaunchKernelGGL(convert_step_f32_kernel, dim3(64), dim3(256), 0, stream, emb, W, ldw, pc, out32, (_Float16*)out16, B, C);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_l2norm_rows_f32(const float* x, float* y, int rows, hipStream_t stream) {
    if (!x || !y || rows <= 0) return EEND_EINVAL;
    hipLaunchKernelGGL(l2norm_rows_f32_kernel, dim3(rows), dim3(64), 0, stream, x, y);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_spk_attn_step_f32(const float* qkv, float* out, int B, int C, float scale, hipStream_t stream) {
    if (!qkv || !out || B <= 0 || C <= 0 || C > 16) return EEND_EINVAL;
    hipLaunchKernelGGL(spk_attn_step_f32_kernel, dim3(B * C, 4), dim3(64), 0, stream, qkv, out, C, scale);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_dwconv_step(const void* x16, float* cache, const float* w, const float* bn_w, const float* bn_b,
                            const float* bn_mean, const float* bn_var, float eps, void* out16, int B, int D, int k,
                            hipStream_t stream) {
    if (B <= 0 || D <= 0 || k < 2) return EEND_EINVAL;
    hipLaunchKernelGGL(dwconv_step_kernel, dim3((B * D + 255) / 256), dim3(256), 0, stream, (const _Float16*)x16, cache, w,
                       bn_w, bn_b, bn_mean, bn_var, eps, (_Float16*)out16, B, D, k);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
