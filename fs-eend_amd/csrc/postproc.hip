// Output-side post-processing: what the reference does on the host with torch /
// scipy after model.test -- FS-EEND/train/utils/make_rttm.py:10-28 (threshold, scipy medfilt(k = 11) along time,
// change points of each speaker track) and train/utils/loss.py:198-236 (frame-level DER counters).  Integer /
// index work on a few hundred KB: bandwidth-trivial, kept on the device so that a 1-hour stream's (T, C)
// activity map never round-trips through host numpy and the DER counters need no per-utterance sync.
// All results are bit exact against the reference (tests/golden/post_*.npz).
#include "common.h"
#include "kernels.h"

namespace {

// pred (T, S) probabilities -> 0/1 after `> thr` and a zero-padded median of k (odd) along time: the median of k
// binary values is 1 exactly when at least k/2 + 1 of them are 1.
__global__ __launch_bounds__(256)
void activity_median_kernel(const float* __restrict__ pred, int ld, int T, int S, float thr, int k, unsigned char* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)T * S) return;
    const int t = (int)(idx / S), s = (int)(idx - (long)t * S);
    const int h = k >> 1;
    int cnt = 0;
    for (int d = -h; d <= h; ++d) {
        const int u = t + d;
        if (u >= 0 && u < T) cnt += pred[(long)u * ld + s] > thr ? 1 : 0;
    }
    out[idx] = cnt >= h + 1 ? 1 : 0;
}

// one wave per speaker: indices i in [0, T] where the zero-padded track changes (padded[i+1] != padded[i]),
// in increasing order (ballot + prefix popcount compaction); even entries are segment starts, odd ones ends.
__global__ __launch_bounds__(64)
void segments_kernel(const unsigned char* __restrict__ act, int T, int S, int* __restrict__ changes, int* __restrict__ counts, int cap) {
    const int s = blockIdx.x, lane = threadIdx.x;
    int base = 0;
    for (int i0 = 0; i0 <= T; i0 += 64) {
        const int i = i0 + lane;
        const int cur = (i < T) ? act[(long)i * S + s] : 0;
        const int prev = (i >= 1 && i - 1 < T) ? act[(long)(i - 1) * S + s] : 0;
        const bool chg = i <= T && cur != prev;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(chg);
        const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (chg && pos < cap) changes[(long)s * cap + pos] = i;
        base += __builtin_popcountll(m);
    }
    if (lane == 0) counts[s] = base;
}

// counters[0..7] += speech_scored, speech_miss, speech_falarm, speaker_scored, speaker_miss, speaker_falarm,
// speaker_error, #(label == decision); decisions = sigmoid(pred[t + delay]) > 0.5 against label[t].
__global__ __launch_bounds__(256)
void der_counters_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ label, int ldl, int T, int C, int delay,
                         unsigned long long* __restrict__ counters) {
    __shared__ unsigned long long red[8];
    if (threadIdx.x < 8) red[threadIdx.x] = 0ull;
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    unsigned v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t < T - delay) {
        int n_sys = 0, n_map = 0, n_eq = 0;
        float ref_sum = 0.f;
        for (int c = 0; c < C; ++c) {
            const float x = pred[(long)(t + delay) * ldp + c], l = label[(long)t * ldl + c];
            const bool dec = 1.0f / (1.0f + expf(-x)) > 0.5f;
            ref_sum += l;
            n_sys += dec;
            n_map += (l == 1.0f && dec);
            n_eq += (l == (dec ? 1.0f : 0.0f));
        }
        const int n_ref = (int)(long)ref_sum;                   // label.sum(-1).long()
        v[0] = n_ref > 0;
        v[1] = n_ref > 0 && n_sys == 0;
        v[2] = n_ref == 0 && n_sys > 0;
        v[3] = (unsigned)n_ref;
        v[4] = n_ref > n_sys ? n_ref - n_sys : 0;
        v[5] = n_sys > n_ref ? n_sys - n_ref : 0;
        v[6] = (unsigned)((n_ref < n_sys ? n_ref : n_sys) - n_map);
        v[7] = (unsigned)n_eq;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        unsigned x = v[q];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&red[q], (unsigned long long)x);
    }
    __syncthreads();
    if (threadIdx.x < 8 && red[threadIdx.x]) atomicAdd(&counters[threadIdx.x], red[threadIdx.x]);
}

}  // namespace

int eend_launch_activity_median(const float* pred, int ld, int T, int S, float thr, int k, unsigned char* out, hipStream_t stream) {
    if (!pred || !out || T <= 0 || S <= 0 || ld < S || k < 1 || (k & 1) == 0 || k > 255) return EEND_EINVAL;
    const long n = (long)T * S;
    hipLaunchKernelGGL(activity_median_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pred, ld, T, S, thr, k, out);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_segments(const unsigned char* act, int T, int S, int* changes, int* counts, int cap, hipStream_t stream) {
    if (!act || !changes || !counts || T <= 0 || S <= 0 || cap <= 0) return EEND_EINVAL;
    hipLaunchKernelGGL(segments_kernel, dim3(S), dim3(64), 0, stream, act, T, S, changes, counts, cap);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_der_counters(const float* pred, int ldp, const float* label, int ldl, int T, int C, int delay,
                             unsigned long long* counters, hipStream_t stream) {
    if (!pred || !label || !counters || T <= 0 || C <= 0 || ldp < C || ldl < C || delay < 0 || delay > T) return EEND_EINVAL;
    if (hipMemsetAsync(counters, 0, 8 * sizeof(unsigned long long), stream) != hipSuccess) return EEND_ELAUNCH;
    const int n = T - delay;
    if (n <= 0) return EEND_OK;
    hipLaunchKernelGGL(der_counters_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pred, ldp, label, ldl, T, C, delay, counters);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
