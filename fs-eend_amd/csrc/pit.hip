// Permutation-invariant label assignment: batch_pit_n_speaker_loss
// (FS-EEND/train/utils/loss.py:257-327) and pit_loss_multispk (LS-EEND/train/utils/loss.py:350-379).  The
// reference stacks C rolled BCE maps and enumerates all C! permutations (FS) or ships a cost matrix to the host
// and runs scipy's Hungarian solver per utterance (LS).  Both reduce to ONE (C x C) matrix per utterance,
//     cost[i][j] = sum_t BCEwithLogits(y[t,i], label[t,j]) = sum_t softplus(y[t,i]) - sum_t y[t,i] label[t,j]
// over the -1-padded batch length, and an assignment on its leading n x n block (the other slots keep their
// place).  pit_cost_kernel: one block per utterance, thread (i,j), frames staged through LDS, fp64 sums.
// pit_assign_kernel: one thread per utterance runs the O(n^3) shortest-augmenting-path (Jonker-Volgenant /
// Hungarian) solver in fp64 -- n <= 16, so this is latency, not throughput, work; no host round trip.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int CMAX = 16;
constexpr int TCH = 128;

__global__ __launch_bounds__(256)
void pit_cost_kernel(const float* __restrict__ y, const float* __restrict__ lab, int T, int C, double* __restrict__ cost) {
    __shared__ float ys[TCH * CMAX], ls[TCH * CMAX];
    __shared__ double sp[CMAX];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int i = tid / CMAX, j = tid % CMAX;
    const float* yb = y + (size_t)b * T * C;
    const float* lb = lab + (size_t)b * T * C;
    double dot = 0.0, spl = 0.0;
    for (int t0 = 0; t0 < T; t0 += TCH) {
        __syncthreads();
        for (int q = tid; q < TCH * C; q += 256) {
            const int t = t0 + q / C;
            ys[(q / C) * CMAX + q % C] = t < T ? yb[(size_t)t0 * C + q] : 0.f;
            ls[(q / C) * CMAX + q % C] = t < T ? lb[(size_t)t0 * C + q] : 0.f;
        }
        __syncthreads();
        const int n = T - t0 < TCH ? T - t0 : TCH;
        if (i < C && j < C) {
            for (int t = 0; t < n; ++t) dot += (double)ys[t * CMAX + i] * (double)ls[t * CMAX + j];
            if (j == 0)
                for (int t = 0; t < n; ++t) {
                    const double v = (double)ys[t * CMAX + i];
                    spl += (v > 0 ? v : 0.0) + log1p(exp(-fabs(v)));
                }
        }
    }
    if (i < C && j == 0) sp[i] = spl;
    __syncthreads();
    if (i < C && j < C) cost[((size_t)b * C + i) * C + j] = sp[i] - dot;
}

// mode 0 (batch_pit): loss_b = (sum_{i<n} cost[i][perm[i]] + sum_{i>=n} cost[i][i]) / C
// perm[b][i] = label column assigned to prediction i (identity for i >= n_b)
__global__ __launch_bounds__(64)
void pit_assign_kernel(const double* __restrict__ cost, const int* __restrict__ nspk, int B, int C, int* __restrict__ perm,
                       double* __restrict__ loss) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const double* a = cost + (size_t)b * C * C;
    int n = nspk[b];
    n = n < 0 ? 0 : (n > C ? C : n);
    double u[CMAX + 1], v[CMAX + 1], minv[CMAX + 1];
    int p[CMAX + 1], way[CMAX + 1];
    bool used[CMAX + 1];
    for (int k = 0; k <= n; ++k) { u[k] = 0.0; v[k] = 0.0; p[k] = 0; way[k] = 0; }
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        for (int k = 0; k <= n; ++k) { minv[k] = 1e300; used[k] = false; }
        do {
            used[j0] = true;
            const int i0 = p[j0];
            double delta = 1e300;
            int j1 = 0;
            for (int j = 1; j <= n; ++j)
                if (!used[j]) {
                    const double cur = a[(i0 - 1) * C + (j - 1)] - u[i0] - v[j];
                    if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                    if (minv[j] < delta) { delta = minv[j]; j1 = j; }
                }
            for (int j = 0; j <= n; ++j)
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            j0 = j1;
        } while (p[j0] != 0);
        do {
            const int j1 = way[j0];
            p[j0] = p[j1];
            j0 = j1;
        } while (j0);
    }
    double tot = 0.0;
    for (int j = 1; j <= n; ++j) {
        perm[(size_t)b * C + (p[j] - 1)] = j - 1;
        tot += a[(p[j] - 1) * C + (j - 1)];
    }
    for (int i = n; i < C; ++i) {
        perm[(size_t)b * C + i] = i;
        tot += a[i * C + i];
    }
    loss[b] = tot / (double)C;
}

}  // namespace

int eend_launch_pit_cost(const float* y, const float* lab, int B, int T, int C, double* cost, hipStream_t stream) {
    if (!y || !lab || !cost || B <= 0 || T <= 0 || C < 1 || C > CMAX) return EEND_EINVAL;
    hipLaunchKernelGGL(pit_cost_kernel, dim3(B), dim3(256), 0, stream, y, lab, T, C, cost);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_pit_assign(const double* cost, const int* nspk, int B, int C, int* perm, double* loss, hipStream_t stream) {
    if (!cost || !nspk || !perm || !loss || B <= 0 || C < 1 || C > CMAX) return EEND_EINVAL;
    hipLaunchKernelGGL(pit_assign_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, cost, nspk, B, C, perm, loss);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
