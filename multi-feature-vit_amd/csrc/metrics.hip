// Epoch metrics on the device: argmax / confusion matrix and the pair counts of the one-vs-rest ROC AUC, so that
// the evaluation loop keeps logits and labels in HBM instead of copying every batch to the host (MAIN_CA:886-899) and calling
// scikit-learn at the end of the epoch (MAIN_CA:901-909).  Integer results: bit-exact against the oracle.
#include "common.cuh"
#include "kernels.h"

namespace mfvit {

namespace {

// conf[t][p] += 1 for every sample (first maximum wins, like torch.max / np.argmax); preds optional
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ scores, long ld, const int64_t* __restrict__ labels, int n,
                                                        int C, unsigned long long* __restrict__ conf, int64_t* __restrict__ preds) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* s = scores + (long)i * ld;
    int best = 0;
    float bv = s[0];
    for (int c = 1; c < C; ++c) {
        const float v = s[c];
        if (v > bv) { bv = v; best = c; }
    }
    if (preds) preds[i] = best;
    const int t = (int)labels[i];
    if (t >= 0 && t < C) atomicAdd(conf + (long)t * C + best, 1ull);
}

// One-vs-rest ROC AUC of class c = blockIdx.y, as exact integers: for every positive i (label == c) count the negatives j with a
// smaller score (weight 2) or an equal one (weight 1): u2[c] = 2 #{s_i > s_j} + #{s_i == s_j}; AUC = u2 / (2 n_pos n_neg), the
// trapezoid area under sklearn's roc_curve (ties share a threshold -> half credit).  O(n^2) compares; n is an epoch's sample count
// (thousands), the negatives' scores are streamed through LDS 1024 at a time.
__global__ __launch_bounds__(256) void auc_pairs_kernel(const float* __restrict__ scores, long ld, const int64_t* __restrict__ labels, int n,
                                                        unsigned long long* __restrict__ u2, unsigned long long* __restrict__ npos) {
    __shared__ float sc[1024];
    __shared__ int neg[1024];
    const int c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool is_pos = i < n && labels[i] == c;
    const float si = i < n ? scores[(long)i * ld + c] : 0.f;
    unsigned long long cnt = 0;
    for (int j0 = 0; j0 < n; j0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024; t += 256) {
            const int j = j0 + t;
            sc[t] = j < n ? scores[(long)j * ld + c] : 0.f;
            neg[t] = j < n && labels[j] != c;
        }
        __syncthreads();
        if (is_pos) {
            const int m = n - j0 < 1024 ? n - j0 : 1024;
            for (int t = 0; t < m; ++t)
                if (neg[t]) cnt += sc[t] < si ? 2u : (sc[t] == si ? 1u : 0u);
        }
    }
    // block reduction, one atomic per block and class
    __shared__ unsigned long long red[256];
    __shared__ unsigned int pc[256];
    red[threadIdx.x] = cnt;
    pc[threadIdx.x] = is_pos ? 1u : 0u;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[threadIdx.x] += red[threadIdx.x + o];
            pc[threadIdx.x] += pc[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (red[0]) atomicAdd(u2 + c, red[0]);
        if (pc[0]) atomicAdd(npos + c, (unsigned long long)pc[0]);
    }
}

}  // namespace

int eval_counts(const float* scores, long ld, const int64_t* labels, int n, int C, unsigned long long* conf, unsigned long long* u2,
                unsigned long long* npos, int64_t* preds, hipStream_t st) {
    if (n <= 0 || C <= 0) return MFVIT_EINVAL;
    if (conf) {
        MFVIT_LAUNCH(confusion_kernel, dim3((n + 255) / 256), dim3(256), 0, st, scores, ld, labels, n, C, conf, preds);
        MFVIT_CHECK_LAUNCH();
    }
    if (u2 && npos) {
        MFVIT_LAUNCH(auc_pairs_kernel, dim3((n + 255) / 256, C), dim3(256), 0, st, scores, ld, labels, n, u2, npos);
        MFVIT_CHECK_LAUNCH();
    }
    return MFVIT_OK;
}

}  // namespace mfvit
