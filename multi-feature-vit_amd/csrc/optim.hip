// Multi-tensor optimizer steps on gfx950: one (or two) launches over a device-resident chunk table instead of a Python
// loop of per-tensor ATen ops.  HBM-bound: every parameter / gradient / state element is touched once.
//   LARS  - moco_pretraining/moco/moco/optimizer.py:10-43 of the reference (trust ratio only for ndim > 1 tensors)
//   Adam / AdamW - torch.optim.Adam / AdamW semantics (MAIN_CA:455-459, MAIN_MOCO:338-345)
//   SGD   - torch.optim.SGD with momentum and L2 weight decay (MAIN_CA:445-448)
// Chunk table (int64, device): per chunk [tensor_id, p_ptr, g_ptr, s0_ptr, s1_ptr, count, flag]; pointers are to the
// chunk's first element.
#include "common.cuh"

namespace mfvit {

constexpr int CH = 7;

__global__ __launch_bounds__(256) void lars_norms_kernel(const long* __restrict__ tab, float wd, float* __restrict__ norms) {
    const long* e = tab + (long)blockIdx.x * CH;
    if (!e[6]) return;  // ndim <= 1: no trust ratio
    const float* p = (const float*)e[1];
    const float* g = (const float*)e[2];
    const long n = e[5];
    float a = 0.f, b = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float pv = p[i], d = fmaf(wd, pv, g[i]);
        a = fmaf(pv, pv, a);
        b = fmaf(d, d, b);
    }
    __shared__ float sc[4];
    a = block_sum<256>(a, sc);
    b = block_sum<256>(b, sc);
    if (threadIdx.x == 0) {
        atomicAdd(norms + 2 * e[0], a);
        atomicAdd(norms + 2 * e[0] + 1, b);
    }
}
__global__ __launch_bounds__(256) void lars_update_kernel(const long* __restrict__ tab, const float* __restrict__ norms, float lr, float wd,
                                                          float momentum, float trust) {
    const long* e = tab + (long)blockIdx.x * CH;
    float* p = (float*)e[1];
    const float* g = (const float*)e[2];
    float* mu = (float*)e[3];
    const long n = e[5];
    const bool big = e[6] != 0;
    float q = 1.f;
    if (big) {
        const float pn = sqrtf(norms[2 * e[0]]), un = sqrtf(norms[2 * e[0] + 1]);
        q = (pn > 0.f && un > 0.f) ? trust * pn / un : 1.f;     // optimizer.py:31-35
    }
    for (long i = threadIdx.x; i < n; i += 256) {
        float d = g[i];
        if (big) d = fmaf(wd, p[i], d) * q;
        const float m = fmaf(mu[i], momentum, d);
        mu[i] = m;
        p[i] = fmaf(-lr, m, p[i]);
    }
}
// flag bit0: decoupled weight decay (AdamW); step-dependent constants are precomputed on the host
__global__ __launch_bounds__(256) void adam_kernel(const long* __restrict__ tab, float lr, float beta1, float beta2, float eps, float wd,
                                                   float bc1, float bc2_sqrt) {
    const long* e = tab + (long)blockIdx.x * CH;
    float* p = (float*)e[1];
    const float* g = (const float*)e[2];
    float* m = (float*)e[3];
    float* v = (float*)e[4];
    const long n = e[5];
    const bool decoupled = e[6] & 1;
    for (long i = threadIdx.x; i < n; i += 256) {
        float pv = p[i], gv = g[i];
        if (decoupled) pv *= 1.f - lr * wd; else gv = fmaf(wd, pv, gv);
        const float mv = fmaf(beta1, m[i], (1.f - beta1) * gv);
        const float vv = fmaf(beta2, v[i], (1.f - beta2) * gv * gv);
        m[i] = mv;
        v[i] = vv;
        p[i] = pv - (lr / bc1) * mv / (sqrtf(vv) / bc2_sqrt + eps);
    }
}
__global__ __launch_bounds__(256) void sgd_kernel(const long* __restrict__ tab, float lr, float momentum, float wd, int first) {
    const long* e = tab + (long)blockIdx.x * CH;
    float* p = (float*)e[1];
    const float* g = (const float*)e[2];
    float* buf = (float*)e[3];
    const long n = e[5];
    for (long i = threadIdx.x; i < n; i += 256) {
        float d = fmaf(wd, p[i], g[i]);
        if (momentum != 0.f) {
            d = first ? d : fmaf(momentum, buf[i], d);
            buf[i] = d;
        }
        p[i] = fmaf(-lr, d, p[i]);
    }
}

// GradScaler.unscale_ (MAIN_MOCO:546-548 -> torch.cuda.amp.GradScaler.step): g *= inv_scale over every chunk of the table and
// *found_inf = 1 as soon as one gradient element is not finite (one flag store per wave that saw one; the flag is never cleared here).
__global__ __launch_bounds__(256) void amp_unscale_kernel(const long* __restrict__ tab, float inv_scale, float* __restrict__ found_inf) {
    const long* e = tab + (long)blockIdx.x * CH;
    float* g = (float*)e[2];
    const long n = e[5];
    bool bad = false;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.402823466e+38f);     // inf or nan
        g[i] = v * inv_scale;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.0f;
}

}  // namespace mfvit

using namespace mfvit;

extern "C" {

int mfvit_lars_step(const int64_t* table, int nchunks, int ntensors, float* norms, float lr, float weight_decay, float momentum,
                    float trust_coefficient, mfvit_stream_t stream) {
    if (!table || !norms || nchunks <= 0 || ntensors <= 0) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(norms, 0, sizeof(float) * 2 * ntensors, st) != hipSuccess) return MFVIT_ELAUNCH;
    MFVIT_LAUNCH(lars_norms_kernel, dim3(nchunks), dim3(256), 0, st, (const long*)table, weight_decay, norms);
    MFVIT_CHECK_LAUNCH();
    MFVIT_LAUNCH(lars_update_kernel, dim3(nchunks), dim3(256), 0, st, (const long*)table, norms, lr, weight_decay, momentum,
                       trust_coefficient);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_adam_step(const int64_t* table, int nchunks, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    mfvit_stream_t stream) {
    if (!table || nchunks <= 0 || step <= 0) return MFVIT_EINVAL;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    MFVIT_LAUNCH(adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const long*)table, lr, beta1, beta2, eps, weight_decay,
                       bc1, bc2s);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_amp_unscale(const int64_t* table, int nchunks, float inv_scale, float* found_inf, mfvit_stream_t stream) {
    if (!table || !found_inf || nchunks <= 0) return MFVIT_EINVAL;
    MFVIT_LAUNCH(amp_unscale_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const long*)table, inv_scale, found_inf);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_sgd_step(const int64_t* table, int nchunks, float lr, float momentum, float weight_decay, int first_step,
                   mfvit_stream_t stream) {
    if (!table || nchunks <= 0) return MFVIT_EINVAL;
    MFVIT_LAUNCH(sgd_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const long*)table, lr, momentum, weight_decay, first_step);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

}  // extern "C"
