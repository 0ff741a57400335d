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
__device__ __forceinline__ void adam_elem(float& pv, float gv, float& mv, float& vv, bool decoupled, float lr, float beta1, float beta2, float eps,
                                          float wd, float bc1, float bc2_sqrt) {
    if (decoupled) pv *= 1.f - lr * wd; else gv = fmaf(wd, pv, gv);
    mv = fmaf(beta1, mv, (1.f - beta1) * gv);
    vv = fmaf(beta2, vv, (1.f - beta2) * gv * gv);
    pv = pv - (lr / bc1) * mv / (sqrtf(vv) / bc2_sqrt + eps);
}
// 28 bytes of HBM traffic per element (p, g, m, v in; p, m, v out) and nothing else: 16-byte accesses, two independent groups per thread in
// flight (the 4-byte form ran at 2.2 TB/s: 0.56 ms per step for the 45 M parameters of the two-encoder model)
__global__ __launch_bounds__(256) void adam_kernel(const long* __restrict__ tab, float lr, float beta1, float beta2, float eps, float wd,
                                                   float bc1, float bc2_sqrt) {
    const long* e = tab + (long)blockIdx.x * CH;
    float* p = (float*)e[1];
    const float* g = (const float*)e[2];
    float* m = (float*)e[3];
    float* v = (float*)e[4];
    const long n = e[5];
    const bool decoupled = e[6] & 1;
    long done = 0;
    if ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0) {
        const long n4 = n >> 2;
        float4* p4 = (float4*)p;
        const float4* g4 = (const float4*)g;
        float4* m4 = (float4*)m;
        float4* v4 = (float4*)v;
        for (long i = threadIdx.x; i < n4; i += 512) {
            const long j = i + 256;
            const bool two = j < n4;
            float4 pa = p4[i], ga = g4[i], ma = m4[i], va = v4[i];
            float4 pb = pa, gb = ga, mb = ma, vb = va;
            if (two) { pb = p4[j]; gb = g4[j]; mb = m4[j]; vb = v4[j]; }
            adam_elem(pa.x, ga.x, ma.x, va.x, decoupled, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
            adam_elem(pa.y, ga.y, ma.y, va.y, decoupled, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
            adam_elem(pa.z, ga.z, ma.z, va.z, decoupled, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
            adam_elem(pa.w, ga.w, ma.w, va.w, decoupled, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
            p4[i] = pa; m4[i] = ma; v4[i] = va;
            if (two) {
                adam_elem(pb.x, gb.x, mb.x, vb.x, decoupled, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
                adam_elem(pb.y, gb.y, mb.y, vb.y, decoupled, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
                adam_elem(pb.z, gb.z, mb.z, vb.z, decoupled, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
                adam_elem(pb.w, gb.w, mb.w, vb.w, decoupled, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
                p4[j] = pb; m4[j] = mb; v4[j] = vb;
            }
        }
        done = n4 << 2;
    }
    for (long i = done + threadIdx.x; i < n; i += 256) {
        float pv = p[i], mv = m[i], vv = v[i];
        adam_elem(pv, g[i], mv, vv, decoupled, lr, beta1, beta2, eps, wd, bc1, bc2_sqrt);
        m[i] = mv;
        v[i] = vv;
        p[i] = pv;
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
