// MoCo (v3 structure, v2 queue loss) step kernels for gfx950: BatchNorm1d of the projector / predictor MLPs (local
// statistics -> cross-rank combine -> apply, and the backward), L2 row normalisation, positive logits, row-wise
// cross entropy over the (1 + K)-wide InfoNCE logits, momentum (EMA) update over flat arenas.
//
// Reference (relative to /root/reference/moco_pretraining/moco/moco/builder_vit_mocov3structure_mocov2loss.py):
//   :62-78   _build_mlp: Linear(no bias) -> BatchNorm1d -> ReLU ... last BatchNorm1d(affine=False)   (SyncBN: MAIN_MOCO:297)
//   :83-89   momentum update  p_k = p_k * m + p_q * (1 - m)
//   :165,175 F.normalize(dim=1)     :183 l_pos = einsum('nc,nc->n')     MAIN_MOCO:330,535  CrossEntropyLoss over (n, 1+K)
// All of these are HBM / latency bound (n x 4096 activations, 64 MiB queue, 43 M parameter floats); the Linear layers
// themselves run on the MFMA GEMM kernels (gemm.hip).
#include "kernels.h"

namespace mfvit {

// ----------------------------------------------------------------------------------------------- BatchNorm1d
// local per-column mean and M2 = sum (x - mean_local)^2 over the n local rows; thread = column (coalesced rows)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int n, int C, float* __restrict__ mean,
                                                       float* __restrict__ m2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < n; ++r) s += to_f32(x[(long)r * C + c]);
    const float mu = s / (float)n;
    float q = 0.f;
    for (int r = 0; r < n; ++r) { const float d = to_f32(x[(long)r * C + c]) - mu; q = fmaf(d, d, q); }
    mean[c] = mu;
    m2[c] = q;
}
// The same with the ROWS spread over the workgroup as well (round 4): 256 threads = 16 row groups x 16 column quads, one workgroup per 64
// columns - with a thread per column and a serial loop over the rows the two passes were 2 n dependent loads per thread on 16 workgroups
// (48 us for n = 128, C = 4096; the backward sums 107 us).  Partial sums meet in LDS; still two passes (mean first, then the squared
// deviations: no cancellation).  Needs C % 4 == 0 (vector loads of four elements).
template <typename T> struct Quad { T v[4]; };
template <typename T> __device__ __forceinline__ void load_quad(const T* p, float (&f)[4]) {
    const Quad<T> q = *(const Quad<T>*)p;
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = to_f32(q.v[j]);
}
__device__ __forceinline__ void quad_reduce(float (&v)[4], float* red, int rg, int cg) {       // sums over the 16 row groups, result in every thread
    __syncthreads();                                                                           // (red may still be read from a previous reduction)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rg * 64 + 4 * cg + j] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g * 64 + 4 * cg + j];
        v[j] = t;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_tiled_kernel(const T* __restrict__ x, int n, int C, float* __restrict__ mean, float* __restrict__ m2) {
    __shared__ float red[16 * 64];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + 4 * cg;
    const bool ok = c < C;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int r = rg; r < n; r += 16) {
            float f[4];
            load_quad(x + (long)r * C + c, f);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += f[j];
        }
    quad_reduce(s, red, rg, cg);
    float mu[4], q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) mu[j] = s[j] / (float)n;
    if (ok)
        for (int r = rg; r < n; r += 16) {
            float f[4];
            load_quad(x + (long)r * C + c, f);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = f[j] - mu[j]; q[j] = fmaf(d, d, q[j]); }
        }
    quad_reduce(q, red, rg, cg);
    if (ok && rg == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { mean[c + j] = mu[j]; m2[c + j] = q[j]; }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_sums_tiled_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd, int relu, int n, int C,
                                                                float* __restrict__ s1, float* __restrict__ s2) {
    __shared__ float red[16 * 64];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + 4 * cg;
    const bool ok = c < C;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        float mu[4], is[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; is[j] = invstd[c + j]; }
        for (int r = rg; r < n; r += 16) {
            const long i = (long)r * C + c;
            float g[4], xv[4], yv[4] = {1.f, 1.f, 1.f, 1.f};
            load_quad(dy + i, g);
            load_quad(x + i, xv);
            if (relu) load_quad(y + i, yv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gg = (relu && !(yv[j] > 0.f)) ? 0.f : g[j];
                a[j] += gg;
                b[j] = fmaf(gg, (xv[j] - mu[j]) * is[j], b[j]);
            }
        }
    }
    quad_reduce(a, red, rg, cg);
    quad_reduce(b, red, rg, cg);
    if (ok && rg == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[c + j] = a[j]; s2[c + j] = b[j]; }
    }
}
// Chan's parallel combine of W rank-local (mean, M2, count) triples -> global mean, invstd; running-stat update.
__global__ __launch_bounds__(256) void bn_combine_kernel(const float* __restrict__ means, const float* __restrict__ m2s,
                                                         const float* __restrict__ counts, int W, int C, float eps, float momentum,
                                                         float* __restrict__ mean, float* __restrict__ invstd,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float N = 0.f, mu = 0.f, M2 = 0.f;
    for (int w = 0; w < W; ++w) {
        const float nb = counts[w], mb = means[(long)w * C + c], qb = m2s[(long)w * C + c];
        const float Nn = N + nb, d = mb - mu;
        mu += d * nb / Nn;
        M2 += qb + d * d * N * nb / Nn;
        N = Nn;
    }
    const float var = M2 / N;
    mean[c] = mu;
    invstd[c] = rsqrtf(var + eps);
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (N > 1.f ? M2 / (N - 1.f) : var);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu, T* __restrict__ y, long total, int C) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    float v = (to_f32(x[i]) - mean[c]) * invstd[c];
    if (gamma) v = fmaf(v, gamma[c], beta[c]);
    if (relu) v = fmaxf(v, 0.f);
    y[i] = from_f32<T>(v);
}
// backward pass 1: per-column sums of dy' and dy' * xhat (dy' = dy masked by the fused ReLU)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                          int n, int C, float* __restrict__ s1, float* __restrict__ s2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float mu = mean[c], is = invstd[c];
    float a = 0.f, b = 0.f;
    for (int r = 0; r < n; ++r) {
        const long i = (long)r * C + c;
        float g = to_f32(dy[i]);
        if (relu && !(to_f32(y[i]) > 0.f)) g = 0.f;
        a += g;
        b = fmaf(g, (to_f32(x[i]) - mu) * is, b);
    }
    s1[c] = a;
    s2[c] = b;
}
// backward pass 2: dx = gamma * invstd * (dy' - S1/N - xhat * S2/N) with the GLOBAL sums S1, S2 and count N
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, int relu, const float* __restrict__ S1,
                                                           const float* __restrict__ S2, float invN, T* __restrict__ dx, long total,
                                                           int C) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    float g = to_f32(dy[i]);
    if (relu && !(to_f32(y[i]) > 0.f)) g = 0.f;
    const float is = invstd[c], xh = (to_f32(x[i]) - mean[c]) * is;
    const float w = gamma ? gamma[c] : 1.f;
    dx[i] = from_f32<T>(w * is * (g - S1[c] * invN - xh * S2[c] * invN));
}

// ----------------------------------------------------------------------------------------------- row ops (wave per row)
// y = x / max(||x||, eps)   (F.normalize, dim=1);  also returns 1/norm for the backward
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv,
                                                         int n, int C, float eps) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = x[(long)r * C + c]; s = fmaf(v, v, s); }
    const float iv = 1.f / fmaxf(sqrtf(wave_sum(s)), eps);
    for (int c = lane; c < C; c += 64) y[(long)r * C + c] = x[(long)r * C + c] * iv;
    if (lane == 0) inv[r] = iv;
}
// dx = (dy - y (y . dy)) / norm
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv, float* __restrict__ dx, int n, int C) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(dy[(long)r * C + c], y[(long)r * C + c], s);
    s = wave_sum(s);
    const float iv = inv[r];
    for (int c = lane; c < C; c += 64) dx[(long)r * C + c] = (dy[(long)r * C + c] - y[(long)r * C + c] * s) * iv;
}
// out[r * ldo] = scale * (a[r] . b[r])      (l_pos, written straight into column 0 of the logits)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                     long ldo, float scale, int n, int C) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(a[(long)r * C + c], b[(long)r * C + c], s);
    s = wave_sum(s);
    if (lane == 0) out[(long)r * ldo] = s * scale;
}

// ----------------------------------------------------------------------------------------------- cross entropy over wide rows
// One block of 1024 threads per row: the row maximum, then the sum of exponentials, then dlogits = (softmax - onehot) * gscale; three sweeps
// over a row that stays in L2 (262 KB at K = 65,536), 16-byte accesses between a scalar head / tail (rows start wherever ld puts them).
// (Round 4: 256 threads, 4-byte accesses and an online maximum with two exponentials per element took 154 us for 128 rows.)
__global__ __launch_bounds__(1024) void ce_rows_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ target,
                                                       float* __restrict__ loss_sum, float* __restrict__ lse_out,
                                                       float* __restrict__ dlogits, long ldd, float gscale, int C) {
    __shared__ float red[16];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* z = logits + (long)r * ld;
    // [0, head): scalar, [head, head + 4 nv): float4, [head + 4 nv, C): scalar
    const int head = min(C, (int)((4 - (((uintptr_t)z >> 2) & 3)) & 3)), nv = (C - head) / 4, tail0 = head + 4 * nv;
    const float4* zv = (const float4*)(z + head);
    auto block_reduce = [&](float v, bool is_max) {
        v = is_max ? wave_max(v) : wave_sum(v);
        __syncthreads();
        if (lane == 0) red[w] = v;
        __syncthreads();
        float t = red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) t = is_max ? fmaxf(t, red[i]) : t + red[i];
        return t;
    };
    float m = -INFINITY;
    for (int i = tid; i < nv; i += 1024) { const float4 v = zv[i]; m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w)); }
    for (int c = tid; c < head; c += 1024) m = fmaxf(m, z[c]);
    for (int c = tail0 + tid; c < C; c += 1024) m = fmaxf(m, z[c]);
    const float M = block_reduce(m, true);
    float s = 0.f;
    for (int i = tid; i < nv; i += 1024) { const float4 v = zv[i]; s += __expf(v.x - M) + __expf(v.y - M) + __expf(v.z - M) + __expf(v.w - M); }
    for (int c = tid; c < head; c += 1024) s += __expf(z[c] - M);
    for (int c = tail0 + tid; c < C; c += 1024) s += __expf(z[c] - M);
    const float S = block_reduce(s, false);
    const float lse = M + __logf(S);
    const int t = (int)target[r];
    if (tid == 0) {
        if (lse_out) lse_out[r] = lse;                                  // (with the row log-sum-exps kept, ce_rows_loss_kernel adds the row losses in a fixed order)
        else atomicAdd(loss_sum, (lse - z[t]) * gscale);
    }
    if (dlogits) {
        float* d = dlogits + (long)r * ldd;
        const bool dvec = ((((uintptr_t)(d + head)) & 15) == 0);        // (the gradient row has the same phase when ldd == ld and the bases are aligned alike)
        if (dvec) {
            float4* dv = (float4*)(d + head);
            for (int i = tid; i < nv; i += 1024) {
                const float4 v = zv[i];
                const int c = head + 4 * i;
                float4 o;
                o.x = (__expf(v.x - lse) - (c == t ? 1.f : 0.f)) * gscale;
                o.y = (__expf(v.y - lse) - (c + 1 == t ? 1.f : 0.f)) * gscale;
                o.z = (__expf(v.z - lse) - (c + 2 == t ? 1.f : 0.f)) * gscale;
                o.w = (__expf(v.w - lse) - (c + 3 == t ? 1.f : 0.f)) * gscale;
                dv[i] = o;
            }
        } else {
            for (int c = head + tid; c < tail0; c += 1024) d[c] = (__expf(z[c] - lse) - (c == t ? 1.f : 0.f)) * gscale;
        }
        for (int c = tid; c < head; c += 1024) d[c] = (__expf(z[c] - lse) - (c == t ? 1.f : 0.f)) * gscale;
        for (int c = tail0 + tid; c < C; c += 1024) d[c] = (__expf(z[c] - lse) - (c == t ? 1.f : 0.f)) * gscale;
    }
}

// mean loss from the row log-sum-exps, one block, rows in a fixed order (256 interleaved sequences + a fixed tree): the same bits on every run
__global__ __launch_bounds__(256) void ce_rows_loss_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ target,
                                                           const float* __restrict__ lse, int n, float gscale, float* __restrict__ loss_mean) {
    __shared__ float sm[256];
    float s = 0.f;
    for (int r = threadIdx.x; r < n; r += 256) s += lse[r] - logits[(long)r * ld + target[r]];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_mean = sm[0] * gscale;
}

// ----------------------------------------------------------------------------------------------- EMA over a flat arena
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ dst, const float* __restrict__ src, float m, long n4, long n) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n4) {
        float4 d = ((float4*)dst)[i];
        const float4 s = ((const float4*)src)[i];
        const float om = 1.f - m;
        d.x = d.x * m + s.x * om; d.y = d.y * m + s.y * om; d.z = d.z * m + s.z * om; d.w = d.w * m + s.w * om;
        ((float4*)dst)[i] = d;
    } else {
        const long j = n4 * 4 + (i - n4);
        if (j < n) dst[j] = dst[j] * m + src[j] * (1.f - m);
    }
}

}  // namespace mfvit

using namespace mfvit;

extern "C" {

int mfvit_bn_stats(int dtype, const void* x, int n, int C, float* mean, float* m2, mfvit_stream_t stream) {
    if (!x || !mean || !m2 || n <= 0 || C <= 0) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && (size_t)x % 16 == 0) {                            // rows spread over the workgroup: one workgroup per 64 columns
        const dim3 grid((C + 63) / 64);
        if (dtype == MFVIT_BF16) MFVIT_LAUNCH(bn_stats_tiled_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, n, C, mean, m2);
        else if (dtype == MFVIT_F16) MFVIT_LAUNCH(bn_stats_tiled_kernel<f16>, grid, dim3(256), 0, st, (const f16*)x, n, C, mean, m2);
        else if (dtype == MFVIT_F32) MFVIT_LAUNCH(bn_stats_tiled_kernel<float>, grid, dim3(256), 0, st, (const float*)x, n, C, mean, m2);
        else return MFVIT_EINVAL;
        MFVIT_CHECK_LAUNCH();
        return MFVIT_OK;
    }
    if (dtype == MFVIT_BF16) MFVIT_LAUNCH(bn_stats_kernel<bf16>, dim3((C + 255) / 256), dim3(256), 0, st, (const bf16*)x, n, C, mean, m2);
    else if (dtype == MFVIT_F16) MFVIT_LAUNCH(bn_stats_kernel<f16>, dim3((C + 255) / 256), dim3(256), 0, st, (const f16*)x, n, C, mean, m2);
    else if (dtype == MFVIT_F32) MFVIT_LAUNCH(bn_stats_kernel<float>, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)x, n, C, mean, m2);
    else return MFVIT_EINVAL;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_bn_combine(const float* means, const float* m2s, const float* counts, int W, int C, float eps, float momentum, float* mean,
                     float* invstd, float* running_mean, float* running_var, mfvit_stream_t stream) {
    if (!means || !m2s || !counts || !mean || !invstd || W <= 0) return MFVIT_EINVAL;
    MFVIT_LAUNCH(bn_combine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, means, m2s, counts, W, C, eps, momentum,
                       mean, invstd, running_mean, running_var);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_bn_apply(int dtype, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                   void* y, int n, int C, mfvit_stream_t stream) {
    if (!x || !mean || !invstd || !y) return MFVIT_EINVAL;
    const long total = (long)n * C;
    const dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFVIT_BF16) MFVIT_LAUNCH(bn_apply_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, mean, invstd, gamma, beta, relu, (bf16*)y, total, C);
    else if (dtype == MFVIT_F16) MFVIT_LAUNCH(bn_apply_kernel<f16>, grid, dim3(256), 0, st, (const f16*)x, mean, invstd, gamma, beta, relu, (f16*)y, total, C);
    else if (dtype == MFVIT_F32) MFVIT_LAUNCH(bn_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)x, mean, invstd, gamma, beta, relu, (float*)y, total, C);
    else return MFVIT_EINVAL;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_bn_bwd_sums(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* invstd, int relu, int n,
                      int C, float* s1, float* s2, mfvit_stream_t stream) {
    if (!dy || !x || !mean || !invstd || !s1 || !s2 || (relu && !y)) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && (size_t)dy % 16 == 0 && (size_t)x % 16 == 0 && (!relu || (size_t)y % 16 == 0)) {
        const dim3 grid((C + 63) / 64);
        if (dtype == MFVIT_BF16) MFVIT_LAUNCH(bn_bwd_sums_tiled_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, (const bf16*)y, mean, invstd, relu, n, C, s1, s2);
        else if (dtype == MFVIT_F16) MFVIT_LAUNCH(bn_bwd_sums_tiled_kernel<f16>, grid, dim3(256), 0, st, (const f16*)dy, (const f16*)x, (const f16*)y, mean, invstd, relu, n, C, s1, s2);
        else if (dtype == MFVIT_F32) MFVIT_LAUNCH(bn_bwd_sums_tiled_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)y, mean, invstd, relu, n, C, s1, s2);
        else return MFVIT_EINVAL;
        MFVIT_CHECK_LAUNCH();
        return MFVIT_OK;
    }
    if (dtype == MFVIT_BF16) MFVIT_LAUNCH(bn_bwd_sums_kernel<bf16>, dim3((C + 255) / 256), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, (const bf16*)y, mean, invstd, relu, n, C, s1, s2);
    else if (dtype == MFVIT_F16) MFVIT_LAUNCH(bn_bwd_sums_kernel<f16>, dim3((C + 255) / 256), dim3(256), 0, st, (const f16*)dy, (const f16*)x, (const f16*)y, mean, invstd, relu, n, C, s1, s2);
    else if (dtype == MFVIT_F32) MFVIT_LAUNCH(bn_bwd_sums_kernel<float>, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)y, mean, invstd, relu, n, C, s1, s2);
    else return MFVIT_EINVAL;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_bn_bwd_apply(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* invstd, const float* gamma,
                       int relu, const float* S1, const float* S2, float inv_count, void* dx, int n, int C, mfvit_stream_t stream) {
    if (!dy || !x || !mean || !invstd || !S1 || !S2 || !dx || (relu && !y)) return MFVIT_EINVAL;
    const long total = (long)n * C;
    const dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MFVIT_BF16) MFVIT_LAUNCH(bn_bwd_apply_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, (const bf16*)y, mean, invstd, gamma, relu, S1, S2, inv_count, (bf16*)dx, total, C);
    else if (dtype == MFVIT_F16) MFVIT_LAUNCH(bn_bwd_apply_kernel<f16>, grid, dim3(256), 0, st, (const f16*)dy, (const f16*)x, (const f16*)y, mean, invstd, gamma, relu, S1, S2, inv_count, (f16*)dx, total, C);
    else if (dtype == MFVIT_F32) MFVIT_LAUNCH(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)y, mean, invstd, gamma, relu, S1, S2, inv_count, (float*)dx, total, C);
    else return MFVIT_EINVAL;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_l2norm_fwd(const float* x, float* y, float* inv_norm, int n, int C, float eps, mfvit_stream_t stream) {
    if (!x || !y || !inv_norm) return MFVIT_EINVAL;
    MFVIT_LAUNCH(l2norm_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, n, C, eps);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int n, int C, mfvit_stream_t stream) {
    if (!dy || !y || !inv_norm || !dx) return MFVIT_EINVAL;
    MFVIT_LAUNCH(l2norm_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, y, inv_norm, dx, n, C);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_rowdot(const float* a, const float* b, float* out, int64_t ldo, float scale, int n, int C, mfvit_stream_t stream) {
    if (!a || !b || !out) return MFVIT_EINVAL;
    MFVIT_LAUNCH(rowdot_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)ldo, scale, n, C);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int mfvit_cross_entropy_rows(const float* logits, int64_t ld, const int64_t* target, float* loss_mean, float* lse, float* dlogits,
                             int64_t ldd, int n, int C, mfvit_stream_t stream) {
    if (!logits || !target || !loss_mean || n <= 0 || C <= 0) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(loss_mean, 0, sizeof(float), st) != hipSuccess) return MFVIT_ELAUNCH;
    MFVIT_LAUNCH(ce_rows_kernel, dim3(n), dim3(1024), 0, st, logits, (long)ld, (const long*)target, loss_mean, lse, dlogits, (long)ldd,
                       1.0f / (float)n, C);
    MFVIT_CHECK_LAUNCH();
    if (lse) {       // (row log-sum-exps kept: the mean in a fixed order instead of one float atomic per row)
        MFVIT_LAUNCH(ce_rows_loss_kernel, dim3(1), dim3(256), 0, st, logits, (long)ld, (const long*)target, (const float*)lse, n, 1.0f / (float)n, loss_mean);
        MFVIT_CHECK_LAUNCH();
    }
    return MFVIT_OK;
}
int mfvit_ema_update(float* dst, const float* src, float m, int64_t n, mfvit_stream_t stream) {
    if (!dst || !src || n <= 0) return MFVIT_EINVAL;
    const long n4 = ((uintptr_t)dst % 16 == 0 && (uintptr_t)src % 16 == 0) ? n / 4 : 0;
    const long threads = n4 + (n - n4 * 4);
    MFVIT_LAUNCH(ema_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, m, n4, (long)n);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

}  // extern "C"
