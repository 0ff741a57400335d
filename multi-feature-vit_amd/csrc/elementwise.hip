// HBM-bound helper kernels of the encoder path (gfx950): patch gather, cls row, LayerNorm rows (fwd / bwd),
// weight cast + transpose, small linear heads, small-C cross entropy.  All wave64, 16-B vector access where the
// layout allows it.  Roofline for every kernel here: HBM bytes (each tensor touched once).
#include "kernels.h"

namespace mfvit {

// ---------------------------------------------------------------------------------------------- im2col
// img (B,3,H,W) f32 NCHW -> P [B*gh*gw][3*16*16] (k = c*256 + py*16 + px, the flatten order of Conv2d weight
// (D,3,16,16)); one thread moves 8 consecutive pixels (32 B in, 16/32 B out).
template <typename T>
__global__ void im2col16_kernel(const float* __restrict__ img, T* __restrict__ P, int B, int Hh, int Ww) {
    const int gh = Hh / 16, gw = Ww / 16;
    const long total = (long)B * gh * gw * 3 * 16 * 2;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int half = q & 1;
        long r = q >> 1;
        const int py = r % 16; r /= 16;
        const int c = r % 3; r /= 3;
        const int pw = r % gw; r /= gw;
        const int ph = r % gh; r /= gh;
        const int b = (int)r;
        const float* s = img + (((long)b * 3 + c) * Hh + ph * 16 + py) * Ww + pw * 16 + half * 8;
        const float4 v0 = *(const float4*)s, v1 = *(const float4*)(s + 4);
        const int k = c * 256 + py * 16 + half * 8;             // logical column (a multiple of 8: inside one 32-group)
        T* d = P + ((long)(b * gh + ph) * gw + pw) * 768 * elems_per<T>::value + (is_split<T>::value ? split_col(k) : k);
        if constexpr (sizeof(T) == 2) {
            typedef typename Vec4<T>::elem E;
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            typename Vec8<T>::type o, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const E h = (E)f[j];
                o[j] = h;
                if constexpr (is_split<T>::value) l[j] = (E)(f[j] - (float)h);
            }
            *(typename Vec8<T>::type*)d = o;
            if constexpr (is_split<T>::value) *(typename Vec8<T>::type*)((E*)d + 32) = l;
        } else {
            *(float4*)d = v0;
            *(float4*)(d + 4) = v1;
        }
    }
}
int im2col16(int dtype, const float* img, void* P, int B, int H, int W, hipStream_t st) {
    if (H % 16 || W % 16) return MFVIT_EINVAL;
    const long total = (long)B * (H / 16) * (W / 16) * 96;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == MFVIT_BF16)
        MFVIT_LAUNCH(im2col16_kernel<bf16>, dim3(blocks), dim3(256), 0, st, img, (bf16*)P, B, H, W);
    else if (dtype == MFVIT_BF16X3)
        MFVIT_LAUNCH(im2col16_kernel<sbf16>, dim3(blocks), dim3(256), 0, st, img, (sbf16*)P, B, H, W);
    else if (dtype == MFVIT_F16)
        MFVIT_LAUNCH(im2col16_kernel<f16>, dim3(blocks), dim3(256), 0, st, img, (f16*)P, B, H, W);
    else if (dtype == MFVIT_F32)
        MFVIT_LAUNCH(im2col16_kernel<float>, dim3(blocks), dim3(256), 0, st, img, (float*)P, B, H, W);
    else
        return MFVIT_EINVAL;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// ------------------------------------------------------------------------------------- LayerNorm rows (fwd)
// One wave per row.  x_row = in0[row*ld0] (+ in1[(row % mod1)*ld1] if in1).  Writes optional xout (f32), y (T or f32),
// mean, rstd.  Rows are addressed through (row_stride, row_off): global row g = r * row_stride + row_off.
template <typename T, int NPL>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ in0, long ld0, const float* __restrict__ in1, long ld1,
                                                      int mod1, float* __restrict__ xout, long ldx, void* __restrict__ y, long ldy, int y_f32,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int rows, int row_stride,
                                                      int row_off, int in0_bcast) {
    constexpr int N = NPL * 64;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long g = (long)r * row_stride + row_off;
    float v[NPL];
    const float* p0 = in0 + (in0_bcast ? 0 : g * ld0);
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = p0[lane + 64 * i];
    if (in1) {
        const float* p1 = in1 + (long)(mod1 ? g % mod1 : g) * ld1;
#pragma unroll
        for (int i = 0; i < NPL; ++i) v[i] += p1[lane + 64 * i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) s += v[i];
    const float mu = wave_sum(s) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / N) + eps);
    if (lane == 0) {
        if (mean) mean[g] = mu;
        if (rstd) rstd[g] = rs;
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = lane + 64 * i;
        if (xout) xout[g * ldx + n] = v[i];
        const float o = (v[i] - mu) * rs * gamma[n] + beta[n];
        if (y) {
            if (y_f32) ((float*)y)[g * ldy + n] = o;
            else store_elem<T>((T*)y + g * ldy, n, o);
        }
    }
}
int ln_rows(int dtype, int N, const float* in0, long ld0, const float* in1, long ld1, int mod1, float* xout, long ldx, void* y, long ldy,
            int y_f32, const float* gamma, const float* beta, float eps, float* mean, float* rstd, int rows, int row_stride, int row_off,
            int in0_bcast, hipStream_t st) {
    if (rows <= 0) return MFVIT_OK;
    const dim3 grid((rows + 3) / 4), blk(256);
#define MFVIT_LN(TT, NPL)                                                                                                        \
    MFVIT_LAUNCH((ln_rows_kernel<TT, NPL>), grid, blk, 0, st, in0, ld0, in1, ld1, mod1, xout, ldx, y, ldy, y_f32, gamma, beta, eps, \
                       mean, rstd, rows, row_stride, row_off, in0_bcast)
#define MFVIT_LN_N(NPL)                                  \
    switch (dtype) {                                     \
        case MFVIT_BF16: MFVIT_LN(bf16, NPL); break;     \
        case MFVIT_BF16X3: MFVIT_LN(sbf16, NPL); break;  \
        case MFVIT_F16: MFVIT_LN(f16, NPL); break;       \
        case MFVIT_F32: MFVIT_LN(float, NPL); break;     \
        default: return MFVIT_EINVAL;                    \
    }
    if (N == 384) { MFVIT_LN_N(6) }
    else if (N == 768) { MFVIT_LN_N(12) }
    else return MFVIT_EINVAL;
#undef MFVIT_LN_N
#undef MFVIT_LN
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// ------------------------------------------------------------------------------------- GEMM output + residual -> LayerNorm rows (unfused path)
// What the row-complete GEMM kernels fuse (gemm_rowp.hip / gemm_nt_row: N = 384 only), as a separate row pass behind a plain tile GEMM - the path of
// every embed dim the row kernels are not built for (vit_base: 768) and an A/B partner at small M:
//   v = tin[g] (operand type T: the GEMM's output incl. bias) (+ pos[(go % pmod)] f32) (+ res[go] f32);  x[go] = v;  y[go] = LN(v);  mean, rstd [go]
// with the output row go = (g / rin) * rout + roff + g % rin  (rin = 0: go = g) - the patch embedding writes the np patch rows of image b behind
// its cls row (rin = np, rout = T, roff = 1; pos = pos_embed, pmod = T).  One wave per row.
template <typename T, int NPL>
__global__ __launch_bounds__(256) void add_ln_rows_kernel(const T* __restrict__ tin, long ldt, const float* __restrict__ pos, long ldp, int pmod,
                                                          const float* __restrict__ res, long ldres, int rin, int rout, int roff,
                                                          float* __restrict__ xout, long ldx, void* __restrict__ y, long ldy, int y_f32,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int rows) {
    constexpr int N = NPL * 64;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= rows) return;
    const long go = rin ? (long)(g / rin) * rout + roff + g % rin : g;
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = lane + 64 * i;
        float t = load_elem<T>(tin + (long)g * ldt, n);
        if (pos) t += pos[(pmod ? go % pmod : go) * ldp + n];
        v[i] = res ? res[go * ldres + n] + t : t;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) s += v[i];
    const float mu = wave_sum(s) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { const float dd = v[i] - mu; q += dd * dd; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / N) + eps);
    if (lane == 0) {
        if (mean) mean[go] = mu;
        if (rstd) rstd[go] = rs;
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = lane + 64 * i;
        if (xout) xout[go * ldx + n] = v[i];
        const float o = (v[i] - mu) * rs * gamma[n] + beta[n];
        if (y_f32) ((float*)y)[go * ldy + n] = o;
        else store_elem<T>((T*)y + go * ldy, n, o);
    }
}
int add_ln_rows(int dtype, int N, const void* tin, long ldt, const float* pos, long ldp, int pmod, const float* res, long ldres, int rin, int rout,
                int roff, float* xout, long ldx, void* y, long ldy, int y_f32, const float* gamma, const float* beta, float eps, float* mean,
                float* rstd, int rows, hipStream_t st) {
    if (rows <= 0) return MFVIT_OK;
    if (!tin || !y || !gamma || !beta || (rin && (rout < rin || roff < 0))) return MFVIT_EINVAL;
    const dim3 grid((rows + 3) / 4), blk(256);
#define MFVIT_ALN(TT, NPL)                                                                                                                 \
    MFVIT_LAUNCH((add_ln_rows_kernel<TT, NPL>), grid, blk, 0, st, (const TT*)tin, ldt, pos, ldp, pmod, res, ldres, rin, rout, roff, xout, ldx, y, \
                 ldy, y_f32, gamma, beta, eps, mean, rstd, rows)
#define MFVIT_ALN_N(NPL)                                  \
    switch (dtype) {                                      \
        case MFVIT_BF16: MFVIT_ALN(bf16, NPL); break;     \
        case MFVIT_BF16X3: MFVIT_ALN(sbf16, NPL); break;  \
        case MFVIT_F16: MFVIT_ALN(f16, NPL); break;       \
        case MFVIT_F32: MFVIT_ALN(float, NPL); break;     \
        default: return MFVIT_EINVAL;                     \
    }
    if (N == 384) { MFVIT_ALN_N(6) }
    else if (N == 768) { MFVIT_ALN_N(12) }
    else return MFVIT_EINVAL;
#undef MFVIT_ALN_N
#undef MFVIT_ALN
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// ------------------------------------------------------------------------------------- dropout + residual + LayerNorm rows (TransFuser GPT)
// One wave per row:  v = drop(t) with t = tin[row] (operand type, a GEMM output incl. bias)  or  a0[row] (+ a1[row % mod1]) (f32);
// x = res[row] + v (res optional); writes x (f32), y = LN(x) (T or f32), mean, rstd.  The keep mask of element (row, n) is
// drop_mul(drop, row * N + n).  This is `x = x + resid_drop(proj(y))` / `x + mlp(..)[Dropout]` / `drop(pos_emb + tokens)` of
// fuseattention.py:57,71-72,187 with the LayerNorm that follows; used only while a dropout site is active.
template <typename T, int NPL>
__global__ __launch_bounds__(256) void drop_add_ln_rows_kernel(const T* __restrict__ tin, long ldt, const float* __restrict__ a0, long lda0,
                                                               const float* __restrict__ a1, long lda1, int mod1, const float* __restrict__ res,
                                                               long ldres, DropP drop, float* __restrict__ xout, long ldx, void* __restrict__ y,
                                                               long ldy, int y_f32, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float* __restrict__ mean,
                                                               float* __restrict__ rstd, int rows) {
    constexpr int N = NPL * 64;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= rows) return;
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = lane + 64 * i;
        float t = tin ? load_elem<T>(tin + (long)g * ldt, n) : a0[(long)g * lda0 + n];
        if (!tin && a1) t += a1[(long)(mod1 ? g % mod1 : g) * lda1 + n];
        t *= drop_mul(drop, (unsigned)g * (unsigned)N + (unsigned)n);
        v[i] = res ? res[(long)g * ldres + n] + t : t;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) s += v[i];
    const float mu = wave_sum(s) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { const float dd = v[i] - mu; q += dd * dd; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / N) + eps);
    if (lane == 0) {
        if (mean) mean[g] = mu;
        if (rstd) rstd[g] = rs;
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int n = lane + 64 * i;
        if (xout) xout[(long)g * ldx + n] = v[i];
        const float o = (v[i] - mu) * rs * gamma[n] + beta[n];
        if (y_f32) ((float*)y)[(long)g * ldy + n] = o;
        else store_elem<T>((T*)y + (long)g * ldy, n, o);
    }
}
int drop_add_ln_rows(int dtype, int N, const void* tin, long ldt, const float* a0, long lda0, const float* a1, long lda1, int mod1,
                     const float* res, long ldres, DropP drop, float* xout, long ldx, void* y, long ldy, int y_f32, const float* gamma,
                     const float* beta, float eps, float* mean, float* rstd, int rows, hipStream_t st) {
    if (rows <= 0) return MFVIT_OK;
    if (N != 384 || (!tin && !a0) || !y || (double)rows * N >= 4294967296.0) return MFVIT_EINVAL;
    const dim3 grid((rows + 3) / 4), blk(256);
#define MFVIT_DAL(TT)                                                                                                                      \
    MFVIT_LAUNCH((drop_add_ln_rows_kernel<TT, 6>), grid, blk, 0, st, (const TT*)tin, ldt, a0, lda0, a1, lda1, mod1, res, ldres, drop, xout, ldx, y, \
                 ldy, y_f32, gamma, beta, eps, mean, rstd, rows)
    switch (dtype) {
        case MFVIT_BF16: MFVIT_DAL(bf16); break;
        case MFVIT_BF16X3: MFVIT_DAL(sbf16); break;
        case MFVIT_F16: MFVIT_DAL(f16); break;
        default: return MFVIT_EINVAL;
    }
#undef MFVIT_DAL
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
// dst[row][n] = src[row][n] * mask(row * N + n) / (1 - p): the gradient entering a dropped branch (operand type -> operand type), or
// d tokens = d x_0 * mask (f32 -> f32; T ignored)
template <typename T, bool F32>
__global__ __launch_bounds__(256) void mask_scale_rows_kernel(const void* __restrict__ src, long lds_, void* __restrict__ dst, long ldd, DropP drop,
                                                              int rows, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * N) return;
    const int g = (int)(i / N), n = (int)(i % N);
    const float mk = drop_mul(drop, (unsigned)i);
    if constexpr (F32) ((float*)dst)[(long)g * ldd + n] = ((const float*)src)[(long)g * lds_ + n] * mk;
    else store_elem<T>((T*)dst + (long)g * ldd, n, load_elem<T>((const T*)src + (long)g * lds_, n) * mk);
}
int mask_scale_rows(int dtype, bool f32, const void* src, long lds_, void* dst, long ldd, DropP drop, int rows, int N, hipStream_t st) {
    if (rows <= 0) return MFVIT_OK;
    if ((double)rows * N >= 4294967296.0) return MFVIT_EINVAL;
    const dim3 grid((unsigned)(((long)rows * N + 255) / 256)), blk(256);
    if (f32) MFVIT_LAUNCH((mask_scale_rows_kernel<float, true>), grid, blk, 0, st, src, lds_, dst, ldd, drop, rows, N);
    else if (dtype == MFVIT_BF16) MFVIT_LAUNCH((mask_scale_rows_kernel<bf16, false>), grid, blk, 0, st, src, lds_, dst, ldd, drop, rows, N);
    else if (dtype == MFVIT_BF16X3) MFVIT_LAUNCH((mask_scale_rows_kernel<sbf16, false>), grid, blk, 0, st, src, lds_, dst, ldd, drop, rows, N);
    else if (dtype == MFVIT_F16) MFVIT_LAUNCH((mask_scale_rows_kernel<f16, false>), grid, blk, 0, st, src, lds_, dst, ldd, drop, rows, N);
    else return MFVIT_EINVAL;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// ------------------------------------------------------------------------------------- LayerNorm rows (bwd)
// dx = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)) (+ dres);  column sums: dgamma, dbeta, sum(dx).
// One wave per row, rows grid-strided; column partials are kept per lane and reduced over the block's 4 waves in LDS.
template <typename T, int NPL>
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ dres, long ldres,
                                                          float* __restrict__ dx, long lddx, T* __restrict__ dxT, long lddxT,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dcol,
                                                          float* __restrict__ cpart, int rows, int row_stride, int row_off,
                                                          const T* __restrict__ dyT, long lddyT) {
    // (dyT != NULL: dy comes in the operand type - the output of a plain tile GEMM, the unfused path - instead of f32)
    constexpr int N = NPL * 64;
    __shared__ float red[3][4][N];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float gm[NPL], ag[NPL], ab[NPL], ax[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) { gm[i] = gamma[lane + 64 * i]; ag[i] = ab[i] = ax[i] = 0.f; }
    for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
        const long g = (long)r * row_stride + row_off;
        const float mu = mean[g], rs = rstd[g];
        float d[NPL], h[NPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int n = lane + 64 * i;
            d[i] = dyT ? load_elem<T>(dyT + g * lddyT, n) : dy[g * lddy + n];
            h[i] = (x[g * ldx + n] - mu) * rs;
            const float t = d[i] * gm[i];
            s1 += t;
            s2 += t * h[i];
            ag[i] += d[i] * h[i];
            ab[i] += d[i];
        }
        const float c1 = wave_sum(s1) * (1.0f / N), c2 = wave_sum(s2) * (1.0f / N);
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int n = lane + 64 * i;
            float o = rs * (d[i] * gm[i] - c1 - h[i] * c2);
            if (dres) o += dres[g * ldres + n];
            if (dx) dx[g * lddx + n] = o;
            if (dxT) store_elem<T>(dxT + g * lddxT, n, o);
            ax[i] += o;
        }
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        red[0][w][lane + 64 * i] = ag[i];
        red[1][w][lane + 64 * i] = ab[i];
        red[2][w][lane + 64 * i] = ax[i];
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        const float a = red[0][0][n] + red[0][1][n] + red[0][2][n] + red[0][3][n];
        const float b = red[1][0][n] + red[1][1][n] + red[1][2][n] + red[1][3][n];
        const float c = red[2][0][n] + red[2][1][n] + red[2][2][n] + red[2][3][n];
        if (cpart) {   // per-block partials [block][3][N] (reduced by colpart_reduce): no contended atomics
            cpart[((long)blockIdx.x * 3 + 0) * N + n] = a;
            cpart[((long)blockIdx.x * 3 + 1) * N + n] = b;
            cpart[((long)blockIdx.x * 3 + 2) * N + n] = c;
        } else {
            if (dgamma) atomicAdd(dgamma + n, a);
            if (dbeta) atomicAdd(dbeta + n, b);
            if (dcol) atomicAdd(dcol + n, c);
        }
    }
}
int ln_bwd_rows(int dtype, int N, const float* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd,
                const float* gamma, const float* dres, long ldres, float* dx, long lddx, void* dxT, long lddxT, float* dgamma, float* dbeta,
                float* dcol, float* cpart, int rows, int row_stride, int row_off, hipStream_t st, const void* dyT, long lddyT) {
    if (rows <= 0) return MFVIT_OK;
    if (!dy && !dyT) return MFVIT_EINVAL;
    int blocks = (rows + 3) / 4;
    const int cap = cpart ? 256 : 1024;
    if (blocks > cap) blocks = cap;
#define MFVIT_LNB(TT, NPL)                                                                                                         \
    MFVIT_LAUNCH((ln_bwd_rows_kernel<TT, NPL>), dim3(blocks), dim3(256), 0, st, dy, lddy, x, ldx, mean, rstd, gamma, dres, ldres, dx, \
                       lddx, (TT*)dxT, lddxT, dgamma, dbeta, dcol, cpart, rows, row_stride, row_off, (const TT*)dyT, lddyT)
#define MFVIT_LNB_N(NPL)                                  \
    switch (dtype) {                                      \
        case MFVIT_BF16: MFVIT_LNB(bf16, NPL); break;     \
        case MFVIT_BF16X3: MFVIT_LNB(sbf16, NPL); break;  \
        case MFVIT_F16: MFVIT_LNB(f16, NPL); break;       \
        case MFVIT_F32: MFVIT_LNB(float, NPL); break;     \
        default: return MFVIT_EINVAL;                     \
    }
    if (N == 384) { MFVIT_LNB_N(6) }
    else if (N == 768) { MFVIT_LNB_N(12) }
    else return MFVIT_EINVAL;
#undef MFVIT_LNB_N
#undef MFVIT_LNB
    MFVIT_CHECK_LAUNCH();
    if (cpart) return colpart_reduce(cpart, blocks, N, 3, dgamma, dbeta, dcol, st);
    return MFVIT_OK;
}

// --------------------------------------------------------------------------------------- cast + transpose
// src f32 [R][C] -> dst T [R][C] (optional) and dstT T [C][R] (optional); 64x64 tiles through LDS.
// blockIdx.z = batch index (same-shaped matrices at fixed strides: the 12 blocks of an encoder in one launch)
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, T* __restrict__ dst, T* __restrict__ dstT, int R,
                                                             int C, long s_src, long s_dst, long s_dstT) {
    __shared__ float tile[64][65];
    src += blockIdx.z * s_src;
    if (dst) dst = (T*)((char*)dst + blockIdx.z * s_dst);
    if (dstT) dstT = (T*)((char*)dstT + blockIdx.z * s_dstT);
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int q = threadIdx.x; q < 64 * 64; q += 256) {
        const int r = q >> 6, c = q & 63;
        float v = 0.f;
        if (r0 + r < R && c0 + c < C) {
            v = src[(long)(r0 + r) * C + c0 + c];
            if (dst) store_elem<T>(dst + (long)(r0 + r) * C * elems_per<T>::value, c0 + c, v);
        }
        tile[r][c] = v;
    }
    __syncthreads();
    if (dstT) {
        for (int q = threadIdx.x; q < 64 * 64; q += 256) {
            const int c = q >> 6, r = q & 63;
            if (r0 + r < R && c0 + c < C) store_elem<T>(dstT + (long)(c0 + c) * R * elems_per<T>::value, r0 + r, tile[r][c]);
        }
    }
}
// four logical columns n .. n+3 (n % 4 == 0) of one row: one 8-byte store per part (16 bytes for float)
template <typename T> __device__ __forceinline__ void store_elem4(T* row, int n, float v0, float v1, float v2, float v3) {
    if constexpr (std::is_same<T, float>::value) {
        *(float4*)(row + n) = make_float4(v0, v1, v2, v3);
    } else if constexpr (is_split<T>::value) {
        bf16 h[4], l[4];
        cvt_pair<bf16, true>(v0, v1, h[0], h[1], l[0], l[1]);
        cvt_pair<bf16, true>(v2, v3, h[2], h[3], l[2], l[3]);
        bf16* r = (bf16*)row + split_col(n);
        *(bf16x4*)r = bf16x4{h[0], h[1], h[2], h[3]};
        *(bf16x4*)(r + 32) = bf16x4{l[0], l[1], l[2], l[3]};
    } else {
        T h[4], l[4];
        cvt_pair<T, false>(v0, v1, h[0], h[1], l[0], l[1]);
        cvt_pair<T, false>(v2, v3, h[2], h[3], l[2], l[3]);
        typedef typename Vec4<T>::type V;
        *(V*)(row + n) = V{h[0], h[1], h[2], h[3]};
    }
}
// The same with four elements per thread (R % 4 == 0, C % 4 == 0, 16-byte aligned bases and strides): a float4 load, 8-byte stores
// on both outputs; the transposed read of the tile is conflict-free ((4 r' + j) * 65 + c over a wave's 16 r' x 4 c = 64 banks).  The
// element-wise kernel above moved 510 MB per optimizer step (two encoders, split bf16) in 0.31 ms.
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose4_kernel(const float* __restrict__ src, T* __restrict__ dst, T* __restrict__ dstT, int R,
                                                              int C, long s_src, long s_dst, long s_dstT) {
    __shared__ float tile[64][65];
    src += blockIdx.z * s_src;
    if (dst) dst = (T*)((char*)dst + blockIdx.z * s_dst);
    if (dstT) dstT = (T*)((char*)dstT + blockIdx.z * s_dstT);
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = threadIdx.x + 256 * k, r = q >> 4, c = (q & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + r < R && c0 + c < C) {
            v = *(const float4*)(src + (long)(r0 + r) * C + c0 + c);
            if (dst) store_elem4<T>(dst + (long)(r0 + r) * C * elems_per<T>::value, c0 + c, v.x, v.y, v.z, v.w);
        }
        tile[r][c] = v.x;
        tile[r][c + 1] = v.y;
        tile[r][c + 2] = v.z;
        tile[r][c + 3] = v.w;
    }
    __syncthreads();
    if (dstT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = threadIdx.x + 256 * k, c = q >> 4, r = (q & 15) * 4;
            if (r0 + r < R && c0 + c < C)
                store_elem4<T>(dstT + (long)(c0 + c) * R * elems_per<T>::value, r0 + r, tile[r][c], tile[r + 1][c], tile[r + 2][c], tile[r + 3][c]);
        }
    }
}
// strides: s_src in floats, s_dst / s_dstT in BYTES (shadow layouts are byte-addressed)
int cast_transpose_batched(int dtype, const float* src, void* dst, void* dstT, int R, int C, int nb, long s_src, long s_dst, long s_dstT,
                           hipStream_t st) {
    const dim3 grid((C + 63) / 64, (R + 63) / 64, nb);
    if (R % 4 == 0 && C % 4 == 0 && s_src % 4 == 0 && s_dst % 16 == 0 && s_dstT % 16 == 0 &&
        (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)dstT) & 15) == 0 && !(dtype == MFVIT_BF16X3 && ((dst && C % 32) || (dstT && R % 32)))) {
        switch (dtype) {
            case MFVIT_BF16: MFVIT_LAUNCH(cast_transpose4_kernel<bf16>, grid, dim3(256), 0, st, src, (bf16*)dst, (bf16*)dstT, R, C, s_src, s_dst, s_dstT); break;
            case MFVIT_BF16X3: MFVIT_LAUNCH(cast_transpose4_kernel<sbf16>, grid, dim3(256), 0, st, src, (sbf16*)dst, (sbf16*)dstT, R, C, s_src, s_dst, s_dstT); break;
            case MFVIT_F16: MFVIT_LAUNCH(cast_transpose4_kernel<f16>, grid, dim3(256), 0, st, src, (f16*)dst, (f16*)dstT, R, C, s_src, s_dst, s_dstT); break;
            case MFVIT_F32: MFVIT_LAUNCH(cast_transpose4_kernel<float>, grid, dim3(256), 0, st, src, (float*)dst, (float*)dstT, R, C, s_src, s_dst, s_dstT); break;
            default: return MFVIT_EINVAL;
        }
        MFVIT_CHECK_LAUNCH();
        return MFVIT_OK;
    }
    if (dtype == MFVIT_BF16)
        MFVIT_LAUNCH(cast_transpose_kernel<bf16>, grid, dim3(256), 0, st, src, (bf16*)dst, (bf16*)dstT, R, C, s_src, s_dst, s_dstT);
    else if (dtype == MFVIT_BF16X3) {   // split layout: both R and C are column counts of one of the outputs -> multiples of 32
        if ((dst && C % 32) || (dstT && R % 32)) return MFVIT_EINVAL;
        MFVIT_LAUNCH(cast_transpose_kernel<sbf16>, grid, dim3(256), 0, st, src, (sbf16*)dst, (sbf16*)dstT, R, C, s_src, s_dst, s_dstT);
    } else if (dtype == MFVIT_F16)
        MFVIT_LAUNCH(cast_transpose_kernel<f16>, grid, dim3(256), 0, st, src, (f16*)dst, (f16*)dstT, R, C, s_src, s_dst, s_dstT);
    else if (dtype == MFVIT_F32)
        MFVIT_LAUNCH(cast_transpose_kernel<float>, grid, dim3(256), 0, st, src, (float*)dst, (float*)dstT, R, C, s_src, s_dst, s_dstT);
    else
        return MFVIT_EINVAL;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int cast_transpose(int dtype, const float* src, void* dst, void* dstT, int R, int C, hipStream_t st) {
    return cast_transpose_batched(dtype, src, dst, dstT, R, C, 1, 0, 0, 0, st);
}

// --------------------------------------------------------------------------------------- small linear heads
// y[m][n] = x[m*ldx .. +K] . W[n][K] + b[n]   (f32; one wave per output; for heads with a handful of classes)
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ W,
                                                               const float* __restrict__ b, float* __restrict__ y, long ldy, int M, int N,
                                                               int K, int accumulate) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= M * N) return;
    const int m = o / N, n = o % N;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(x[(long)m * ldx + k], W[(long)n * K + k], s);
    s = wave_sum(s);
    if (lane == 0) {
        s += b ? b[n] : 0.f;
        if (accumulate) y[(long)m * ldy + n] += s; else y[(long)m * ldy + n] = s;
    }
}
// dx[m][k] (+)= sum_n dy[m][n] W[n][k];  dW[n][k] += sum_m dy[m][n] x[m][k];  db[n] += sum_m dy[m][n]
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                               const float* __restrict__ W, float* __restrict__ dx, long lddx,
                                                               int dx_accumulate, float* __restrict__ dW, float* __restrict__ db, int M,
                                                               int N, int K) {
    const long id = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (dx && id < (long)M * K) {
        const int m = id / K, k = id % K;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s = fmaf(dy[(long)m * lddy + n], W[(long)n * K + k], s);
        if (dx_accumulate) dx[(long)m * lddx + k] += s; else dx[(long)m * lddx + k] = s;
    }
    if (dW && id < (long)N * K) {
        const int n = id / K, k = id % K;
        float s = 0.f;
        for (int m = 0; m < M; ++m) s = fmaf(dy[(long)m * lddy + n], x[(long)m * ldx + k], s);
        dW[(long)n * K + k] += s;
    }
    if (db && id < N) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += dy[(long)m * lddy + id];
        db[id] += s;
    }
}
int linear_small_fwd(const float* x, long ldx, const float* W, const float* b, float* y, long ldy, int M, int N, int K, int accumulate,
                     hipStream_t st) {
    MFVIT_LAUNCH(linear_small_fwd_kernel, dim3((M * N + 3) / 4), dim3(256), 0, st, x, ldx, W, b, y, ldy, M, N, K, accumulate);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int linear_small_bwd(const float* dy, long lddy, const float* x, long ldx, const float* W, float* dx, long lddx, int dx_accumulate, float* dW,
                     float* db, int M, int N, int K, hipStream_t st) {
    long work = (long)M * K;
    if ((long)N * K > work) work = (long)N * K;
    MFVIT_LAUNCH(linear_small_bwd_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, dy, lddy, x, ldx, W, dx, lddx,
                       dx_accumulate, dW, db, M, N, K);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// --------------------------------------------------------------------------------------- small-C cross entropy
// nn.CrossEntropyLoss (mean) over logits [B][C], C <= 64: one thread per sample, ONE block (samples strided over its 1,024 threads); the mean is a fixed-order
// sum (per-thread sequences, wave sums, the 16 wave totals in order): the same bits on every run (round 6: one float atomic per wave before).
// Writes dlogits = (softmax - onehot) / B and preds (argmax, first maximum like torch.max).
__global__ __launch_bounds__(1024) void ce_small_kernel(const float* __restrict__ logits, const long* __restrict__ target, float* __restrict__ loss_mean,
                                                        float* __restrict__ dlogits, long* __restrict__ preds, int B, int C) {
    __shared__ float red[16];
    float li = 0.f;
    for (int b = threadIdx.x; b < B; b += 1024) {
        const float* z = logits + (long)b * C;
        float m = z[0];
        int am = 0;
        for (int c = 1; c < C; ++c) if (z[c] > m) { m = z[c]; am = c; }
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += __expf(z[c] - m);
        const float lse = m + __logf(s);
        const int t = (int)target[b];
        li += (lse - z[t]) / (float)B;
        if (dlogits)
            for (int c = 0; c < C; ++c) dlogits[(long)b * C + c] = (__expf(z[c] - lse) - (c == t ? 1.f : 0.f)) / (float)B;
        if (preds) preds[b] = am;
    }
    li = wave_sum(li);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = li;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i];
        *loss_mean = t;
    }
}
int ce_small(const float* logits, const long* target, float* loss_mean, float* dlogits, long* preds, int B, int C, hipStream_t st) {
    if (C > 64 || C < 1) return MFVIT_EINVAL;
    MFVIT_LAUNCH(ce_small_kernel, dim3(1), dim3(1024), 0, st, logits, target, loss_mean, dlogits, preds, B, C);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// --------------------------------------------------------------------------------------- misc
__global__ void add_rows_kernel(float* __restrict__ dst, long ldd, const float* __restrict__ src, long lds_, int rows, int N) {
    const long id = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (id < (long)rows * N) dst[(id / N) * ldd + id % N] += src[(id / N) * lds_ + id % N];
}
int add_rows(float* dst, long ldd, const float* src, long lds_, int rows, int N, hipStream_t st) {
    MFVIT_LAUNCH(add_rows_kernel, dim3((unsigned)(((long)rows * N + 255) / 256)), dim3(256), 0, st, dst, ldd, src, lds_, rows, N);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) y[i] = fmaf(a, x[i], y[i]);
}
int axpy(float* y, const float* x, float a, long n, hipStream_t st) {
    MFVIT_LAUNCH(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, x, a, n);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
// column sum of selected rows: out[n] += sum_r x[(r*row_stride + row_off)*ld + n]   (f32).  One block per 64 columns adds its rows in a fixed order (sixteen
// interleaved sequences, combined 0 + 1 + ... + 15) and is the only writer of its columns: the same bits on every run (round 6: up to 64 float atomics per column before)
__global__ __launch_bounds__(1024) void colsum_rows_kernel(const float* __restrict__ x, long ld, float* __restrict__ out, int rows,
                                                           int row_stride, int row_off, int N) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    __shared__ float sm[16][64];
    float s = 0.f;
    if (n < N) {
#pragma unroll 4
        for (int r = sub; r < rows; r += 16) s += x[((long)r * row_stride + row_off) * ld + n];
    }
    sm[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && n < N) {
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += sm[i][threadIdx.x];
        out[n] += s;
    }
}
int colsum_rows(const float* x, long ld, float* out, int rows, int row_stride, int row_off, int N, hipStream_t st) {
    MFVIT_LAUNCH(colsum_rows_kernel, dim3((N + 63) / 64), dim3(1024), 0, st, x, ld, out, rows, row_stride, row_off, N);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// out[i] += sum_b x[b * n + i]   (d pos_emb of the token-input encoder: the position table is shared by the batch)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, long n) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += x[(long)b * n + i];
    out[i] += s;
}
int batch_sum(const float* x, float* out, int B, long n, hipStream_t st) {
    MFVIT_LAUNCH(batch_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, out, B, n);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(DropP drop, long n, unsigned char* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = drop_mul(drop, (unsigned)i) != 0.f;
}
int dropout_mask(DropP drop, long n, unsigned char* out, hipStream_t st) {
    if (n <= 0 || n >= (1L << 32)) return MFVIT_EINVAL;
    MFVIT_LAUNCH(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, drop, n, out);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

}  // namespace mfvit
