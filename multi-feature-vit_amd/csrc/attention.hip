// Multi-head self-attention core of the ViT block:  O = softmax(Q K^T / sqrt(d)) V  per (image, head), and its backward.
// Replaces timm Attention.forward's  (q @ k.transpose) * scale -> softmax -> @ v  (same math as
// moco/model/module.py:57-61 of the reference) without materialising the (B, h, T, T) score tensor.
//
// qkv layout (written by the qkv GEMM epilogue): [B][T][3][H][HD]   out: [B][T][H*HD]   lse: [B][H][T] (natural log)
//
// v1 "exact" kernels: K/V (fwd) or Q/dO (bwd) of one (image, head) live in LDS as f32, one query (or key) row per
// thread, online softmax in registers.  f32 arithmetic for both storage types; the f32 instantiation is the
// parity path (bit-for-bit deterministic, no atomics except the bias-gradient column sums).
#include "common.cuh"
#include "prof.h"

#include <stdlib.h>

namespace mfvit {

template <typename T, int HD> __device__ __forceinline__ void load_row(const T* p, float (&r)[HD]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < HD / 8; ++i) {
            const typename Vec8<T>::type v = *(const typename Vec8<T>::type*)(p + 8 * i);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[8 * i + j] = (float)v[j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const float4 v = *(const float4*)(p + 4 * i);
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
    }
}
template <typename T, int HD> __device__ __forceinline__ void store_row(T* p, const float (&r)[HD]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < HD / 8; ++i) {
            typename Vec8<T>::type v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = from_f32<T>(r[8 * i + j]);
            *(typename Vec8<T>::type*)(p + 8 * i) = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) *(float4*)(p + 4 * i) = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    }
}

// cooperative copy of rows [0,T) of one (b, which, h) slice into LDS as f32 [T][HD]
template <typename T, int HD> __device__ __forceinline__ void stage_rows(const T* base, long row_stride, int Tn, float* dst) {
    constexpr int CH = HD / 4;  // 4-element chunks per row
    for (int q = threadIdx.x; q < Tn * CH; q += blockDim.x) {
        const int t = q / CH, c = q % CH;
        const T* s = base + (long)t * row_stride + 4 * c;
        float4 v;
        if constexpr (sizeof(T) == 2) {
            const typename Vec4<T>::type w = *(const typename Vec4<T>::type*)s;
            v = make_float4((float)w[0], (float)w[1], (float)w[2], (float)w[3]);
        } else {
            v = *(const float4*)s;
        }
        *(float4*)(dst + t * HD + 4 * c) = v;
    }
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_fwd_exact_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse,
                                                             int Tn, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float* Ks = (float*)lds_raw;
    float* Vs = Ks + Tn * HD;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const long rs = 3L * H * HD;
    const T* base = qkv + (long)b * Tn * rs + h * HD;
    stage_rows<T, HD>(base + (long)H * HD, rs, Tn, Ks);
    stage_rows<T, HD>(base + 2L * H * HD, rs, Tn, Vs);
    __syncthreads();
    const float qs = scale * 1.4426950408889634f;  // log2(e): exp(x) = exp2(x * log2e)
    for (int i = threadIdx.x; i < Tn; i += blockDim.x) {
        float q[HD], o[HD];
        load_row<T, HD>(base + (long)i * rs, q);
#pragma unroll
        for (int d = 0; d < HD; ++d) { q[d] *= qs; o[d] = 0.f; }
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < Tn; ++j) {
            const float* kr = Ks + j * HD;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s = fmaf(q[d], kr[d], s);
            if (s > m) {
                const float a = exp2f(m - s);
                l *= a;
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] *= a;
                m = s;
            }
            const float p = exp2f(s - m);
            l += p;
            const float* vr = Vs + j * HD;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] = fmaf(p, vr[d], o[d]);
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= inv;
        store_row<T, HD>(out + ((long)b * Tn + i) * H * HD + h * HD, o);
        lse[((long)b * H + h) * Tn + i] = (m + log2f(l)) * 0.6931471805599453f;
    }
}

// Backward.  Phase A (thread = query i): dQ_i = scale * sum_j p_ij (dO_i.V_j - D_i) K_j, D_i = dO_i.O_i.
//            Phase B (thread = key j):   dV_j = sum_i p_ij dO_i ;  dK_j = scale * sum_i p_ij (dO_i.V_j - D_i) Q_i.
// p_ij = exp(scale q_i.k_j - lse_i) is recomputed from the forward's log-sum-exp.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_exact_kernel(const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout,
                                                             const float* __restrict__ lse, T* __restrict__ dqkv, float* __restrict__ dbias,
                                                             int Tn, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float* S0 = (float*)lds_raw;   // K then Q
    float* S1 = S0 + Tn * HD;      // V then dO
    float* Dl = S1 + Tn * HD;      // D_i
    float* Ll = Dl + Tn;           // lse_i
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const long rs = 3L * H * HD, os = (long)H * HD;
    const T* base = qkv + (long)b * Tn * rs + h * HD;
    T* dbase = dqkv + (long)b * Tn * rs + h * HD;
    const T* obase = out + (long)b * Tn * os + h * HD;
    const T* dobase = dout + (long)b * Tn * os + h * HD;
    const float* lrow = lse + ((long)b * H + h) * Tn;
    stage_rows<T, HD>(base + (long)H * HD, rs, Tn, S0);
    stage_rows<T, HD>(base + 2L * H * HD, rs, Tn, S1);
    for (int i = threadIdx.x; i < Tn; i += blockDim.x) Ll[i] = lrow[i];
    __syncthreads();
    float colsum[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) colsum[d] = 0.f;
    // ---- phase A
    for (int i0 = 0; i0 < Tn; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        if (i < Tn) {
            float q[HD], dO[HD], dq[HD];
            load_row<T, HD>(base + (long)i * rs, q);
            load_row<T, HD>(dobase + (long)i * os, dO);
            {
                float o[HD];
                load_row<T, HD>(obase + (long)i * os, o);
                float D = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) D = fmaf(dO[d], o[d], D);
                Dl[i] = D;
            }
            const float D = Dl[i], L = Ll[i];
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = 0.f;
            for (int j = 0; j < Tn; ++j) {
                const float* kr = S0 + j * HD;
                const float* vr = S1 + j * HD;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) { s = fmaf(q[d], kr[d], s); dp = fmaf(dO[d], vr[d], dp); }
                const float p = __expf(s * scale - L);
                const float ds = p * (dp - D) * scale;
#pragma unroll
                for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
            }
            store_row<T, HD>(dbase + (long)i * rs, dq);
#pragma unroll
            for (int d = 0; d < HD; ++d) colsum[d] += dq[d];
        }
    }
    if (dbias) {
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const float s = wave_sum(colsum[d]);
            if ((threadIdx.x & 63) == 0) atomicAdd(dbias + h * HD + d, s);
        }
    }
    __syncthreads();
    // ---- phase B: restage Q and dO over K and V
    stage_rows<T, HD>(base, rs, Tn, S0);
    stage_rows<T, HD>(dobase, os, Tn, S1);
    __syncthreads();
    float cs_k[HD], cs_v[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) cs_k[d] = cs_v[d] = 0.f;
    for (int j0 = 0; j0 < Tn; j0 += blockDim.x) {
        const int j = j0 + threadIdx.x;
        if (j < Tn) {
            float k[HD], v[HD], dk[HD], dv[HD];
            load_row<T, HD>(base + (long)j * rs + (long)H * HD, k);
            load_row<T, HD>(base + (long)j * rs + 2L * H * HD, v);
#pragma unroll
            for (int d = 0; d < HD; ++d) dk[d] = dv[d] = 0.f;
            for (int i = 0; i < Tn; ++i) {
                const float* qr = S0 + i * HD;
                const float* dor = S1 + i * HD;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) { s = fmaf(qr[d], k[d], s); dp = fmaf(dor[d], v[d], dp); }
                const float p = __expf(s * scale - Ll[i]);
                const float ds = p * (dp - Dl[i]) * scale;
#pragma unroll
                for (int d = 0; d < HD; ++d) { dv[d] = fmaf(p, dor[d], dv[d]); dk[d] = fmaf(ds, qr[d], dk[d]); }
            }
            store_row<T, HD>(dbase + (long)j * rs + (long)H * HD, dk);
            store_row<T, HD>(dbase + (long)j * rs + 2L * H * HD, dv);
#pragma unroll
            for (int d = 0; d < HD; ++d) { cs_k[d] += dk[d]; cs_v[d] += dv[d]; }
        }
    }
    if (dbias) {
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            const float sk = wave_sum(cs_k[d]);
            const float sv = wave_sum(cs_v[d]);
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(dbias + (long)H * HD + h * HD + d, sk);
                atomicAdd(dbias + 2L * H * HD + h * HD + d, sv);
            }
        }
    }
}

template <typename T, int HD>
static int launch_fwd(const void* qkv, void* out, float* lse, int B, int Tn, int H, hipStream_t st) {
    const int bytes = 2 * Tn * HD * 4;
    if (bytes > 160 * 1024) return MFVIT_EINVAL;
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_exact_kernel<T, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    ProfScope ps(PROF_ATTN_FWD, 4.0 * B * H * (double)Tn * Tn * HD, 0, st);
    MFVIT_LAUNCH((attn_fwd_exact_kernel<T, HD>), dim3(B * H), dim3(256), bytes, st, (const T*)qkv, (T*)out, lse, Tn, H,
                       1.0f / sqrtf((float)HD));
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
template <typename T, int HD>
static int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn, int H,
                      hipStream_t st) {
    const int bytes = 2 * Tn * HD * 4 + 2 * Tn * 4;
    if (bytes > 160 * 1024) return MFVIT_EINVAL;
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_exact_kernel<T, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    ProfScope ps(PROF_ATTN_BWD, 8.0 * B * H * (double)Tn * Tn * HD, 0, st);
    MFVIT_LAUNCH((attn_bwd_exact_kernel<T, HD>), dim3(B * H), dim3(256), bytes, st, (const T*)qkv, (const T*)out, (const T*)dout, lse,
                       (T*)dqkv, dbias, Tn, H, 1.0f / sqrtf((float)HD));
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// exact (VALU, thread-per-row) kernels: f32 parity path, and the fallback of the 16-bit types when a head does not fit the MFMA
// kernels' LDS images.  Split tensors (MFVIT_BF16X3) have no exact variant: MFVIT_ENOSYS.
#define MFVIT_EXACT_BY_DTYPE(CALL32, CALL64)                                                       \
    switch (dtype) {                                                                               \
        case MFVIT_F32: { typedef float TT; return HD == 32 ? CALL32 : CALL64; }                   \
        case MFVIT_BF16: { typedef bf16 TT; return HD == 32 ? CALL32 : CALL64; }                   \
        case MFVIT_F16: { typedef f16 TT; return HD == 32 ? CALL32 : CALL64; }                     \
        case MFVIT_BF16X3: return MFVIT_ENOSYS;                                                    \
        default: return MFVIT_EINVAL;                                                              \
    }
int attn_fwd_exact(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, int HD, hipStream_t st) {
    if (HD != 32 && HD != 64) return MFVIT_EINVAL;
    MFVIT_EXACT_BY_DTYPE((launch_fwd<TT, 32>(qkv, out, lse, B, Tn, H, st)), (launch_fwd<TT, 64>(qkv, out, lse, B, Tn, H, st)))
}
int attn_bwd_exact(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn,
                   int H, int HD, hipStream_t st) {
    if (HD != 32 && HD != 64) return MFVIT_EINVAL;
    MFVIT_EXACT_BY_DTYPE((launch_bwd<TT, 32>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st)),
                         (launch_bwd<TT, 64>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st)))
}

// ---- dispatch: 16-bit types / head_dim 32 -> MFMA kernels (attention_mfma.hip); everything else -> exact kernels
bool attn_mfma_supported(int dtype, int Tn, int HDim, bool backward);
int attn_fwd_mfma(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, hipStream_t st);
int attn_bwd_mfma(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn, int H,
                  hipStream_t st, const unsigned* domax = nullptr);
bool attn_tiled_supported(int dtype, int Tn, int HDim);      // attention_tiled.hip: streaming kernels, any T, head_dim 32 / 64 / 96
int attn_fwd_tiled(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, int HDim, hipStream_t st);
int attn_bwd_tiled(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int Tn, int H, int HDim,
                   hipStream_t st);
int attn_colsum(int dtype, const void* dqkv, int M, int N, float* dbias, hipStream_t st);
static bool force_exact() {
    static const bool v = [] { const char* e = getenv("MFVIT_ATTN_EXACT"); return e && e[0] == '1'; }();
    return v;
}
static bool force_tiled() {   // MFVIT_ATTN_TILED=1: the streaming kernels also where the whole-head-in-LDS kernels would fit (tests, profiling)
    static const bool v = [] { const char* e = getenv("MFVIT_ATTN_TILED"); return e && e[0] == '1'; }();
    return v;
}
// whole-(image, head)-in-LDS MFMA kernels where they fit (ViT-S at 224^2: fastest) -> streaming MFMA kernels (long sequences, wide
// heads, split bf16 beyond the LDS limits) -> exact VALU kernels (f32)
int attn_qkv_dtype(int dtype, int Tn, int HDim) {
    if (dtype == MFVIT_BF16X3 && !force_tiled() && attn_mfma_supported(MFVIT_X3F16, Tn, HDim, false) && attn_mfma_supported(MFVIT_X3F16, Tn, HDim, true))
        return MFVIT_X3F16;
    return dtype;
}
int attn_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, int HDim, hipStream_t st) {
    if (dtype == MFVIT_X3F16) return attn_mfma_supported(dtype, Tn, HDim, false) ? attn_fwd_mfma(dtype, qkv, out, lse, B, Tn, H, st) : MFVIT_ENOSYS;
    const bool exact = force_exact() && dtype != MFVIT_BF16X3;
    if (!exact && !force_tiled() && attn_mfma_supported(dtype, Tn, HDim, false)) return attn_fwd_mfma(dtype, qkv, out, lse, B, Tn, H, st);
    if (!exact && attn_tiled_supported(dtype, Tn, HDim)) return attn_fwd_tiled(dtype, qkv, out, lse, B, Tn, H, HDim, st);
    return attn_fwd_exact(dtype, qkv, out, lse, B, Tn, H, HDim, st);
}
int attn_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn, int H,
             int HDim, hipStream_t st, const unsigned* domax) {
    if (dtype == MFVIT_X3F16)
        return attn_mfma_supported(dtype, Tn, HDim, true) ? attn_bwd_mfma(dtype, qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st, domax) : MFVIT_ENOSYS;
    const bool exact = force_exact() && dtype != MFVIT_BF16X3;
    if (!exact && !force_tiled() && attn_mfma_supported(dtype, Tn, HDim, true))
        return attn_bwd_mfma(dtype, qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st);
    if (!exact && attn_tiled_supported(dtype, Tn, HDim)) {
        const int rc = attn_bwd_tiled(dtype, qkv, out, dout, lse, dqkv, B, Tn, H, HDim, st);
        if (rc != MFVIT_OK || !dbias) return rc;
        return attn_colsum(dtype, dqkv, B * Tn, 3 * H * HDim, dbias, st);
    }
    return attn_bwd_exact(dtype, qkv, out, dout, lse, dqkv, dbias, B, Tn, H, HDim, st);
}

}  // namespace mfvit
