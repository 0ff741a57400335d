// f32 GEMMs with a handful of rows (round 4): the fusion module's projections of ONE row per sample - M = batch (128) x 384 x 384 or
// x 128 per head - and their weight gradients (reduction over those 128 rows).
//
// Why: on the 128 x 128 tile kernels (gemm.hip) such a product is 3 - 9 workgroups on 256 CUs, each running 768 dependent
// v_mfma_f32_32x32x2_f32 (64 cycles apiece) per wave: 27 - 36 us per launch whatever the size, eleven launches per train step of the
// two-stream model.  Here a workgroup owns ONE 32 x 32 output tile and its four waves split the reduction index; the partial tiles are
// added through LDS.  M = 128, N = K = 384, two directions: 96 workgroups x 48 MFMAs per wave.
//   * NT (y = x W^T + b): a lane loads 16 contiguous bytes of its row (k .. k+3 for lanes 0 - 31, k+4 .. k+7 for lanes 32 - 63) of both
//     operands and feeds four MFMAs; which k an MFMA's two lane halves hold is free as long as both operands agree.
//   * TN (dW[n][k] += sum_m dy[m][n] x[m][k], float atomics like gemm_tn_kernel): 4-byte loads, 128 bytes contiguous per half-wave.
// Same GemmP batch addressing as the tile kernels (apply_batch); no epilogue beyond the bias.
#include "gemm.cuh"
#include "kernels.h"
#include "prof.h"

namespace mfvit {

namespace {

constexpr int SM_MAX_ROWS = 512;          // above that the 128 x 128 tiles fill the chip better

__device__ __forceinline__ f32x16 mma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// sum of the four waves' accumulators in wave 0 (red: 3 x 16 x 64 floats)
__device__ __forceinline__ void reduce_waves(f32x16& acc, float* red, int wave, int lane) {
    if (wave) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (!wave) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[(w * 16 + r) * 64 + lane];
    }
}

// STEPS: 8-wide k steps per wave when known at compile time (all loads issued up front: the kernel is latency-bound), 0 = any
template <int STEPS>
__global__ __launch_bounds__(256) void gemm_nt_small_kernel(GemmP p) {
    __shared__ float red[3 * 16 * 64];
    apply_batch<float>(p, 4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int kc = p.K >> 2;                                           // (K % 32 == 0: checked by the launcher)
    const bool aok = m0 + r < p.M, wok = n0 + r < p.N;                  // rows past the end: a valid row is read, zeros are multiplied
    const float* ap = (const float*)p.A + (long)(aok ? m0 + r : 0) * p.lda + wave * kc + hh * 4;
    const float* wp = (const float*)p.W + (long)(wok ? n0 + r : 0) * p.ldw + wave * kc + hh * 4;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if constexpr (STEPS > 0) {
        float4 a[STEPS], w[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            a[s] = *(const float4*)(ap + 8 * s);
            w[s] = *(const float4*)(wp + 8 * s);
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            acc = mma_f32(aok ? a[s].x : 0.f, wok ? w[s].x : 0.f, acc);
            acc = mma_f32(aok ? a[s].y : 0.f, wok ? w[s].y : 0.f, acc);
            acc = mma_f32(aok ? a[s].z : 0.f, wok ? w[s].z : 0.f, acc);
            acc = mma_f32(aok ? a[s].w : 0.f, wok ? w[s].w : 0.f, acc);
        }
    } else {
        // any K: chunks of four 8-wide steps, the next chunk's eight loads in flight under the sixteen MFMAs of this one
        auto mma4 = [&](const float4& a, const float4& w) __attribute__((always_inline)) {
            acc = mma_f32(aok ? a.x : 0.f, wok ? w.x : 0.f, acc);
            acc = mma_f32(aok ? a.y : 0.f, wok ? w.y : 0.f, acc);
            acc = mma_f32(aok ? a.z : 0.f, wok ? w.z : 0.f, acc);
            acc = mma_f32(aok ? a.w : 0.f, wok ? w.w : 0.f, acc);
        };
        const int nch = kc >> 5;                                        // whole chunks of 32 k
        float4 a[4], w[4], an[4], wn[4];
        if (nch > 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { a[s] = *(const float4*)(ap + 8 * s); w[s] = *(const float4*)(wp + 8 * s); }
        }
        for (int c = 0; c < nch; ++c) {
            const int kn = (c + 1 < nch ? c + 1 : c) * 32;              // (last chunk: a harmless reload of itself)
#pragma unroll
            for (int s = 0; s < 4; ++s) { an[s] = *(const float4*)(ap + kn + 8 * s); wn[s] = *(const float4*)(wp + kn + 8 * s); }
#pragma unroll
            for (int s = 0; s < 4; ++s) mma4(a[s], w[s]);
#pragma unroll
            for (int s = 0; s < 4; ++s) { a[s] = an[s]; w[s] = wn[s]; }
        }
        for (int k = nch * 32; k < kc; k += 8) mma4(*(const float4*)(ap + k), *(const float4*)(wp + k));
    }
    reduce_waves(acc, red, wave, lane);
    if (wave) return;
    const int n = n0 + r;
    if (n >= p.N) return;
    const float b = p.bias ? p.bias[n] : 0.f;
    float* out = (float*)p.out0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + acc_row(i, lane);
        if (m < p.M) out[(long)m * p.ldo0 + n] = acc[i] + b;
    }
}

__global__ __launch_bounds__(256) void gemm_tn_small_kernel(GemmP p) {
    __shared__ float red[3 * 16 * 64];
    apply_batch<float>(p, 4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int mc = ((p.M + 3) / 4 + 1) & ~1;                           // reduction rows per wave, even
    const int mb = wave * mc, me = min(p.M, mb + mc);
    const bool aok = n0 + r < p.N, wok = k0 + r < p.K;
    const float* ap = (const float*)p.A + (aok ? n0 + r : 0);
    const float* wp = (const float*)p.W + (wok ? k0 + r : 0);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int m = mb + hh; m < mb + mc + hh; m += 8) {                  // (uniform trip count: the MFMA needs every lane)
        float a[4], w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mj = m + 2 * j;
            const bool in = mj < me;
            const long mm = in ? mj : 0;
            a[j] = ap[mm * p.lda];
            if (!in || !aok) a[j] = 0.f;
            w[j] = wok ? wp[mm * p.ldw] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mma_f32(a[j], w[j], acc);
    }
    reduce_waves(acc, red, wave, lane);
    if (wave) return;
    const int k = k0 + r;
    if (k >= p.K) return;
    float* out = (float*)p.out0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = n0 + acc_row(i, lane);
        if (n < p.N) atomicAdd(out + (long)n * p.ldo0 + k, acc[i]);
    }
}

bool plain(const GemmP& p) {
    return !p.out1 && !p.aux && !p.res && !p.res_t && !p.orow_in && !p.gamma && !p.cs0 && !p.cs1 && !p.cs2 && !p.cpart && !p.omax && p.out0 && p.A && p.W;
}
bool small_switch() { return true; }   // (round 4 A/B: 27 - 36 us -> 6 - 9 us per launch against the 128 x 128 tile kernels)

}  // namespace

bool gemm_nt_small_supported(int dtype, int epi, const GemmP& p) {
    if (dtype != MFVIT_F32 || (epi != EPI_NONE && epi != EPI_BIAS) || !plain(p) || !small_switch()) return false;
    if (p.M <= 0 || p.M > SM_MAX_ROWS || p.K % 32 || p.lda % 4 || p.ldw % 4) return false;
    if ((((uintptr_t)p.A | (uintptr_t)p.W) & 15) || ((p.sAo | p.sAi | p.sWo | p.sWi) & 3)) return false;
    return epi == EPI_NONE || p.bias != nullptr;
}
int gemm_nt_small(int epi, GemmP p, hipStream_t st) {
    if (epi == EPI_NONE) p.bias = nullptr;
    const int nb = p.nb > 1 ? p.nb : 1;
    ProfScope ps(PROF_OTHER, 2.0 * p.M * p.N * p.K * nb, 0, st);
    const dim3 grid((p.N + 31) / 32, (p.M + 31) / 32, nb);
    if (p.K == 384)
        MFVIT_LAUNCH(gemm_nt_small_kernel<12>, grid, dim3(256), 0, st, p);
    else if (p.K == 128)
        MFVIT_LAUNCH(gemm_nt_small_kernel<4>, grid, dim3(256), 0, st, p);
    else
        MFVIT_LAUNCH(gemm_nt_small_kernel<0>, grid, dim3(256), 0, st, p);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

bool gemm_tn_small_supported(int dtype, const GemmP& p) {
    return dtype == MFVIT_F32 && plain(p) && !p.bias && small_switch() && p.M > 0 && p.M <= SM_MAX_ROWS;
}
int gemm_tn_small(GemmP p, hipStream_t st) {
    const int nb = p.nb > 1 ? p.nb : 1;
    ProfScope ps(PROF_OTHER, 2.0 * p.M * p.N * p.K * nb, 0, st);
    MFVIT_LAUNCH(gemm_tn_small_kernel, dim3((p.K + 31) / 32, (p.N + 31) / 32, nb), dim3(256), 0, st, p);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

}  // namespace mfvit
