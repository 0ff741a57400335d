// Shared device helpers for the MF-ViT gfx950 kernels (wave64, MFMA, LDS).
// gfx950 only: no CUDA / multi-backend paths.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/mfvit.h"

namespace mfvit {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// error codes (MFVIT_OK / MFVIT_E*) and dtype tags (MFVIT_F32 / MFVIT_BF16 / MFVIT_BF16X3 / MFVIT_F16) come from the public header

constexpr int WAVE = 64;

// Element of a SPLIT-bf16 tensor (dtype MFVIT_BF16X3): a logical f32 value x is kept as hi = bf16(x), lo = bf16(x - hi), i.e. 16
// mantissa bits, and a product of two such tensors runs as three bf16 MFMAs (hi*hi + lo*hi + hi*lo, f32 accumulate; the dropped
// lo*lo term is 2^-16 relative).  Storage layout "I32": a logical row-major [M][N] matrix (N % 32 == 0) is a bf16 [M][2N] array in
// which every group of 32 logical columns occupies 64 consecutive elements, [hi x 32 | lo x 32]; logical (m, n) has its hi part at
// column 64 (n / 32) + n % 32 and its lo part 32 elements further.  A 128-byte LDS row of a K-contiguous tile therefore holds one
// 32-wide k group of both parts, and a head_dim-32 attention head's row piece is one such group.  Leading dimensions of split
// tensors are given in STORAGE elements (twice the logical width).
struct sbf16 { bf16 v; };
// SPLIT fp16 (dtype tag MFVIT_X3F16, the qkv tensor of the attention core in bf16x3 mode only): the same I32 layout with hi = f16(x),
// lo = f16(x - hi) - 22 mantissa bits for |x| >= 2^-2, an absolute floor of 2^-25 below that, |x| <= 65504.  The attention kernels take q, k, v
// in this form so that the probabilities P (in [0, 2^14]) and the scaled score gradients dS can feed the f16 MFMA: their hi / lo split is
// one packed conversion + one mixed-precision FMA per element (v_cvt_pk_f16_f32, v_fma_mixlo/hi_f16) instead of convert, shift / mask,
// subtract, convert (attention_mfma.hip).
struct sf16 { _Float16 v; };
template <typename T> struct is_split { static constexpr bool value = false; };
template <> struct is_split<sbf16> { static constexpr bool value = true; };
template <> struct is_split<sf16> { static constexpr bool value = true; };
// storage elements per logical element (1, or 2 for split tensors)
template <typename T> struct elems_per { static constexpr int value = is_split<T>::value ? 2 : 1; };
// storage column of the hi part of logical column n (lo part: + 32)
__device__ __host__ __forceinline__ constexpr int split_col(int n) { return ((n >> 5) << 6) + (n & 31); }
__device__ __forceinline__ void split2(float x, bf16& hi, bf16& lo) {
    hi = (bf16)x;
    lo = (bf16)(x - (float)hi);
}
// Two values at once: hi = E(x), lo = E(x - hi) as ONE packed conversion each (v_cvt_pk_bf16_f32 / v_cvt_pkrtz-free f16 pair).  Element
// by element every conversion is a packed instruction with half of it unused (the attention files are built without SLP
// vectorisation): 66 of the ~200 VALU instructions per 24 MFMAs of the attention backward's inner loop before this helper.
template <typename E, bool SPLIT> __device__ __forceinline__ void cvt_pair(float x0, float x1, E& h0, E& h1, E& l0, E& l1) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef E ex2_t __attribute__((ext_vector_type(2)));
    const f32x2_t x = {x0, x1};
    const ex2_t h = __builtin_convertvector(x, ex2_t);
    h0 = h[0];
    h1 = h[1];
    if constexpr (SPLIT) {
        const ex2_t l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2_t), ex2_t);
        l0 = l[0];
        l1 = l[1];
    }
}
// ---- dropout (TransFuser GPT, fuseattention.py:33-34,71,112): counter-based masks, regenerated in the backward from (seed, site, element
// index) - nothing is stored.  keep(idx) = lowbias32(idx ^ key) >= thr with thr = p * 2^32 and key = hash of (seed, site); kept values are
// multiplied by 1 / (1 - p).  thr == 0 switches a site off.  (A two-round multiply-xorshift hash, not Philox: the masks need to be
// reproducible and well mixed, not cryptographic; mfvit_dropout_mask exports them so that the parity tests use the very same masks.)
struct DropP { unsigned thr; float scale; unsigned key; };
__host__ __device__ inline unsigned lowbias32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float drop_mul(const DropP& d, unsigned idx) { return lowbias32(idx ^ d.key) >= d.thr ? d.scale : 0.f; }
inline DropP make_drop(float p, unsigned long long seed, unsigned site) {
    DropP d;
    if (!(p > 0.f)) { d.thr = 0; d.scale = 1.f; d.key = 0; return d; }
    const double t = (double)p * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    d.scale = 1.0f / (1.0f - p);
    d.key = lowbias32((unsigned)seed ^ lowbias32((unsigned)(seed >> 32) + 0x9E3779B9u * (site + 1u)));
    return d;
}
// storage type of a saved activation derivative (gelu' / relu'): the tensor's own type, plain fp16 for split tensors (gemm.hip)
template <typename T> struct act_grad_type { typedef T type; };
template <> struct act_grad_type<sbf16> { typedef f16 type; };
// 8 x 16-bit MFMA operand fragment of element type T
template <typename T> struct Vec8;
template <> struct Vec8<bf16> { typedef bf16x8 type; };
template <> struct Vec8<sbf16> { typedef bf16x8 type; };
template <> struct Vec8<f16> { typedef f16x8 type; };
template <> struct Vec8<sf16> { typedef f16x8 type; };
template <typename T> struct Vec4;
template <> struct Vec4<bf16> { typedef bf16x4 type; typedef bf16 elem; };
template <> struct Vec4<sbf16> { typedef bf16x4 type; typedef bf16 elem; };
template <> struct Vec4<f16> { typedef f16x4 type; typedef f16 elem; };
template <> struct Vec4<sf16> { typedef f16x4 type; typedef f16 elem; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
__device__ __forceinline__ float to_f32(f16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }  // v_cvt_pk_bf16_f32 (RNE, NaN-safe)
template <> __device__ __forceinline__ f16 from_f32<f16>(float x) { return (f16)x; }     // v_cvt_f16_f32 (RNE; overflow -> inf)

// Typed element access for tensors of element type T at a LOGICAL column (row pointer given in storage elements):
// plain types store one value; split tensors store the hi / lo pair at the I32 positions.
template <typename T> __device__ __forceinline__ void store_elem(T* row, int n, float v) { row[n] = from_f32<T>(v); }
template <> __device__ __forceinline__ void store_elem<sbf16>(sbf16* row, int n, float v) {
    bf16 hi, lo;
    split2(v, hi, lo);
    bf16* r = (bf16*)row + split_col(n);
    r[0] = hi;
    r[32] = lo;
}
// the same for two values that go to column n of two different rows: one packed conversion per part instead of one per element
template <typename T> __device__ __forceinline__ void store_elem_pair(T* row0, T* row1, int n, float v0, float v1) {
    if constexpr (std::is_same<T, float>::value) {
        row0[n] = v0;
        row1[n] = v1;
    } else if constexpr (is_split<T>::value) {
        typedef typename Vec4<T>::elem E;
        E h0, h1, l0, l1;
        cvt_pair<E, true>(v0, v1, h0, h1, l0, l1);
        E* r0 = (E*)row0 + split_col(n);
        E* r1 = (E*)row1 + split_col(n);
        r0[0] = h0;
        r0[32] = l0;
        r1[0] = h1;
        r1[32] = l1;
    } else {
        T h0, h1, l0, l1;
        cvt_pair<T, false>(v0, v1, h0, h1, l0, l1);
        row0[n] = h0;
        row1[n] = h1;
    }
}
template <typename T> __device__ __forceinline__ float load_elem(const T* row, int n) { return to_f32(row[n]); }
template <> __device__ __forceinline__ float load_elem<sbf16>(const sbf16* row, int n) {
    const bf16* r = (const bf16*)row + split_col(n);
    return (float)r[0] + (float)r[32];
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// d/dx [ x * Phi(x) ] = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Fast erf-GELU for the bf16 path: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, far below bf16's 2^-9), one v_exp_f32.
// e = exp(-x^2/2) is shared between erf(x/sqrt2) and the normal pdf in the gradient.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2/2) = exp(-z^2)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erf_abs = fmaf(-poly * t, e, 1.0f);
    cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float gelu_fast(float x) {
    float cdf, e;
    gelu_parts_fast(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float cdf, e;
    gelu_parts_fast(x, cdf, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}
// gelu(x) and gelu'(x) together: the erf / exp evaluation is shared (the fc1 epilogue needs both; computed separately they cost two
// transcendental evaluations per element in a VALU-bound epilogue)
template <typename T> __device__ __forceinline__ void gelu_both_t(float x, float& g, float& dg) {
    if constexpr (sizeof(T) == 2) {
        float cdf, e;
        gelu_parts_fast(x, cdf, e);
        g = x * cdf;
        dg = fmaf(x * 0.39894228040143267794f, e, cdf);
    } else {
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        g = x * cdf;
        dg = cdf + x * (0.39894228040143267794f * __expf(-0.5f * x * x));
    }
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) { return sizeof(T) == 2 ? gelu_fast(x) : gelu_erf(x); }
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) { return sizeof(T) == 2 ? gelu_grad_fast(x) : gelu_erf_grad(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// The wave's maximum of NON-NEGATIVE values as a scalar, without LDS permutes: four DPP steps inside the rows of 16 lanes, then the four rows by readlane.
__device__ __forceinline__ float wave_max_nonneg_dpp(float v) {
    auto dpp = [](float x, auto ctrl) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}));     // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}));     // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}));    // row_half_mirror
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>{}));    // row_mirror: every lane of a row of 16 holds the row's maximum
    const int b = __builtin_bit_cast(int, v);
    const int r0 = __builtin_amdgcn_readlane(b, 0), r1 = __builtin_amdgcn_readlane(b, 16), r2 = __builtin_amdgcn_readlane(b, 32), r3 = __builtin_amdgcn_readlane(b, 48);
    const int m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;   // (bit patterns of non-negative floats order like integers)
    return __builtin_bit_cast(float, m01 > m23 ? m01 : m23);
}

// Block-wide sum for blockDim.x == NT (multiple of 64); scratch must hold NT/64 floats. All threads get the result.
template <int NT> __device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += scratch[i];
    return t;
}
template <int NT> __device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float t = scratch[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, scratch[i]);
    return t;
}

// Streaming 16-byte OUTPUT store (`sc0 sc1 nt`: system scope, non-temporal).  Round 5, measured with FETCH_SIZE: with plain or nt-only stores the L2
// FETCHES every output line it allocates (write-allocate: fc1 + GELU read 135 MB from HBM for 41 MB of operands, the fc2 data gradient 221 for 119 -
// the "1.4 - 1.5 x wasted traffic" of rounds 3 - 4 was never operand re-reads); with the scope bits the line goes through without the fetch: 56 / 152 MB,
// -3 ... -4 % per launch (profiles/r05_tile_gemm_experiments.txt).  For tensors the NEXT kernel streams from HBM anyway (hundreds of MB per launch).
typedef unsigned int u32x4_st __attribute__((ext_vector_type(4)));
// (s_nop behind it: a vector-memory store of more than 64 bits reads its data registers late - the next write of those registers needs wait states
// (CDNA3 ISA 4.5, "VMEM store more than 8 bytes followed by a write of the write-data VGPRs").  The compiler inserts them for the stores it knows;
// it cannot see into an asm statement.  Without them the fc1 epilogue's second phase overwrote gelu' values still waiting to be read: ~8,000 wrong
// elements per launch at M = 25,216, different ones every run - tests/test_precision_gpu.py::test_linear_fwd caught it.)
__device__ __forceinline__ void store16_stream(void* gp, u32x4_st v) {
#ifdef MFVIT_PLAIN_OUTPUT_STORES        // A/B builds only (tools/build_variant_lib.sh): the round-4 behaviour
    __builtin_nontemporal_store(v, (u32x4_st*)gp);
#elif defined(MFVIT_STREAM_POLICY) && MFVIT_STREAM_POLICY == 1      // A/B builds: system scope without the non-temporal hint
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(gp), "v"(v) : "memory");
#elif defined(MFVIT_STREAM_POLICY) && MFVIT_STREAM_POLICY == 2      // A/B builds: sc1 only
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(gp), "v"(v) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(gp), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ void store8_stream(void* gp, uint2 v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(gp), "v"(v) : "memory");
}
__device__ __forceinline__ void store4_stream(void* gp, float v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(gp), "v"(v) : "memory");
}

// XCD-aware bijective block remap (8 XCDs, blocks dealt round-robin): blocks that are neighbours in the
// remapped id share an XCD / L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

// A one-time action PER DEVICE (hipFuncSetAttribute is a per-device setting: a process-wide flag left a second GPU with the 64 KB LDS default).
struct PerDeviceOnce {
    unsigned long long done = 0;
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;      // unknown device: just do it again
        const unsigned long long bit = 1ull << dev;
        if (done & bit) return false;
        done |= bit;
        return true;
    }
};
// compute units of the CURRENT device (cached per device)
inline int device_cus() {
    static int cache[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
    if (cache[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev] = n;
    }
    return cache[dev];
}
// How many independent kernel streams the caller runs side by side on this GPU (mfvit_set_stream_share; 1 = a kernel may take the whole chip).
// The two-stream CA model sets 2: the one-workgroup-per-CU kernels then size their grids for HALF the chip when the problem is small - a row tile
// streams the whole W[384][K] out of L2 whatever its height and a weight-gradient split costs a prologue and a round of atomics whatever its length,
// so fewer, longer workgroups cost a launch little, and the other encoder's kernels run BESIDE it instead of behind it (round 5: B = 16 8.13 -> 7.23 ms,
// B = 32 9.86 -> 8.78 ms per step; B = 128 unchanged: its tiles are taller than the floor anyway).
int stream_share();

// A/B switch from the environment: read ONCE per process into the caller's static cache (INT_MIN = unread), or at every launch when
// MFVIT_AB_LIVE=1 (the in-process A/B tools and the kernel-variant tests set it) - no getenv on the launch path otherwise.
inline int env_switch(const char* name, int dflt, int& cache) {
    static const bool live = [] { const char* e = getenv("MFVIT_AB_LIVE"); return e && e[0] == '1'; }();
    if (cache == INT_MIN || live) { const char* e = getenv(name); cache = e ? atoi(e) : dflt; }
    return cache;
}

// Launch wrapper: clears whatever sticky error code earlier, unrelated runtime calls of this thread left behind (the host
// framework's event queries, device probing at load time), so that MFVIT_CHECK_LAUNCH reports THIS launch only.
#define MFVIT_LAUNCH(kernel, grid, block, lds_bytes, stream, ...)                       \
    do {                                                                               \
        (void)hipGetLastError();                                                       \
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, __VA_ARGS__);       \
    } while (0)

// hipGetLastError() also reports benign sticky codes left behind by the host framework's own calls on this thread
// (hipErrorNotReady from event / stream queries of the caching allocator): those are not launch failures.
#define MFVIT_CHECK_LAUNCH()                                                       \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess && e__ != hipErrorNotReady) {                        \
            fprintf(stderr, "[mfvit] HIP error %d (%s) at %s:%d\n", (int)e__, hipGetErrorString(e__), __FILE__, __LINE__); \
            return MFVIT_ELAUNCH;                                                  \
        }                                                                          \
    } while (0)

}  // namespace mfvit
