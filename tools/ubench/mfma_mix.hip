// What would a cheaper operand format buy the GEMM classes?  Pure matrix-pipe loops (operands in registers, no LDS, no epilogue: the ceiling, clock and power cap
// included) for the same ALGORITHMIC work - one 32 x 32 output tile per wave over K = 64 k values per iteration and accumulator set:
//   x3   : split bf16 (shipping): hi.hi + hi.lo + lo.hi = 3 v_mfma_f32_32x32x16_bf16 per 16 k  -> 12 MFMAs per 64 k
//   f16+2f8 : hi in fp16 (one v_mfma_f32_32x32x16_f16 per 16 k) + the two correction products on the fp8 path (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3,
//          unit block scales): 4 + 2 instructions per 64 k - "2 MFMA units instead of 3"
//   f16+2f6 : the same with the correction products in fp6 (e2m3; the 4-bit rate)
//   f16  : plain fp16, 4 MFMAs per 64 k (the floor of any 16-bit format)
// Prints algorithmic TFLOP/s (2 * 32 * 32 * 64 flops per tile step), the shader clock the loop ran at, and the ratio to x3.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_mix.hip -o tools/ubench/mfma_mix && tools/ubench/mfma_mix [waves_per_simd] [random|zero]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int NACC = 4;      // independent 32 x 32 accumulator tiles per wave (a 64 x 64 wave tile, like the tile GEMM)

template <int MODE>
__global__ __launch_bounds__(256) void mix_loop(const int* __restrict__ in, float* __restrict__ out, int iters, long long* cyc) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const long long c0 = __builtin_readcyclecounter();
    // operand registers: 4 k-steps of 16 (hi / lo fragments of 4 dwords each) and the 8-dword fp8 fragments of the 64-wide step, for A and B
    i32x8 raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 8; ++r) raw[i][r] = in[((gid * 8 + i) * 8 + r) & 0xffff];
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
            const i32x8 a = raw[t & 3], b = raw[4 + (t >> 1)], a2 = raw[(t + 1) & 3], b2 = raw[4 + ((t + 1) & 3)];
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            const i32x4 alo = {a[0], a[1], a[2], a[3]}, ahi = {a[4], a[5], a[6], a[7]}, blo = {b[0], b[1], b[2], b[3]}, bhi = {b[4], b[5], b[6], b[7]};
            if constexpr (MODE == 0) {                     // split bf16: 4 k-steps x 3 terms
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 A0 = __builtin_bit_cast(bf16x8, (ks & 1) ? ahi : alo), A1 = __builtin_bit_cast(bf16x8, (ks & 1) ? alo : ahi);
                    const bf16x8 B0 = __builtin_bit_cast(bf16x8, (ks & 2) ? bhi : blo), B1 = __builtin_bit_cast(bf16x8, (ks & 2) ? blo : bhi);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {           // hi . hi in fp16
                    const f16x8 A0 = __builtin_bit_cast(f16x8, (ks & 1) ? ahi : alo), B0 = __builtin_bit_cast(f16x8, (ks & 2) ? bhi : blo);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B0, acc[t], 0, 0, 0);
                }
                if constexpr (MODE == 1) {                 // + two fp8 (e4m3) correction products over the 64 k values, unit scales (E8M0 127)
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b2, acc[t], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a2, b, acc[t], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                } else if constexpr (MODE == 2) {          // ... in fp6 (e2m3): 6 of the 8 dwords carry data
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b2, acc[t], 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a2, b, acc[t], 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[gid] = s;
    if (gid == 0) *cyc = __builtin_readcyclecounter() - c0;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const bool zero = argc > 2 && !strcmp(argv[2], "zero");
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * wps, iters = 4000;
    int* in; float* out; long long* cyc;
    hipMalloc(&cyc, 8); hipMalloc(&in, 65536 * 4); hipMalloc(&out, (size_t)blocks * 256 * 4);
    int* h = (int*)malloc(65536 * 4);
    unsigned x = 12345u;
    for (int i = 0; i < 65536; ++i) {
        // finite small values in every interpretation: bf16 / fp16 halves with exponents near 1, fp8 bytes below the NaN codes
        x = x * 1664525u + 1013904223u;
        const unsigned v = zero ? 0u : (x & 0x03ff03ffu) | 0x38003800u;
        h[i] = (int)v;
    }
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"x3 (3 bf16 MFMAs per 16 k)", "f16 + 2 fp8 corrections", "f16 + 2 fp6 corrections", "plain f16"};
    double base = 0;
    for (int mode = 0; mode < 4; ++mode) {
        double best = 1e30; long long bc = 0;
        for (int rep = 0; rep < 4; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(mix_loop<0>, dim3(blocks), dim3(256), 0, 0, in, out, iters, cyc);
            if (mode == 1) hipLaunchKernelGGL(mix_loop<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters, cyc);
            if (mode == 2) hipLaunchKernelGGL(mix_loop<2>, dim3(blocks), dim3(256), 0, 0, in, out, iters, cyc);
            if (mode == 3) hipLaunchKernelGGL(mix_loop<3>, dim3(blocks), dim3(256), 0, 0, in, out, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            if (rep > 0 && ms < best) { best = ms; bc = c; }
        }
        const double flops = 2.0 * 32 * 32 * 64 * NACC * (double)iters * blocks * 4;      // algorithmic
        const double tf = flops / (best * 1e-3) / 1e12;
        if (mode == 0) base = tf;
        // matrix-pipe cycles one iteration of one wave needs if every instruction issues back to back: 32 per 32x32x16 (bf16 / f16), 64 per fp8 32x32x64, 32 per fp6
        const double pipe = NACC * (mode == 0 ? 12 * 32.0 : mode == 1 ? 4 * 32.0 + 2 * 64.0 : mode == 2 ? 4 * 32.0 + 2 * 32.0 : 4 * 32.0);
        const double mhz = pipe * wps * iters / (best * 1e-3) / 1e6;          // the clock at which that schedule would fill `best` (an upper bound of the real clock)
        printf("%-30s %s operands, %d waves/SIMD: %8.1f algorithmic TFLOP/s  (%.2f x the split-bf16 loop)  %.3f ms; back-to-back issue would need %.0f MHz\n",
               names[mode], zero ? "zero" : "random", wps, tf, tf / base, best, mhz);
        (void)bc;
    }
    return 0;
}
