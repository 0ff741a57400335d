// Pure-MFMA ceiling of gfx950 for v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16 (5th argument: 32 | 16; same FLOPs per iteration:
// the 16x16 loop issues twice as many instructions on 4-register accumulators): no LDS, no global loads in the loop, no epilogue -
// what the matrix cores sustain under load (clock included).  Every GEMM roofline fraction in DESIGN.md is also quoted against THIS
// number (the "practical ceiling"), next to the 2.5 PFLOP/s datasheet peak.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o tools/mfma_ceiling && tools/mfma_ceiling [waves_per_simd] [random|zero] [accumulators]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(4))) float f32x4;
// the 16x16x32 twin: 2 * NACC independent 16x16 accumulators, 2 * NACC MFMAs per iteration = the same FLOPs and the same output elements
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop16(const bf16x8* __restrict__ in, float* __restrict__ out, int iters, long long* cyc) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const long long c0 = __builtin_readcyclecounter();
    bf16x8 a[NACC], b[NACC];
    f32x4 acc[4 * NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        a[i] = in[(gid * 2 * NACC + 2 * i) & 0xffff];
        b[i] = in[(gid * 2 * NACC + 2 * i + 1) & 0xffff];
    }
#pragma unroll
    for (int i = 0; i < 4 * NACC; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        // one 32x32x16 MFMA = 32 K flops; a 16x16x32 MFMA = 16 K flops: two per (a, b) pair and iteration on distinct accumulators
#pragma unroll
        for (int i = 0; i < 2 * NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i % NACC], b[i % NACC], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NACC; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[gid] = s;
    if (gid == 0) *cyc = __builtin_readcyclecounter() - c0;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const bf16x8* __restrict__ in, float* __restrict__ out, int iters, long long* cyc) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const long long c0 = __builtin_readcyclecounter();       // s_memtime: shader-clock cycles
    bf16x8 a[NACC], b[NACC];
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        a[i] = in[(gid * 2 * NACC + 2 * i) & 0xffff];
        b[i] = in[(gid * 2 * NACC + 2 * i + 1) & 0xffff];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[gid] = s;
    if (gid == 0) *cyc = __builtin_readcyclecounter() - c0;  // cycles one wave spent in the loop -> the clock it ran at
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;                 // waves per SIMD (4 SIMDs per CU)
    const bool zero = argc > 2 && !strcmp(argv[2], "zero");
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * wps;                                   // 256 threads = 4 waves = one per SIMD
    const int iters = 20000;
    const int NACC = argc > 3 ? atoi(argv[3]) : 8;                  // independent accumulators per wave: 8 (default), 4, 2, 1
    const int shape = argc > 4 ? atoi(argv[4]) : 32;                // 32: v_mfma_f32_32x32x16_bf16, 16: v_mfma_f32_16x16x32_bf16
    bf16x8* in;
    float* out;
    long long* cyc;
    hipMalloc(&cyc, 8);
    hipMalloc(&in, 65536 * sizeof(bf16x8));
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    unsigned short* h = (unsigned short*)malloc(65536 * 16);
    unsigned x = 12345u;
    for (int i = 0; i < 65536 * 8; ++i) {
        x = x * 1664525u + 1013904223u;
        const float f = zero ? 0.f : ((int)(x >> 8) % 2001 - 1000) * 1e-3f;      // values in [-1, 1]
        unsigned u;
        memcpy(&u, &f, 4);
        h[i] = (unsigned short)(u >> 16);
    }
    hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&](int it) {
        if (shape == 16) {
            if (NACC == 8) hipLaunchKernelGGL(mfma_loop16<8>, dim3(blocks), dim3(256), 0, 0, in, out, it, cyc);
            else if (NACC == 4) hipLaunchKernelGGL(mfma_loop16<4>, dim3(blocks), dim3(256), 0, 0, in, out, it, cyc);
            else hipLaunchKernelGGL(mfma_loop16<2>, dim3(blocks), dim3(256), 0, 0, in, out, it, cyc);
            return;
        }
        if (NACC == 8) hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, 0, in, out, it, cyc);
        else if (NACC == 4) hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, in, out, it, cyc);
        else if (NACC == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, 0, in, out, it, cyc);
        else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, in, out, it, cyc);
    };
    launch(2000);                                                                                  // warm-up (clock ramp)
    hipDeviceSynchronize();
    float best = 1e30f, ms;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        launch(iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2.0 * 32 * 32 * 16 * (double)NACC * iters * blocks * 4;
    long long hc = 0;
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    // last repetition: cycles wave 0 spent in its loop / the launch time.  With more than one wave per SIMD the oldest wave keeps the
    // matrix pipe (33 cycles per own MFMA either way) and finishes after 1/wps of the launch: scale accordingly.
    const double mhz = (double)hc * wps / (ms * 1e-3) / 1e6;
    printf("{\"shape\": %d, \"kernel\": \"pure bf16 MFMA loop (32: v_mfma_f32_32x32x16_bf16, 16: two v_mfma_f32_16x16x32_bf16 per 32x32 one), %d independent 32x32-equivalent accumulators per wave, %d wave(s) per SIMD, %s operands\", "
           "\"cus\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"frac_of_2500\": %.3f, \"mfma_cycles_per_instr_at_2400MHz\": %.1f, \"shader_clock_mhz_under_load\": %.0f, \"cycles_per_mfma_at_that_clock\": %.1f}\n",
           shape, NACC, wps, zero ? "zero" : "random", cus, best, flop / best / 1e9, flop / best / 1e9 / 2500.0,
           best * 1e-3 * 2.4e9 / ((double)NACC * iters * wps), mhz, (double)hc / ((double)NACC * iters));
    return 0;
}
