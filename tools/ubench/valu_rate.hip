// VALU issue capacity of one SIMD with 1 / 2 / 3 / 4 waves on it, for the instruction kinds of the attention softmax, alone and beside MFMAs
// (hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate).  Prints cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int KIND, int MFMA>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    f32x16 acc = {};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(1.0001f));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
                if (KIND == 3) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a[i]));
                if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(1.0f));
            }
            if (KIND == 5) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(double*)&a[i]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND, int MFMA> void run(const char* name, int waves_per_simd) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 16 * 8);
    k<KIND, MFMA><<<blocks, threads>>>(out, cyc, iters);
    k<KIND, MFMA><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
    const int valu_per_wave = iters * 4 * (KIND == 5 ? 4 : 8);
    const int mfma_per_wave = MFMA ? iters * 4 : 0;
    printf("%-12s mfma=%d waves/SIMD=%d: %.2f cycles per VALU instr per SIMD (per wave %.2f)%s\n", name, MFMA, waves_per_simd,
           mx / (valu_per_wave * waves_per_simd), mx / valu_per_wave, MFMA ? "" : "");
    if (MFMA) printf("             -> %.1f cycles per (1 MFMA + %d VALU) group per wave, %.1f per SIMD\n", mx / mfma_per_wave, KIND == 5 ? 4 : 8, mx / mfma_per_wave / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<0, 0>("v_fma_f32", w);
        run<1, 0>("v_exp_f32", w);
        run<2, 0>("v_cvt_pk", w);
        run<3, 0>("v_and_b32", w);
        run<4, 0>("v_max3_f32", w);
        run<5, 0>("v_pk_add_f32", w);
    }
    for (int w = 1; w <= 2; ++w) {
        run<0, 1>("v_fma_f32", w);
        run<1, 1>("v_exp_f32", w);
        run<2, 1>("v_cvt_pk", w);
        run<5, 1>("v_pk_add_f32", w);
    }
    return 0;
}
