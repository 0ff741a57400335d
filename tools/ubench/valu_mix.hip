// Do two waves on one SIMD overlap DIFFERENT instruction kinds?  Waves 0-3 (one per SIMD) run kind A, waves 4-7 kind B; time for both vs each alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND> __device__ __forceinline__ void body(float (&a)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(1.0001f));
        if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
        if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(1.0f));
    }
}
template <int KA, int KB>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int mode) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1) for (int it = 0; it < iters; ++it) { body<KA>(a); body<KA>(a); body<KA>(a); body<KA>(a); }
    } else {
        if (mode & 2) for (int it = 0; it < iters; ++it) { body<KB>(a); body<KB>(a); body<KB>(a); body<KB>(a); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int KA, int KB> void run(const char* name) {
    const int iters = 2000;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    for (int mode = 1; mode <= 3; ++mode) {
        k<KA, KB><<<256, 512>>>(out, cyc, iters, mode);
        k<KA, KB><<<256, 512>>>(out, cyc, iters, mode);
        (void)hipDeviceSynchronize();
        unsigned long long h[8];
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-22s mode %d (1: A alone, 2: B alone, 3: both): wave0 %.2f  wave4 %.2f cycles per instr\n", name, mode, h[0] / (iters * 32.0), h[4] / (iters * 32.0));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<1, 0>("A=exp B=fma");
    run<1, 2>("A=exp B=cvt_pk");
    run<2, 0>("A=cvt_pk B=fma");
    run<1, 4>("A=exp B=max3");
    run<2, 4>("A=cvt_pk B=max3");
    return 0;
}
