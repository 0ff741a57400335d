// Prototype for the operand format proposed for the next round (Next (1)): the SAME simple LDS-tiled GEMM  C[M][N] = A[M][K] W[N][K]^T  (f32 out)
// in two operand formats of 4 bytes per element, so that global / LDS traffic is identical and only the matrix-pipe work differs:
//   x3  : split bf16 (shipping): a 128-byte row piece = one 32-wide k group [hi bf16 x 32 | lo bf16 x 32]; 3 v_mfma_f32_32x32x16_bf16 per 16 k
//   hf8 : per 64-wide k group TWO 128-byte pieces: [hi f16 x 64] and [a8 x 64 | lo8 x 64] (e4m3; a8 = fp8 of the whole value, lo8 = fp8 of (x - hi) * 2^12);
//         4 v_mfma_f32_32x32x16_f16 (hi . hi) + 2 v_mfma_scale_f32_32x32x64_f8f6f4 (a8 . wl8 and al8 . w8, the power-of-two factors as E8M0 scales) per 64 k
// The kernel is deliberately plain (128 x 128 tile, 4 waves of 64 x 64, double-buffered LDS filled through registers one K tile ahead, XOR-swizzled 16-byte
// chunks, one barrier per K tile, two workgroups per CU): NOT the tuned tile kernel of csrc/gemm.hip - the question is the RATIO between the two formats on a
// real LDS-fed loop at the power cap, and the accuracy of hf8 on the real instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_format_proto.hip -o tools/ubench/gemm_format_proto && tools/ubench/gemm_format_proto [M N K]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128, ROWB = 128;            // tile rows, bytes of k per row and K tile
constexpr int TILE_BYTES = BM * ROWB;                    // 16 KB per operand and stage

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + 16 * (chunk ^ (row & 7)); }

// FORMAT 0: x3, 1: hf8.  K tiles: ktiles = K / 32 for both (x3: one 32-group per tile; hf8: tiles alternate hi / 8-bit, two per 64-group)
template <int FORMAT, int ABL = 0>      // ABL (timing only, results invalid): 1 no MFMAs, 2 no global loads after the first tile, 4 no LDS stores, 8 no fragment reads; 16 (VALID results): staging by LDS-DMA
// (global_load_lds_dwordx4: L2 -> LDS without the register round trip and the ds_write; the XOR swizzle goes on the per-lane SOURCE address)
__global__ __launch_bounds__(256, 2) void gemm_proto(const char* __restrict__ A, const char* __restrict__ W, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const long ldb = (long)K * 4;                        // bytes per operand row (4 B per element in both formats)
    const int nkt = K / 32;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging: 128 rows x 8 chunks per operand = 1024 chunks -> 4 per thread and operand
    i32x4 ra[4], rb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 3, c = q & 7;
            int gm = m0 + row;
            gm = gm < M ? gm : M - 1;
            ra[i] = *(const i32x4*)(A + (long)gm * ldb + (long)kt * ROWB + 16 * c);
            rb[i] = *(const i32x4*)(W + (long)(n0 + row) * ldb + (long)kt * ROWB + 16 * c);
        }
    };
    auto lstore = [&](char* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 3, c = q & 7;
            *(i32x4*)(st + lds_off(row, c)) = ra[i];
            *(i32x4*)(st + TILE_BYTES + lds_off(row, c)) = rb[i];
        }
    };
    auto frag = [&](const char* t, int rowbase, int chunk) { return *(const i32x4*)(t + lds_off(rowbase + (lane & 31), chunk)); };
    const int h = lane >> 5;
    i32x4 keep[2][2][4];
    (void)keep;
    // LDS-DMA: one wave instruction fills 1 KB = 8 rows x 128 B (lane l -> row 8 i + l / 8, slot l % 8, which must hold chunk slot ^ (row & 7))
    auto dma = [&](int kt, char* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int blk = wave * 4 + i, row = 8 * blk + (lane >> 3), c = (lane & 7) ^ (row & 7);
            int gm = m0 + row;
            gm = gm < M ? gm : M - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (long)gm * ldb + (long)kt * ROWB + 16 * c),
                                             (__attribute__((address_space(3))) void*)(st + blk * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (long)(n0 + row) * ldb + (long)kt * ROWB + 16 * c),
                                             (__attribute__((address_space(3))) void*)(st + TILE_BYTES + blk * 1024), 16, 0, 0);
        }
    };
    if constexpr (ABL & 16) {
        dma(0, lds);
    } else {
        gload(0);
        lstore(lds);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const char* ta = lds + (kt & 1) * 2 * TILE_BYTES;
        const char* tb = ta + TILE_BYTES;
        if constexpr (ABL & 16) { if (kt + 1 < nkt) dma(kt + 1, lds + ((kt + 1) & 1) * 2 * TILE_BYTES); }
        else if (kt + 1 < nkt && !(ABL & 2)) gload(kt + 1);
        if constexpr (FORMAT == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if ((ABL & 8) && kt > 0) {        // keep the first tile's fragments (made opaque so that nothing is hoisted or folded)
                        asm volatile("" : "+v"(keep[s][i][0]), "+v"(keep[s][i][1]), "+v"(keep[s][i][2]), "+v"(keep[s][i][3]));
                        ah[i] = __builtin_bit_cast(bf16x8, keep[s][i][0]); al[i] = __builtin_bit_cast(bf16x8, keep[s][i][1]);
                        bh[i] = __builtin_bit_cast(bf16x8, keep[s][i][2]); bl[i] = __builtin_bit_cast(bf16x8, keep[s][i][3]);
                        continue;
                    }
                    ah[i] = __builtin_bit_cast(bf16x8, frag(ta, (wm * 2 + i) * 32, 2 * s + h));
                    al[i] = __builtin_bit_cast(bf16x8, frag(ta, (wm * 2 + i) * 32, 4 + 2 * s + h));
                    bh[i] = __builtin_bit_cast(bf16x8, frag(tb, (wn * 2 + i) * 32, 2 * s + h));
                    bl[i] = __builtin_bit_cast(bf16x8, frag(tb, (wn * 2 + i) * 32, 4 + 2 * s + h));
                    if (ABL & 8) { keep[s][i][0] = __builtin_bit_cast(i32x4, ah[i]); keep[s][i][1] = __builtin_bit_cast(i32x4, al[i]);
                                   keep[s][i][2] = __builtin_bit_cast(i32x4, bh[i]); keep[s][i][3] = __builtin_bit_cast(i32x4, bl[i]); }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (ABL & 1) {                // consume the fragments with two VALU ops instead of three MFMAs
                            const i32x4 u = __builtin_bit_cast(i32x4, al[i]) ^ __builtin_bit_cast(i32x4, bh[j]) ^ __builtin_bit_cast(i32x4, ah[i]) ^ __builtin_bit_cast(i32x4, bl[j]);
                            acc[i][j][0] += __int_as_float((u[0] ^ u[1] ^ u[2] ^ u[3]) & 0x3fffffff);
                            continue;
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            if ((kt & 1) == 0) {                          // hi tile: 64 k values of fp16 = four k steps
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f16x8 a[2], b[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        a[i] = __builtin_bit_cast(f16x8, frag(ta, (wm * 2 + i) * 32, 2 * s + h));
                        b[i] = __builtin_bit_cast(f16x8, frag(tb, (wn * 2 + i) * 32, 2 * s + h));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            } else {                                      // 8-bit tile: [x8 x 64 | lo8 x 64]; lane half h takes bytes 32 h .. 32 h + 31 of each part
                i32x8 a8[2], al8[2], w8[2], wl8[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const i32x4 p0 = frag(ta, (wm * 2 + i) * 32, 2 * h), p1 = frag(ta, (wm * 2 + i) * 32, 2 * h + 1);
                    const i32x4 q0 = frag(ta, (wm * 2 + i) * 32, 4 + 2 * h), q1 = frag(ta, (wm * 2 + i) * 32, 4 + 2 * h + 1);
                    a8[i] = i32x8{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
                    al8[i] = i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
                    const i32x4 u0 = frag(tb, (wn * 2 + i) * 32, 2 * h), u1 = frag(tb, (wn * 2 + i) * 32, 2 * h + 1);
                    const i32x4 v0 = frag(tb, (wn * 2 + i) * 32, 4 + 2 * h), v1 = frag(tb, (wn * 2 + i) * 32, 4 + 2 * h + 1);
                    w8[i] = i32x8{u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
                    wl8[i] = i32x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                }
                // E8M0 scales: a8 2^0 (127), al8 2^-12 (115), w8 2^-4 (123), wl8 2^-16 (111): the same byte in all four positions
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], wl8[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x6f6f6f6f);
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(al8[i], w8[j], acc[i][j], 0, 0, 0, 0x73737373, 0, 0x7b7b7b7b);
                    }
            }
        }
        if (kt + 1 < nkt && !(ABL & 4) && !(ABL & 16)) lstore(lds + ((kt + 1) & 1) * 2 * TILE_BYTES);
        __syncthreads();
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + (wn * 2 + j) * 32 + (lane & 31);
                if (m < M) C[(long)m * N + n] = acc[i][j][r];
            }
}

// ---- host-side packing
static uint16_t f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2h(float x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
static uint8_t f2e4m3(float x) {      // OCP e4m3fn: bias 7, max 448, subnormals 2^-9 steps, round to nearest even, saturating
    const uint8_t s = x < 0 ? 0x80 : 0;
    float a = fabsf(x);
    if (!(a == a)) return 0x7f;
    if (a >= 448.f) return s | 0x7e;
    if (a < 0.0009765625f) return s;                                   // < 2^-10: rounds to zero
    int e; float m = frexpf(a, &e);                                    // a = m 2^e, m in [0.5, 1)
    int E = e - 1 + 7;                                                 // biased exponent of 1.f form
    if (E <= 0) {                                                      // subnormal: units of 2^-9
        const int q = (int)nearbyintf(a * 512.f);
        return s | (uint8_t)(q > 7 ? 8 : q);                           // q == 8 is the smallest normal
    }
    int q = (int)nearbyintf((m * 2.f - 1.f) * 8.f);                    // 3 mantissa bits
    if (q == 8) { q = 0; ++E; }
    if (E > 15 || (E == 15 && q > 6)) return s | 0x7e;
    return s | (uint8_t)(E << 3) | (uint8_t)q;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 25216, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 384;
    if (M % 1 || N % 128 || K % 64) { printf("N %% 128 and K %% 64 required\n"); return 1; }
    std::vector<float> a((size_t)M * K), w((size_t)N * K);
    unsigned x = 1234567u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (float)((x >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : a) { float s = rnd() + rnd() + rnd() + rnd(); v = 1.7f * s; }          // ~ N(0, 1)-ish activations
    for (size_t i = 0; i < a.size(); i += 97) a[i] *= 6.f;                                 // a few outliers
    for (auto& v : w) { float s = rnd() + rnd() + rnd() + rnd(); v = 0.09f * s; }
    std::vector<char> pa[2], pw[2];
    auto pack = [&](const std::vector<float>& src, int rows, float pre, int fmt) {
        std::vector<char> out((size_t)rows * K * 4);
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < K; ++k) {
                const float v = src[(size_t)r * K + k];
                char* row = out.data() + (size_t)r * K * 4;
                if (fmt == 0) {                           // group of 32: [hi x 32 | lo x 32]
                    const uint16_t hi = f2bf(v), lo = f2bf(v - bf2f(hi));
                    char* g = row + (k / 32) * 128;
                    memcpy(g + 2 * (k % 32), &hi, 2);
                    memcpy(g + 64 + 2 * (k % 32), &lo, 2);
                } else {                                  // group of 64: [hi f16 x 64][x8 x 64 | lo8 x 64]
                    const uint16_t hi = f2h(v);
                    char* g = row + (k / 64) * 256;
                    memcpy(g + 2 * (k % 64), &hi, 2);
                    g[128 + (k % 64)] = (char)f2e4m3(v * pre);
                    g[192 + (k % 64)] = (char)f2e4m3((v - h2f(hi)) * pre * 4096.f);
                }
            }
        return out;
    };
    for (int f = 0; f < 2; ++f) { pa[f] = pack(a, M, 1.f, f); pw[f] = pack(w, N, 16.f, f); }
    char *dA, *dW; float* dC;
    (void)hipMalloc(&dA, pa[0].size()); (void)hipMalloc(&dW, pw[0].size()); (void)hipMalloc(&dC, (size_t)M * N * 4);
    std::vector<float> c((size_t)M * N);
    const int grid = ((M + BM - 1) / BM) * (N / BN), ldsb = 4 * TILE_BYTES;
    (void)hipFuncSetAttribute((const void*)gemm_proto<0>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    (void)hipFuncSetAttribute((const void*)gemm_proto<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    (void)hipFuncSetAttribute((const void*)gemm_proto<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); (void)hipFuncSetAttribute((const void*)gemm_proto<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    (void)hipFuncSetAttribute((const void*)gemm_proto<0, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); (void)hipFuncSetAttribute((const void*)gemm_proto<0, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    (void)hipFuncSetAttribute((const void*)gemm_proto<0, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); (void)hipFuncSetAttribute((const void*)gemm_proto<0, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    (void)hipFuncSetAttribute((const void*)gemm_proto<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    (void)hipFuncSetAttribute((const void*)gemm_proto<0, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); (void)hipFuncSetAttribute((const void*)gemm_proto<0, 17>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    double t0 = 0;
    for (int f = 0; f < 2; ++f) {
        (void)hipMemcpy(dA, pa[f].data(), pa[f].size(), hipMemcpyHostToDevice);
        (void)hipMemcpy(dW, pw[f].data(), pw[f].size(), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 30; ++rep) {
            (void)hipEventRecord(e0);
            if (f == 0) hipLaunchKernelGGL(gemm_proto<0>, dim3(grid), dim3(256), ldsb, 0, dA, dW, dC, M, N, K);
            else hipLaunchKernelGGL(gemm_proto<1>, dim3(grid), dim3(256), ldsb, 0, dA, dW, dC, M, N, K);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 5 && ms < best) best = ms;
        }
        (void)hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
        // accuracy on 64 sampled rows against float64
        double emax = 0, ymax = 0, e2 = 0, y2 = 0;
        for (int s = 0; s < 64; ++s) {
            const int m = (int)(((long)s * 7919) % M);
            for (int n = 0; n < N; ++n) {
                double y = 0;
                for (int k = 0; k < K; ++k) y += (double)a[(size_t)m * K + k] * (double)w[(size_t)n * K + k];
                const double d = fabs((double)c[(size_t)m * N + n] - y);
                emax = d > emax ? d : emax; ymax = fabs(y) > ymax ? fabs(y) : ymax; e2 += d * d; y2 += y * y;
            }
        }
        if (f == 0 && argc > 4) {           // ablation sweep of the x3 loop (timing only)
            auto timeit = [&](auto kern, const char* what) {
                float b2 = 1e30f;
                for (int rep = 0; rep < 20; ++rep) {
                    (void)hipEventRecord(e0);
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, 0, dA, dW, dC, M, N, K);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    if (rep >= 5 && ms < b2) b2 = ms;
                }
                printf("   x3 ablation %-58s %8.1f us\n", what, b2 * 1e3);
            };
            timeit(gemm_proto<0, 1>, "no MFMAs (loads + LDS stores + fragment reads)");
            timeit(gemm_proto<0, 2>, "no global loads after the first tile");
            timeit(gemm_proto<0, 2 | 4>, "no global loads, no LDS stores (fragment reads + MFMAs)");
            timeit(gemm_proto<0, 2 | 4 | 8>, "MFMAs only");
            timeit(gemm_proto<0, 1 | 8>, "loads + LDS stores only (no fragment reads, no MFMAs)");
            timeit(gemm_proto<0, 1 | 4 | 8>, "global loads only");
            timeit(gemm_proto<0, 8>, "no fragment reads (loads + LDS stores + MFMAs)");
            timeit(gemm_proto<0, 16>, "FULL kernel with LDS-DMA staging (valid results)");
            timeit(gemm_proto<0, 16 | 1>, "LDS-DMA staging, no MFMAs");
            (void)hipMemset(dC, 0, (size_t)M * N * 4);
            hipLaunchKernelGGL((gemm_proto<0, 16>), dim3(grid), dim3(256), ldsb, 0, dA, dW, dC, M, N, K);
            (void)hipDeviceSynchronize();
            {
                std::vector<float> c2((size_t)64 * N);
                double emax = 0, ymax = 0;
                for (int s2 = 0; s2 < 64; ++s2) {
                    const int m = (int)(((long)s2 * 7919) % M);
                    (void)hipMemcpy(c2.data(), dC + (size_t)m * N, (size_t)N * 4, hipMemcpyDeviceToHost);
                    for (int n = 0; n < N; ++n) {
                        double y = 0;
                        for (int k = 0; k < K; ++k) y += (double)a[(size_t)m * K + k] * (double)w[(size_t)n * K + k];
                        const double d = fabs((double)c2[n] - y);
                        emax = d > emax ? d : emax; ymax = fabs(y) > ymax ? fabs(y) : ymax;
                    }
                }
                printf("   LDS-DMA staging: max-rel error against float64 %.2e\n", emax / ymax);
            }
        }
        const double tf = 2.0 * M * N * K / (best * 1e-3) / 1e12;
        if (f == 0) t0 = best;
        printf("%-4s M %d N %d K %d: %8.1f us  %7.1f algorithmic TFLOP/s  (%.2f x)   max-rel %.2e  rms-rel %.2e\n", f == 0 ? "x3" : "hf8", M, N, K, best * 1e3, tf,
               t0 / best, emax / ymax, sqrt(e2 / y2));
    }
    return 0;
}
