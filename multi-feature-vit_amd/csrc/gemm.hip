// MFMA GEMM kernels of the MF-ViT encoder path (gfx950).
//
//   gemm_nt_tile  : C[M][N] = epi(A[M][K] * W[N][K]^T), 128x128 block tile, 4 waves (2x2), 64x64 per wave.
//                   epilogues: bias | bias+GELU (pre & act) | GELU' * (.) + column sums | none
//                   replaces torch nn.Linear fwd / dgrad of timm Block (qkv, fc1, fc2-dgrad, proj-dgrad)
//   gemm_nt_row   : same product with a ROW-COMPLETE 64x384 tile (N == 384 == embed dim), 4 waves (1x4), so the
//                   epilogue owns whole token rows:  +bias +residual -> LayerNorm (forward)  or
//                   LayerNorm-backward + residual-gradient add + dgamma/dbeta/dbias column sums (backward).
//                   replaces proj/fc2/patch-embed (+ the following nn.LayerNorm) and qkv-/fc1-dgrad (+ LN bwd)
//   gemm_tn       : dW[N][K] += dY[m][N]^T * X[m][K] over an m-range (split over grid.y, f32 atomics),
//                   operands K-strided -> transposing LDS reads.  replaces nn.Linear wgrad
// Both element types: bf16 (v_mfma_f32_32x32x16_bf16) and f32 (v_mfma_f32_32x32x2_f32, exact f32).
#include "kernels.h"
#include "prof.h"

#include <stdlib.h>

#include <type_traits>

// A/B builds only (tools/build_variant_lib.sh): 1 = plain (L2-resident) stores for EVERY output of the tile kernel (default: qkv only, see copy_tile), 2 = for the fc2 data gradient, 3 = for gelu(h)
#ifndef MFVIT_TILE_PLAIN_STORE
#define MFVIT_TILE_PLAIN_STORE 0
#endif

namespace mfvit {




// ------------------------------------------------------------------------------------------ tile kernel
// (Round 5 tried an N-wide 128 x 256 tile of 8 waves, one workgroup per CU - half the A re-reads across the N tiles of a row block, 3/4 of the staged
// operand bytes per MFMA: bit-identical, L2 requests - 17 %, and 4 - 9 % SLOWER (fc1 + GELU 130.0 -> 135.7 us, fc2 data gradient 112.9 -> 122.9 us);
// like the M-tall 256 x 128 tile of round 4 it gives up the second independent workgroup per CU.  Removed; profiles/r05_tile_gemm_experiments.txt.)
template <typename T, int EPI, int DEEP = 0>     // DEEP: K tiles kept in flight by the main loop (0: NtLoop's one; 2: NtLoopDeep, 16-bit types; 12: interleaved, split)
__global__ __launch_bounds__(256, 2) void gemm_nt_tile_kernel(GemmP p) {   // 2 workgroups per CU (64 - 68 KB of LDS each): register budget 256 per wave
    constexpr int BM = 128, BN = 128, BKB = 128, WM = 2, WN = 2, NTHR = WM * WN * 64;
    typedef NtLoop<T, BM, BN, BKB, WM, WN> Loop;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    apply_batch<T>(p, sizeof(T));
    const int ntn = p.N / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, ntn * ntm);
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    f32x16 acc[Loop::TM][Loop::TN];
    if constexpr (DEEP == 12) {          // two K tiles in flight, loads / LDS stores interleaved with the MFMAs (split bf16)
        NtLoopDeep<T, BM, BN, BKB, WM, WN, 2, true>::run(p, m0, n0, lds, acc);
    } else if constexpr (DEEP > 0) {
        NtLoopDeep<T, BM, BN, BKB, WM, WN, DEEP>::run(p, m0, n0, lds, acc);
    } else {
        Loop::run(p, m0, n0, lds, acc);
    }

    // ---- epilogue through LDS: per-element math in registers -> [128][128 + pad] tile in LDS -> 16-byte coalesced stores.
    // Rows >= M replicate row M-1 exactly (the A loads are clamped), so they are stored as identical duplicates: no branches.
    // The activation derivative (out0 of the GELU / ReLU epilogues, aux of EPI_GELU_BWD) is kept in AX: the tensor's own type for bf16 /
    // fp16 / f32, plain fp16 for split tensors - it only ever multiplies a gradient (2^-11 relative on values in [-0.13, 1.13], 1e-5 after
    // the k-sums of the following GEMMs), and the split copy was 155 MB of stores in fc1 and as many loads in the fc2 dgrad per launch.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    typedef typename act_grad_type<T>::type AX;
    constexpr int EP = elems_per<T>::value;             // storage elements per logical element (2 for split tensors)
    constexpr int ROWB = BN * (int)sizeof(T) * EP;      // bytes of one tile row in the output tensor
    constexpr int AROWB = BN * (int)sizeof(AX);         // ... in the activation-derivative tensor
    char* tile = lds;                                   // the staging buffers are free after the main loop's last barrier
    auto trow = [&](int i, int r, int pitch) -> char* { return tile + ((wm * Loop::TM + i) * 32 + acc_row(r, lane)) * pitch; };
    auto ecol = [&](int j) -> int { return (wn * Loop::TN + j) * 32 + (lane & 31); };   // LOGICAL column inside the tile
    // tile rows of RB bytes (LDS pitch RB + 16)  <->  global rows (ld_bytes apart, the tile's columns start col_bytes into the row)
    auto copy_tile = [&](auto rb_c, auto to_global_c, char* g, long ld_bytes, long col_bytes) {
        constexpr int RB = decltype(rb_c)::value, CP = RB / 16, NC = BM * CP / NTHR;
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int q = tid + i * NTHR, row = q / CP, c = q % CP;
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            u32x4* gp = (u32x4*)(g + (long)m * ld_bytes + col_bytes + 16 * c);
            u32x4* lp = (u32x4*)(tile + row * (RB + 16) + 16 * c);
            if constexpr (decltype(to_global_c)::value) {
                // qkv (the attention core reads it NEXT, 116 MB): a PLAIN store - the lines stay in the L2 / Infinity Cache on their way out and the
                // forward attention launch behind it runs 54.5 -> 50.5 us in the step (same-box A/B, round 5; an `nt` store does not: 54.4).  Every
                // other output of this kernel leaves through system-scope streaming stores (no write-allocate fetch, common.cuh).
                if constexpr (EPI == EPI_BIAS_X3F16 || EPI == EPI_BIAS || MFVIT_TILE_PLAIN_STORE == 1 || (MFVIT_TILE_PLAIN_STORE == 2 && EPI == EPI_GELU_BWD) ||
                              (MFVIT_TILE_PLAIN_STORE == 3 && EPI == EPI_BIAS_GELU && RB == ROWB && ROWB != AROWB)) *gp = *lp;
                else
                store16_stream(gp, *lp);                                         // system-scope streaming store: no write-allocate fetch (common.cuh)
            } else {
                *lp = *gp;
            }
        }
    };
    constexpr std::integral_constant<int, ROWB> rb_t{};
    constexpr std::integral_constant<int, AROWB> rb_a{};
    auto store_tile = [&](void* out, long ldo) {
        __syncthreads();
        copy_tile(rb_t, std::true_type(), (char*)out, ldo * (long)sizeof(T), (long)n0 * EP * (long)sizeof(T));
    };
    float bj[Loop::TN], csums[Loop::TN];
#pragma unroll
    for (int j = 0; j < Loop::TN; ++j) {
        csums[j] = 0.f;
        const int n = n0 + (wn * Loop::TN + j) * 32 + (lane & 31);
        bj[j] = ((EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_X3F16) && p.bias) ? p.bias[n] : 0.f;
    }
    if constexpr (EPI == EPI_GELU_BWD) {
        // aux = act'(pre-activation) saved by the forward: its tile comes in with 16-byte reads, every lane multiplies its accumulators in
        // place, and only then (barrier) the product goes into the same LDS as a tile of T
        copy_tile(rb_a, std::false_type(), (char*)const_cast<void*>(p.aux), p.ldaux * (long)sizeof(AX), (long)n0 * (long)sizeof(AX));
        __syncthreads();
#pragma unroll
        for (int j = 0; j < Loop::TN; ++j) {
            float csum = 0.f;
#pragma unroll
            for (int i = 0; i < Loop::TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r] * load_elem<AX>((const AX*)trow(i, r, AROWB + 16), ecol(j));
                    acc[i][j][r] = v;
                    csum += m0 + (wm * Loop::TM + i) * 32 + acc_row(r, lane) < p.M ? v : 0.f;
                }
            if (p.cs0) {
                csum += __shfl_xor(csum, 32, 64);
                if (p.cpart) csums[j] = csum;
                else if (lane < 32) atomicAdd(p.cs0 + n0 + (wn * Loop::TN + j) * 32 + lane, csum);
            }
        }
        __syncthreads();
    }
    if constexpr (EPI == EPI_NONE) {
        // The proj data gradient is the attention backward's dO, and its split-fp16 core scales every (image, head) by a power of two taken from the
        // largest |dO| of the pair (attention_mfma.hip, pow2_scale): that maximum leaves HERE, where the values are in registers - a wave owns 64 rows x
        // two 32-column slices (one head each, or the halves of a 64-wide head), its rows belong to at most two images - instead of a prefetch of all of
        // dO's hi parts by the consumer in front of each pair.
        if (p.omax) {
            const int r0 = m0 + wm * (Loop::TM * 32);
            const int img = r0 / p.omax_rows;
            const int mb = (img + 1) * p.omax_rows - r0;          // rows of this wave at and past mb belong to the next image (or repeat row M - 1: dropped)
            const int nimg = p.M / p.omax_rows, nh = p.N / p.omax_hd;
            const bool two = mb < Loop::TM * 32;                    // (wave-uniform: two of three waves lie inside one image and skip the row tests)
#pragma unroll
            for (int j = 0; j < Loop::TN; ++j) {
                float mx0 = 0.f, mx1 = 0.f;
                if (two) {
#pragma unroll
                    for (int i = 0; i < Loop::TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float a = fabsf(acc[i][j][r]);
                            if (i * 32 + acc_row(r, lane) < mb) mx0 = fmaxf(mx0, a);
                            else mx1 = fmaxf(mx1, a);
                        }
                    mx1 = wave_max_nonneg_dpp(mx1);
                } else {
#pragma unroll
                    for (int i = 0; i < Loop::TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) mx0 = fmaxf(mx0, fmaxf(fabsf(acc[i][j][r]), fabsf(acc[i][j][r + 1])));
                }
                mx0 = wave_max_nonneg_dpp(mx0);
                const int head = (n0 + (wn * Loop::TN + j) * 32) / p.omax_hd;
                if (lane == 0) {
                    if (img < nimg) atomicMax(p.omax + img * nh + head, __builtin_bit_cast(unsigned, mx0));
                    if (two && img + 1 < nimg) atomicMax(p.omax + (img + 1) * nh + head, __builtin_bit_cast(unsigned, mx1));
                }
            }
        }
    }
    const bool want_grad = p.out0 != nullptr;           // GELU / ReLU epilogues: no-grad forwards pass out0 = NULL and skip the derivative
    if ((EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU) ? want_grad : true) {
#pragma unroll
        for (int j = 0; j < Loop::TN; ++j)
#pragma unroll
            for (int i = 0; i < Loop::TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {     // two rows at a time: one packed conversion per pair of values
                    const float v0 = acc[i][j][r] + bj[j], v1 = acc[i][j][r + 1] + bj[j];
                    if (EPI == EPI_BIAS_GELU) {
                        float g0, g1, d0, d1;
                        gelu_both_t<T>(v0, g0, d0);                         // one erf / exp evaluation for both outputs
                        gelu_both_t<T>(v1, g1, d1);
                        acc[i][j][r] = g0;                                  // gelu(pre): the second output, stored below
                        acc[i][j][r + 1] = g1;
                        store_elem_pair<AX>((AX*)trow(i, r, AROWB + 16), (AX*)trow(i, r + 1, AROWB + 16), ecol(j), d0, d1);   // out0 = gelu'(pre)
                    } else if (EPI == EPI_BIAS_RELU) {
                        acc[i][j][r] = fmaxf(v0, 0.f);
                        acc[i][j][r + 1] = fmaxf(v1, 0.f);
                        store_elem_pair<AX>((AX*)trow(i, r, AROWB + 16), (AX*)trow(i, r + 1, AROWB + 16), ecol(j), v0 > 0.f ? 1.f : 0.f,
                                            v1 > 0.f ? 1.f : 0.f);         // out0 = relu'(pre) (fuseattention.py:69: nn.ReLU)
                    } else if (EPI == EPI_BIAS_X3F16) {             // the same tile bytes, hi / lo parts in fp16 (the attention core's operand format)
                        store_elem_pair<sf16>((sf16*)trow(i, r, ROWB + 16), (sf16*)trow(i, r + 1, ROWB + 16), ecol(j), v0, v1);
                    } else {
                        store_elem_pair<T>((T*)trow(i, r, ROWB + 16), (T*)trow(i, r + 1, ROWB + 16), ecol(j), v0, v1);
                    }
                }
        if (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU) {
            __syncthreads();
            copy_tile(rb_a, std::true_type(), (char*)p.out0, p.ldo0 * (long)sizeof(AX), (long)n0 * (long)sizeof(AX));
        } else {
            store_tile(p.out0, p.ldo0);
        }
    } else {
        // activation only (no-grad forward)
#pragma unroll
        for (int j = 0; j < Loop::TN; ++j)
#pragma unroll
            for (int i = 0; i < Loop::TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r] + bj[j];
                    acc[i][j][r] = EPI == EPI_BIAS_RELU ? fmaxf(v, 0.f) : gelu_t<T>(v);
                }
    }
    if (EPI == EPI_GELU_BWD && p.cs0 && p.cpart) {   // per-workgroup partial column sums: [m-tile][N], plain stores
        __syncthreads();
        float* sc = (float*)tile;                       // [WM][BN]
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < Loop::TN; ++j) sc[wm * BN + (wn * Loop::TN + j) * 32 + lane] = csums[j];
        }
        __syncthreads();
        if (tid < BN) p.cpart[(long)(m0 / BM) * p.N + n0 + tid] = sc[tid] + sc[BN + tid];
    }
    if (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < Loop::TN; ++j)
#pragma unroll
            for (int i = 0; i < Loop::TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2)
                    store_elem_pair<T>((T*)trow(i, r, ROWB + 16), (T*)trow(i, r + 1, ROWB + 16), ecol(j), acc[i][j][r], acc[i][j][r + 1]);
        store_tile(p.out1, p.ldo1);
    }
}

// ------------------------------------------------------------------------------------------- row kernel
constexpr int ROW_BN = 384, ROW_BKB = 64, ROW_RS = 132;

// Full-row totals of per-lane partials p[i][r] (row = i*32 + acc_row(r)), summed over the 32 column lanes of the
// 4 waves, through LDS (red: [64][132] floats, tot: [64]).  Three barriers; all 256 threads must call it.
template <int TM, int BM, int NTHREADS, bool READBACK = true>
__device__ __forceinline__ void row_reduce(float (&p)[TM][16], float* red, float* tot, int lane, int wm, int wn, int tid) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wm * TM + i) * 32 + acc_row(r, lane)) * ROW_RS + wn * 32 + (lane & 31)] = p[i][r];
    __syncthreads();
    for (int t2 = tid; t2 < BM * 4; t2 += NTHREADS) {
        const int row = t2 >> 2, q = t2 & 3;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 v = *(const float4*)&red[row * ROW_RS + 32 * q + 4 * k];
            s += (v.x + v.y) + (v.z + v.w);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (q == 0) tot[row] = s;
    }
    __syncthreads();
    if constexpr (READBACK) {       // (otherwise the caller reads tot[] when it needs a total; red is free again after the barrier above)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) p[i][r] = tot[(wm * TM + i) * 32 + acc_row(r, lane)];
        __syncthreads();
    }
}

// Epilogues of the TWO-WORKGROUPS-PER-CU row kernel (4 waves side by side, 64 x 96 per wave: TM = 2, TN = 3; 256 registers).  Same
// arithmetic as the epilogues inside gemm_nt_row_kernel below, organised for the register budget: row partials go straight into the LDS
// reduction arrays and the row totals / saved statistics are read back from LDS where they are used (no [TM][16] register copies), the
// LayerNorm backward recomputes the normalised input from a second read of x (L2: this workgroup fetched it a moment ago) instead of
// keeping it in 96 registers, and global loads are issued in batches of 24 - 48 with the arithmetic behind a scheduling barrier.
constexpr int ROW_LEAN_LDS = 2 * 64 * ROW_RS * 4 + 4 * 64 * 4;
// (Measured and not kept: staging each wave's 32 x 96 output block in a wave-private LDS slice and writing it with 16-byte stores instead of
// the 96 four-byte + 192 two-byte stores per thread below - proj + LN 50 -> 53.5 us, fc2 + LN 155 -> 162, LayerNorm-backward 130 -> 129 /
// 174 -> 160: the stores cost their BYTES, not their instruction count; what pays is not writing a tensor at all, see `res_t`.)
// REMAP: the patch-embedding launch only (output rows re-indexed past the cls row, residual = pos_embed[row % patches]); everything
// else gets straight-line code without the per-row `orow_in ? ... : m` selects and their integer divisions
template <typename T, int REPI, int BM, int TM, int TN, bool REMAP>
__device__ __forceinline__ void row_epilogue_lean(const GemmP& p, int m0, f32x16 (&acc)[TM][TN], char* lds) {
    static_assert(BM == 64 && TM == 2, "one row of four waves over a 64-row tile");
    constexpr int BN = ROW_BN;
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    float* red1 = (float*)lds;
    float* red2 = red1 + BM * ROW_RS;
    float* tot1 = red2 + BM * ROW_RS;
    float* tot2 = tot1 + BM;
    float* smu = tot2 + BM;
    float* srs = smu + BM;
    const float invN = 1.0f / (float)BN;
    int ncol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) ncol[j] = (wn * TN + j) * 32 + (lane & 31);
    auto lrow = [&](int i, int r) { return i * 32 + acc_row(r, lane); };
    auto put = [&](float* red, int i, int r, float v) { red[lrow(i, r) * ROW_RS + wn * 32 + (lane & 31)] = v; };
    // totals of the 128 partials of every row: thread (row, quarter) sums 32 of them, two shuffles finish the row
    auto finish = [&](const float* ra, float* ta, const float* rb, float* tb) {
        __syncthreads();
        const int row = tid >> 2, q = tid & 3;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const float* red = w ? rb : ra;
            if (!red) continue;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 v = *(const float4*)&red[row * ROW_RS + 32 * q + 4 * k];
                sum += (v.x + v.y) + (v.z + v.w);
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            if (q == 0) (w ? tb : ta)[row] = sum;
        }
        __syncthreads();
    };

    if constexpr (REPI == REPI_RES_LN) {
        float bj[TN], gj[TN], btj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bj[j] = p.bias ? p.bias[ncol[j]] : 0.f;
            gj[j] = p.gamma[ncol[j]];
            btj[j] = p.beta[ncol[j]];
        }
        const float* __restrict__ resp = p.res;
#pragma unroll
        for (int i = 0; i < TM; ++i) {           // v = acc + bias + residual: 48 loads in flight, then the adds
            if (resp) {
                float t[16][TN];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + lrow(i, r);
                    const int mm = m < p.M ? m : p.M - 1;
                    const long rrow = !REMAP ? mm : (p.res_mod ? (mm % p.res_mod) + p.res_off : out_row(p, mm));
#pragma unroll
                    for (int j = 0; j < TN; ++j) t[r][j] = resp[rrow * p.ldres + ncol[j]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][r] += bj[j] + t[r][j];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][r] += bj[j];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) sum += acc[i][j][r];
                put(red1, i, r, sum);
            }
        finish(red1, tot1, nullptr, nullptr);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {       // two-pass variance
                const float mu = tot1[lrow(i, r)] * invN;
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float d = acc[i][j][r] - mu;
                    sum += d * d;
                }
                put(red1, i, r, sum);
            }
        finish(red1, tot2, nullptr, nullptr);
        int m0c = m0;
        asm volatile("" : "+s"(m0c));            // fresh row arithmetic for the stores (no 32 row offsets kept alive from the loads)
        float* __restrict__ xo = (float*)p.out0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {    // stores only, two rows at a time (one packed conversion per pair); padded rows replicate row M-1
                int orow[2];
                float mu[2], rs[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int mraw = m0c + lrow(i, r + e);
                    const int m = mraw < p.M ? mraw : p.M - 1;
                    orow[e] = REMAP ? out_row(p, m) : m;
                    mu[e] = tot1[lrow(i, r + e)] * invN;
                    rs[e] = rsqrtf(tot2[lrow(i, r + e)] * invN + p.eps);
                    if (wn == 0 && (lane & 31) == 0 && p.mean) {
                        p.mean[orow[e]] = mu[e];
                        p.rstd[orow[e]] = rs[e];
                    }
                }
                if (xo) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int j = 0; j < TN; ++j) xo[(long)orow[e] * p.ldo0 + ncol[j]] = acc[i][j][r + e];
                }
                if (p.y_f32) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            ((float*)p.out1)[(long)orow[e] * p.ldo1 + ncol[j]] = (acc[i][j][r + e] - mu[e]) * rs[e] * gj[j] + btj[j];
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        store_elem_pair<T>((T*)p.out1 + (long)orow[0] * p.ldo1, (T*)p.out1 + (long)orow[1] * p.ldo1, ncol[j],
                                           (acc[i][j][r] - mu[0]) * rs[0] * gj[j] + btj[j], (acc[i][j][r + 1] - mu[1]) * rs[1] * gj[j] + btj[j]);
                }
            }
    } else {  // REPI_LNBWD_RES: acc = dL/dy (y = LN output); aux = saved LN input x (f32)
        float gj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) gj[j] = p.gamma[ncol[j]];
        const float* __restrict__ auxp = (const float*)p.aux;
        const float* __restrict__ resp = p.res;
        if (tid < BM) {                          // the saved statistics of the workgroup's rows, once
            const int mm = m0 + tid < p.M ? m0 + tid : p.M - 1;
            smu[tid] = p.mean[mm];
            srs[tid] = p.rstd[mm];
        }
        __syncthreads();
        float cs_g[TN], cs_b[TN], cs_x[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) cs_g[j] = cs_b[j] = cs_x[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float xv[16][TN];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + lrow(i, r);
                const int mm = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int j = 0; j < TN; ++j) xv[r][j] = auxp[(long)mm * p.ldaux + ncol[j]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = m0 + lrow(i, r) < p.M;
                const float mu = smu[lrow(i, r)], rs = srs[lrow(i, r)];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float h = (xv[r][j] - mu) * rs;
                    const float dy = acc[i][j][r];      // padded rows replicate row M-1; only the column sums mask them
                    const float g = dy * gj[j];
                    s1 += g;
                    s2 += g * h;
                    cs_g[j] += ok ? dy * h : 0.f;
                    cs_b[j] += ok ? dy : 0.f;
                }
                put(red1, i, r, s1);
                put(red2, i, r, s2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // pin the column sums here: left alone, the compiler sinks their whole accumulation chain below the branches of phase 2 and carries
        // the 96 h values (in scratch) and the 96 dy values to get there
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(cs_g[j]), "+v"(cs_b[j]));
        finish(red1, tot1, red2, tot2);
        // the second read of x: launder the row base and the pointer, or the compiler proves the loads redundant and keeps the 96
        // values of the first read alive across the reduction - in scratch
        int m0b = m0;
        const float* x2 = auxp;
        asm volatile("" : "+s"(m0b), "+s"(x2));
        float gj2[TN];                            // likewise dy * gamma: one multiply again instead of 96 values carried over
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            gj2[j] = gj[j];
            asm volatile("" : "+v"(gj2[j]));
        }
        const T* __restrict__ rest = (const T*)p.res_t;      // residual gradient in the operand type (hi + lo = the f32 value to 2^-17)
        auto phase2 = [&](auto has_res) {
            constexpr int RES = decltype(has_res)::value;     // 0 none, 1 f32, 2 operand type
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float xv[8][TN], rv[8][RES ? TN : 1];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int m = m0b + lrow(i, hf * 8 + e);
                        const int mm = m < p.M ? m : p.M - 1;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            xv[e][j] = x2[(long)mm * p.ldaux + ncol[j]];
                            if constexpr (RES == 1) rv[e][j] = resp[(long)mm * p.ldres + ncol[j]];
                            if constexpr (RES == 2) rv[e][j] = load_elem<T>(rest + (long)mm * p.ldres_t, ncol[j]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = hf * 8 + e;
                        const bool ok = m0b + lrow(i, r) < p.M;
                        const float mu = smu[lrow(i, r)], rs = srs[lrow(i, r)];
                        const float c1 = tot1[lrow(i, r)] * invN, c2 = tot2[lrow(i, r)] * invN;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float h = (xv[e][j] - mu) * rs;
                            float dx = rs * (acc[i][j][r] * gj2[j] - c1 - h * c2);
                            if constexpr (RES) dx += rv[e][j];
                            acc[i][j][r] = dx;
                            cs_x[j] += ok ? dx : 0.f;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        if (resp)
            phase2(std::integral_constant<int, 1>());
        else if (rest)
            phase2(std::integral_constant<int, 2>());
        else
            phase2(std::integral_constant<int, 0>());
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(cs_x[j]));
        int m0c = m0;
        asm volatile("" : "+s"(m0c));
        float* __restrict__ dxo = (float*)p.out0;
        T* __restrict__ dxt = (T*)p.out1;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {    // stores only, two rows at a time (one packed conversion per pair)
                int m[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int mraw = m0c + lrow(i, r + e);
                    m[e] = mraw < p.M ? mraw : p.M - 1;
                    if (dxo) {       // the f32 copy is optional: inside the encoder only the embedding stage reads it (dropping it: -14 % per launch)
#pragma unroll
                        for (int j = 0; j < TN; ++j) dxo[(long)m[e] * p.ldo0 + ncol[j]] = acc[i][j][r + e];
                    }
                }
                if (dxt) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        store_elem_pair<T>(dxt + (long)m[0] * p.ldo1, dxt + (long)m[1] * p.ldo1, ncol[j], acc[i][j][r], acc[i][j][r + 1]);
                }
            }
        // column sums: red1 is free (the totals live behind red2); [3][BN] scratch
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float a = cs_g[j], b = cs_b[j], c = cs_x[j];
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 32, 64);
            c += __shfl_xor(c, 32, 64);
            if (lane < 32) {
                if (p.cpart) {
                    p.cpart[(long)blockIdx.x * 3 * BN + 0 * BN + ncol[j]] = a;      // one wave row: the per-workgroup partial IS the wave's
                    p.cpart[(long)blockIdx.x * 3 * BN + 1 * BN + ncol[j]] = b;
                    p.cpart[(long)blockIdx.x * 3 * BN + 2 * BN + ncol[j]] = c;
                } else {
                    if (p.cs0) atomicAdd(p.cs0 + ncol[j], a);
                    if (p.cs1) atomicAdd(p.cs1 + ncol[j], b);
                    if (p.cs2) atomicAdd(p.cs2 + ncol[j], c);
                }
            }
        }
    }
}

// WM == 1 (4 waves, 64 x 96 per wave, 16-bit types with a 64-byte K tile = 56 KB of LDS): built for TWO workgroups per CU - the register
// budget is forced to 256 and the LayerNorm-backward epilogue recomputes the normalised input instead of keeping it (RECOMP).
template <typename T, int REPI, int WM, int BKB, int BM>
__global__ __launch_bounds__(WM * 256, (WM == 1 && sizeof(T) == 2) ? 2 : 1) void gemm_nt_row_kernel(GemmP p) {
    constexpr int BN = ROW_BN, WN = 4;
    typedef NtLoop<T, BM, BN, BKB, WM, WN> Loop;
    constexpr int TM = Loop::TM, TN = Loop::TN;  // (2 | 1) x 3
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // A workgroup owns rows [m0, m0 + rows_per_wg) of its BM-row MFMA tile; the rest of the tile is padding (clamped loads,
    // duplicate stores).  rows_per_wg is chosen so that the grid is just under a multiple of the CU count: at M = 25,216 full
    // 128-row tiles make 197 workgroups for 256 CUs (77 %), 99 rows each make 255.
    const int rpw = p.rows_per_wg > 0 ? p.rows_per_wg : BM;
    const int m0 = blockIdx.x * rpw;
    p.M = p.M < m0 + rpw ? p.M : m0 + rpw;   // everything below treats rows >= p.M as padding
    f32x16 acc[TM][TN];
    // (NtLoopDeep, two K tiles in flight, measured on the two-workgroups-per-CU variant and not used: 2 - 4 % SLOWER (50.0 -> 52.5,
    // 148.6 -> 154.2, 132.4 -> 136.5, 173.4 -> 176.3 us), and with the LayerNorm-backward epilogue at the register limit the
    // allocator spills around the in-flight sets - wrong results.  The row tiles are LDS-read / MFMA paced, not latency paced.  An LDS-DMA
    // ring for the 128-row bf16 variant (round 2, opt-in until round 5): 71 vs 75 us on operands in the Infinity Cache, 84 vs 76 us inside the step.)
    Loop::run(p, m0, 0, lds, acc);  // ends with a barrier: the staging LDS is free from here on

    constexpr bool LEAN = WM == 1 && sizeof(T) == 2 && BM == 64;      // the two-workgroups-per-CU variant: 256 registers
    if constexpr (LEAN) {
        if (REPI == REPI_RES_LN && (p.orow_in || p.res_mod)) row_epilogue_lean<T, REPI, BM, TM, TN, true>(p, m0, acc, lds);
        else row_epilogue_lean<T, REPI, BM, TM, TN, false>(p, m0, acc, lds);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    float* red = (float*)lds;
    float* tot = red + BM * ROW_RS;
    const float invN = 1.0f / (float)BN;
    int ncol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) ncol[j] = (wn * TN + j) * 32 + (lane & 31);

    if (REPI == REPI_RES_LN) {
        float bj[TN], gj[TN], btj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bj[j] = p.bias ? p.bias[ncol[j]] : 0.f;
            gj[j] = p.gamma[ncol[j]];
            btj[j] = p.beta[ncol[j]];
        }
        // pass A: v = acc + bias + residual.  LOADS ONLY (restrict-qualified, no store in this loop) so that the 48 residual
        // loads of a lane are all in flight together; the stores come in pass C.
        const float* __restrict__ resp = p.res;
        float part[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + acc_row(r, lane);
                const int mm = m < p.M ? m : p.M - 1;
                const long rrow = p.res_mod ? (mm % p.res_mod) + p.res_off : out_row(p, mm);
                if (resp) {       // branch hoisted out of the j loop: hipcc serialises loads that sit in their own basic block
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][r] += bj[j] + resp[rrow * p.ldres + ncol[j]];
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][r] += bj[j];
                }
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) s += acc[i][j][r];
                part[i][r] = s;
            }
        row_reduce<TM, BM, WM * 256>(part, red, tot, lane, wm, wn, tid);
        float mu[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mu[i][r] = part[i][r] * invN;
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float d = acc[i][j][r] - mu[i][r];
                    s += d * d;
                }
                part[i][r] = s;
            }
        row_reduce<TM, BM, WM * 256>(part, red, tot, lane, wm, wn, tid);
        // pass C: stores only (x_out f32, y, statistics); padded rows replicate row M-1 -> identical duplicate stores
        float* __restrict__ xo = (float*)p.out0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mraw = m0 + (wm * TM + i) * 32 + acc_row(r, lane);
                const int m = mraw < p.M ? mraw : p.M - 1;
                const int orow = out_row(p, m);
                const float rs = rsqrtf(part[i][r] * invN + p.eps);
                if (wn == 0 && (lane & 31) == 0 && p.mean) {
                    p.mean[orow] = mu[i][r];
                    p.rstd[orow] = rs;
                }
                if (xo) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) xo[(long)orow * p.ldo0 + ncol[j]] = acc[i][j][r];
                }
                if (p.y_f32) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        ((float*)p.out1)[(long)orow * p.ldo1 + ncol[j]] = (acc[i][j][r] - mu[i][r]) * rs * gj[j] + btj[j];
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        store_elem<T>((T*)p.out1 + (long)orow * p.ldo1, ncol[j], (acc[i][j][r] - mu[i][r]) * rs * gj[j] + btj[j]);
                }
            }
    } else {  // REPI_LNBWD_RES: acc = dL/dy (y = LN output); aux = saved LN input x (f32)
        float gj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) gj[j] = p.gamma[ncol[j]];
        const float* __restrict__ auxp = (const float*)p.aux;
        const float* __restrict__ resp = p.res;
        float cs_g[TN], cs_b[TN], cs_x[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) cs_g[j] = cs_b[j] = cs_x[j] = 0.f;
        f32x16 xh[TM][TN];
        float p1[TM][16], p2[TM][16], rsv[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + acc_row(r, lane);
                const bool ok = m < p.M;
                const int mm = ok ? m : p.M - 1;
                const float mu = p.mean[mm], rs = p.rstd[mm];
                rsv[i][r] = rs;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float x = auxp[(long)mm * p.ldaux + ncol[j]];
                    const float h = (x - mu) * rs;
                    const float dy = acc[i][j][r];      // padded rows replicate row M-1; only the column sums mask them
                    xh[i][j][r] = h;
                    const float g = dy * gj[j];
                    s1 += g;
                    s2 += g * h;
                    cs_g[j] += ok ? dy * h : 0.f;
                    cs_b[j] += ok ? dy : 0.f;
                }
                p1[i][r] = s1;
                p2[i][r] = s2;
            }
        row_reduce<TM, BM, WM * 256>(p1, red, tot, lane, wm, wn, tid);
        row_reduce<TM, BM, WM * 256>(p2, red, tot, lane, wm, wn, tid);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mraw = m0 + (wm * TM + i) * 32 + acc_row(r, lane);
                const bool ok = mraw < p.M;
                const int m = ok ? mraw : p.M - 1;
                const float c1 = p1[i][r] * invN, c2 = p2[i][r] * invN, rs = rsv[i][r];
                if (resp) {                      // loads only: dx (+ residual gradient) back into acc
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j][r] = rs * (acc[i][j][r] * gj[j] - c1 - xh[i][j][r] * c2) + resp[(long)m * p.ldres + ncol[j]];
                } else if (p.res_t) {            // ... residual gradient kept in the operand type
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j][r] = rs * (acc[i][j][r] * gj[j] - c1 - xh[i][j][r] * c2) + load_elem<T>((const T*)p.res_t + (long)m * p.ldres_t, ncol[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j][r] = rs * (acc[i][j][r] * gj[j] - c1 - xh[i][j][r] * c2);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) cs_x[j] += ok ? acc[i][j][r] : 0.f;
            }
    
        float* __restrict__ dxo = (float*)p.out0;
        T* __restrict__ dxt = (T*)p.out1;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {       // stores only
                const int mraw = m0 + (wm * TM + i) * 32 + acc_row(r, lane);
                const int m = mraw < p.M ? mraw : p.M - 1;
                if (dxo) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) dxo[(long)m * p.ldo0 + ncol[j]] = acc[i][j][r];
                }
                if (dxt) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) store_elem<T>(dxt + (long)m * p.ldo1, ncol[j], acc[i][j][r]);
                }
            }
        if (p.cpart) __syncthreads();                  // red / tot are free again; reuse as [WM][3][BN] scratch
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float a = cs_g[j], b = cs_b[j], c = cs_x[j];
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 32, 64);
            c += __shfl_xor(c, 32, 64);
            if (lane < 32) {
                if (p.cpart) {
                    red[(wm * 3 + 0) * BN + ncol[j]] = a;
                    red[(wm * 3 + 1) * BN + ncol[j]] = b;
                    red[(wm * 3 + 2) * BN + ncol[j]] = c;
                } else {
                    if (p.cs0) atomicAdd(p.cs0 + ncol[j], a);
                    if (p.cs1) atomicAdd(p.cs1 + ncol[j], b);
                    if (p.cs2) atomicAdd(p.cs2 + ncol[j], c);
                }
            }
        }
        if (p.cpart) {                                 // per-workgroup partials [block][3][384], plain coalesced stores
            __syncthreads();
            for (int q = tid; q < 3 * BN; q += WM * 256) {
                float s = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WM; ++w2) s += red[w2 * 3 * BN + q];
                p.cpart[(long)blockIdx.x * 3 * BN + q] = s;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- wgrad (TN)
template <typename T, bool REMAP>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmP p) {
    // out0[n][k] (f32, atomicAdd) += sum_{m in split} A[m][n] * W[m][k];  p.N = n extent, p.K = k extent, p.M = reduction
    constexpr int BN = 128, BK2 = 128, NT = 256;
    constexpr int KR = 128 / (int)sizeof(T);  // 64 bf16 / 32 f32 reduction rows per stage (16 MFMAs per wave per barrier)
    typedef STile<T, BN, KR> TA;
    typedef STile<T, BK2, KR> TB;
    constexpr int STAGE = TA::BYTES + TB::BYTES;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    apply_batch<T>(p, 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // split tensors (sbf16): the tile grid runs over the STORAGE columns of dY [m][2N] and X [m][2K].  A wave's 64 x 64 storage
    // sub-tile is [hi x 32 | lo x 32] of 32 logical n against the same of 32 logical k, so its four accumulators are the hi*hi,
    // hi*lo, lo*hi (and lo*lo, skipped) blocks of ONE 32 x 32 logical tile: they are summed in the epilogue.
    constexpr bool SPLIT = is_split<T>::value;
    constexpr int EP = elems_per<T>::value;
    const int ntk = p.K * EP / BK2;
    // XCD-aware placement (speed only): blocks are dealt round-robin over the 8 XCDs, each with a private L2.  All tiles of
    // one m-split read the same dY / X rows, so a split's tiles are kept on ONE XCD (split = xcd + 8 * ...): the operands are
    // then fetched into that L2 once and re-read from it by the other tiles instead of 3-12 times from HBM / Infinity Cache.
    int tile = blockIdx.x, split = blockIdx.y;
    if (gridDim.y == 1 && p.splits > 1) {          // 1-D launch: each XCD gets a contiguous run of the split-major block order
        const int tiles = ntk * (p.N * EP / BN);
        const int lin = xcd_remap(blockIdx.x, gridDim.x);
        split = lin / tiles;
        tile = lin % tiles;
    }
    const int n0 = (tile / ntk) * BN, k0 = (tile % ntk) * BK2;
    int chunk = (p.M + p.splits - 1) / p.splits;
    chunk = (chunk + KR - 1) / KR * KR;
    const int mbeg = split * chunk;
    const int mend = min(p.M, mbeg + chunk);
    if (mbeg >= mend) return;
    const T* A = (const T*)p.A;
    const T* X = (const T*)p.W;
    SStage<T, BN, KR, NT> sa;
    SStage<T, BK2, KR, NT> sb;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // optional column sums of A (= bias gradient): one extra MFMA per A fragment against an all-ones B fragment,
    // done by the k-tile-0 / wn-0 waves only (every output column of that product equals sum_m A[m][n])
    const bool do_cs = p.cs0 != nullptr && k0 == 0 && wn == 0;
    f32x16 bacc[2];
    typename MmaTraits<T>::frag_t ones;
    if constexpr (sizeof(T) == 2) {
        typedef typename Vec4<T>::elem E16;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (E16)1.0f;
    } else {
        ones = 1.0f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[i][r] = 0.f;
    const int nst = (mend - mbeg + KR - 1) / KR;
    const int nfull = (mend - mbeg) / KR;          // stages that lie completely inside [mbeg, mend): loaded without range checks
    auto load_stage = [&](int stg) {
        if (stg < nfull) {
            sa.template load<REMAP, false>(A, p.lda, mbeg + stg * KR, mend, n0, tid, p.orow_in, p.orow_out, p.orow_off);
            sb.template load<false, false>(X, p.ldw, mbeg + stg * KR, mend, k0, tid);
        } else {
            sa.template load<REMAP, true>(A, p.lda, mbeg + stg * KR, mend, n0, tid, p.orow_in, p.orow_out, p.orow_off);
            sb.template load<false, true>(X, p.ldw, mbeg + stg * KR, mend, k0, tid);
        }
    };
    // the bias-column-sum MFMAs are selected ONCE per kernel (CS template flag), not per k-step: a branch inside the hot loop
    // splits it into basic blocks and serialises the ds_read -> MFMA software pipeline
    auto main_loop = [&](auto cs_tag) {
        constexpr bool CS = decltype(cs_tag)::value;
        load_stage(0);
        sa.store(lds, tid);
        sb.store(lds + TA::BYTES, tid);
        __syncthreads();
        int cur = 0;
        for (int st = 0; st < nst; ++st) {
            const char* ta = lds + cur * STAGE;
            const char* tb = ta + TA::BYTES;
            if (st + 1 < nst) load_stage(st + 1);
            // hand double-buffered fragments (see NtLoop): reads of sub-step s+1 go out before the MFMAs of sub-step s
            typename MmaTraits<T>::frag_t a[2][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[0][i] = TA::frag(ta, (wm * 2 + i) * 32, 0, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[0][j] = TB::frag(tb, (wn * 2 + j) * 32, 0, lane);
#pragma unroll
            for (int s = 0; s < TA::KSTEPS; ++s) {
                if (s + 1 < TA::KSTEPS) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[(s + 1) & 1][i] = TA::frag(ta, (wm * 2 + i) * 32, s + 1, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[(s + 1) & 1][j] = TB::frag(tb, (wn * 2 + j) * 32, s + 1, lane);
                }
                if (MFVIT_TN_PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (!(SPLIT && i == 1 && j == 1)) acc[i][j] = MmaTraits<T>::mma(a[s & 1][i], b[s & 1][j], acc[i][j]);
                if constexpr (CS) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) bacc[i] = MmaTraits<T>::mma(a[s & 1][i], ones, bacc[i]);
                }
                if (MFVIT_TN_PIPE) __builtin_amdgcn_sched_barrier(0);
            }
            if (st + 1 < nst) {
                char* na = lds + (cur ^ 1) * STAGE;
                sa.store(na, tid);
                sb.store(na + TA::BYTES, tid);
            }
            __syncthreads();
            cur ^= 1;
        }
    };
    if (do_cs) main_loop(std::true_type{});
    else main_loop(std::false_type{});
    if constexpr (SPLIT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[0][0][r] += acc[0][1][r] + acc[1][0][r];
            bacc[0][r] += bacc[1][r];
        }
    }
    constexpr int NI = SPLIT ? 1 : 2;                  // accumulator tiles left per wave after the merge
    const int nw = SPLIT ? n0 / 2 + wm * 32 : n0 + wm * 64, kw = SPLIT ? k0 / 2 + wn * 32 : k0 + wn * 64;   // LOGICAL tile origin of this wave
    if (p.cpart) {
        // split partials as PLAIN stores into scratch [split][N][K] (summed into out0 by tn_reduce_kernel): float atomics run at
        // ~1.3 TB/s chip-wide and one 256-B wave-instruction per ~50 ns per CU, i.e. ~13 us for the 256 of a 128x128 tile
        float* part = p.cpart + (long)split * p.N * p.K;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = nw + i * 32 + acc_row(r, lane);
                    const int k = kw + j * 32 + (lane & 31);
                    part[(long)n * p.K + k] = acc[i][j][r];
                }
    } else {
        float* out = (float*)p.out0;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = nw + i * 32 + acc_row(r, lane);
                    const int k = kw + j * 32 + (lane & 31);
                    atomicAdd(out + (long)n * p.ldo0 + k, acc[i][j][r]);
                }
    }
    if (do_cs && (lane & 31) == 0) {
        // (one writer per split and row: into the split's row of the bias partials behind the tile partials when the launch has them - summed in a fixed order
        // by the reduce pass, like gemm_tn2.hip - else a float atomic)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nw + i * 32 + acc_row(r, lane);
                if (p.kpart) p.kpart[(long)split * p.N + n] = bacc[i][r];
                else atomicAdd(p.cs0 + n, bacc[i][r]);
            }
    }
}

// out[n][k] += sum_s part[s * stride + n * K + k]   (K % 4 == 0; one float4 per thread; the splits in a FIXED order: run-to-run identical sums)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, int splits, long stride, int N, int K, float* __restrict__ out,
                                                        long ldo) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const long total4 = (long)N * K / 4;
    if (q >= total4) return;
    const long e = q * 4;
    float4 s = *(const float4*)(part + e);
    for (int t = 1; t < splits; ++t) {
        const float4 v = *(const float4*)(part + (long)t * stride + e);
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    const int n = (int)(e / K), k = (int)(e % K);
    float4* o = (float4*)(out + (long)n * ldo + k);
    float4 c = *o;
    c.x += s.x, c.y += s.y, c.z += s.z, c.w += s.w;
    *o = c;
}
// the same for every job of a batch in one launch (job of a float4 index: binary search over the prefix counts)
__global__ __launch_bounds__(256) void tn_reduce_batch_kernel(TnPartBatch b) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= b.total4) return;
    int lo = 0, hi = b.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.job[mid].first4 <= q) lo = mid; else hi = mid - 1;
    }
    const TnPartJob& j = b.job[lo];
    const long e = (q - j.first4) * 4;
    float4 s = *(const float4*)(j.part + e);
    for (int t = 1; t < j.splits; ++t) {
        const float4 v = *(const float4*)(j.part + (long)t * j.stride + e);
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    const int n = (int)(e / j.K), k = (int)(e % j.K);
    float4* o = (float4*)(j.out + (long)n * j.ldo + k);
    float4 c = *o;
    c.x += s.x, c.y += s.y, c.z += s.z, c.w += s.w;
    *o = c;
}

// ------------------------------------------------------------------------------------------- launchers
template <typename T, int EPI> static int launch_tile(const GemmP& pin, hipStream_t st) {
    typedef NtLoop<T, 128, 128, 128, 2, 2> Loop;
    GemmP p = pin;
    constexpr int EP = elems_per<T>::value;
    if (p.N % 128 || p.K * EP % Loop::BK || p.M <= 0) return MFVIT_EINVAL;
    if (p.omax && (EPI != EPI_NONE || p.omax_rows < 64 || p.M % p.omax_rows || (p.omax_hd != 32 && p.omax_hd != 64) || p.nb > 1)) return MFVIT_EINVAL;   // (a wave's 64 rows: at most two images)
    // (output tiles leave through common.cuh::store16_stream: system-scope streaming stores)
    const int nwg = (p.N / 128) * ((p.M + 127) / 128);
    constexpr int epi_bytes = 128 * (128 * (int)sizeof(T) * EP + 16);
    constexpr int lds_bytes = Loop::LDS_BYTES > epi_bytes ? Loop::LDS_BYTES : epi_bytes;
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_tile_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if constexpr (sizeof(T) == 2)
            (void)hipFuncSetAttribute((const void*)gemm_nt_tile_kernel<T, EPI, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    }
    ProfScope ps(PROF_GEMM_TILE, 2.0 * p.M * p.N * p.K * (p.nb > 1 ? p.nb : 1), 0, st);
    if constexpr (sizeof(T) == 2) {
        // two K tiles in flight per workgroup (NtLoopDeep): needs an even number of K tiles and 32-bit byte offsets inside both operands
        const bool fits = (unsigned long long)(p.M - 1) * p.lda * 2 + 256 < (1ull << 32) && (unsigned long long)(p.N - 1) * p.ldw * 2 + 256 < (1ull << 32);
        const bool deep_ok = p.K * EP / Loop::BK >= 2 && (p.K * EP / Loop::BK) % 2 == 0 && fits;
        if constexpr (is_split<T>::value) {
            // interleaved loads / LDS stores (round 3; the burst form stays for the plain 16-bit types).
            // Measured inside the step (rocprofv3, serialized streams): fc1 + GELU 138.3 -> 133.1 us, fc2-dgrad 122.7 -> 114.7, qkv 84.0 -> 79.6,
            // proj-dgrad 35.0 -> 32.2; results bit-identical (same MFMA order).
            if (deep_ok) {
                static PerDeviceOnce a12;
                if (a12.first())
                    (void)hipFuncSetAttribute((const void*)gemm_nt_tile_kernel<T, EPI, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                MFVIT_LAUNCH((gemm_nt_tile_kernel<T, EPI, 12>), dim3(nwg, 1, p.nb > 1 ? p.nb : 1), dim3(256), lds_bytes, st, p);
                MFVIT_CHECK_LAUNCH();
                return MFVIT_OK;
            }
        }
        if (deep_ok) {
            MFVIT_LAUNCH((gemm_nt_tile_kernel<T, EPI, 2>), dim3(nwg, 1, p.nb > 1 ? p.nb : 1), dim3(256), lds_bytes, st, p);
            MFVIT_CHECK_LAUNCH();
            return MFVIT_OK;
        }
    }
    MFVIT_LAUNCH((gemm_nt_tile_kernel<T, EPI>), dim3(nwg, 1, p.nb > 1 ? p.nb : 1), dim3(256), lds_bytes, st, p);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
static int row_grid(const GemmP& p, int BM) {
    const int rpw = p.rows_per_wg > 0 ? p.rows_per_wg : BM;
    return (p.M + rpw - 1) / rpw;
}
template <typename T, int REPI, int WM, int BKB, int BM> static int launch_row_v(const GemmP& pin, hipStream_t st) {
    GemmP p = pin;
    p.rows_per_wg = 0;      // full tiles (rows spread evenly over one workgroup per CU measured 6 - 15 % slower in round 2: every workgroup streams the whole W)
    typedef NtLoop<T, BM, ROW_BN, BKB, WM, 4> Loop;
    if (p.N != ROW_BN || p.K * elems_per<T>::value % Loop::BK || p.M <= 0) return MFVIT_EINVAL;
    constexpr int need = BM * ROW_RS * 4 + BM * 4;
    constexpr int need2 = WM * 3 * ROW_BN * 4;
    constexpr int bytes0 = Loop::LDS_BYTES > need ? Loop::LDS_BYTES : need;
    constexpr int bytes1 = bytes0 > need2 ? bytes0 : need2;
    constexpr bool lean = WM == 1 && sizeof(T) == 2 && BM == 64;
    constexpr int bytes = lean && ROW_LEAN_LDS > bytes1 ? ROW_LEAN_LDS : bytes1;
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_row_kernel<T, REPI, WM, BKB, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
    ProfScope ps(REPI == REPI_RES_LN ? PROF_GEMM_ROW_FWD : PROF_GEMM_ROW_BWD, 2.0 * p.M * p.N * p.K, 0, st);
    MFVIT_LAUNCH((gemm_nt_row_kernel<T, REPI, WM, BKB, BM>), dim3(row_grid(p, BM)), dim3(WM * 256), bytes, st, p);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
// gemm_nt_row serves what gemm_rowp.hip does not take: the patch embedding (row remap), f32, K not a multiple of a tile.  Variants (rounds 2 - 3):
// 16-bit types above 16,384 rows: two co-resident 4-wave workgroups per CU on 64-byte K tiles; otherwise 8 waves (2 x 4) on 128-byte K tiles, 128
// rows per workgroup for the forward at K >= 768 and M >= 16,384 (half the W traffic per row).
template <typename T, int REPI> static int launch_row(const GemmP& p, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (p.M > 256 * 64 || p.K * elems_per<T>::value % (128 / (int)sizeof(T))) return launch_row_v<T, REPI, 1, 64, 64>(p, st);
    } else {
        if (p.K % (128 / (int)sizeof(T))) return launch_row_v<T, REPI, 1, 64, 64>(p, st);
    }
    if constexpr (REPI == REPI_RES_LN) {
        if (p.K >= 768 && p.M >= 128 * 128) return launch_row_v<T, REPI, 2, 128, 128>(p, st);
    }
    return launch_row_v<T, REPI, 2, 128, 64>(p, st);
}
template <typename T> static int launch_tn(GemmP p, hipStream_t st) {
    constexpr int EP = elems_per<T>::value;
    if (p.N * EP % 128 || p.K * EP % 128 || p.M <= 0) return MFVIT_EINVAL;
    constexpr int KR = 128 / (int)sizeof(T);
    const int tiles = (p.N * EP / 128) * (p.K * EP / 128);
    if (p.splits <= 0) {
        constexpr int target = 384;
        // Fill the chip evenly: two workgroups fit a CU (2 x 80 KB of LDS), so aim just BELOW a multiple of 256 blocks - 288
        // blocks on 256 CUs leave 224 CUs idle while 32 run two (the makespan is the slowest CU's).
        int s = (tiles >= 16 ? target : (target * 2) / 3) / tiles;
        const int maxs = (p.M + 4 * KR - 1) / (4 * KR);  // at least 4 stages per split
        p.splits = s < 1 ? 1 : (s > maxs ? maxs : s);
    }
    {   // no empty split (the kernel derives the same chunk from p.splits): with the plain-store partial path every split must write its tile -
        // an empty one used to return early and the reduce pass summed whatever the scratch held (MFVIT_TN_PART=1 at M where 28 splits of
        // KR-rounded chunks overshoot M: caught by the switch-matrix run of round 3)
        int chunk = (p.M + p.splits - 1) / p.splits;
        chunk = (chunk + KR - 1) / KR * KR;
        p.splits = (p.M + chunk - 1) / chunk;
    }
    const bool xcd1d = p.splits > 1 && p.nb <= 1;
    constexpr int bytes = 2 * (STile<T, 128, KR>::BYTES * 2);
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
    // p.cpart (optional, >= splits * N * K floats): split partials go there as plain stores and are summed by a second kernel
    if (p.nb > 1 || p.splits < 2 || p.ldo0 % 4 || (long)tiles * p.splits > 384) p.cpart = nullptr;   // scratch holds 384 tiles
    p.kpart = p.cpart && p.cs0 && (long)tiles * p.splits + 2 <= 384 && p.N % 4 == 0 && (size_t)p.cs0 % 16 == 0 ? p.cpart + (long)p.splits * p.N * p.K : nullptr;
    ProfScope ps(PROF_GEMM_TN, 2.0 * p.M * p.N * p.K * (p.nb > 1 ? p.nb : 1), 0, st);
    const dim3 grid = xcd1d ? dim3(tiles * p.splits, 1, 1) : dim3(tiles, p.splits, p.nb > 1 ? p.nb : 1);
    if (p.orow_in)
        MFVIT_LAUNCH((gemm_tn_kernel<T, true>), grid, dim3(256), bytes, st, p);
    else
        MFVIT_LAUNCH((gemm_tn_kernel<T, false>), grid, dim3(256), bytes, st, p);
    MFVIT_CHECK_LAUNCH();
    if (p.cpart) {
        const int rc = tn_partial_reduce(p.cpart, p.splits, (long)p.N * p.K, p.N, p.K, (float*)p.out0, p.ldo0, st);
        if (rc != MFVIT_OK || !p.kpart) return rc;
        return tn_partial_reduce(p.kpart, p.splits, p.N, 1, p.N, p.cs0, p.N, st);
    }
    return MFVIT_OK;
}

// out[n][k] += sum over `splits` partial matrices [N][K] (the weight-gradient kernels' plain-store path)
static thread_local TnPartBatch* g_tnpart_batch = nullptr;
TnPartBatch* tnpart_batch_begin(TnPartBatch* b) {
    TnPartBatch* prev = g_tnpart_batch;
    if (b) { b->n = 0; b->total4 = 0; }
    g_tnpart_batch = b;
    return prev;
}
int tnpart_batch_flush(hipStream_t st) {
    TnPartBatch* b = g_tnpart_batch;
    if (!b || b->n == 0) return MFVIT_OK;
    MFVIT_LAUNCH(tn_reduce_batch_kernel, dim3((unsigned)((b->total4 + 255) / 256)), dim3(256), 0, st, *b);
    b->n = 0;
    b->total4 = 0;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int tn_partial_reduce(const float* part, int splits, long stride, int N, int K, float* out, long ldo, hipStream_t st) {
    if (TnPartBatch* b = g_tnpart_batch) {                    // deferred (every job has its OWN partial buffer until the flush)
        if (b->n == TnPartBatch::MAXJ) {
            const int rc = tnpart_batch_flush(st);
            if (rc != MFVIT_OK) return rc;
        }
        TnPartJob& j = b->job[b->n++];
        j.part = part; j.out = out; j.stride = stride; j.ldo = ldo; j.splits = splits; j.N = N; j.K = K; j.first4 = b->total4;
        b->total4 += (long)N * K / 4;
        return MFVIT_OK;
    }
    const long total4 = (long)N * K / 4;
    MFVIT_LAUNCH(tn_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, part, splits, stride, N, K, out, ldo);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// dst_q[c] += sum_g part[g][q * ncols + c]  for q < nq.  Grid: column groups of 64; ONE block adds all G partial rows of its columns in a fixed order (sixteen
// interleaved row sequences, combined 0 + 1 + ... + 15) and adds the total to the destination with one atomic per column: the sum is the same bits on every run
// (round 6; before, chunks of 32 partial rows went to the destination by float atomics in arrival order - G / 32 adds per address, 8 at the bench shape).  The
// destination itself is zero or holds an earlier launch's total: one or two commutative adds.
constexpr int CPR_SUB = 16;      // row sequences per block (1,024 threads: G = 396 partial rows of the x act' epilogue are 25 loads per thread)
__device__ __forceinline__ float colpart_rows(const float* __restrict__ part, int G, long pitch, int c, int sub, float (*sm)[64]) {
    float s = 0.f;
#pragma unroll 4
    for (int g2 = sub; g2 < G; g2 += CPR_SUB) s += part[(long)g2 * pitch + c];
    sm[sub][threadIdx.x & 63] = s;
    __syncthreads();
    s = 0.f;
    if (sub == 0) {
#pragma unroll
        for (int i = 0; i < CPR_SUB; ++i) s += sm[i][threadIdx.x];
    }
    return s;
}
__global__ __launch_bounds__(64 * CPR_SUB) void colpart_reduce_kernel(const float* __restrict__ part, int G, int ncols, int nq, float* d0, float* d1,
                                                                     float* d2) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    __shared__ float sm[CPR_SUB][64];
    const bool ok = c < nq * ncols;
    float s = colpart_rows(part, ok ? G : 0, (long)nq * ncols, c, sub, sm);
    if (sub == 0 && ok) {
        float* d = c / ncols == 0 ? d0 : (c / ncols == 1 ? d1 : d2);
        if (d) atomicAdd(d + c % ncols, s);
    }
}
#define MFVIT_TRY_RC(expr) do { int rc__ = (expr); if (rc__ != MFVIT_OK) return rc__; } while (0)
// The same for a BATCH of partial buffers in one launch (blockIdx.z = job): the encoder backward gives every row-kernel launch of a call its
// own partial buffer and reduces them all at the end of the call - one launch instead of 25 (round 3: 50 launches of 4.4 us + their
// boundaries per step).
__global__ __launch_bounds__(64 * CPR_SUB) void colpart_reduce_batch_kernel(ColpartBatch b) {
    const ColpartJob& j = b.job[blockIdx.z];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    __shared__ float sm[CPR_SUB][64];
    const bool ok = c < j.nq * j.ncols;
    float s = colpart_rows(j.part, ok ? j.G : 0, (long)j.nq * j.ncols, c, sub, sm);
    if (sub == 0 && ok) {
        float* d = j.d[c / j.ncols];
        if (d) atomicAdd(d + c % j.ncols, s);
    }
}
static thread_local ColpartBatch* g_colpart_batch = nullptr;
ColpartBatch* colpart_batch_begin(ColpartBatch* b) {
    ColpartBatch* prev = g_colpart_batch;
    if (b) b->n = 0;
    g_colpart_batch = b;
    return prev;
}
int colpart_batch_flush(hipStream_t st) {
    ColpartBatch* b = g_colpart_batch;
    if (!b || b->n == 0) return MFVIT_OK;
    int cmax = 0;
    for (int i = 0; i < b->n; ++i) cmax = b->job[i].nq * b->job[i].ncols > cmax ? b->job[i].nq * b->job[i].ncols : cmax;
    MFVIT_LAUNCH(colpart_reduce_batch_kernel, dim3((cmax + 63) / 64, 1, b->n), dim3(64 * CPR_SUB), 0, st, *b);
    b->n = 0;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int colpart_reduce(const float* part, int G, int ncols, int nq, float* d0, float* d1, float* d2, hipStream_t st) {
    if (ColpartBatch* b = g_colpart_batch) {                  // deferred: reduced by colpart_batch_flush (every job has its OWN partial buffer)
        if (b->n == ColpartBatch::MAXJ) MFVIT_TRY_RC(colpart_batch_flush(st));
        ColpartJob& j = b->job[b->n++];
        j.part = part; j.G = G; j.ncols = ncols; j.nq = nq; j.d[0] = d0; j.d[1] = d1; j.d[2] = d2;
        return MFVIT_OK;
    }
    MFVIT_LAUNCH(colpart_reduce_kernel, dim3((nq * ncols + 63) / 64), dim3(64 * CPR_SUB), 0, st, part, G, ncols, nq, d0, d1, d2);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// dtype -> element type dispatch (MFVIT_F32 float | MFVIT_BF16 bf16 | MFVIT_BF16X3 sbf16 (split) | MFVIT_F16 f16)
#define MFVIT_BY_DTYPE(dtype, CALL)                       \
    switch (dtype) {                                      \
        case MFVIT_F32: { typedef float TT; return CALL; }    \
        case MFVIT_BF16: { typedef bf16 TT; return CALL; }    \
        case MFVIT_BF16X3: { typedef sbf16 TT; return CALL; } \
        case MFVIT_F16: { typedef f16 TT; return CALL; }      \
        default: return MFVIT_EINVAL;                     \
    }
template <int EPI> static int tile_by_dtype(int dtype, const GemmP& p, hipStream_t st) { MFVIT_BY_DTYPE(dtype, (launch_tile<TT, EPI>(p, st))) }
template <int REPI> static int row_by_dtype(int dtype, const GemmP& p, hipStream_t st) { MFVIT_BY_DTYPE(dtype, (launch_row<TT, REPI>(p, st))) }
static int tn_by_dtype(int dtype, const GemmP& p, hipStream_t st) { MFVIT_BY_DTYPE(dtype, (launch_tn<TT>(p, st))) }

int gemm_nt_tile(int dtype, int epi, const GemmP& p, hipStream_t st) {
    if (gemm_nt_pp_supported(dtype, epi, p)) return gemm_nt_pp(dtype, epi, p, st);   // gemm_pp.hip: 16-bit types at large M (round 6)
    if (epi == EPI_GELU_BWD && p.cpart && p.cs0) {
        const int rc = tile_by_dtype<EPI_GELU_BWD>(dtype, p, st);
        if (rc != MFVIT_OK) return rc;
        return colpart_reduce(p.cpart, (p.M + 127) / 128, p.N, 1, p.cs0, nullptr, nullptr, st);
    }
    if (gemm_nt_small_supported(dtype, epi, p)) return gemm_nt_small(epi, p, st);   // gemm_small.hip: f32, M <= 512
    switch (epi) {
        case EPI_BIAS: return tile_by_dtype<EPI_BIAS>(dtype, p, st);
        case EPI_BIAS_GELU: return tile_by_dtype<EPI_BIAS_GELU>(dtype, p, st);
        case EPI_GELU_BWD: return tile_by_dtype<EPI_GELU_BWD>(dtype, p, st);
        case EPI_NONE: return tile_by_dtype<EPI_NONE>(dtype, p, st);
        case EPI_BIAS_RELU: return tile_by_dtype<EPI_BIAS_RELU>(dtype, p, st);
        case EPI_BIAS_X3F16: return dtype == MFVIT_BF16X3 ? launch_tile<sbf16, EPI_BIAS_X3F16>(p, st) : MFVIT_EINVAL;
    }
    return MFVIT_EINVAL;
}
int gemm_nt_row(int dtype, int repi, const GemmP& p, hipStream_t st) {
    if (gemm_nt_rowp_supported(dtype, repi, p)) return gemm_nt_rowp(dtype, repi, p, st);
    if (repi == REPI_RES_LN) return row_by_dtype<REPI_RES_LN>(dtype, p, st);
    if (repi == REPI_LNBWD_RES) {
        const int rc = row_by_dtype<REPI_LNBWD_RES>(dtype, p, st);
        if (rc != MFVIT_OK || !p.cpart) return rc;
        GemmP q = p;
        q.rows_per_wg = 0;                          // the LN-backward variants all use full 64-row tiles (see launch_row)
        return colpart_reduce(p.cpart, row_grid(q, 64), ROW_BN, 3, p.cs0, p.cs1, p.cs2, st);
    }
    return MFVIT_EINVAL;
}
int gemm_tn(int dtype, const GemmP& p, hipStream_t st) {
    if (gemm_tn_glds_supported(dtype, p)) return gemm_tn_glds(dtype, p, st);   // gemm_tn2.hip: LDS-DMA ring (16-bit types, large M)
    {   // gemm_small.hip: f32, M <= 512 - one workgroup per output tile, no splits: it needs no partial scratch (and is deterministic without one)
        GemmP q = p;
        q.cpart = nullptr;
        if (gemm_tn_small_supported(dtype, q)) return gemm_tn_small(q, st);
    }
    return tn_by_dtype(dtype, p, st);
}

}  // namespace mfvit
