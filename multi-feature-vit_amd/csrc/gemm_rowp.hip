// Row-complete NT GEMM for SPLIT-bf16 tensors, one tall tile per CU ("rowp"):  out[M][384] = epi(A[M][K] W[384][K]^T) with the epilogues of
// gemm_nt_row -  + bias + residual -> LayerNorm (proj / fc2 of timm's Block, crossvit_2vits_..._sum.py:128-135)  and
// LayerNorm backward + residual-gradient add + dgamma / dbeta / dbias column sums (the qkv / fc1 data gradients).
//
// Why (round 3).  gemm_nt_row_kernel runs 394 workgroups of 64 rows, two per CU, on K tiles of HALF a split k group (64 bytes per row, so
// that two double-buffered workgroups fit the LDS).  Measured consequences: (1) every workgroup streams the whole W[384][K] from L2 - 930 MB
// per fc2 launch against 155 MB of activations; (2) a 64-byte K tile takes 32-byte pieces out of every 128-byte line and fetches the line
// again for the next tile: the L2 -> CU traffic in lines is twice the bytes used (gemm_pp.hip measured the same effect on its LDS-DMA ring:
// half the rate); (3) 394 workgroups on 512 slots fill 77 % of the chip.  The kernels are bound by exactly that traffic, not by the matrix
// pipe (17 - 26 % MFMA-busy) and hardly react to the clock (tools/power_probe.py: fc2 + LN 134 us on zeros vs 139 - 158 us on random data).
// Here:
//   * ONE 8-wave workgroup per CU and per launch, one tile of ceil(M / #CUs) rows (99 at M = 25,216: 7 row fragments of 16) x all 384
//     columns: W is streamed once per CU (602 MB instead of 930), every CU works (255 of 256), v_mfma_f32_16x16x32_bf16 so that the row
//     granularity is 16 (the 32x32 shape would pad 99 rows to 128; both shapes sustain the same FLOP rate at two waves per SIMD,
//     profiles/r03_mfma_ceiling.jsonl).
//   * stages of one whole k group (128-byte lines, 128 A rows + 384 W rows = 64 KB) by LDS-DMA into a 2-slot ring; wave w owns output columns
//     48 w .. 48 w + 47 (3 column fragments x 7 row fragments = 21 accumulator tiles of 4 registers).
//   * one barrier per stage, placed BEHIND the fourth row fragment: by then every fragment of the stage is in registers (the slot is free)
//     and stage s + 1 - issued a stage earlier - has landed; the LDS-DMA of stage s + 2 and the first fragment reads of stage s + 1 are
//     interleaved with the MFMAs of the last three row fragments, the remaining reads of stage s with those of the first four: nothing but
//     MFMAs between MFMAs for more than a few instructions.
//   * MFMA operands swapped (W fragment first): a lane owns ONE output row (lane & 15) and 4 consecutive columns per accumulator tile, so
//     the row statistics are in-lane sums + two cross-lane adds + one LDS exchange between the 8 waves.
// Products in the term order of the other split kernels (a_lo w_hi + a_hi w_lo + a_hi w_hi, f32 accumulate).
#include "gemm.cuh"
#include "kernels.h"
#include "prof.h"

#include <stdlib.h>

#include <type_traits>

namespace mfvit {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((may_alias)) stg_b4;

constexpr int RP_N = 384;                       // output columns (embed dim): 8 waves x 48
constexpr int RP_MF = 7;                        // row fragments of 16 per tile
constexpr int RP_TH = 16 * RP_MF;               // 112 rows at most
constexpr int RP_AROWS = 128;                   // A rows of a stage (the last 16 are never read: keeps 64 pieces = 8 per wave)
constexpr int RP_ROWS = RP_AROWS + RP_N;        // 512
constexpr int RP_STAGE = RP_ROWS * 128;         // 64 KB
constexpr int RP_RING = 2 * RP_STAGE;           // 128 KB
constexpr int RP_SCR = 32768;                   // reductions / statistics
constexpr int RP_LDS = RP_RING + RP_SCR;        // 163,840 B
constexpr int RP_L = 8;                         // LDS-DMA instructions per wave and stage
constexpr int RP_YP = 1536 + 16;                // staging pitch of a split output row (768 storage elements + pad)

__device__ __forceinline__ const char* rp_uniform_ptr(const void* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void rp_dma16(unsigned voff, const char* sbase, unsigned m0v) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(m0v)
                 : "memory");
}

template <int REPI>
__global__ __launch_bounds__(512, 1) void gemm_rowp_kernel(GemmP p, int rpt) {
    typedef sbf16 T;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * rpt;
    const int rows = p.M - m0 < rpt ? p.M - m0 : rpt;                 // valid rows of this tile (1 .. RP_TH)
    const int nk = p.K / 32;
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);

    // ---- LDS-DMA: stage row R (A rows 0 .. 127, W rows 128 .. 511) = 8 chunks of 16 B, positions XOR (R >> 1) & 7; wave w copies the
    // 8-row pieces w + 8 i (i = 0, 1: A rows, clamped to the tile's last valid row; i = 2 .. 7: W rows)
    const int lrow = lane >> 3;
    const unsigned coff = (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 16);
    unsigned voff[RP_L];
#pragma unroll
    for (int i = 0; i < RP_L; ++i) {
        const int pc = wave + 8 * i;
        if (i < 2) {
            int r = 8 * pc + lrow;
            r = r < rows ? r : rows - 1;
            voff[i] = (unsigned)(m0 + r) * (unsigned)(p.lda * 2) + coff;
        } else {
            voff[i] = (unsigned)(8 * (pc - 16) + lrow) * (unsigned)(p.ldw * 2) + coff;
        }
    }
    const char* gA = rp_uniform_ptr(p.A);
    const char* gW = rp_uniform_ptr(p.W);
    auto issue_piece2 = [&](int stage, int slot, int i) __attribute__((always_inline)) {     // k group `stage` -> ring slot `slot`
        const char* base = rp_uniform_ptr((i < 2 ? gA : gW) + (long)stage * 128);
        rp_dma16(voff[i], base, __builtin_amdgcn_readfirstlane(lbase + (unsigned)slot * RP_STAGE + (unsigned)(wave + 8 * i) * 1024u));
    };
    auto issue_piece = [&](int stage, int i) __attribute__((always_inline)) { issue_piece2(stage, stage & 1, i); };

    // ---- fragments: lane (r = lane & 15, q = lane >> 4) holds k = 8 q .. 8 q + 7 of row base + r: chunk q (hi) / 4 + q (lo)
    const int fr = lane & 15, fq = lane >> 4, fsw = (fr >> 1) & 7;
    const int f_hi = fr * 128 + 16 * (fq ^ fsw), f_lo = fr * 128 + 16 * ((4 + fq) ^ fsw);
    auto frag_a = [&](int slot, int i, int off) __attribute__((always_inline)) { return *(const bf16x8*)(lds + slot * RP_STAGE + (16 * i) * 128 + off); };
    auto frag_w = [&](int slot, int j, int off) __attribute__((always_inline)) {
        return *(const bf16x8*)(lds + slot * RP_STAGE + (RP_AROWS + 48 * wave + 16 * j) * 128 + off);
    };

    f32x4v acc[RP_MF][3];
#pragma unroll
    for (int i = 0; i < RP_MF; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // prologue: stages 0 and 1 in flight, stage 0 landed, its W fragments and the first two row fragments in registers
#pragma unroll
    for (int i = 0; i < RP_L; ++i) issue_piece(0, i);
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < RP_L; ++i) issue_piece(1, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RP_L) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 wh[2][3], wl[2][3], ah[RP_MF], al[RP_MF];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        wh[0][j] = frag_w(0, j, f_hi);
        wl[0][j] = frag_w(0, j, f_lo);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ah[i] = frag_a(0, i, f_hi);
        al[i] = frag_a(0, i, f_lo);
    }

    // one stage; `side(slot index 0 .. 62)` runs behind every MFMA
    auto stage_body = [&](int s, bf16x8 (&wH)[3], bf16x8 (&wL)[3], bf16x8 (&wHn)[3], bf16x8 (&wLn)[3]) __attribute__((always_inline)) {
        const int slot = s & 1, nslot = slot ^ 1;
        // (past the end of K the refills repeat the last stage and the reads take the other slot: harmless, and no branches in the MFMA stream)
        const int s2 = s + 2 < nk ? s + 2 : nk - 1;
#pragma unroll
        for (int i = 0; i < RP_MF; ++i) {
            if (i == 4) {
                // every fragment of stage s is in registers (requested during row fragments 0 - 3), stage s + 1 has landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int j = t % 3, term = t / 3;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(term == 1 ? wL[j] : wH[j], term == 0 ? al[i] : ah[i], acc[i][j], 0, 0, 0);
                const int idx = 9 * i + t;
                if (idx < 36) {
                    // row fragments 2 .. 6 of this stage: one read every third MFMA (10 reads in 30 slots)
                    if (idx % 3 == 0 && idx / 3 < 10) {
                        const int k = idx / 3, fi = 2 + k / 2;
                        if (k % 2 == 0) ah[fi] = frag_a(slot, fi, f_hi);
                        else al[fi] = frag_a(slot, fi, f_lo);
                    }
                } else {
                    // behind the barrier: this wave's pieces of stage s + 2 into the slot just freed (8, every third slot), then the W
                    // fragments and row fragments 0, 1 of stage s + 1 (10 reads)
                    const int u = idx - 36;                                  // 0 .. 26
                    if (u % 3 == 0 && u / 3 < RP_L) {
                        issue_piece2(s2, slot, u / 3);
                    } else if (u % 3 == 1 && u / 3 < 6) {
                        const int k = u / 3;
                        if (k < 3) wHn[k] = frag_w(nslot, k, f_hi);
                        else wLn[k - 3] = frag_w(nslot, k - 3, f_lo);
                    } else if (u % 3 == 2 && u / 3 < 4) {
                        const int k = u / 3;
                        if (k < 2) ah[k] = frag_a(nslot, k, f_hi);
                        else al[k - 2] = frag_a(nslot, k - 2, f_lo);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int s = 0; s < nk; s += 2) {
        stage_body(s, wh[0], wl[0], wh[1], wl[1]);
        if (s + 1 < nk) stage_body(s + 1, wh[1], wl[1], wh[0], wl[0]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                   // the ring is free: epilogue scratch

    // ------------------------------------------------------------------------------------------------ epilogues
    // acc[i][j][r] = out[m0 + 16 i + fr][48 wave + 16 j + 4 fq + r]; rows past `rows` replicate the tile's last valid row exactly (clamped
    // A rows, clamped residual rows): they are stored to that row's address again - identical duplicates, no branches
    float* red = (float*)(lds + RP_RING);                              // [8 waves][RP_TH]
    float* red2 = red + 8 * RP_TH;
    const int ncol0 = 48 * wave + 4 * fq;                              // + 16 j + r
    auto row_of = [&](int i) __attribute__((always_inline)) {
        const int r = 16 * i + fr;
        return m0 + (r < rows ? r : rows - 1);
    };
    // per-row totals over the 384 columns of `part[i]` (this lane's partial over its 12 values): all 512 threads call it
    auto row_total = [&](float (&part)[RP_MF], float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RP_MF; ++i) {
            float v = part[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (fq == 0) buf[wave * RP_TH + 16 * i + fr] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RP_MF; ++i) {
            float t = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) t += buf[w8 * RP_TH + 16 * i + fr];
            part[i] = t;
        }
    };
    // split output tile rows [16 i0, 16 i1) of this workgroup -> out (storage ld `ldo`), through LDS (ring region) as whole lines
    auto store_split = [&](void* out, long ldo, int i0, int i1) __attribute__((always_inline)) {
        char* ybuf = lds;
#pragma unroll
        for (int i = 0; i < RP_MF; ++i) {
            if (i < i0 || i >= i1) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = ncol0 + 16 * j;                          // 4 consecutive logical columns n .. n + 3 (inside one 32-group)
                bf16 h0, h1, h2, h3, l0, l1, l2, l3;
                cvt_pair<bf16, true>(acc[i][j][0], acc[i][j][1], h0, h1, l0, l1);
                cvt_pair<bf16, true>(acc[i][j][2], acc[i][j][3], h2, h3, l2, l3);
                bf16x4 hv, lv;
                hv[0] = h0; hv[1] = h1; hv[2] = h2; hv[3] = h3;
                lv[0] = l0; lv[1] = l1; lv[2] = l2; lv[3] = l3;
                char* rowp = ybuf + (16 * (i - i0) + fr) * RP_YP + 128 * (n >> 5) + 2 * (n & 31);
                *(stg_b4*)rowp = hv;
                *(stg_b4*)(rowp + 64) = lv;
            }
        }
        __syncthreads();
        const int nrow = 16 * (i1 - i0);
        for (int q = tid; q < nrow * 96; q += 512) {
            const int row = q / 96, ch = q % 96;
            const u32x4 v = *(const u32x4 __attribute__((may_alias))*)(ybuf + row * RP_YP + 16 * ch);
            int r = 16 * i0 + row;
            r = r < rows ? r : rows - 1;
            __builtin_nontemporal_store(v, (u32x4*)((char*)out + (long)(m0 + r) * ldo * 2 + 16 * ch));
        }
        __syncthreads();
    };

    if constexpr (REPI == REPI_RES_LN) {
        const float invN = 1.0f / (float)RP_N;
        float part[RP_MF];
        // v = acc + bias + residual
#pragma unroll
        for (int i = 0; i < RP_MF; ++i) {
            const long m = row_of(i);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = ncol0 + 16 * j;
                f32x4v b = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) b = *(const f32x4v*)(p.bias + n);
                if (p.res) rv = *(const f32x4v*)(p.res + m * p.ldres + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[i][j][r] += b[r] + rv[r];
                    s += acc[i][j][r];
                }
            }
            part[i] = s;
        }
        row_total(part, red);
        float mu[RP_MF];
#pragma unroll
        for (int i = 0; i < RP_MF; ++i) {
            mu[i] = part[i] * invN;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[i][j][r] - mu[i];
                    s += d * d;
                }
            part[i] = s;
        }
        row_total(part, red2);
        float* xo = (float*)p.out0;
#pragma unroll
        for (int i = 0; i < RP_MF; ++i) {
            const long m = row_of(i);
            const float rs = rsqrtf(part[i] * invN + p.eps);
            if (wave == 0 && fq == 0 && p.mean) {
                p.mean[m] = mu[i];
                p.rstd[m] = rs;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = ncol0 + 16 * j;
                if (xo) *(f32x4v*)(xo + m * p.ldo0 + n) = acc[i][j];
                const f32x4v g = *(const f32x4v*)(p.gamma + n), be = *(const f32x4v*)(p.beta + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = (acc[i][j][r] - mu[i]) * rs * g[r] + be[r];
                if (p.y_f32) *(f32x4v*)((float*)p.out1 + m * p.ldo1 + n) = acc[i][j];
            }
        }
        if (!p.y_f32) {
            __syncthreads();                                           // (red / red2 live outside the ring; the ring itself is free)
            store_split(p.out1, p.ldo1, 0, 4);
            store_split(p.out1, p.ldo1, 4, RP_MF);
        }
    }
}

int rp_cus() {
    static const int n = [] {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return n;
}

// rows per tile: the smallest whole number of rounds of one tile per CU that covers M with tiles of at most 112 rows, rows spread evenly
int rp_rows_per_tile(int M) {
    const int cus = rp_cus();
    const int rounds = (M + cus * RP_TH - 1) / (cus * RP_TH);
    return (M + cus * rounds - 1) / (cus * rounds);
}

template <int REPI> int launch_rowp(const GemmP& p, hipStream_t st) {
    const int rpt = rp_rows_per_tile(p.M);
    const int grid = (p.M + rpt - 1) / rpt;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_rowp_kernel<REPI>, hipFuncAttributeMaxDynamicSharedMemorySize, RP_LDS);
        attr = true;
    }
    ProfScope ps(REPI == REPI_RES_LN ? PROF_GEMM_ROW_FWD : PROF_GEMM_ROW_BWD, 2.0 * p.M * p.N * p.K, 0, st);
    MFVIT_LAUNCH((gemm_rowp_kernel<REPI>), dim3(grid), dim3(512), RP_LDS, st, p, rpt);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

int rowp_mode() {   // MFVIT_ROWP: 0 off, 1 (default) on.  Read at every launch (A/B runs in one process)
    const char* e = getenv("MFVIT_ROWP");
    return e ? atoi(e) : 1;
}

}  // namespace

bool gemm_nt_rowp_supported(int dtype, int repi, const GemmP& p) {
    if (dtype != MFVIT_BF16X3 || rowp_mode() == 0) return false;
    if (repi != REPI_RES_LN) return false;
    if (p.N != RP_N || p.K % 32 || p.K < 64 || p.M < 4096 || p.nb > 1) return false;
    if (p.orow_in || p.res_mod || p.rows_per_wg) return false;                  // patch-embedding row remap stays with gemm_nt_row
    if (!p.out1 || !p.gamma || !p.beta) return false;
    if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 32) || (unsigned long long)p.N * p.ldw * 2 >= (1ull << 32)) return false;
    if (p.lda % 8 || p.ldw % 8 || (p.out0 && p.ldo0 % 4) || p.ldo1 % (p.y_f32 ? 4 : 8) || (p.res && p.ldres % 4)) return false;
    if ((size_t)p.A % 16 || (size_t)p.W % 16 || (size_t)p.out1 % 16 || (p.out0 && (size_t)p.out0 % 16) || (p.res && (size_t)p.res % 16) ||
        (p.bias && (size_t)p.bias % 16) || (size_t)p.gamma % 16 || (size_t)p.beta % 16)
        return false;
    return true;
}

int gemm_nt_rowp(int repi, const GemmP& p, hipStream_t st) {
    if (repi == REPI_RES_LN) return launch_rowp<REPI_RES_LN>(p, st);
    return MFVIT_EINVAL;
}

}  // namespace mfvit
