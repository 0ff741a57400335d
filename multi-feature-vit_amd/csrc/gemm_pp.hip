// Ping-pong persistent NT GEMM for SPLIT-bf16 tensors (dtype MFVIT_BF16X3):  out[M][N] = epi(A[M][K] W[N][K]^T), the short-K linears of the
// encoder (qkv, fc1 + GELU, fc2-dgrad * gelu', proj-dgrad).  Replaces gemm_nt_tile for these shapes (reference: timm Block's nn.Linear
// calls, crossvit_2vits_..._sum.py:128-135).
//
// Why a third structure (round 3).  At K = 384 the 128x128 / two-workgroups-per-CU kernel spends per tile ~2.5 us in dispatch + first-load
// latency and ~3 us in its epilogue against 5.4 us of MFMAs, and the two co-resident workgroups only overlap those phases by chance (37 %
// MFMA-busy, profiles/pmc/r02_pmc_bf16x3_qkv.txt).  Here ONE 8-wave workgroup owns a CU for the whole launch and the overlap is by
// construction:
//   * two groups of four waves (one wave of each group per SIMD).  Units (BM x 128 output tiles, BM = 256 or 128) alternate between the
//     groups: while group X runs the MFMAs of unit j (one wave per SIMD, fragments fetched just in time), group Y ("assist") converts and
//     stores the outputs of the unit it finished before - and the other way round for unit j + 1.  The matrix pipe sees one uninterrupted
//     MFMA stream; no epilogue arithmetic, LDS staging or store sits in front of an MFMA.
//   * the K steps of ALL units of the workgroup form one flat stream of stages (one 32-wide k group = [hi x 32 | lo x 32] = one 128-byte line
//     per row; BM + 128 rows = 48 KB or 32 KB) through a 144 / 128 KB LDS ring filled by LDS-DMA (global_load_lds_dwordx4) NSLOT - 1 stages
//     ahead: no per-tile fill bubble.  Every wave issues its eighth of each stage (6 or 4 instructions, spread between MFMA groups in a
//     consume turn): the CU's vector-memory path (64 B/clk) needs ~400 cycles per stage, which no single group of four waves could issue
//     beside its other work.  256-row units move 24 KB per 128x128x32 step instead of 32.  (First version: half-k-group stages of 64 B per
//     row - every 128-byte line was fetched twice, in 32-byte pieces, and the L2 -> LDS rate halved: 11 TB/s, the pipeline alone 51 us.)
//   * one s_barrier per stage for all 8 waves; before it every wave waits for ITS pieces of stage q + 1 with a COUNTED vmcnt (vmcnt retires
//     in order: "at most (NSLOT - 2) stages' worth outstanding"; epilogue stores in between only make the wait stricter).  Every
//     vector-memory LOAD of the kernel is inline asm: the compiler's own waits would count only what it sees and drain the ring.
//   * MFMA operands swapped (W fragment first): a lane owns one output row and 4 consecutive columns per register quad, so outputs leave
//     as packed hi / lo quads through a wave-private 4 KB LDS tile and are stored as whole 128-byte lines.
//   * the bias is the accumulators' INITIAL value (scalar loads at the start of a consume turn, one select per register): the epilogue of
//     the plain linear is conversion + stores only.  (acc = bias + sum instead of sum + bias: same value to one f32 rounding.)
// Without a bias the results are bit-identical to gemm_nt_tile_kernel (same MFMA term order, same conversions): tests/test_ops_gpu.py.
#include "gemm.cuh"
#include "kernels.h"
#include "prof.h"

#include <stdlib.h>

#include <type_traits>

// MFVIT_PP_TRACE (debug builds only, tools/build_trace_lib.sh): MFVIT_PP_ABLATE bits switch pieces off (timing only), and unless
// MFVIT_PP_NOTICKS every wave accumulates s_memtime cycles per phase into 8 counters written to the buffer whose address the environment
// variable MFVIT_PP_TRACE_PTR holds ([workgroup][wave][8] floats): consume turns 0 = first two products + waits, 1 = barrier, 2 = fragment
// reads + last product; assist turns 3 = LDS-DMA issue, 4 = epilogue steps, 5 = counted wait, 6 = barrier; 7 = the wave's whole life.
#if defined(MFVIT_PP_TRACE) && !defined(MFVIT_PP_NOTICKS)
#define PP_TICK(i) do { const long long t__ = __builtin_readcyclecounter(); tacc[i] += (float)(t__ - tlast); tlast = t__; } while (0)
#else
#define PP_TICK(i) do { } while (0)
#endif

#ifndef MFVIT_PP_CONS_DMA
#define MFVIT_PP_CONS_DMA 1       // 1: every wave issues its eighth of each stage; 0: the four assisting waves issue a quarter each
#endif
#ifndef MFVIT_PP_PRIO
#define MFVIT_PP_PRIO 0           // 1: s_setprio 1 during consume turns
#endif

namespace mfvit {

namespace {

typedef bf16x4 __attribute__((may_alias)) stg_b4;
typedef f16x4 __attribute__((may_alias)) stg_h4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int QBN = 128;                    // unit columns
constexpr int QSTG = 4096;                  // epilogue staging per assisting wave: 32 rows x 128 B

template <int TM> struct QCfg {
    static constexpr int BM = 64 * TM;                 // unit rows: 2 waves x TM x 32
    static constexpr int ROWS = BM + QBN;              // rows of a stage: the A rows, then the W rows
    static constexpr int STAGE = ROWS * 128;           // one 32-wide k group per row
    static constexpr int NSLOT = TM == 4 ? 3 : 4;      // 3 x 48 KB, 4 x 32 KB
    static constexpr int RING = NSLOT * STAGE;
    static constexpr int LDS = RING + 4 * QSTG;        // 163,840 B (the whole LDS of a CU) / 147,456 B
    static constexpr int NISSUE = MFVIT_PP_CONS_DMA ? 8 : 4;   // waves that issue the LDS-DMA of a stage
    static constexpr int L = ROWS / 8 / NISSUE;        // LDS-DMA instructions (8 rows = 1 KB each) per issuing wave and stage
    static_assert(ROWS % 64 == 0 && LDS <= 163840, "pieces dealt evenly to the issuing waves; the ring fits");
};

// wait until at most n * L of this wave's vector-memory operations are outstanding (n wave-uniform, 0 <= n <= 3)
template <int L> __device__ __forceinline__ void wait_stages(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * L) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * L) : "memory"); break;
    }
}

// a wave-uniform pointer the compiler can PROVE uniform (an "s" asm operand it believes divergent is silently given VGPRs)
__device__ __forceinline__ const char* uniform_ptr(const void* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// one LDS-DMA: 64 lanes x 16 B from sbase + voff (per lane) to LDS bytes [m0v, m0v + 1024), lane-linear
__device__ __forceinline__ void dma16(unsigned voff, const char* sbase, unsigned m0v) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(m0v)
                 : "memory");
}
__device__ __forceinline__ void aload8(u32x2& dst, unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase));
}

template <int TM, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(GemmP p, int ntm, int ntn) {
    typedef QCfg<TM> C;
    typedef sbf16 T;
    typedef typename act_grad_type<T>::type AX;         // plain fp16: the saved activation derivative of split tensors (gemm.hip)
    constexpr int TN = 2, NSLOT = C::NSLOT, L = C::L, BM = C::BM, STAGE = C::STAGE;
    constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w = wave & 3, wm = w >> 1, wn = w & 1;
    const int G = gridDim.x, ntiles = ntm * ntn;
    // the workgroups of one XCD (blockIdx & 7) get consecutive units of every round: they share activation rows and W in that L2
    const int cslot = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
    if (cslot >= ntiles) return;
    const int nloc = (ntiles - cslot + G - 1) / G;
    const int nk = p.K / 32;                            // stages (k groups) per unit
    const int S = nloc * nk;
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
#ifdef MFVIT_PP_ABL
    constexpr int abl = MFVIT_PP_ABL;   // compile-time ablation bits (timing only, results invalid): 1 no LDS-DMA, 2 no MFMAs, 4 no global stores,
#else                                   // 8 no epilogue, 16 no fragment reads in the MFMA stream
    constexpr int abl = 0;
#endif
#if defined(MFVIT_PP_TRACE) && !defined(MFVIT_PP_NOTICKS)
    float tacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long long tstart = __builtin_readcyclecounter();
    long long tlast = tstart;
#endif

    // ------------------------------------------------------------------------------------------------ issue side (every wave, every stage)
    // Stage image: row R (A rows 0 .. BM - 1, then the 128 W rows) = 128 B = 8 chunks of 16 B (0 - 3: hi k 0-7 .. 24-31, 4 - 7: lo), chunk
    // positions XOR-swizzled by (R >> 1) & 7 (KTile16, 128-byte rows: conflict-free ds_read_b128 fragment reads).  An LDS-DMA writes 1 KB
    // lane-linear = 8 rows; the swizzle goes on the per-lane SOURCE chunk: position lane & 7 of row R receives chunk (lane & 7) ^ ((R >> 1) & 7).
    // Wave `wave` copies the pieces wave + 8 i: (R >> 1) & 7 = (4 (wave & 1) + (lane >> 4)) & 7 for all of them.
    constexpr int NISSUE = C::NISSUE;
    const int pw = NISSUE == 8 ? wave : w;               // this wave copies the pieces pw + NISSUE i (when it issues)
    const int lrow = lane >> 3;
    const unsigned coff = (unsigned)(((lane & 7) ^ ((4 * (pw & 1) + (lane >> 4)) & 7)) * 16);
    unsigned voff[L];
    int it_unit = cslot, it_kt = 0;
    auto set_offsets = [&](int unit) __attribute__((always_inline)) {
        const int m0 = (unit / ntn) * BM, n0 = (unit % ntn) * QBN;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int pc = pw + NISSUE * i;             // 8-row piece of the stage
            if (pc < BM / 8) {                          // pieces 0 .. BM / 8 - 1: A rows
                int gr = m0 + 8 * pc + lrow;
                gr = gr < p.M ? gr : p.M - 1;
                voff[i] = (unsigned)gr * (unsigned)(p.lda * 2) + coff;
            } else {
                voff[i] = (unsigned)(n0 + 8 * (pc - BM / 8) + lrow) * (unsigned)(p.ldw * 2) + coff;
            }
        }
    };
    const char* it_ga = nullptr;
    const char* it_gw = nullptr;
    unsigned it_sb = 0;
    // the stage at the iterator goes to ring slot `slot`: issue_begin once, then issue_piece(i) for i = 0 .. L - 1 (anywhere in the phase)
    auto issue_begin = [&](int slot) __attribute__((always_inline)) {
        const long koff = (long)it_kt * 128;
        it_ga = uniform_ptr((const char*)p.A + koff);
        it_gw = uniform_ptr((const char*)p.W + koff);
        it_sb = __builtin_amdgcn_readfirstlane(lbase + (unsigned)slot * STAGE + (unsigned)pw * 1024u);
    };
    auto issue_piece = [&](int i) __attribute__((always_inline)) {
        if (!(abl & 1)) dma16(voff[i], NISSUE * i < BM / 8 ? it_ga : it_gw, it_sb + (unsigned)i * (NISSUE * 1024u));
    };
    auto issue_end = [&]() __attribute__((always_inline)) {
        if (++it_kt == nk) {
            it_kt = 0;
            // past the workgroup's last unit the stream repeats it: the refills land in slots nobody reads any more, every phase of the
            // kernel issues the same number of pieces and every wait is the same counted one (no branches in the MFMA stream)
            if (it_unit + G < ntiles) {
                it_unit += G;
                set_offsets(it_unit);
            }
        }
    };

    // ------------------------------------------------------------------------------------------------ compute side (consume turns)
    // MFMA operand fragments of a stage row R for k step s (16 of the group's 32 k): lane (r = lane & 31, h = lane >> 5) holds k = 16 s + 8 h
    // .. + 8 of the hi part (chunk 2 s + h) or of the lo part (chunk 4 + 2 s + h); (R >> 1) & 7 = (r >> 1) & 7 for every tile base used here
    const int r32 = lane & 31, hh = lane >> 5, sw = (r32 >> 1) & 7;
    int f_hi[2], f_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f_hi[s] = r32 * 128 + 16 * ((2 * s + hh) ^ sw);
        f_lo[s] = r32 * 128 + 16 * ((4 + 2 * s + hh) ^ sw);
    }
    auto frag_a = [&](int slot, int i, int off) __attribute__((always_inline)) {
        return *(const bf16x8*)(lds + slot * STAGE + (wm * 32 * TM + 32 * i) * 128 + off);
    };
    auto frag_w = [&](int slot, int j, int off) __attribute__((always_inline)) {
        return *(const bf16x8*)(lds + slot * STAGE + (BM + wn * 64 + 32 * j) * 128 + off);
    };

    // ------------------------------------------------------------------------------------------------ epilogue pieces
    char* const stg0 = lds + C::RING + w * QSTG;
    // (per-lane epilogue addresses are recomputed from `lanev` inside every step - see epi_step - instead of living in ~25 registers
    // through the consume turns)
    int lanev = lane;
    char* stg = stg0;
    int ml = lane & 31, hl = lane >> 5;
    // a 32 x 32 logical tile (v[r]: row ml, column 8 (r >> 2) + 4 hl + (r & 3)) -> [32 rows][hi x 32 | lo x 32], 16-B chunks XOR (row & 7)
    auto stage_split = [&](const float (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16 h0, h1, h2, h3, l0, l1, l2, l3;
            cvt_pair<bf16, true>(v[4 * g], v[4 * g + 1], h0, h1, l0, l1);
            cvt_pair<bf16, true>(v[4 * g + 2], v[4 * g + 3], h2, h3, l2, l3);
            bf16x4 hv, lv;
            hv[0] = h0; hv[1] = h1; hv[2] = h2; hv[3] = h3;
            lv[0] = l0; lv[1] = l1; lv[2] = l2; lv[3] = l3;
            *(stg_b4*)(stg + ml * 128 + 16 * (g ^ (ml & 7)) + 8 * hl) = hv;
            *(stg_b4*)(stg + ml * 128 + 16 * ((4 + g) ^ (ml & 7)) + 8 * hl) = lv;
        }
    };
    // the staged 32 rows x 128 B -> global rows mbase .. mbase + 31 (clamped: rows past M replicate row M - 1 exactly), whole lines
    auto flush = [&](void* out, long ld_bytes, long col_bytes, int mbase) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");   // lanes exchange data through LDS inside one wave (in-order LDS, no barrier): compiler fence only
        u32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 8 * q + (lanev >> 3), ch = lanev & 7;
            v[q] = *(const u32x4 __attribute__((may_alias))*)(stg + row * 128 + 16 * (ch ^ (row & 7)));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 8 * q + (lanev >> 3), ch = lanev & 7;
            int m = mbase + row;
            m = m < p.M ? m : p.M - 1;
            if (!(abl & 4)) __builtin_nontemporal_store(v[q], (u32x4*)((char*)out + (long)m * ld_bytes + col_bytes + 16 * ch));
        }
        asm volatile("" ::: "memory");
    };

    // gelu'(pre) of the GELU backward: fetched by inline-asm loads at the start of an assist turn, complete - in-order vmcnt - once the
    // turn's first counted wait has passed
    u32x2 ax[EPI == EPI_GELU_BWD ? TM : 1][EPI == EPI_GELU_BWD ? TN : 1][4];   // 4 halves per quad
    f16x4 dq[2][4];                                                        // GELU with derivative: gelu' of one row group, both j
    auto fetch_epi_operands = [&](int m0, int n0) __attribute__((always_inline)) {
        if constexpr (EPI == EPI_GELU_BWD) {
            const char* gx = uniform_ptr(p.aux);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int m = m0 + wm * 32 * TM + 32 * i + ml;
                m = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        aload8(ax[i][j][g], (unsigned)m * (unsigned)(p.ldaux * 2) + (unsigned)(n0 + wn * 64 + 32 * j + 8 * g + 4 * hl) * 2u, gx);
            }
        }
    };
    auto claim_epi_operands = [&]() __attribute__((always_inline)) {   // the loads above have completed: hand the registers back to the compiler
        if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(ax[i][j][g]));
        }
    };

    const bool want_grad = EPI == EPI_BIAS_GELU && p.out0 != nullptr;
    const int nstep = want_grad ? TM * 3 : TM * TN;
    // epilogue steps per assist phase.  They start in phase NSLOT - 2: the operands fetched at the start of the turn are older than the turn's
    // first LDS-DMA, so they are complete once a counted wait leaves only LDS-DMA of THIS turn outstanding - the wait of phase NSLOT - 3
    constexpr int EPI_START = MFVIT_PP_CONS_DMA ? NSLOT - 2 : NSLOT - 1;   // (assist-only issue: the first counted wait of a turn is in phase NSLOT - 2)
    const int spp = (nstep + nk - EPI_START - 1) / (nk - EPI_START);

    // epilogue step `step` of the unit at (m0, n0) held in acc (bias already inside)
    auto epi_step = [&](auto step_c, f32x16 (&acc)[TM][TN], int m0, int n0) __attribute__((always_inline)) {
        constexpr int STEP = decltype(step_c)::value;
        // fresh address arithmetic in every step: the store addresses are invariant in the assist loop, and hoisted out of it they would
        // sit in ~64 registers across the whole turn (spills; scratch reloads count on vmcnt and would drain the LDS-DMA ring)
        asm volatile("" : "+v"(m0), "+v"(n0), "+v"(lanev));
        ml = lanev & 31;
        hl = lanev >> 5;
        const long colT = (long)(n0 + wn * 64) * 4;                          // byte column of this wave's 64 logical columns in a split row
        if constexpr (EPI == EPI_BIAS_GELU) {
            if (want_grad) {
                if constexpr (STEP < TM * 3) {
                    constexpr int i = STEP / 3, sub = STEP % 3;
                    const int mbase = m0 + wm * 32 * TM + 32 * i;
                    if constexpr (sub < 2) {
                        constexpr int j = sub;
                        float gv[16];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float d[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) gelu_both_t<T>(acc[i][j][4 * g + e], gv[4 * g + e], d[e]);
                            f16 a0, a1, a2, a3, u0, u1;
                            cvt_pair<f16, false>(d[0], d[1], a0, a1, u0, u1);
                            cvt_pair<f16, false>(d[2], d[3], a2, a3, u0, u1);
                            dq[j][g][0] = a0; dq[j][g][1] = a1; dq[j][g][2] = a2; dq[j][g][3] = a3;
                        }
                        stage_split(gv);
                        flush(p.out1, p.ldo1 * 2, colT + 128 * j, mbase);
                    } else {
                        // gelu'(pre) of both column tiles: [32 rows][64 halves] = 128 B per row
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) *(stg_h4*)(stg + ml * 128 + 16 * ((4 * j + g) ^ (ml & 7)) + 8 * hl) = dq[j][g];
                        flush(p.out0, p.ldo0 * (long)sizeof(AX), (long)(n0 + wn * 64) * (long)sizeof(AX), mbase);
                    }
                }
                return;
            }
        }
        if constexpr (STEP < TM * TN) {
            constexpr int i = STEP / TN, j = STEP % TN;
            const int mbase = m0 + wm * 32 * TM + 32 * i;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
            if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = gelu_t<T>(v[r]);
                stage_split(v);
                flush(p.out1, p.ldo1 * 2, colT + 128 * j, mbase);
            } else {
                if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        union { u32x2 u; f16x4 h; } cv;
                        cv.u = ax[i][j][g];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * g + e] *= (float)cv.h[e];
                    }
                }
                stage_split(v);
                flush(p.out0, p.ldo0 * 2, colT + 128 * j, mbase);
            }
        }
    };
    auto epi_dispatch = [&](int step, f32x16 (&acc)[TM][TN], int m0, int n0) __attribute__((always_inline)) {
        switch (step) {
#define MFVIT_PP_CASE(s) case s: epi_step(std::integral_constant<int, s>(), acc, m0, n0); break;
            MFVIT_PP_CASE(0) MFVIT_PP_CASE(1) MFVIT_PP_CASE(2) MFVIT_PP_CASE(3) MFVIT_PP_CASE(4) MFVIT_PP_CASE(5)
            MFVIT_PP_CASE(6) MFVIT_PP_CASE(7) MFVIT_PP_CASE(8) MFVIT_PP_CASE(9) MFVIT_PP_CASE(10) MFVIT_PP_CASE(11)
#undef MFVIT_PP_CASE
            default: break;
        }
    };

    // ------------------------------------------------------------------------------------------------ the two turns
    // Phase q (one per stage, closed by barrier B_q) of every wave: issue its pieces of stage q + NSLOT - 1 into the slot of stage q - 1 (every
    // wave finished reading it before B_{q-1}); before B_q wait until its pieces of stage q + 1 have landed - younger operations that may stay
    // in flight: the pieces of stages q + 2 .. q + NSLOT - 1 (and nothing else: epilogue stores in between only make the wait stricter).
    int q0 = 0, slot0 = 0;                               // global stage index / ring slot of the current unit's first stage
    auto phase_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * L) : "memory"); };

    // assist turn beside the consumption of a unit: the epilogue steps of the unit in `acc` (HAS_EPI)
    auto assist = [&](auto has_epi_c, f32x16 (&acc)[TM][TN], int pm0, int pn0) __attribute__((always_inline)) {
        constexpr bool HAS_EPI = decltype(has_epi_c)::value;
        if constexpr (HAS_EPI) fetch_epi_operands(pm0, pn0);      // older than every LDS-DMA of this turn: complete after the first wait
        if (!MFVIT_PP_CONS_DMA) {
            // assist-only issue: this turn issues stages q0 + NSLOT - 1 .. q0 + nk + NSLOT - 2 (the rest of the unit being consumed, then the
            // first NSLOT - 1 stages of this group's own next unit); past the workgroup's last unit the stream repeats it (issue_end)
            it_unit = cslot + (q0 / nk) * G;
            it_kt = NSLOT - 1;
            set_offsets(it_unit);
        }
        int islot = slot0 == 0 ? NSLOT - 1 : slot0 - 1;  // slot of stage q0 - 1 = slot of stage q0 + NSLOT - 1
        int step = 0;
        for (int kt = 0; kt < nk; ++kt) {
            PP_TICK(6);
            issue_begin(islot);
#pragma unroll
            for (int i = 0; i < L; ++i) issue_piece(i);
            issue_end();
            islot = islot + 1 == NSLOT ? 0 : islot + 1;
            PP_TICK(3);
            if constexpr (HAS_EPI) {
                if (kt >= EPI_START && !(abl & 8)) {
                    if (kt == EPI_START) claim_epi_operands();
                    for (int s = 0; s < spp && step < nstep; ++s, ++step) epi_dispatch(step, acc, pm0, pn0);
                }
            }
            PP_TICK(4);
            // assist-only issue: stage q + 1 is this wave's from phase NSLOT - 2 on (before that the consuming group issued it and waits for it)
            if (MFVIT_PP_CONS_DMA || kt >= NSLOT - 2 || !HAS_EPI) phase_wait();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_TICK(5);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        PP_TICK(6);
    };

    auto consume = [&](f32x16 (&acc)[TM][TN], int n0) __attribute__((always_inline)) {
        // accumulators start at the bias (column 8 (r >> 2) + 4 h + (r & 3) of tile j: scalar loads of the tile's 32 values, one select each)
        bool zero = true;
        if constexpr (HAS_BIAS) {
            if (p.bias) {
                zero = false;
                typedef float f32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x8 b0, b1, b2, b3;
                    const char* bp = uniform_ptr(p.bias + __builtin_amdgcn_readfirstlane(n0) + wn * 64 + 32 * j);
                    asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %4, 0x40\n\t"
                                 "s_load_dwordx8 %3, %4, 0x60\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&s"(b0), "=&s"(b1), "=&s"(b2), "=&s"(b3)
                                 : "s"(bp)
                                 : "memory");
                    const f32x8 bg[4] = {b0, b1, b2, b3};
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float bv = hh ? bg[r >> 2][4 + (r & 3)] : bg[r >> 2][r & 3];
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i][j][r] = bv;
                    }
                }
            }
        }
        if (zero) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        int slot = slot0;
        int islot = slot0 == 0 ? NSLOT - 1 : slot0 - 1;
        // One wave per SIMD feeds the matrix pipe: an MFMA occupies it for 32 cycles and issues in 4, so whatever else the wave has to issue
        // (fragment reads, its LDS-DMA pieces, address and loop arithmetic) is placed BETWEEN single MFMAs, at most a few instructions per
        // gap (MI355X_MICROARCH.md: <= 5 single-issue instructions hide in a gap) - in blocks between MFMA groups the same instructions left
        // the pipe idle (first version: 46 % MFMA-busy).  Per k step t the three products run in the term order of NtLoop::compute:
        //   P1 a_lo w_hi : + this wave's LDS-DMA pieces; in the stage's LAST k step it ends with the counted wait and the stage's barrier
        //                    (every fragment of the stage is in registers by then)
        //   P2 a_hi w_lo : + the reads of a_lo, w_hi of step t + 1 (the next stage once the barrier has passed)
        //   P3 a_hi w_hi : + the reads of a_hi, w_lo of step t + 1
        // so every fragment is requested >= TM * TN MFMAs before its first use.  Only w_hi is double-buffered (its successor is read during P2,
        // it is used again in P3); P3 runs row tile by row tile and refreshes each a_hi right behind its last use: 56 (40) fragment registers
        // beside the 128 (64) accumulators.
        bf16x8 alo[TM], wlo[TN], whi[2][TN], ahi[TM];
#pragma unroll
        for (int j = 0; j < TN; ++j) whi[0][j] = frag_w(slot, j, f_hi[0]);
#pragma unroll
        for (int i = 0; i < TM; ++i) alo[i] = frag_a(slot, i, f_lo[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) wlo[j] = frag_w(slot, j, f_lo[0]);
#pragma unroll
        for (int i = 0; i < TM; ++i) ahi[i] = frag_a(slot, i, f_hi[0]);
        auto product = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TN], auto side) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!(abl & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                    side(i * TN + j);
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        auto kstep = [&](auto ks_c, int kt, bf16x8 (&wh)[TN], bf16x8 (&whn)[TN]) __attribute__((always_inline)) {
            constexpr int KS = decltype(ks_c)::value;
            constexpr int HALF = (L + 1) / 2;            // LDS-DMA pieces of this wave per k step
            product(alo, wh, [&](int idx) __attribute__((always_inline)) {
                if (MFVIT_PP_CONS_DMA && idx < HALF && KS * HALF + idx < L) issue_piece(KS * HALF + idx);
            });
            int nslot = slot;
            if constexpr (KS == 1) {
                if (MFVIT_PP_CONS_DMA) {
                    issue_end();
                    phase_wait();
                } else if (kt <= NSLOT - 3) {
                    // this group issued the first NSLOT - 1 stages of its unit at the end of its assist turn: stage kt + 1 must have landed,
                    // the NSLOT - 3 - kt stages behind it may stay in flight (stores among them only make the wait stricter)
                    wait_stages<L>(NSLOT - 3 - kt);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PP_TICK(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                PP_TICK(1);
                nslot = slot + 1 == NSLOT ? 0 : slot + 1;
            }
            // (behind the unit's last k step these read the other group's next stage: valid LDS, results unused - no branch here)
            product(ahi, wlo, [&](int idx) __attribute__((always_inline)) {
                if (abl & 16) return;
                if (idx < TM) alo[idx] = frag_a(nslot, idx, f_lo[1 - KS]);
                else if (idx < TM + TN) whn[idx - TM] = frag_w(nslot, idx - TM, f_hi[1 - KS]);
            });
            product(ahi, wh, [&](int idx) __attribute__((always_inline)) {      // idx = i * TN + j: a_hi[i] is dead behind j = TN - 1
                if (abl & 16) return;
                if (idx % TN == TN - 1) ahi[idx / TN] = frag_a(nslot, idx / TN, f_hi[1 - KS]);
                else if (idx / TN < TN) wlo[idx / TN] = frag_w(nslot, idx / TN, f_lo[1 - KS]);
            });
            slot = nslot;
        };
        PP_TICK(2);
        if (MFVIT_PP_PRIO) __builtin_amdgcn_s_setprio(1);
        for (int kt = 0; kt < nk; ++kt) {
            if (MFVIT_PP_CONS_DMA) issue_begin(islot);
            islot = islot + 1 == NSLOT ? 0 : islot + 1;
            kstep(std::integral_constant<int, 0>(), kt, whi[0], whi[1]);
            kstep(std::integral_constant<int, 1>(), kt, whi[1], whi[0]);
            PP_TICK(2);
        }
        if (MFVIT_PP_PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // ------------------------------------------------------------------------------------------------ schedule
    // prologue: every wave issues its pieces of stages 0 .. NSLOT - 2 and waits for those of stage 0
    if (MFVIT_PP_CONS_DMA || grp == 1) {
        set_offsets(cslot);
        for (int s = 0; s < NSLOT - 1; ++s) {
            issue_begin(s);
#pragma unroll
            for (int i = 0; i < L; ++i) issue_piece(i);
            issue_end();
        }
        phase_wait();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    auto advance = [&]() __attribute__((always_inline)) {
        q0 += nk;
        slot0 = (slot0 + nk) % NSLOT;
    };
    int jj = 0;
    if (grp == 1) {
        f32x16 none[TM][TN];
        assist(std::false_type(), none, 0, 0);
        advance();
        jj = 1;
    }
    for (; jj < nloc; jj += 2) {
        // the accumulators live for one consume turn + the assist turn behind it: no loop-carried copies of 128 registers
        f32x16 acc[TM][TN];
        const int unit = cslot + jj * G;
        const int m0 = __builtin_amdgcn_readfirstlane((unit / ntn) * BM), n0 = __builtin_amdgcn_readfirstlane((unit % ntn) * QBN);
        consume(acc, n0);
        advance();
        if (jj + 1 < nloc) {
            assist(std::true_type(), acc, m0, n0);
            advance();
        } else {
            // last unit of the workgroup: the other group has left, nothing more to issue - the whole epilogue at once
            fetch_epi_operands(m0, n0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            claim_epi_operands();
#pragma nounroll
            for (int step = 0; step < nstep; ++step) {
                if (!(abl & 8)) epi_dispatch(step, acc, m0, n0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the refills issued past the end of the stream must land before the LDS is released
#if defined(MFVIT_PP_TRACE) && !defined(MFVIT_PP_NOTICKS)
    if (lane == 0 && p.cpart) {
        tacc[7] = (float)(__builtin_readcyclecounter() - tstart);
        for (int i = 0; i < 8; ++i) p.cpart[((long)blockIdx.x * 8 + wave) * 8 + i] = tacc[i];
    }
#endif
}

int n_cus() {
    static const int n = [] {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return n;
}

template <int TM, int EPI> int launch_pp(const GemmP& p, hipStream_t st) {
    typedef QCfg<TM> C;
    const int ntm = (p.M + C::BM - 1) / C::BM, ntn = p.N / QBN;
    const int ntiles = ntm * ntn;
    int G = n_cus();
    G &= ~7;
    if (G <= 0) G = 8;
    if (ntiles < G) G = ntiles;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<TM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        attr = true;
    }
    ProfScope ps(PROF_GEMM_TILE, 2.0 * p.M * p.N * p.K, 0, st);
#ifdef MFVIT_PP_TRACE
    GemmP q = p;
    { const char* e = getenv("MFVIT_PP_TRACE_PTR"); q.cpart = e ? (float*)strtoull(e, nullptr, 0) : nullptr; }
    MFVIT_LAUNCH((gemm_pp_kernel<TM, EPI>), dim3(G), dim3(512), C::LDS, st, q, ntm, ntn);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
#endif
    MFVIT_LAUNCH((gemm_pp_kernel<TM, EPI>), dim3(G), dim3(512), C::LDS, st, p, ntm, ntn);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// MFVIT_PP: 0 (default) off, 2 / 4 = rows per unit / 64 forced, 1 = chosen per shape.  Read at every launch (A/B runs in one process).
// OPT-IN: measured on MI355X (tools/pp_check.py, sustained launches, M = 25,216) it ties or loses against the 128x128 kernel on the wide
// linears (qkv 92 - 96 vs 82 us, fc1 + GELU 132 - 143 vs 134 us, fc2-dgrad 122 - 130 vs 120 us) and wins only on proj-dgrad (30.1 vs 33.8 us):
// the MFMA-issuing wave of a SIMD reaches ~50 % of the pipe even with every other instruction interleaved between single MFMAs, and
// what the structure saves in fill / epilogue bubbles it pays in LDS-DMA issue time on the CU's one vector-memory path.
int pp_mode() {
    const char* e = getenv("MFVIT_PP");
    return e ? atoi(e) : 0;
}

// rows per unit / 64: the 256-row unit moves 25 % fewer operand bytes; the 128-row unit quantises better when there are few units per CU
int pick_tm(const GemmP& p, int epi) {
    const int mode = pp_mode();
    // the 256-row unit keeps 64 registers of gelu'(pre) in flight beside 128 accumulators: the allocator spills them while the loads are
    // still outstanding (garbage) - the GELU backward runs 128-row units
    if (epi == EPI_GELU_BWD) return 2;
    if (mode == 2 || mode == 4) return mode;
    const long u4 = (long)((p.M + 255) / 256) * (p.N / QBN);
    const int cus = n_cus();
    const double rounds4 = (double)u4 / cus;
    // efficiency of the last round with 256-row units
    const double eff4 = rounds4 / (double)((u4 + cus - 1) / cus);
    return eff4 >= 0.80 ? 4 : 2;
}

}  // namespace

bool gemm_nt_pp_supported(int dtype, int epi, const GemmP& p) {
    if (dtype != MFVIT_BF16X3 || pp_mode() == 0) return false;
    if (epi != EPI_BIAS && epi != EPI_BIAS_GELU && epi != EPI_GELU_BWD && epi != EPI_NONE) return false;
    if (p.nb > 1 || p.M < 4096 || p.N % QBN || p.K % 32) return false;
    if (epi == EPI_GELU_BWD && (p.cs0 || !p.aux)) return false;               // column sums stay with the 128x128 kernel
    if (epi == EPI_BIAS_GELU && !p.out1) return false;
    if (p.K / 32 < 8) return false;                                             // at most two epilogue steps (of <= 12) in each of the phases 2 ..
    if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 32) || (unsigned long long)p.N * p.ldw * 2 >= (1ull << 32)) return false;
    if (p.aux && (unsigned long long)p.M * p.ldaux * 2 >= (1ull << 32)) return false;
    if (p.lda % 8 || p.ldw % 8 || (p.out0 && p.ldo0 % 8) || (p.out1 && p.ldo1 % 8) || (p.aux && p.ldaux % 4)) return false;
    if ((size_t)p.A % 16 || (size_t)p.W % 16 || (p.out0 && (size_t)p.out0 % 16) || (p.out1 && (size_t)p.out1 % 16) ||
        (p.aux && (size_t)p.aux % 8) || (p.bias && (size_t)p.bias % 4))
        return false;
    return true;
}

int gemm_nt_pp(int epi, const GemmP& p, hipStream_t st) {
    const int tm = pick_tm(p, epi);
#define MFVIT_PP_LAUNCH(E) return tm == 4 ? launch_pp<4, E>(p, st) : launch_pp<2, E>(p, st);
    switch (epi) {
        case EPI_BIAS: MFVIT_PP_LAUNCH(EPI_BIAS)
        case EPI_BIAS_GELU: MFVIT_PP_LAUNCH(EPI_BIAS_GELU)
        case EPI_GELU_BWD: MFVIT_PP_LAUNCH(EPI_GELU_BWD)
        case EPI_NONE: MFVIT_PP_LAUNCH(EPI_NONE)
    }
#undef MFVIT_PP_LAUNCH
    return MFVIT_EINVAL;
}

}  // namespace mfvit
