// Row-complete NT GEMM for SPLIT-bf16 tensors, one tall tile per CU ("rowp"):  out[M][384] = epi(A[M][K] W[384][K]^T) with the epilogues of
// gemm_nt_row -  + bias + residual -> LayerNorm (proj / fc2 of timm's Block, crossvit_2vits_..._sum.py:128-135)  and
// LayerNorm backward + residual-gradient add + dgamma / dbeta / dbias column sums (the qkv / fc1 data gradients).
//
// Why (round 3).  gemm_nt_row_kernel runs 394 workgroups of 64 rows, two per CU, on K tiles of HALF a split k group (64 bytes per row, so
// that two double-buffered workgroups fit the LDS).  Measured consequences: (1) every workgroup streams the whole W[384][K] from L2 - 930 MB
// per fc2 launch against 155 MB of activations; (2) a 64-byte K tile takes 32-byte pieces out of every 128-byte line and fetches the line
// again for the next tile: the L2 -> CU traffic in lines is twice the bytes used (the ping-pong kernel of round 3 measured the same effect on its LDS-DMA ring:
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

#include <mutex>

#include <type_traits>

#ifndef MFVIT_ROWX_STORE
#define MFVIT_ROWX_STORE 1
#endif
#ifndef MFVIT_ROWY_STORE
#define MFVIT_ROWY_STORE 1
#endif
// timing ablations of the forward mode (tools/build_variant_lib.sh; WRONG results): 1 no x_out store, 2 no y store, 4 no main loop, 8 no residual prefetch
#ifndef MFVIT_RP_ABL
#define MFVIT_RP_ABL 0
#endif

namespace mfvit {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((may_alias)) stg_b4;

constexpr int RP_N = 384;                       // output columns (embed dim): 8 waves x 48
constexpr int RP_MF = 7;                        // row fragments of 16 per tile
constexpr int RP_TH = 16 * RP_MF;               // 112 rows at most
// LDS: W ring 2 x 48 KB | A ring 4 x 14 KB | 8 KB of reduction scratch.  The W rows come out of L2 (every CU streams the same 384 x K
// matrix), the activation rows come from HBM once (inside a training step they are cold): their ring is four stages deep - a piece is
// requested three stage times (~3.5 us) before its first use; with one 64 KB stage of both in flight the kernel stalled on every stage
// inside the step (fc2 + LN 98 us isolated, where the 155 MB of activations sit in the Infinity Cache, but no faster than gemm_nt_row in it).
constexpr int RP_WSTAGE = RP_N * 128;           // 48 KB
constexpr int RP_WRING = 2 * RP_WSTAGE;         // 96 KB
constexpr int RP_ASTAGE = RP_TH * 128;          // 14 KB
constexpr int RP_ASLOTS = 4;
constexpr int RP_RING = RP_WRING + RP_ASLOTS * RP_ASTAGE;   // 152 KB
constexpr int RP_SCR = 8192;                    // reductions / statistics
constexpr int RP_LDS = RP_RING + RP_SCR;        // 163,840 B
constexpr int RP_TOUCH = 2;                     // L2 warm-up loads per thread in the prologue
constexpr int RP_LW = 6, RP_LA = 2;             // LDS-DMA instructions per wave and stage: W pieces wave + 8 i; A pieces wave, wave + 7
                                                // (14 pieces of 8 rows: wave 7 repeats wave 6's - identical bytes - so that every wave
                                                // counts the same number of operations)
constexpr int RP_XROWS = 100;                   // LayerNorm-backward mode: tile rows whose x (1536 B each) fit the ring
constexpr int RP_XDMA = (RP_XROWS * 96 + 511) / 512;   // 16-byte LDS-DMA instructions per thread for them (19: exactly the 152 KB ring)
constexpr int RP_MINM_FWD = 1, RP_MINM_BWD = 1;   // smallest M per epilogue (tools/rowp_small_m.py)
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

// 16 bytes per lane into registers, issued from asm: the compiler does not know the load is in flight (no s_waitcnt of its own - it would be a vmcnt(0) that
// also drains the LDS-DMA queue); the caller waits with a counted s_waitcnt and launders the registers behind it (tools/check_vmem_hazards.py scans the window)
__device__ __forceinline__ void rp_gload16(f32x4v& dst, unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

__device__ __forceinline__ f32x4v rp_mfma(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4v rp_mfma(f16x8 a, f16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// MODE: the two row-complete epilogues (REPI_RES_LN = 0, REPI_LNBWD_RES = 1; N = 384: one pass, npass = 1).  (Round 3 also ran the plain
// linears on this kernel as `npass` passes of 384 columns, MFVIT_ROWT=1: qkv 107 - 119 us against 80 of the 128 x 128 tile kernel, fc1 + GELU
// 144 - 152 against 135 - the 8-byte-per-lane partial-line stores of this accumulator layout cost 20 - 45 us per launch; removed in round 4,
// the measurements stay in DESIGN.md 5.)
// the same with 4 bytes per lane (used as a register-free "touch": the data lands in a dummy LDS line nobody reads)
__device__ __forceinline__ void rp_dma4(unsigned voff, const char* sbase, unsigned m0v) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(m0v)
                 : "memory");
}

// MF: row fragments of 16 that carry rows (the launcher picks the smallest that covers its rows per tile: the ring always holds RP_TH rows - past
// the tile they replicate its last row - but only MF fragments are multiplied, reduced and stored: at M = 3,152 a tile has 13 rows)
// T: sbf16 (split bf16: a 128-byte row = ONE 32-wide k group as [hi x 32 | lo x 32], three MFMAs per product) or a plain 16-bit type (bf16 / f16,
// round 4: a 128-byte row = 64 k values = TWO k steps, one MFMA each).  The fragment addressing is the same in both - chunk q of a row is the hi part /
// k step 0, chunk 4 + q the lo part / k step 1 - only the term table of the MFMAs and the output formats differ.
// (the body of the kernels below: tile `tile` of the launch owns rows m0 .. m0 + rpt - 1)
template <int MODE, int MF, typename T>
__device__ __forceinline__ void rowp_body(GemmP& p, const int tile, const int m0, const int rpt, const int npass) {
    constexpr int REPI = MODE;
    constexpr bool SPLIT = is_split<T>::value;
    constexpr int KG = SPLIT ? 32 : 64;                                // logical k values per 128-byte row
    constexpr int TERMS = SPLIT ? 3 : 2;                               // MFMAs per (row fragment, column fragment) and stage
    constexpr int PER = 3 * TERMS;                                     // MFMAs per row fragment and stage
    typedef typename Vec8<T>::type frag_t;
    typedef typename Vec4<T>::elem E16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = p.M - m0 < rpt ? p.M - m0 : rpt;                 // valid rows of this tile (1 .. RP_TH)
    // K splits (small M, round 5): S workgroups share a row tile, workgroup ks multiplies k groups [ks nk, (ks + 1) nk) of it, the partial accumulator
    // tiles meet in scratch and the workgroup that arrives LAST adds them up and runs the epilogue (no workgroup ever waits for another one: nothing
    // to deadlock beside a second kernel stream).  Why: a tile streams the whole W[384][K] through its CU's LDS whatever its height - at M = 3,152 that
    // stream (2.4 MB at K = 1,536) is all a launch does (48 us for 13 - 25 rows per CU); with S = 4 a CU streams a quarter of W for four times the rows.
    const int S = p.splits > 1 ? p.splits : 1;
    const int ks = S > 1 ? (int)blockIdx.x % S : 0;
    const int nk = p.K / KG / S;
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);

    // ---- LDS-DMA: a row = 8 chunks of 16 B, positions XOR (R >> 1) & 7 (R = row inside its slot); a piece = 8 rows = one 1 KB instruction:
    // position lane & 7 of row 8 pc + (lane >> 3) receives chunk (lane & 7) ^ ((4 (pc & 1) + (lane >> 4)) & 7)
    const int lrow = lane >> 3;
    auto coff_of = [&](int pc) __attribute__((always_inline)) { return (unsigned)(((lane & 7) ^ ((4 * (pc & 1) + (lane >> 4)) & 7)) * 16); };
    const int pa0 = wave < 7 ? wave : 6, pa1 = pa0 + 7;                // this wave's two A pieces
    unsigned voffw[RP_LW], voffa[RP_LA];
#pragma unroll
    for (int i = 0; i < RP_LW; ++i) voffw[i] = (unsigned)(8 * (wave + 8 * i) + lrow) * (unsigned)(p.ldw * 2) + coff_of(wave);
#pragma unroll
    for (int i = 0; i < RP_LA; ++i) {
        const int pc = i ? pa1 : pa0;
        int r = 8 * pc + lrow;
        r = r < rows ? r : rows - 1;                                   // rows past the tile replicate its last valid row
        voffa[i] = (unsigned)(m0 + r) * (unsigned)(p.lda * 2) + coff_of(pc);
    }
    const char* gW0 = rp_uniform_ptr(p.W);                             // (the L2 warm-up touches walk the WHOLE matrix)
    const char* gA = rp_uniform_ptr((const char*)p.A + (long)ks * nk * 128);
    const char* gW = rp_uniform_ptr((const char*)p.W + (long)ks * nk * 128);
    const long wpass = (long)RP_N * p.ldw * 2;                         // bytes between the W rows of two passes
    auto issue_w = [&](int pass, int stage, int slot, int i) __attribute__((always_inline)) {   // W rows of `pass`, k group `stage` -> W slot
        rp_dma16(voffw[i], rp_uniform_ptr(gW + pass * wpass + (long)stage * 128),
                 __builtin_amdgcn_readfirstlane(lbase + (unsigned)slot * RP_WSTAGE + (unsigned)(wave + 8 * i) * 1024u));
    };
    auto issue_a = [&](int stage, int slot, int i) __attribute__((always_inline)) {          // A k group `stage` -> A slot
        rp_dma16(voffa[i], rp_uniform_ptr(gA + (long)stage * 128),
                 __builtin_amdgcn_readfirstlane(lbase + RP_WRING + (unsigned)slot * RP_ASTAGE + (unsigned)(i ? pa1 : pa0) * 1024u));
    };

    // ---- fragments: lane (r = lane & 15, q = lane >> 4) holds k = 8 q .. 8 q + 7 of row base + r: chunk q (hi) / 4 + q (lo)
    const int fr = lane & 15, fq = lane >> 4, fsw = (fr >> 1) & 7;
    const int f_hi = fr * 128 + 16 * (fq ^ fsw), f_lo = fr * 128 + 16 * ((4 + fq) ^ fsw);
    auto frag_a = [&](int stage, int i, int off) __attribute__((always_inline)) {
        return *(const frag_t*)(lds + RP_WRING + (stage & (RP_ASLOTS - 1)) * RP_ASTAGE + (16 * i) * 128 + off);
    };
    auto frag_w = [&](int stage, int j, int off) __attribute__((always_inline)) {
        return *(const frag_t*)(lds + (stage & 1) * RP_WSTAGE + (48 * wave + 16 * j) * 128 + off);
    };

    // prologue.  Inside a training step W is cold, and every CU walks through it in lockstep: each stage would be a first touch served at HBM
    // latency (measured inside the step: fc2 + LN 123 us against 98 us on a warm W).  So first the CUs of an XCD (blockIdx & 7: round-robin
    // dispatch - an assumption for speed only) touch one dword of every 128-byte line of W once, two lines per thread: the stages behind the
    // first ones find their lines in that XCD's L2 (123 -> 105 us).  The touches are LDS-DMA loads into a dummy 256-byte line of the scratch
    // area: no destination register (a register destination was copied away and reused by the compiler while the load was still in
    // flight - caught by tools/check_vmem_hazards.py - the compiler does not know an asm load is outstanding).
    const int G = npass * nk;                                          // stages of the whole tile
    {
        const unsigned lpr = (unsigned)(p.K / KG);                             // 128-byte lines per W row
        const unsigned lines = (unsigned)p.N * lpr;                            // all passes
        const unsigned t0 = (unsigned)(tile >> 3) * 512u + (unsigned)tid;
        const unsigned nthr = ((gridDim.x + 7u) >> 3) * 512u;
#pragma unroll
        for (int k = 0; k < RP_TOUCH; ++k) {
            unsigned ln = t0 + (unsigned)k * nthr;
            ln = ln < lines ? ln : lines - 1;
            rp_dma4((ln / lpr) * (unsigned)(p.ldw * 2) + (ln % lpr) * 128u, gW0, __builtin_amdgcn_readfirstlane(lbase + RP_RING + RP_SCR - 256));
        }
    }
    // Forward epilogue mode: the accumulators START as the residual rows, so that the 39 MB of residual rows are fetched under the latency of the first operand
    // stages instead of in the epilogue, where nothing overlaps them (116 MB of exposed epilogue streaming became 77).  Sum order: res + products + bias instead
    // of products + bias + res - one f32 rounding apart.  Round 6: the loads are issued from asm and FIRST - the oldest operations of the workgroup - so that the
    // counted wait for stage 0 below covers them and nothing younger: as compiler-visible loads behind the prologue they needed a vmcnt(0), which also waited for
    // W(1) and A(1 .. 3), 90 KB per CU that the first stage does not need (profiles/r06_row_fwd_ablation.txt: 9 us of the launch).
    f32x4v acc[MF][3];
    {
        const bool with_res = REPI == REPI_RES_LN && p.res && ks == 0 && !(MFVIT_RP_ABL & 8);      // (one K split carries the residual)
        const char* gR = rp_uniform_ptr(p.res);
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        if constexpr (REPI == REPI_RES_LN) {
            if (with_res) {
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    int r0 = 16 * i + fr;
                    r0 = r0 < rows ? r0 : rows - 1;
                    const unsigned rowoff = (unsigned)(m0 + r0) * (unsigned)p.ldres + (unsigned)(48 * wave + 4 * fq);
                    // (the load result IS the accumulator: no arithmetic on it - the bias joins in the epilogue)
#pragma unroll
                    for (int j = 0; j < 3; ++j) rp_gload16(acc[i][j], (rowoff + 16u * (unsigned)j) * 4u, gR);
                }
            }
        }
    }
    // W stages 0, 1 and A stages 0 .. 3 in flight (clamped to the last stage for very short K), stage 0 landed
    // (requires nk >= 4: checked by the launcher)
#pragma unroll
    for (int i = 0; i < RP_LW; ++i) issue_w(0, 0, 0, i);
#pragma unroll
    for (int i = 0; i < RP_LA; ++i) issue_a(0, 0, i);
#pragma unroll
    for (int i = 0; i < RP_LW; ++i) issue_w(0, 1, 1, i);
#pragma unroll
    for (int st = 1; st < RP_ASLOTS; ++st)
#pragma unroll
        for (int i = 0; i < RP_LA; ++i) issue_a(st, st, i);
    // the touches, W(0), A(0) - and the residual rows, older than all of them - have completed; W(1) and A(1 .. 3) stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RP_LW + (RP_ASLOTS - 1) * RP_LA) : "memory");
    if constexpr (REPI == REPI_RES_LN) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(acc[i][j]));
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    frag_t wh[2][3], wl[2][3], ah[MF > 2 ? MF : 2], al[MF > 2 ? MF : 2];            // ah / al [0], [1]: unused (a01h / a01l hold those fragments)
    frag_t a01h[2][2], a01l[2][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        wh[0][j] = frag_w(0, j, f_hi);
        wl[0][j] = frag_w(0, j, f_lo);
    }
#pragma unroll
    for (int i = 0; i < (MF > 1 ? 2 : 1); ++i) {
        a01h[0][i] = frag_a(0, i, f_hi);
        a01l[0][i] = frag_a(0, i, f_lo);
    }

    // one stage; `side(slot index 0 .. 62)` runs behind every MFMA
    // lookahead of the two streams: W is at stage s + 2 = (wp, wk), A at stage s + 4 = k group ak (the same rows in every pass); past the end
    // of the tile both repeat its last stage (harmless refills of free slots: no branches in the MFMA stream)
    int wp = nk > 2 ? 0 : 1, wk = nk > 2 ? 2 : 2 - nk, ak = RP_ASLOTS % nk;
    if (wp >= npass) { wp = npass - 1; wk = nk - 1; }
    auto advance_streams = [&]() __attribute__((always_inline)) {
        if (++wk == nk) { wk = 0; ++wp; }
        if (wp >= npass) { wp = npass - 1; wk = nk - 1; }
        if (++ak == nk) ak = 0;
    };
    // Schedule of one stage (9 MF MFMAs, a side job behind each).  Before the barrier, behind the MFMAs of row fragments 0 .. NB - 1: the reads
    // of row fragments 2 .. MF - 1 of this stage (one every third MFMA).  The barrier stands behind row fragment NB - 1: every fragment of the
    // stage is in registers by then (the slot is free) and stage s + 1 - issued a stage earlier - has landed.  Behind it, NJ jobs: this wave's 8
    // LDS-DMA pieces of W(s + 2) / A(s + 4) into the slots just freed, the 6 W fragments and row fragments 0, 1 of stage s + 1.  MF = 7: barrier
    // behind the fourth row fragment, jobs on 27 slots (the schedule the round-3 numbers were measured on); MF <= 4: 18 slots, one job each;
    // MF = 1: 9 slots, two jobs each - such a tile is bound by its 48 KB W stage out of L2, not by the matrix pipe.
    constexpr int NB = SPLIT ? (MF >= 7 ? 4 : MF == 6 ? 3 : MF >= 4 ? 2 : MF == 3 ? 1 : 0) : (MF > 2 ? MF - 2 : 0);   // (plain: 6 MFMAs per fragment, two reads per fragment)
    constexpr int NPRE = MF > 2 ? 2 * (MF - 2) : 0;                   // fragment reads in front of the barrier
    constexpr int NAF = MF > 1 ? 4 : 2;                                // A fragment registers of stage s + 1 read behind it
    constexpr int POST = PER * (MF - NB);                                // MFMA slots behind the barrier
    constexpr bool SPACED = POST >= 27;                                // one DMA every third slot (MF >= 5), else the jobs back to back
    constexpr int NJ = RP_LW + RP_LA + 6 + NAF;
    constexpr int JPS = SPACED ? 1 : (NJ + POST - 1) / POST;           // jobs per slot
    static_assert(PER * NB / 3 >= NPRE, "fragment reads must fit in front of the barrier");
    // (MF <= 4: row fragments 0, 1 of stage s + 1 are read while those of stage s still feed MFMAs - a second register set, like W's)
    auto stage_body = [&](int s, frag_t (&wH)[3], frag_t (&wL)[3], frag_t (&wHn)[3], frag_t (&wLn)[3], frag_t (&aH)[2], frag_t (&aL)[2],
                          frag_t (&aHn)[2], frag_t (&aLn)[2]) __attribute__((always_inline)) {
        // job k of the back-to-back order: D0 W0 A0 D1 W1 A1 ... (D = LDS-DMA piece, W / A = fragment read of stage s + 1)
        auto dma_job = [&](int k) __attribute__((always_inline)) {
            if (k < RP_LW) issue_w(wp, wk, s & 1, k);
            else issue_a(ak, s & (RP_ASLOTS - 1), k - RP_LW);
        };
        auto w_job = [&](int k) __attribute__((always_inline)) {
            if (k < 3) wHn[k] = frag_w(s + 1, k, f_hi);
            else wLn[k - 3] = frag_w(s + 1, k - 3, f_lo);
        };
        auto a_job = [&](int k) __attribute__((always_inline)) {
            if constexpr (MF > 1) {
                if (k < 2) aHn[k] = frag_a(s + 1, k, f_hi);
                else aLn[k - 2] = frag_a(s + 1, k - 2, f_lo);
            } else {
                if (k == 0) aHn[0] = frag_a(s + 1, 0, f_hi);
                else aLn[0] = frag_a(s + 1, 0, f_lo);
            }
        };
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            if (i == NB) {
                // every fragment of stage s is in registers; W(s + 1) - and everything older - must have landed: the only younger operations
                // of this wave are its two pieces of A(s + 3)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RP_LA) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int j = t % 3, term = t / 3;
                // split: a_lo w_hi + a_hi w_lo + a_hi w_hi; plain: k step 0 (the "hi" registers), then k step 1
                const bool a_lo = SPLIT ? term == 0 : term == 1, w_lo = term == 1;
                const frag_t av = i < 2 ? (a_lo ? aL[i] : aH[i]) : (a_lo ? al[i] : ah[i]);
                acc[i][j] = rp_mfma(w_lo ? wL[j] : wH[j], av, acc[i][j]);
                const int idx = PER * i + t;
                if (idx < PER * NB) {
                    if (idx % 3 == 0 && idx / 3 < NPRE) {
                        const int k = idx / 3, fi = 2 + k / 2;
                        if (k % 2 == 0) ah[fi] = frag_a(s, fi, f_hi);
                        else al[fi] = frag_a(s, fi, f_lo);
                    }
                } else if constexpr (SPACED) {
                    const int u = idx - PER * NB;                              // 0 .. 26
                    if (u % 3 == 0 && u / 3 < RP_LW + RP_LA) dma_job(u / 3);
                    else if (u % 3 == 1 && u / 3 < 6) w_job(u / 3);
                    else if (u % 3 == 2 && u / 3 < NAF) a_job(u / 3);
                } else {
#pragma unroll
                    for (int q = 0; q < JPS; ++q) {
                        const int n = (idx - PER * NB) * JPS + q;              // job number: triples (D, W, A) while all three kinds last
                        // order: D0 W0 A0 .. D(NAF-1) W(NAF-1) A(NAF-1) | D W pairs up to W5 | the remaining D
                        if (n < 3 * NAF) {
                            if (n % 3 == 0) dma_job(n / 3);
                            else if (n % 3 == 1) w_job(n / 3);
                            else a_job(n / 3);
                        } else if (n < 3 * NAF + 2 * (6 - NAF)) {
                            const int m = n - 3 * NAF;
                            if (m % 2 == 0) dma_job(NAF + m / 2);
                            else w_job(NAF + m / 2);
                        } else if (n < NJ) {
                            dma_job(6 + (n - 3 * NAF - 2 * (6 - NAF)));
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // (the epilogue's lane-derived values are re-derived from a laundered lane id at the start of every phase: otherwise the compiler
    // computes the addresses of ALL phases up front and carries them - in scratch - across the reductions)
    int lanee = lane;
    int fre = fr, fqe = fq, ncol0 = 48 * wave + 4 * fq;                // ncol0 + 16 j + r = this lane's column
    auto fresh_lane = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(lanee));
        fre = lanee & 15;
        fqe = lanee >> 4;
        ncol0 = 48 * wave + 4 * fqe;
    };
    int m0e = m0;                                                      // laundered before every epilogue phase: fresh row arithmetic instead of
    auto row_of = [&](int i) __attribute__((always_inline)) {          // dozens of 64-bit row addresses carried (in scratch) across the reductions
        const int r = 16 * i + fre;
        return m0e + (r < rows ? r : rows - 1);
    };

    for (int g = 0; g < ((MFVIT_RP_ABL & 4) && REPI == REPI_RES_LN ? 2 : G); g += 2) {
        constexpr int AN = SPACED ? 0 : 1;                             // MF >= 5: one set (the fragments are dead by the time the next are read)
        stage_body(g, wh[0], wl[0], wh[1], wl[1], a01h[0], a01l[0], a01h[AN], a01l[AN]);
        advance_streams();
        stage_body(g + 1, wh[1], wl[1], wh[0], wl[0], a01h[AN], a01l[AN], a01h[0], a01l[0]);   // (nk is even: checked by the launcher)
        advance_streams();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                   // the ring is free: epilogue scratch
    if (S > 1) {
        // Hand-off of the partial tiles (MI355X_MICROARCH.md, cross-CU visibility, the "last arrival" row of its table): every byte is stored and loaded
        // by relaxed agent-scope 8-byte atomics (global_store / global_load ... sc1: through the XCD's L2 to memory), every storing wave waits for its
        // stores, a workgroup barrier, then ONE lane adds to the tile's counter; the workgroup whose add returns S - 1 is the last one - its other waves
        // learn that behind a second barrier - and only it reads.  The counter goes back to zero for the next launch on this stream.
        typedef unsigned long long u64;
        u64* const mine = (u64*)p.kpart + ((long)blockIdx.x * (RP_MF * 6) * 512 + tid);      // (one fixed-size slot per workgroup)
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u64 lo = ((u64)__float_as_uint(acc[i][j][1]) << 32) | __float_as_uint(acc[i][j][0]);
                const u64 hi = ((u64)__float_as_uint(acc[i][j][3]) << 32) | __float_as_uint(acc[i][j][2]);
                __hip_atomic_store(mine + (long)((i * 3 + j) * 2) * 512, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + (long)((i * 3 + j) * 2 + 1) * 512, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* const lastf = (int*)(lds + RP_RING + RP_SCR - 16);
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.kcnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(S - 1);
            if (last) __hip_atomic_store(p.kcnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *lastf = last;
        }
        __syncthreads();
        if (!*lastf) return;
        // the sum in a FIXED order, ((p0 + p1) + p2) + p3, whoever arrived last - its own partial comes back from the scratch like the others - so that
        // the result is the same bits on every run (tests/test_encoder_gpu.py::test_gradients_are_bit_identical_from_run_to_run)
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        for (int o = 0; o < S; ++o) {
            const u64* const theirs = (const u64*)p.kpart + ((long)(tile * S + o) * (RP_MF * 6) * 512 + tid);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u64 lo = __hip_atomic_load(theirs + (long)((i * 3 + j) * 2) * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const u64 hi = __hip_atomic_load(theirs + (long)((i * 3 + j) * 2 + 1) * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const f32x4v v = {__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                                      __uint_as_float((unsigned)(hi >> 32))};
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += v[r];
                }
        }
        __syncthreads();                                               // (lastf sits in the scratch area the epilogue reuses)
    }

    // ------------------------------------------------------------------------------------------------ epilogues
    // acc[i][j][r] = out[m0 + 16 i + fr][48 wave + 16 j + 4 fq + r]; rows past `rows` replicate the tile's last valid row exactly (clamped
    // A rows, clamped residual rows): they are stored to that row's address again - identical duplicates, no branches
    float* red = (float*)(lds + RP_RING);                              // [8 waves][RP_TH]
    float* red2 = red + 8 * RP_TH;
    // per-row totals over the 384 columns of `part[i]` (this lane's partial over its 12 values): all 512 threads call it
    auto row_total = [&](float (&part)[MF], float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float v = part[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (fqe == 0) buf[wave * RP_TH + 16 * i + fre] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float t = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) t += buf[w8 * RP_TH + 16 * i + fre];
            part[i] = t;
        }
    };
    // the same for two partials at once (one barrier)
    auto row_total2 = [&](float (&pa)[MF], float (&pb)[MF]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float a = pa[i], b = pb[i];
            a += __shfl_xor(a, 16, 64);
            b += __shfl_xor(b, 16, 64);
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 32, 64);
            if (fqe == 0) {
                red[wave * RP_TH + 16 * i + fre] = a;
                red2[wave * RP_TH + 16 * i + fre] = b;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float ta = 0.f, tb = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) {
                ta += red[w8 * RP_TH + 16 * i + fre];
                tb += red2[w8 * RP_TH + 16 * i + fre];
            }
            pa[i] = ta;
            pb[i] = tb;
        }
    };
    // output tile rows [16 i0, 16 i1) of this workgroup in the operand type -> out (storage ld `ldo`), through LDS (ring region) as whole lines
    constexpr int YB = RP_N * 2 * (SPLIT ? 2 : 1), YP = YB + 16, YCH = YB / 16;      // bytes of an output row, staging pitch, 16-byte chunks per row
    auto store_split = [&](void* out, long ldo, int i0, int i1) __attribute__((always_inline)) {
        char* ybuf = lds;
        fresh_lane();
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            if (i < i0 || i >= i1) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = ncol0 + 16 * j;                          // 4 consecutive logical columns n .. n + 3 (inside one 32-group)
                typedef typename Vec4<T>::type V4;
                typedef V4 __attribute__((may_alias)) stg4;
                if constexpr (SPLIT) {
                    bf16 h0, h1, h2, h3, l0, l1, l2, l3;
                    cvt_pair<bf16, true>(acc[i][j][0], acc[i][j][1], h0, h1, l0, l1);
                    cvt_pair<bf16, true>(acc[i][j][2], acc[i][j][3], h2, h3, l2, l3);
                    V4 hv, lv;
                    hv[0] = h0; hv[1] = h1; hv[2] = h2; hv[3] = h3;
                    lv[0] = l0; lv[1] = l1; lv[2] = l2; lv[3] = l3;
                    char* rowp = ybuf + (16 * (i - i0) + fre) * YP + 128 * (n >> 5) + 2 * (n & 31);
                    *(stg4*)rowp = hv;
                    *(stg4*)(rowp + 64) = lv;
                } else {
                    V4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = from_f32<E16>(acc[i][j][r]);
                    *(stg4*)(ybuf + (16 * (i - i0) + fre) * YP + 2 * n) = v;
                }
            }
        }
        __syncthreads();
        const int nrow = 16 * (i1 - i0);
        for (int q = tid; q < nrow * YCH; q += 512) {
            const int row = q / YCH, ch = q % YCH;
            const u32x4 v = *(const u32x4 __attribute__((may_alias))*)(ybuf + row * YP + 16 * ch);
            int r = 16 * i0 + row;
            r = r < rows ? r : rows - 1;
            // a PLAIN store for the operand-type output (y of the forward, the residual gradient of the backward): the NEXT launches read it - as the A
            // operand of a tile GEMM, once per 128-column tile - and find it in the L2 / Infinity Cache: tile class 90.4 -> 85.4 us per launch in the
            // step, this kernel + 1 us (same-box A/B, round 5).  MFVIT_ROWY_STORE=0 (A/B builds): system-scope streaming stores (no write-allocate fetch).
#if MFVIT_ROWY_STORE
            *(u32x4*)((char*)out + (long)(m0 + r) * ldo * 2 + 16 * ch) = v;
#else
            store16_stream((char*)out + (long)(m0 + r) * ldo * 2 + 16 * ch, v);
#endif
        }
        __syncthreads();
    };

    if constexpr (REPI == REPI_RES_LN) {
        const float invN = 1.0f / (float)RP_N;
        float part[MF];
        fresh_lane();
        // v = (residual + products) + bias: the residual rows were the accumulators' initial values
        f32x4v bj[3], gj[3], bej[3];                    // (gamma / beta too: requested in front of the x_out stores, or the wait for them waits for the stores)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bj[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (p.bias) bj[j] = *(const f32x4v*)(p.bias + ncol0 + 16 * j);
            gj[j] = *(const f32x4v*)(p.gamma + ncol0 + 16 * j);
            bej[j] = *(const f32x4v*)(p.beta + ncol0 + 16 * j);
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[i][j][r] += bj[j][r];
                    s += acc[i][j][r];
                }
            part[i] = s;
        }
        row_total(part, red);
        float mu[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
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
        asm volatile("" : "+s"(m0e));
        fresh_lane();
        // (the means are laundered: otherwise the compiler keeps the 12 MF differences acc - mu of the variance pass alive for the normalisation below, beside
        // the accumulators themselves, which still have to leave as x_out - one register more than the file holds, a spill and its drain of the store queue)
#pragma unroll
        for (int i = 0; i < MF; ++i) asm volatile("" : "+v"(mu[i]));
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const unsigned m = (unsigned)row_of(i);
            const float rs = rsqrtf(part[i] * invN + p.eps);
            if (wave == 0 && fqe == 0 && p.mean) {
                p.mean[m] = mu[i];
                p.rstd[m] = rs;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = ncol0 + 16 * j;
                // x_out: a plain store, like the operand-type output below - the next row kernel starts its accumulators from these rows (fc2 / proj + LN
                // 79.5 -> 74.3 us per launch in the step, round 5; MFVIT_ROWX_STORE=0 (A/B builds): the streaming store).  (Issued in front of the row
                // statistics instead - the rows are complete there - the class ran 1 us SLOWER, round 6: profiles/r06_row_fwd_ablation.txt)
#if MFVIT_ROWX_STORE
                if (p.out0 && !(MFVIT_RP_ABL & 1)) *(f32x4v*)((float*)p.out0 + m * (unsigned)p.ldo0 + n) = acc[i][j];
#else
                if (p.out0) store16_stream((float*)p.out0 + m * (unsigned)p.ldo0 + n, __builtin_bit_cast(u32x4_st, acc[i][j]));
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = (acc[i][j][r] - mu[i]) * rs * gj[j][r] + bej[j][r];
                if (p.y_f32) *(f32x4v*)((float*)p.out1 + m * (unsigned)p.ldo1 + n) = acc[i][j];
            }
        }
        if (!p.y_f32 && !(MFVIT_RP_ABL & 2)) {
            __syncthreads();                                           // (red / red2 live outside the ring; the ring itself is free)
            store_split(p.out1, p.ldo1, 0, MF < 4 ? MF : 4);
            if constexpr (MF > 4) store_split(p.out1, p.ldo1, 4, MF);
        }
    }

    if constexpr (REPI == REPI_LNBWD_RES) {
        // acc = dL/dy (y = LayerNorm output); aux = the saved LayerNorm input x (f32), mean / rstd its row statistics:
        //   h = (x - mu) rs,  g = dy gamma,  dx = rs (g - mean_n(g) - h mean_n(g h)) + residual gradient
        //   column sums over the VALID rows: dgamma += dy h, dbeta += dy, dcol += dx
        // x is needed on both sides of the row reduction.  Read twice from global memory by 84-accumulator lanes it cost 60 - 90 us per launch
        // (tools/rowp_epi_cost.py: one row fragment of loads in flight at a time, ~40 registers in scratch), three times the forward epilogue.
        // Now the tile's x rows (<= 100 x 1536 B: the launcher caps the tile height of this mode) are fetched ONCE, by LDS-DMA into the free ring:
        // 19 instructions per thread, all in flight at once, no registers; both phases read them from LDS (16-byte chunk c of row r sits at
        // position c ^ (r & 15): the 16 rows of a fragment read hit 16 different bank groups).  The residual gradient is folded in BEFORE the
        // reduction (u = rs g + res, prefetched two row fragments ahead), so the second phase touches no global memory but its output:
        //   dx = u - rs (c1 + h c2)
        const float invN = 1.0f / (float)RP_N;
        const char* gX = rp_uniform_ptr(p.aux);
        {
            const unsigned nchunk = (unsigned)rows * 96u;
#pragma unroll 1
            for (int n = 0; n < (int)((nchunk + 511u) / 512u); ++n) {                      // (<= RP_XDMA)
                unsigned q = (unsigned)n * 512u + (unsigned)tid;
                q = q < nchunk ? q : nchunk - 1;
                const unsigned r = q / 96u, c = (q % 96u) ^ (r & 15u);
                rp_dma16((unsigned)(m0 + (int)r) * (unsigned)(p.ldaux * 4) + 16u * c, gX,
                         __builtin_amdgcn_readfirstlane(lbase + ((unsigned)n * 512u + (unsigned)wave * 64u) * 16u));
            }
        }
        asm volatile("" : "+s"(m0e));
        fresh_lane();
        const float* __restrict__ gamp = p.gamma;
        const bf16* __restrict__ rest = (const bf16*)p.res_t;
        f32x4v gm[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) gm[j] = *(const f32x4v*)(gamp + ncol0 + 16 * j);
        // Row statistics of the tile (mean, rstd of the saved LayerNorm input) into the scratch area, read per row fragment from LDS in both
        // phases.  (Held in 14 registers per lane they were spilled together with the residual-gradient pointer, and every reload - a scratch
        // load - waits for ALL vector-memory operations in flight: the residual-gradient prefetch of the next fragments ran one fragment at a time.)
        float* stat = (float*)(lds + RP_RING + 2 * 8 * RP_TH * 4);      // behind red / red2: [2][RP_TH] (896 B of the 1024 left; the prologue's dummy line is dead)
        if (tid < RP_TH) {
            const unsigned m = (unsigned)(m0 + (tid < rows ? tid : rows - 1));
            stat[tid] = p.mean[m];
            stat[RP_TH + tid] = p.rstd[m];
        }
        auto mu_of = [&](int i) __attribute__((always_inline)) { return stat[16 * i + fre]; };            // (rows past the tile: the last valid row's, like x)
        auto rsd_of = [&](int i) __attribute__((always_inline)) { return stat[RP_TH + 16 * i + fre]; };
        // residual gradient of row fragment i -> rq[i % 3] (f32 rows, or the operand-type copy: hi + lo)
        int xlaunder = 0;                                              // (opaque zero, refreshed before the second phase: its reads of x must not
                                                                       // be merged with the first phase's - that kept 84 values of h alive, in scratch)
        auto x_lds = [&](int i, int j) __attribute__((always_inline)) {
            int r = 16 * i + fre;
            r = r < rows ? r : rows - 1;
            const int c = (12 * wave + 4 * j + fqe) ^ (r & 15);
            return *(const f32x4v*)(lds + xlaunder + r * 1536 + 16 * c);
        };
        // one column quantity: sum over the 16 row lanes, then out (a column belongs to exactly one wave)
        auto col_out = [&](float (&cv)[3][4], int q, float* dst) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) cv[j][r] += __shfl_xor(cv[j][r], o, 64);
                }
            if (fre != 0) return;
            if (p.cpart) {
                float* cp = p.cpart + ((long)tile * 3 + q) * RP_N + ncol0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    f32x4v v4 = {cv[j][0], cv[j][1], cv[j][2], cv[j][3]};
                    *(f32x4v*)(cp + 16 * j) = v4;
                }
            } else if (dst) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(dst + ncol0 + 16 * j + r, cv[j][r]);
            }
        };
        float s1[MF], s2[MF];
        // phase 1, instantiated per residual-gradient kind (0 none, 1 f32 rows, 2 the operand-type copy hi + lo): a run-time test per
        // element turned the phase into 400 branches and 100 spilled registers
        auto phase1 = [&](auto kind) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind)::value;
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            u32x4 rq[3][3];                                            // 16 bytes either way: 4 floats, or 4 hi + 4 lo bf16
            auto res_load = [&](int i) __attribute__((always_inline)) {
                if constexpr (KIND != 0) {
                    const unsigned m = (unsigned)row_of(i);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int n = ncol0 + 16 * j;
                        if constexpr (KIND == 1) {
                            rq[i % 3][j] = *(const u32x4 __attribute__((may_alias))*)(p.res + m * (unsigned)p.ldres + n);
                        } else {
                            const bf16* rp2 = rest + m * (unsigned)p.ldres_t + 64 * (n >> 5) + (n & 31);
                            const u32x2 hv = *(const u32x2 __attribute__((may_alias))*)rp2, lv = *(const u32x2 __attribute__((may_alias))*)(rp2 + 32);
                            rq[i % 3][j][0] = hv[0];
                            rq[i % 3][j][1] = hv[1];
                            rq[i % 3][j][2] = lv[0];
                            rq[i % 3][j][3] = lv[1];
                        }
                    }
                }
            };
            auto res_val = [&](const u32x4& q, int r) __attribute__((always_inline)) {
                if constexpr (KIND == 0) return 0.f;
                if constexpr (KIND == 1) return __uint_as_float(q[r]);
                const unsigned h = q[r >> 1], l = q[2 + (r >> 1)];
                return __uint_as_float((r & 1) ? (h & 0xffff0000u) : (h << 16)) + __uint_as_float((r & 1) ? (l & 0xffff0000u) : (l << 16));
            };
            res_load(0);
            if constexpr (MF > 1) res_load(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                           // every wave's x pieces have landed
            float cg[3][4], cb[3][4];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) cg[j][r] = cb[j][r] = 0.f;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                if (i + 2 < MF) res_load(i + 2);
                const bool ok = 16 * i + fre < rows;
                float a1 = 0.f, a2 = 0.f;
                const float mu_i = mu_of(i), rsd_i = rsd_of(i);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const f32x4v xv = x_lds(i, j);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float h = (xv[r] - mu_i) * rsd_i, dy = acc[i][j][r], g = dy * gm[j][r];
                        a1 += g;
                        a2 += g * h;
                        cg[j][r] += ok ? dy * h : 0.f;
                        cb[j][r] += ok ? dy : 0.f;
                        acc[i][j][r] = rsd_i * g + res_val(rq[i % 3][j], r);
                    }
                    // pinned here: left alone, the compiler sinks these updates behind the reduction (next to their uses) and carries h of
                    // every element there - through scratch
                    asm volatile("" : "+v"(a1), "+v"(a2), "+v"(acc[i][j]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                s1[i] = a1;
                s2[i] = a2;
                __builtin_amdgcn_sched_barrier(0);
            }
            col_out(cg, 0, p.cs0);
            col_out(cb, 1, p.cs1);
        };
        if (p.res) phase1(std::integral_constant<int, 1>{});
        else if (rest) phase1(std::integral_constant<int, 2>{});
        else phase1(std::integral_constant<int, 0>{});
        row_total2(s1, s2);
        asm volatile("" : "+v"(xlaunder));
        fresh_lane();
        float* dxo = (float*)p.out0;
        float cx[3][4];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) cx[j][r] = 0.f;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const bool ok = 16 * i + fre < rows;
            const float mu_i = mu_of(i), rsd_i = rsd_of(i);
            const float k1 = rsd_i * s1[i] * invN, k2 = rsd_i * s2[i] * invN;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f32x4v xv = x_lds(i, j);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dx = acc[i][j][r] - (k1 + (xv[r] - mu_i) * rsd_i * k2);
                    acc[i][j][r] = dx;
                    cx[j][r] += ok ? dx : 0.f;
                }
                // (a PLAIN store: the f32 copy is only asked for by block 0 of the lean split path and by the plain 16-bit types.  As an asm streaming
                // store - 64-bit address registers per statement - this line pushed the seven-fragment kernel into 148 bytes of scratch: 65 spill
                // accesses in the epilogue, each reload a drain of the LDS-DMA queue, 89.8 -> 100.3 us per launch in the step (round 5, same-box A/B
                // against the round-4 library; __graft_entry__.build() now checks the hot kernels for scratch))
                if (dxo) *(f32x4v*)(dxo + (unsigned)row_of(i) * (unsigned)p.ldo0 + ncol0 + 16 * j) = acc[i][j];
            }
        }
        col_out(cx, 2, p.cs2);
        if (p.out1) {
            __syncthreads();                                           // x is dead: the ring turns into the staging buffer of the split rows
            store_split(p.out1, p.ldo1, 0, MF < 4 ? MF : 4);
            if constexpr (MF > 4) store_split(p.out1, p.ldo1, 4, MF);
        }
    }
}

template <int MODE, int MF, typename T>
__global__ __launch_bounds__(512, 1) void gemm_rowp_kernel(GemmP p, int rpt, int npass) {
    const int tile = (int)blockIdx.x / (p.splits > 1 ? p.splits : 1);
    rowp_body<MODE, MF, T>(p, tile, tile * rpt, rpt, npass);
}
// MIXED tile heights (round 4): the first n_lo tiles carry 16 (MF - 1) rows and run the (MF - 1)-fragment body, the others rpt_hi <= 16 MF rows.  With one
// tile per CU, 25,216 rows are 98.5 per tile: 99 rows in 7 fragments of 16 multiply 13 padding rows in EVERY tile (12 % of the MFMAs); 216 tiles of 96 rows
// + 40 of 112 multiply none.  The launch is as long as its 7-fragment tiles either way - but it runs at the power cap (profiles/r04_clock_power_probe.txt), and
// the work the other tiles no longer do is clock for those.
template <int MODE, int MF, typename T>
__global__ __launch_bounds__(512, 1) void gemm_rowp_mixed_kernel(GemmP p, int n_lo, int rpt_hi, int npass) {
    const int tile = (int)blockIdx.x / (p.splits > 1 ? p.splits : 1);
    if (tile < n_lo) rowp_body<MODE, MF - 1, T>(p, tile, tile * 16 * (MF - 1), 16 * (MF - 1), npass);
    else rowp_body<MODE, MF, T>(p, tile, n_lo * 16 * (MF - 1) + (tile - n_lo) * rpt_hi, rpt_hi, npass);
}

int rp_cus() { return device_cus(); }

// rows per tile: the smallest whole number of rounds of one tile per CU that covers M with tiles of at most 112 rows, rows spread evenly
int rp_rows_per_tile(int M, int cap = RP_TH) {
    const int cus = rp_cus();
    const int rounds = (M + cus * cap - 1) / (cus * cap);
    int rpt = (M + cus * rounds - 1) / (cus * rounds);
    // small M with another kernel stream beside this one (stream_share() >= 2, common.cuh): at least 64 rows per tile, i.e. FEWER workgroups than
    // CUs - a tile streams the whole W[384][K] whatever its height, and a launch that leaves CUs free lets the other encoder's kernels run beside it.
    // MFVIT_ROWP_MINROWS overrides the floor (0: none).
    static int sw_min = INT_MIN;
    const int minr_env = env_switch("MFVIT_ROWP_MINROWS", -1, sw_min);
    const int minr = minr_env >= 0 ? minr_env : (stream_share() >= 2 ? 64 : 0);
    if (minr > 0 && rpt < minr) rpt = minr < cap ? minr : cap;
    return rpt;
}

// launch geometry: rows per tile, row tiles, K splits per tile (1: none).  K splits (see rowp_body) when the caller handed scratch in, K >= 1,024 and the
// rows are too few to give every workgroup of the stream's share of the chip (all CUs, or half of them beside a second kernel stream) a tall tile:
// S = that many workgroups per tile of ~ 56 rows, at most 4, each split an even number >= 4 of k groups.  MFVIT_ROWP_KSPLIT=0 switches them off, N forces N.
struct RpGeo { int rpt, tiles, S; };
RpGeo rp_geometry(const GemmP& p, int cap, int kg, bool bwd) {
    RpGeo g;
    g.S = 1;
    static int sw_ks = INT_MIN;
    const int ks_env = env_switch("MFVIT_ROWP_KSPLIT", -1, sw_ks);
    const int nk = p.K / kg;
    // (both epilogues: split in the forward only, the B = 16 step took 6.5 instead of 6.17 ms)
    (void)bwd;
    if (p.kpart && p.kcnt && ks_env != 0 && p.K >= 1024) {
        const int wgs = rp_cus() / (stream_share() >= 2 ? 2 : 1);      // workgroups this launch should put on the chip
        // as many splits as leave a tile ~ 56 rows (3 - 4 row fragments): measured at 16 pairs per step beside a second stream (M = 3,152, 128 workgroups):
        // S = 2 (64 tiles of 49 rows) 6.16 ms per step, 3 (43 x 74) 6.22, 4 (32 x 99) 6.42, none 6.5; at 32 pairs two splits of 99-row tiles lose (9.02 -> 9.17)
        int S = ks_env > 0 ? ks_env : (int)((long)wgs * 56 / p.M);
        S = S > 4 ? 4 : S;
        for (; S >= 2; --S) {
            if (nk % (2 * S) || nk / S < 4) continue;
            const int tiles = (wgs + S - 1) / S;
            const int rpt = (p.M + tiles - 1) / tiles;
            if (rpt > cap || rpt < 16) continue;
            g.S = S;
            g.rpt = rpt;
            g.tiles = (p.M + rpt - 1) / rpt;
            return g;
        }
    }
    g.rpt = rp_rows_per_tile(p.M, cap);
    g.tiles = (p.M + g.rpt - 1) / g.rpt;
    return g;
}
// arrival counters of the K-split launches: one zeroed set of 512 per stream (launches of one stream never overlap; every launch leaves its counters zero)
__device__ unsigned g_rowp_kcnt[16][512];
unsigned* rowp_counters(hipStream_t st) {
    // per DEVICE: the symbol has one instance on every GPU of the process, stream handles belong to a device
    constexpr int MAXDEV = 16;
    static std::mutex mu;
    static hipStream_t owner[MAXDEV][16];
    static int used[MAXDEV] = {};
    static unsigned* base[MAXDEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!base[dev] && hipGetSymbolAddress((void**)&base[dev], HIP_SYMBOL(g_rowp_kcnt)) != hipSuccess) return nullptr;
    for (int i = 0; i < used[dev]; ++i)
        if (owner[dev][i] == st) return base[dev] + i * 512;
    if (used[dev] == 16) return nullptr;                               // more streams than sets: no K splits for the newcomers
    owner[dev][used[dev]] = st;
    return base[dev] + (used[dev]++) * 512;
}
template <int MODE> constexpr int rp_cap() { return MODE == REPI_LNBWD_RES ? RP_XROWS : RP_TH; }

template <int MODE, int MF, typename T> int launch_rowp_mf(const GemmP& q, int grid, int rpt, hipStream_t st) {
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_rowp_kernel<MODE, MF, T>, hipFuncAttributeMaxDynamicSharedMemorySize, RP_LDS);
    }
    MFVIT_LAUNCH((gemm_rowp_kernel<MODE, MF, T>), dim3(grid), dim3(512), RP_LDS, st, q, rpt, q.N / RP_N);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

template <int MODE, typename T> int launch_rowp(const GemmP& p, hipStream_t st) {
    GemmP q = p;
    q.rows_per_wg = 0;
    q.kcnt = q.kpart ? rowp_counters(st) : nullptr;
    const RpGeo geo = rp_geometry(q, rp_cap<MODE>(), is_split<T>::value ? 32 : 64, MODE == REPI_LNBWD_RES);
    const int rpt = geo.rpt, tiles = geo.tiles;
    if (geo.S > 1 && tiles > 512) return MFVIT_EINVAL;                  // (cannot happen: tiles <= #CUs)
    q.splits = geo.S;
    const int grid = tiles * geo.S;                                      // workgroups: tile * S + k split
    ProfScope ps(MODE == REPI_RES_LN ? PROF_GEMM_ROW_FWD : PROF_GEMM_ROW_BWD, 2.0 * p.M * p.N * p.K, 0, st);
    const int mf = (rpt + 15) / 16;
    if (mf >= 2 && tiles > 1 && rpt % 16) {
        // mixed heights: n_lo tiles of 16 (mf - 1) rows, the rest as many rows as it takes, at most 16 mf and the mode's cap
        // rows of the tall tiles at most.  Their epilogue is the launch's critical path, so in the SERIALIZED pass the forward launch is shortest with
        // short tall tiles (74.9 us at 100 rows, 77.2 at 112) - but the timed step, at the power cap, follows the MFMA count: 26.85 ms uniform,
        // 26.63 / 26.55 / 26.56 / 26.52 ms at 100 / 104 / 108 / 112 rows (profiles/r04_kernel_experiments.txt)
        const int lo = 16 * (mf - 1);
        int cap = RP_TH;
        cap = cap > rp_cap<MODE>() ? rp_cap<MODE>() : cap;
        cap = cap > 16 * mf ? 16 * mf : (cap < lo + 1 ? lo + 1 : cap);
        int n_lo = (cap * tiles - p.M) / (cap - lo);                     // the most short tiles that leave <= cap rows for each of the others
        n_lo = n_lo < 0 ? 0 : (n_lo > tiles - 1 ? tiles - 1 : n_lo);
        const int rpt_hi = (p.M - lo * n_lo + (tiles - n_lo) - 1) / (tiles - n_lo);
        if (n_lo > 0 && rpt_hi > lo && rpt_hi <= cap) {
            auto go = [&](auto mtag) {
                constexpr int MFH = decltype(mtag)::value;
                static PerDeviceOnce attr_m;
                if (attr_m.first())
                    (void)hipFuncSetAttribute((const void*)gemm_rowp_mixed_kernel<MODE, MFH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, RP_LDS);
                MFVIT_LAUNCH((gemm_rowp_mixed_kernel<MODE, MFH, T>), dim3(grid), dim3(512), RP_LDS, st, q, n_lo, rpt_hi, q.N / RP_N);
            };
            switch (mf) {
            case 2: go(std::integral_constant<int, 2>{}); break;
            case 3: go(std::integral_constant<int, 3>{}); break;
            case 4: go(std::integral_constant<int, 4>{}); break;
            case 5: go(std::integral_constant<int, 5>{}); break;
            case 6: go(std::integral_constant<int, 6>{}); break;
            default: go(std::integral_constant<int, 7>{}); break;
            }
            MFVIT_CHECK_LAUNCH();
            return MFVIT_OK;
        }
    }
    switch (mf) {
    case 1: return launch_rowp_mf<MODE, 1, T>(q, grid, rpt, st);
    case 2: return launch_rowp_mf<MODE, 2, T>(q, grid, rpt, st);
    case 3: return launch_rowp_mf<MODE, 3, T>(q, grid, rpt, st);
    case 4: return launch_rowp_mf<MODE, 4, T>(q, grid, rpt, st);
    case 5: return launch_rowp_mf<MODE, 5, T>(q, grid, rpt, st);
    case 6: return launch_rowp_mf<MODE, 6, T>(q, grid, rpt, st);
    default: return launch_rowp_mf<MODE, 7, T>(q, grid, rpt, st);
    }
}

}  // namespace

bool gemm_nt_rowp_supported(int dtype, int repi, const GemmP& p) {
    // (the tall-tile kernel is the default for N = 384 in the three 16-bit types since round 4; its measurements against gemm_nt_row: DESIGN.md 5.
    // MFVIT_ROWP=0 keeps gemm_nt_row reachable for its own parity tests - it still serves the patch embedding and f32.)
    static int sw_on = INT_MIN;
    if ((dtype != MFVIT_BF16X3 && dtype != MFVIT_BF16 && dtype != MFVIT_F16) || env_switch("MFVIT_ROWP", 1, sw_on) == 0) return false;
    const bool split = dtype == MFVIT_BF16X3;
    if (repi != REPI_RES_LN && repi != REPI_LNBWD_RES) return false;
    const int minm = repi == REPI_RES_LN ? RP_MINM_FWD : RP_MINM_BWD;
    // an even number of stages (128-byte k groups: 32 logical columns split, 64 plain), at least 4
    if (p.N != RP_N || p.K % (split ? 64 : 128) || p.K < (split ? 128 : 256) || p.M < minm || p.nb > 1) return false;
    if (!split && p.res_t) return false;                                 // (the operand-type residual-gradient copy is a split-bf16 path)
    if (p.orow_in || p.res_mod || p.rows_per_wg) return false;                  // patch-embedding row remap stays with gemm_nt_row
    if (repi == REPI_LNBWD_RES) {
        if (!p.aux || !p.mean || !p.rstd || !p.gamma || p.ldaux % 4 || (size_t)p.aux % 16 || (size_t)p.gamma % 16) return false;
        if ((unsigned long long)p.M * p.ldaux * 4 >= (1ull << 32)) return false;                                      // 32-bit LDS-DMA offsets into x
        if ((p.out0 && (p.ldo0 % 4 || (size_t)p.out0 % 16)) || (p.out1 && (p.ldo1 % 8 || (size_t)p.out1 % 16))) return false;
        if ((p.res && (p.ldres % 4 || (size_t)p.res % 16)) || (p.res_t && (p.ldres_t % 8 || (size_t)p.res_t % 16))) return false;
        if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 32) || (unsigned long long)p.N * p.ldw * 2 >= (1ull << 32)) return false;
        if (p.lda % 8 || p.ldw % 8 || (size_t)p.A % 16 || (size_t)p.W % 16) return false;
        return true;
    }
    if (!p.out1 || !p.gamma || !p.beta) return false;
    if (p.res && (unsigned long long)p.M * p.ldres * 4 >= (1ull << 32)) return false;                                 // 32-bit offsets of the asm residual loads
    if ((unsigned long long)p.M * p.lda * 2 >= (1ull << 32) || (unsigned long long)p.N * p.ldw * 2 >= (1ull << 32)) return false;
    if (p.lda % 8 || p.ldw % 8 || (p.out0 && p.ldo0 % 4) || p.ldo1 % (p.y_f32 ? 4 : 8) || (p.res && p.ldres % 4)) return false;
    if ((size_t)p.A % 16 || (size_t)p.W % 16 || (size_t)p.out1 % 16 || (p.out0 && (size_t)p.out0 % 16) || (p.res && (size_t)p.res % 16) ||
        (p.bias && (size_t)p.bias % 16) || (size_t)p.gamma % 16 || (size_t)p.beta % 16)
        return false;
    return true;
}

template <int MODE> static int launch_rowp_t(int dtype, const GemmP& p, hipStream_t st) {
    if (dtype == MFVIT_BF16X3) return launch_rowp<MODE, sbf16>(p, st);
    if (dtype == MFVIT_BF16) return launch_rowp<MODE, bf16>(p, st);
    if (dtype == MFVIT_F16) return launch_rowp<MODE, f16>(p, st);
    return MFVIT_EINVAL;
}

int gemm_nt_rowp(int dtype, int repi, const GemmP& p, hipStream_t st) {
    if (repi == REPI_RES_LN) return launch_rowp_t<REPI_RES_LN>(dtype, p, st);
    if (repi == REPI_LNBWD_RES) {
        const int rc = launch_rowp_t<REPI_LNBWD_RES>(dtype, p, st);
        if (rc != MFVIT_OK || !p.cpart) return rc;
        GemmP q = p;
        q.kcnt = q.kpart ? rowp_counters(st) : nullptr;               // (the same geometry the launch used)
        const RpGeo geo = rp_geometry(q, RP_XROWS, dtype == MFVIT_BF16X3 ? 32 : 64, true);
        return colpart_reduce(p.cpart, geo.tiles, RP_N, 3, p.cs0, p.cs1, p.cs2, st);   // [tile][3][384] partials -> dgamma, dbeta, dcol
    }
    return MFVIT_EINVAL;
}

}  // namespace mfvit
