// Weight-gradient GEMM with LDS-DMA staging (bf16):  out0[n][k] (f32, accumulated) += sum_m A[m][n] X[m][k].
//
// Why a second wgrad kernel: in gemm_tn_kernel (gemm.hip) every stage is copied global -> VGPR -> LDS with ds_write_b128, which the
// LDS accepts at ~79 B/clk/CU (13 cycles per wave-instruction): 416 LDS cycles per 32 KB stage next to 256 cycles of fragment reads
// and 512 cycles of MFMAs per wave - the staging writes, not the matrix core, set the pace, and with one stage of register
// prefetch a stage must land within one stage time.  Here the operands go global -> LDS directly (global_load_lds_dwordx4, no VGPR
// round trip, no ds_write) through a 3-slot ring with counted vmcnt waits (two stages in flight), one barrier per stage.
//   * operands are reduction-major in HBM ([m][n], [m][k]); the LDS image is [64 m-rows][128 columns] bf16 with a 256-B pitch and
//     NO padding (an LDS-DMA writes 1 KB contiguously = 4 rows); the transposing fragment reads (ds_read_b64_tr_b16: 4 rows x 32 B
//     per 16 lanes) stay conflict-free because the 16-B chunks of row m are XOR-swizzled by 4 * (m & 3) - applied on the SOURCE
//     address of the DMA and on the read address (same involution on both sides).
//   * one workgroup per CU and launch wave (96 KB of LDS): tiles x splits <= 256.
//   * W8 (round 3, default): EIGHT waves per workgroup.  Waves 4 - 7 own the same four 64 x 64 quadrants as waves 0 - 3 and take the second
//     half of every stage's reduction rows; the two halves are added through LDS (the ring is free by then) before the float atomics.  The
//     tile, the LDS traffic and the L2 traffic are unchanged - what changes is that every SIMD holds two MFMA-issuing waves instead of one, so
//     one wave's transposing fragment reads hide behind the other's MFMAs (MFVIT_TN2_W8=0 restores four waves).
//   * a split's last stage may be partial: its rows past the end are fetched clamped (finite data) and the A-operand rows are
//     zeroed in LDS before use, so they add nothing.
#include <stdlib.h>

#include <type_traits>
#include "gemm.cuh"
#include "kernels.h"
#include "prof.h"

#ifndef MFVIT_TNPART_PLAIN
#define MFVIT_TNPART_PLAIN 0
#endif

namespace mfvit {

namespace {

constexpr int T2_NS = 3;
constexpr int T2_NSP = 4;                       // ring slots of the pipelined form (split bf16, eight waves): three stages in flight
constexpr int T2_TILE = 16 * 1024;              // bytes per operand and stage (KR rows x TW columns x 2 B)
constexpr int T2_STAGE = 2 * T2_TILE;           // 32 KB

// Tile geometry by element type.  Plain 16-bit types: 128 x 128 output tile, 64 reduction rows per stage.  Split bf16 (sbf16): the
// operand tiles are 256 STORAGE columns wide = 128 logical columns as [hi x 32 | lo x 32] groups, 32 reduction rows per stage (the
// same 16 KB per operand), so a workgroup still owns a 128 x 128 LOGICAL tile and each wave a 64 x 64 logical one:
// per 16-row k step 8 fragments (hi / lo of 2 + 2 groups) feed 12 MFMAs  (a_hi b_hi + a_lo b_hi + a_hi b_lo per 32 x 32 tile).
template <typename T> struct T2Geo {
    static constexpr bool SPLIT = is_split<T>::value;
    static constexpr int TW = SPLIT ? 256 : 128;   // storage columns per operand tile
    static constexpr int KR = SPLIT ? 32 : 64;     // reduction rows per stage
    static constexpr int ROWB = TW * 2;            // LDS row pitch in bytes (no padding: an LDS-DMA writes 1 KB contiguously)
    static constexpr int CPR = ROWB / 16;          // 16-B chunks per row
    static constexpr int RPI = 64 / CPR;           // rows filled by one LDS-DMA wave instruction
    static constexpr int NF = TW / 2 / 32;         // 32-column fragments per operand and wave
    static constexpr int BL = 128;                 // logical tile edge
};

template <int N> __device__ __forceinline__ void t2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void t2_glds16(const void* gsrc, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_off)
                 : "memory");
}

// transposed fragment of the swizzled image: lane holds column (rowbase + lane & 31), elements k = 16 s + 8 (lane >> 5) .. +8.
// The 16-B chunks of row m sit XOR-swizzled by 4 * (m & 3): the 4 rows of one tr-read block land in 4 disjoint 64-B windows of
// the bank row, whatever the row pitch (256 or 512 B).
template <typename T> __device__ __forceinline__ typename Vec8<T>::type t2_frag(const char* t, int rowbase, int s, int lane) {
    constexpr int ROWB = T2Geo<T>::ROWB;
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
    const int k0 = 16 * s + 8 * h + q;                       // (k0 & 3) == q, also for k0 + 4
    const int col = rowbase + 16 * g1 + 4 * pp;
    const char* a = t + k0 * ROWB + (((col >> 3) ^ (4 * q)) << 4) + (col & 7) * 2;
    typedef __attribute__((address_space(3))) s16x4* lptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a + 4 * ROWB));
    union { struct { s16x4 a, b; } s; typename Vec8<T>::type v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

// T = bf16 | f16 | sbf16 (split bf16, see T2Geo)
// RM: the rows of A are REMAPPED (GemmP::orow_*: the patch-embedding weight gradient reads the token rows 1 .. np of every image of a [B][np + 1][D]
// gradient tensor); instantiated for the default split-bf16 form only
// TWO (opt-in, MFVIT_WGRAD_TERMS=2; pipelined split form only): the dY_lo x_hi term is dropped - two MFMAs per product, dY enters at bf16 precision
// (2^-9 per element, averaged down by the sum over the token rows), x keeps both parts
template <typename T, bool CS, bool W8, bool IL, bool RM = false, bool TWO = false>
__global__ __launch_bounds__(W8 ? 512 : 256) void gemm_tn_glds_kernel(GemmP p) {
    typedef T2Geo<T> G;
    constexpr int NWV = W8 ? 8 : 4;                          // waves
    constexpr int NI = 16 / NWV;                             // LDS-DMA instructions per wave, operand and stage
    constexpr int T2_LPS = 2 * NI;
    constexpr bool SPLIT = G::SPLIT;
    constexpr int EP = elems_per<T>::value, TW = G::TW, KR = G::KR, ROWB = G::ROWB, CPR = G::CPR, RPI = G::RPI, NF = G::NF;
    constexpr int NL = 2;                                    // 32 x 32 LOGICAL output tiles per wave and dimension
    typedef typename Vec8<T>::type frag_t;
    typedef typename Vec4<T>::elem E16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3, kh = wave >> 2;                 // output quadrant; half of the stage's reduction rows (W8)
    const int wm = wq >> 1, wn = wq & 1;
    // PAIRED launch (p.res_mod = N2 > 0): a second weight gradient with the same reduction rows and the same K appended along N - its tiles
    // come behind the first one's, the split count is common.  (qkv: 27 tiles and proj: 9 tiles give 36 x 7 = 252 workgroups in ONE launch instead
    // of 243 + 252 in two: half the float atomics, 12.5 us per launch, and one prologue.)  Second set: A2 = aux / ldaux, X2 = res_t / ldres_t,
    // out2 = out1 / ldo1; the bias column sums belong to the first.
    const int ntk = p.K * EP / TW;
    const int tiles1 = ntk * (p.N * EP / TW);
    const int tiles = tiles1 + ntk * (p.res_mod * EP / TW);
    const int lin = xcd_remap(blockIdx.x, gridDim.x);        // each XCD gets a contiguous run of the split-major block order
    const int split = lin / tiles, tile_all = lin % tiles;
    const bool second = tile_all >= tiles1;                  // (block-uniform)
    const int tile = second ? tile_all - tiles1 : tile_all;
    const int n0 = (tile / ntk) * TW, k0 = (tile % ntk) * TW;   // STORAGE columns
    int chunk = (p.M + p.splits - 1) / p.splits;
    chunk = (chunk + KR - 1) / KR * KR;
    const int mbeg = split * chunk;
    const int mend = min(p.M, mbeg + chunk);
    if (mbeg >= mend) return;
    const int nst = (mend - mbeg + KR - 1) / KR;
    const E16* A = (const E16*)(second ? p.aux : p.A);
    const E16* X = (const E16*)(second ? p.res_t : p.W);
    const long lda_ = second ? p.ldaux : p.lda, ldw_ = second ? p.ldres_t : p.ldw;

    // LDS-DMA g = wave + NWV i (i = 0 .. NI-1) of an operand fills rows RPI g .. RPI g + RPI - 1: lane -> row RPI g + lane / CPR, chunk
    // position lane % CPR, which receives source chunk (lane % CPR) ^ (4 * (row & 3))   (NWV RPI i is a multiple of 4)
    const int lrow = RPI * wave + lane / CPR;                // + NWV RPI i
    const int lch = (lane % CPR) ^ (4 * (lrow & 3));
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)wave * 1024u);
    auto issue = [&](int st, int slot) {
        const unsigned sa = lbase + (unsigned)slot * T2_STAGE, sb = sa + T2_TILE;
        const int mrow = mbeg + st * KR + lrow;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int m = mrow + NWV * RPI * i;
            m = m < p.M ? m : p.M - 1;                        // never read past the tensor (clamped rows are zeroed below)
            if constexpr (RM) m = out_row(p, m);
            t2_glds16(A + (long)m * lda_ + n0 + lch * 8, sa + i * NWV * 1024);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int m = mrow + NWV * RPI * i;
            m = m < p.M ? m : p.M - 1;
            t2_glds16(X + (long)m * ldw_ + k0 + lch * 8, sb + i * NWV * 1024);
        }
    };

    // IL: the same instructions ONE at a time, spread between the MFMAs of the stage (d = 0 .. 2 NI - 1: A pieces, then X pieces).  Issued
    // in one burst behind the barrier they hold the wave's instruction stream for the ~500 cycles the CU's vector-memory path needs to take
    // 32 KB - with the matrix pipe idle: DMA phase + MFMA phase = the measured 2.1 x of the MFMA time per stage.  Stages past the end of
    // the split re-fetch its last stage into the free slot (harmless; no branch in the MFMA stream).
    auto dma_one = [&](int st, int slot, int d) __attribute__((always_inline)) {
        const int stc = st < nst ? st : nst - 1;
        const int i = d % NI;
        int m = mbeg + stc * KR + lrow + NWV * RPI * i;
        m = m < p.M ? m : p.M - 1;
        const unsigned dst = lbase + (unsigned)slot * T2_STAGE + (d >= NI ? T2_TILE : 0) + i * NWV * 1024;
        if (d < NI) t2_glds16(A + (long)(RM ? out_row(p, m) : m) * lda_ + n0 + lch * 8, dst);
        else t2_glds16(X + (long)m * ldw_ + k0 + lch * 8, dst);
    };

    f32x16 acc[NL][NL], bacc[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
        for (int j = 0; j < NL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[i][r] = 0.f;
    }
    frag_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (E16)1.0f;
    // column sums of A (= bias gradient): the workgroups of k-tile 0; the two waves that hold the same A fragments (wn = 0 / 1) take one 32-row block each
    // (round 6: until then wn = 0 summed both - a third more MFMAs on half the waves of those workgroups, and the launch waits for its slowest workgroup)
    const bool do_cs = CS && !second && k0 == 0;

    // the bias-column-sum MFMAs are selected ONCE per wave (template flag), not per k-step: a branch inside the hot loop splits it
    // into basic blocks and serialises the ds_read -> MFMA pipeline (measured: 72 vs 52 us on the kernels that carry a bias sum)
    // Split bf16, eight waves, interleaved issue ("pipelined" form): the stage barrier sits in the MIDDLE of the stage.  A wave enters stage st
    // with its 8 fragments already in registers and starts on the MFMAs at once; behind MFMA 6 it waits for stage st + 1 (requested a stage
    // earlier), meets the other waves, and then reads the fragments of stage st + 1 into the second register set BETWEEN MFMAs 7 - 12.  The
    // barrier -> 16 transposing reads -> first MFMA chain of the plain form (every wave of the CU in that phase at the same time, the matrix
    // pipe idle) is gone; the LDS-DMA pieces of stage st + 2 go out one per three MFMAs as before.
    auto main_loop_pipe = [&](auto cs_tag) {
        constexpr bool WITH_CS = decltype(cs_tag)::value;
        static_assert(!(SPLIT && W8 && IL) || (KR == 32 && NF == 4 && NI == 2), "pipelined wgrad loop: geometry");
        frag_t fa[2][NF], fb[2][NF];
        auto zero_partial = [&](int st_, int slot_) __attribute__((always_inline)) {
            const int valid = mend - mbeg - st_ * KR;            // rows of the stage inside the split
            if (valid < KR) {                                    // partial last stage: zero the A rows past the end (block-uniform branch)
                char* ta = lds + slot_ * T2_STAGE;
                for (int q = tid; q < (KR - valid) * CPR; q += NWV * 64)
                    *(uint4*)(ta + (valid + q / CPR) * ROWB + (q % CPR) * 16) = make_uint4(0, 0, 0, 0);
                __syncthreads();
            }
        };
        auto load_one = [&](auto set_tag, int slot_, int q) __attribute__((always_inline)) {
            constexpr int SET = decltype(set_tag)::value;
            const char* ta = lds + slot_ * T2_STAGE;
            if (q < NF) fa[SET][q] = t2_frag<T>(ta, (wm * NF + q) * 32, kh, lane);
            else fb[SET][q - NF] = t2_frag<T>(ta + T2_TILE, (wn * NF + q - NF) * 32, kh, lane);
        };
        auto mfma_t = [&](auto set_tag, int t) __attribute__((always_inline)) {
            constexpr int SET = decltype(set_tag)::value;
            const int term = t / (NL * NL), i = (t % (NL * NL)) / NL, j = t % NL;
            if (TWO && term == 0) return;                        // (a_lo b_hi: its fragment reads become dead code as well)
            acc[i][j] = MmaTraits<T>::mma(fa[SET][2 * i + (term == 0 ? 1 : 0)], fb[SET][2 * j + (term == 1 ? 1 : 0)], acc[i][j]);
        };
        auto stage = [&](int st, auto set_tag, auto nset_tag, int slot) __attribute__((always_inline)) {
            const int nslot = slot == 0 ? T2_NSP - 1 : slot - 1; // slot of stage st + 3 = that of st - 1: its last readers passed the previous mid-stage barrier
            const int rslot = slot == T2_NSP - 1 ? 0 : slot + 1; // slot of stage st + 1
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                mfma_t(set_tag, t);
                if (t == 2) dma_one(st + T2_NSP - 1, nslot, 0);
                if (t == 5) dma_one(st + T2_NSP - 1, nslot, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            t2_wait_vm<2 + (T2_NSP - 3) * T2_LPS>();             // all but the two pieces just issued and stage st + 2's: stage st + 1 has landed
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < nst) zero_partial(st + 1, rslot);
#pragma unroll
            for (int t = 6; t < 12; ++t) {
                mfma_t(set_tag, t);
                // 8 fragment reads of stage st + 1 (past the end: a stale slot, never used) and the other two LDS-DMA pieces
                if (t == 6) { load_one(nset_tag, rslot, 0); load_one(nset_tag, rslot, 1); }
                if (t == 7) { load_one(nset_tag, rslot, 2); dma_one(st + T2_NSP - 1, nslot, 2); }
                if (t == 8) { load_one(nset_tag, rslot, 3); load_one(nset_tag, rslot, 4); }
                if (t == 9) { load_one(nset_tag, rslot, 5); dma_one(st + T2_NSP - 1, nslot, 3); }
                if (t == 10) load_one(nset_tag, rslot, 6);
                if (t == 11) load_one(nset_tag, rslot, 7);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (WITH_CS) {
                constexpr int SET = decltype(set_tag)::value;
                // (wave-uniform select of this wave's row block: both candidates are in registers, no branch in the MFMA stream)
                const frag_t ch = wn ? fa[SET][2] : fa[SET][0], cl = wn ? fa[SET][3] : fa[SET][1];
                bacc[0] = MmaTraits<T>::mma(ch, ones, bacc[0]);
                bacc[0] = MmaTraits<T>::mma(cl, ones, bacc[0]);
            }
        };
#pragma unroll
        for (int q = 0; q < T2_NSP - 1; ++q) issue(q < nst ? q : nst - 1, q);
        t2_wait_vm<(T2_NSP - 2) * T2_LPS>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        zero_partial(0, 0);
#pragma unroll
        for (int q = 0; q < 2 * NF; ++q) load_one(std::integral_constant<int, 0>{}, 0, q);
        int slot = 0;
        for (int st = 0; st < nst; st += 2) {
            stage(st, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, slot);
            slot = slot == T2_NSP - 1 ? 0 : slot + 1;
            if (st + 1 < nst) {
                stage(st + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, slot);
                slot = slot == T2_NSP - 1 ? 0 : slot + 1;
            }
        }
        t2_wait_vm<0>();                                         // the refills behind the end have landed too (the ring is reused below)
    };
    auto main_loop = [&](auto cs_tag) {
        constexpr bool WITH_CS = decltype(cs_tag)::value;
        issue(0, 0);
        if (IL || nst > 1) issue(nst > 1 ? 1 : 0, 1);
        int slot = 0;
        for (int st = 0; st < nst; ++st) {
            if (IL || st + 1 < nst) t2_wait_vm<T2_LPS>();        // stage st landed; stage st+1 (IL: or the refill behind the end) may stay in flight
            else t2_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int nslot = slot == 0 ? 2 : slot - 1;          // (st + 2) % 3: read last in step st-1, free since the barrier
            if constexpr (!IL)
                if (st + 2 < nst) issue(st + 2, nslot);
            char* ta = lds + slot * T2_STAGE;
            const char* tb = ta + T2_TILE;
            const int valid = mend - mbeg - st * KR;             // rows of this stage inside the split
            if (valid < KR) {                                    // partial last stage: zero the A rows past the end (block-uniform branch)
                for (int q = tid; q < (KR - valid) * CPR; q += NWV * 64)
                    *(uint4*)(ta + (valid + q / CPR) * ROWB + (q % CPR) * 16) = make_uint4(0, 0, 0, 0);
                __syncthreads();
            }
            constexpr int KS = KR / 16 / (W8 ? 2 : 1);          // k steps of 16 rows per wave and stage
            const int sb0 = W8 ? kh * KS : 0;                    // first k step of this wave
            frag_t a[2][NF], b[2][NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) a[0][i] = t2_frag<T>(ta, (wm * NF + i) * 32, sb0, lane);
#pragma unroll
            for (int j = 0; j < NF; ++j) b[0][j] = t2_frag<T>(tb, (wn * NF + j) * 32, sb0, lane);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
#pragma unroll
                    for (int i = 0; i < NF; ++i) a[(s + 1) & 1][i] = t2_frag<T>(ta, (wm * NF + i) * 32, sb0 + s + 1, lane);
#pragma unroll
                    for (int j = 0; j < NF; ++j) b[(s + 1) & 1][j] = t2_frag<T>(tb, (wn * NF + j) * 32, sb0 + s + 1, lane);
                }
                // product MFMAs of this k step, one at a time (t): split bf16 - fragments 2 l / 2 l + 1 = hi / lo part of logical group l,
                // terms a_lo b_hi, a_hi b_lo, a_hi b_hi over the 2 x 2 tiles; plain types - the 2 x 2 tiles
                constexpr int MPK = SPLIT ? 3 * NL * NL : NL * NL;
                constexpr int NM = KS * MPK, ND = 2 * NI;
#pragma unroll
                for (int t = 0; t < MPK; ++t) {
                    const int term = t / (NL * NL), i = (t % (NL * NL)) / NL, j = t % NL;
                    if constexpr (SPLIT)
                        acc[i][j] = MmaTraits<T>::mma(a[s & 1][2 * i + (term == 0 ? 1 : 0)], b[s & 1][2 * j + (term == 1 ? 1 : 0)], acc[i][j]);
                    else
                        acc[i][j] = MmaTraits<T>::mma(a[s & 1][i], b[s & 1][j], acc[i][j]);
                    if constexpr (IL) {
                        const int m = s * MPK + t;
                        if ((m + 1) * ND / NM != m * ND / NM) dma_one(st + 2, nslot, m * ND / NM);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (WITH_CS) {
                    {   // this wave's 32-row block (wn = 0 / 1), selected without a branch
                        if constexpr (SPLIT) {
                            const frag_t ch = wn ? a[s & 1][2] : a[s & 1][0], cl = wn ? a[s & 1][3] : a[s & 1][1];
                            bacc[0] = MmaTraits<T>::mma(ch, ones, bacc[0]);
                            bacc[0] = MmaTraits<T>::mma(cl, ones, bacc[0]);
                        } else {
                            const frag_t ch = wn ? a[s & 1][1] : a[s & 1][0];
                            bacc[0] = MmaTraits<T>::mma(ch, ones, bacc[0]);
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot are done before it reaches the next barrier
            slot = slot == T2_NS - 1 ? 0 : slot + 1;
        }
        if constexpr (IL) t2_wait_vm<0>();                       // the refills behind the end have landed too (the ring is reused below)
    };
    if constexpr (SPLIT && W8 && IL) {
        if (do_cs) main_loop_pipe(std::true_type{});
        else main_loop_pipe(std::false_type{});
    } else {
        if (do_cs) main_loop(std::true_type{});
        else main_loop(std::false_type{});
    }
    const int nw = n0 / EP + wm * 64, kw = k0 / EP + wn * 64;   // LOGICAL origin of this wave's 64 x 64 tile
    float* out = (float*)(second ? p.out1 : p.out0);
    const long ldo_ = second ? p.ldo1 : p.ldo0;
    if constexpr (W8) {
        // the second k half hands its 64 x 64 quadrant to the first through LDS (4 quadrants x 4 tiles x 16 registers x 64 lanes x 4 B = 64 KB
        // of the ring: every wave is past its last fragment read and every LDS-DMA has landed)
        __syncthreads();
        float* xch = (float*)lds + (size_t)wq * (NL * NL * 16 * 64) + lane;
        float* bxch = (float*)lds + (size_t)4 * (NL * NL * 16 * 64) + (size_t)wq * (16 * 64) + lane;     // (behind the four quadrants: the bias column sums)
        if (CS && do_cs && kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bxch[r * 64] = bacc[0][r];
        }
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
#pragma unroll
                for (int j = 0; j < NL; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((i * NL + j) * 16 + r) * 64] = acc[i][j][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
#pragma unroll
                for (int j = 0; j < NL; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += xch[((i * NL + j) * 16 + r) * 64];
            if (CS && do_cs) {
#pragma unroll
                for (int r = 0; r < 16; ++r) bacc[0][r] += bxch[r * 64];
            }
        }
    }
    if (!W8 || kh == 0) {
        if (p.cpart) {
            // split partials as PLAIN stores into scratch [split][N][K] (a register of a 32 x 32 accumulator = two 128-byte row segments: full store
            // rate), summed into out0 by tn_reduce_kernel: the float atomics of this tile (64 KB per workgroup, 16.5 MB per launch at the
            // ~1.3 TB/s atomics run at) were 10 - 13 us at the end of every launch, with nothing left to overlap them
            // (paired launch: a split's region holds the first gradient's [N][K] tile rows, then the second's [N2][K])
            float* part = p.cpart + (long)split * (p.N + p.res_mod) * p.K + (second ? (long)p.N * p.K : 0);
#pragma unroll
            for (int i = 0; i < NL; ++i)
#pragma unroll
                for (int j = 0; j < NL; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = nw + i * 32 + acc_row(r, lane);
                        const int k = kw + j * 32 + (lane & 31);
#if MFVIT_TNPART_PLAIN        // A/B builds only: plain stores for the split partials (the reduce launch reads them back within microseconds)
                        part[(long)n * p.K + k] = acc[i][j][r];
#else
                        __builtin_nontemporal_store(acc[i][j][r], part + (long)n * p.K + k);
#endif
                    }
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i)
#pragma unroll
                for (int j = 0; j < NL; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = nw + i * 32 + acc_row(r, lane);
                        const int k = kw + j * 32 + (lane & 31);
                        atomicAdd(out + (long)n * ldo_ + k, acc[i][j][r]);
                    }
        }
    }
    if (CS) {
        // (the two k halves of a workgroup were added above: ONE value per split and row - into the split's row of the bias partials when the launch has
        // them (p.kpart: [split][N] behind the tile partials, summed in a fixed order by the same reduce launch as the tiles: the bias gradient is the same bits
        // on every run, round 6), else one float atomic)
        if (do_cs && (!W8 || kh == 0) && (lane & 31) == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nw + wn * 32 + acc_row(r, lane);
                if (p.kpart) p.kpart[(long)split * p.N + n] = bacc[0][r];
                else atomicAdd(p.cs0 + n, bacc[0][r]);
            }
        }
    }
}

}  // namespace

bool gemm_tn_glds_supported(int dtype, const GemmP& p) {
    static int sw_on = INT_MIN;                                 // MFVIT_TN_GLDS=0 keeps gemm_tn (gemm.hip: f32, small M, odd shapes) reachable at every M for its parity tests
    if (env_switch("MFVIT_TN_GLDS", 1, sw_on) == 0 || (dtype != MFVIT_BF16 && dtype != MFVIT_BF16X3 && dtype != MFVIT_F16) || p.nb > 1) return false;
    if (p.orow_in && (dtype != MFVIT_BF16X3 || p.cs0 || p.res_mod)) return false;    // remapped A rows: the split-bf16 form without bias sums only
    // (M >= 2,048 since round 5: at B = 16 (M = 3,152) the LDS-DMA kernel with the paired dWqkv + dWproj launch is ~3 % of the step ahead of gemm_tn)
    if (p.N % 128 || p.K % 128 || p.M < 2048) return false;     // 128 x 128 LOGICAL tiles in every mode
    if (p.lda % 8 || p.ldw % 8) return false;
    return true;
}

// may `b` ride along with `a` in one launch?  (same reduction rows and K, no bias sums / partial scratch / batch on b)
bool gemm_tn_pair_supported(int dtype, const GemmP& a, const GemmP& b) {
    // (policy - when to pair - lives with the caller, vit.hip.  Measured inside the step, profiles/r03_wgrad_ab.txt: the weight-gradient class 7.77 -> 7.32 ms per
    // step in the serialized pass, as the saved atomics predict; with the weight gradients on a side stream the TIMED step loses 0.1 ms - dWproj alone was a short
    // kernel that filled gaps beside the data-gradient chain - on the caller's stream it gains 0.13 ms.)
    if (!gemm_tn_glds_supported(dtype, a) || !gemm_tn_glds_supported(dtype, b)) return false;
    if (a.M != b.M || a.K != b.K || b.cs0 || b.cpart || a.splits > 0 || b.splits > 0 || a.res_mod || b.res_mod) return false;
    return true;
}

template <typename T> static int launch_t2(GemmP p, hipStream_t st) {
    typedef T2Geo<T> G;
    const int tiles = ((p.N + p.res_mod) / 128) * (p.K / 128);          // res_mod = N of the paired second GEMM (0: none)
    if (p.splits <= 0) {
        static const int target_env = [] { const char* e = getenv("MFVIT_TN2_TARGET"); return e ? atoi(e) : 0; }();
        // one workgroup per CU (96 KB of LDS): tiles x splits <= 256; with a second kernel stream beside this one (stream_share(), common.cuh) and a
        // reduction short enough that the splits are latency (M < 8,192 rows) half of that, so that the other encoder's launch fits beside it
        const int target = target_env > 0 ? target_env : (stream_share() >= 2 && p.M < 8192 ? 128 : 256);
        int s = target / tiles;
        const int maxs = (p.M + 4 * G::KR - 1) / (4 * G::KR);
        p.splits = s < 1 ? 1 : (s > maxs ? maxs : s);
    }
    {   // no empty split (the kernel derives the same chunk from p.splits): every split writes its partial tile
        int chunk = (p.M + p.splits - 1) / p.splits;
        chunk = (chunk + G::KR - 1) / G::KR * G::KR;
        p.splits = (p.M + chunk - 1) / chunk;
    }
    // p.cpart (optional scratch of >= 384 tiles of 128 x 128 floats): split partials as plain stores + one reduce pass
    if (p.splits < 2 || p.ldo0 % 4 || (p.res_mod && p.ldo1 % 4) || (long)tiles * p.splits > 384) p.cpart = nullptr;
    // ... and the bias column sums of the splits as [split][N] behind them (at most two more tile units: splits x N <= 256 x 128 floats)
    p.kpart = p.cpart && p.cs0 && (long)tiles * p.splits + 2 <= 384 && p.N % 4 == 0 && (size_t)p.cs0 % 16 == 0 ? p.cpart + (long)p.splits * (p.N + p.res_mod) * p.K : nullptr;
    constexpr int bytes = (is_split<T>::value ? T2_NSP : T2_NS) * T2_STAGE;   // (the pipelined form's ring; the other split forms use three of the four slots)
    p.rows_per_wg = 0;
    // (eight waves per workgroup and the LDS-DMA issue interleaved with the MFMAs - the W8 / IL template flags of the kernel - are the only forms
    // launched since round 5; their four-wave / burst-issue alternatives and measurements: DESIGN.md 5, round 3)
    ProfScope ps(PROF_GEMM_TN, 2.0 * p.M * (p.N + p.res_mod) * p.K, 0, st);
    static int sw2 = INT_MIN;
    const bool two = is_split<T>::value && !p.orow_in && env_switch("MFVIT_WGRAD_TERMS", 3, sw2) == 2;
    auto go = [&](auto cs, auto w, auto i) {
        constexpr bool CS = decltype(cs)::value, W8 = decltype(w)::value, IL = decltype(i)::value;
        if constexpr (is_split<T>::value && W8 && IL) {
            if (two) {
                static PerDeviceOnce attr2;
                if (attr2.first())
                    (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<T, CS, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                MFVIT_LAUNCH((gemm_tn_glds_kernel<T, CS, true, true, false, true>), dim3(tiles * p.splits), dim3(512), bytes, st, p);
                return;
            }
        }
        static PerDeviceOnce attr;
        if (attr.first()) {
            (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<T, CS, W8, IL>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        }
        MFVIT_LAUNCH((gemm_tn_glds_kernel<T, CS, W8, IL>), dim3(tiles * p.splits), dim3(W8 ? 512 : 256), bytes, st, p);
    };
    auto go1 = [&](auto cs) { go(cs, std::true_type{}, std::true_type{}); };
    if (p.orow_in) {
        if constexpr (is_split<T>::value) {
            static PerDeviceOnce attr_rm;
            if (attr_rm.first())
                (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<T, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            MFVIT_LAUNCH((gemm_tn_glds_kernel<T, false, true, true, true>), dim3(tiles * p.splits), dim3(512), bytes, st, p);
        } else {
            return MFVIT_EINVAL;
        }
    } else if (p.cs0) go1(std::true_type{});
    else go1(std::false_type{});
    MFVIT_CHECK_LAUNCH();
    if (p.cpart) {
        const long stride = (long)(p.N + p.res_mod) * p.K;
        int rc = tn_partial_reduce(p.cpart, p.splits, stride, p.N, p.K, (float*)p.out0, p.ldo0, st);
        if (rc == MFVIT_OK && p.kpart) rc = tn_partial_reduce(p.kpart, p.splits, p.N, 1, p.N, p.cs0, p.N, st);
        if (rc != MFVIT_OK || !p.res_mod) return rc;
        return tn_partial_reduce(p.cpart + (long)p.N * p.K, p.splits, stride, p.res_mod, p.K, (float*)p.out1, p.ldo1, st);
    }
    return MFVIT_OK;
}

// a and b in one launch (gemm_tn_pair_supported)
int gemm_tn_glds_pair(int dtype, GemmP a, const GemmP& b, hipStream_t st) {
    a.aux = b.A; a.ldaux = b.lda;
    a.res_t = b.W; a.ldres_t = b.ldw;
    a.out1 = b.out0; a.ldo1 = b.ldo0;
    a.res_mod = b.N;
    return gemm_tn_glds(dtype, a, st);
}

int gemm_tn_glds(int dtype, GemmP p, hipStream_t st) {
    if (dtype == MFVIT_BF16) return launch_t2<bf16>(p, st);
    if (dtype == MFVIT_BF16X3) return launch_t2<sbf16>(p, st);
    if (dtype == MFVIT_F16) return launch_t2<f16>(p, st);
    return MFVIT_EINVAL;
}

}  // namespace mfvit
