// Tile GEMM, round 6:  C[M][N] = epi(A[M][K] W[N][K]^T)  for the 16-bit operand types (split bf16, bf16, fp16) at large M - the Linear forward / data-gradient
// products of the ViT block that gemm.hip::gemm_nt_tile_kernel served alone until round 5 (timm Block: qkv, fc1 + GELU, fc2 data gradient x gelu', proj data gradient;
// call sites of the reference: moco_pretraining/moco/model/crossvit_2vits_..._std002_sum.py:128-135 through `vits.vit_small`).
//
// Why a second tile kernel.  The round-5 ablation of the 128 x 128 / two-workgroups-per-CU structure (profiles/r05_gemm_format_prototype.txt) found its three phases -
// matrix work, operand staging through registers into LDS, output stores - ADDING instead of overlapping, and every scheduling experiment on it moved nothing: the
// launches sit at the package power cap, where time follows ENERGY, not stalls.  This kernel removes work and pairs what is left (prototype against the old structure on
// one box: 120 -> 88 - 92 us for the fc1 shape, profiles/r06_pp_proto_*.txt):
//   * ONE persistent 512-thread workgroup per CU walks 256 x 128 output tiles (3/4 of the L2 -> LDS bytes of 128 x 128 tiles: the W tile is shared by both halves).
//     Waves 0-3 (group 0) own rows 0..127 of the tile, waves 4-7 (group 1) rows 128..255, each wave a 64 x 64 quadrant; every SIMD hosts one wave of each group.
//   * operands go global -> LDS by LDS-DMA (no VGPR round trip, no ds_write_b128 - the 13-cycle-per-instruction path that set the old loop's pace) into a 3-slot ring of
//     whole 128-byte row pieces (one 32-wide k group of a split tensor, 64 k of a plain one): A 32 KB + W 16 KB per stage, two stages in flight, counted vmcnt.  The 16-byte
//     chunks of LDS row r sit XOR-swizzled by (r >> 1) & 7 - applied on the SOURCE address of the DMA and on the fragment reads (conflict-free ds_read_b128).
//   * the two groups run HALF A STAGE apart ("ping-pong", MI355X_MICROARCH.md 'Two waves per SIMD'): while one group issues the MFMAs of a stage,
//     the other reads its 16 fragments of the next stage and issues its 6 LDS-DMA pieces (v_mfma_f32_16x16x32: 48 / 32 per stage at 16 cycles); two s_barrier per stage separate the phases.
//   * the stage stream runs ACROSS tile boundaries: the next tile's first two stages are in flight during the epilogue, and the epilogue's stores stay in flight behind
//     counted waits (they are older than the stage issued after them, so the first load phase of the next tile waits for `its stage + the stores` only).
//   * the epilogue needs no workgroup barrier INSIDE (two extra ones around it line the groups up, so that both run it at the same time): every wave turns its accumulators into whole 128-byte output lines through a
//     PRIVATE 2 KB LDS patch - 16 rows at a time - and stores 16 bytes per lane; per-element math as in gemm.hip (bias | bias + GELU / ReLU with the saved derivative |
//     x act' + column sums | none | bias with split-fp16 output for the attention core).
// Synchronisation (stage g, ring slot g % 3; group 0's load phase L(g) and compute phase C(g) are the barrier intervals 2g+1 and 2g+2, group 1's 2g+2 and 2g+3):
//   RAW  a wave waits for ITS pieces of stage g+1 (vmcnt) at the end of L(g) and then passes a barrier; stage g+1 is first read in L(g+1), i.e. after every wave of both
//        groups has waited and passed a barrier.
//   WAR  L(g) refills the slot of stage g-1.  Its fragment reads were issued in L(g-1) and RETIRED (lgkmcnt(0)) before the barrier that ends L(g-1) - for group 1 that is
//        interval 2g, over before group 0's L(g) starts.
#include "kernels.h"
#include "prof.h"

#include <stdlib.h>

#include <type_traits>

namespace mfvit {

namespace {

constexpr int PP_BM = 256, PP_BN = 128;
constexpr int PP_SROW = 128;                                   // bytes per row and stage
constexpr int PP_A = PP_BM * PP_SROW, PP_W = PP_BN * PP_SROW;  // 32 KB + 16 KB
constexpr int PP_SLOT = PP_A + PP_W;
constexpr int PP_NSLOT = 3, PP_AHEAD = 2;
constexpr int PP_RING = PP_NSLOT * PP_SLOT;                    // 144 KB
constexpr int PP_PATCH = 16 * 128;                             // per-wave epilogue patch: 16 rows of 128 B
constexpr int PP_LDS = PP_RING + 8 * PP_PATCH;                 // 160 KB: the whole LDS of a CU
constexpr int PP_NDA = 4, PP_NDW = 2, PP_NDMA = PP_NDA + PP_NDW;   // LDS-DMA instructions (1 KB = 8 rows) per wave and stage
#ifndef PP_NC_PLAIN
#define PP_NC_PLAIN 0                                           // plain 16-bit types: pieces of a stage issued inside the compute phase (A/B builds; 3 measured: no change)
#endif

template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// 16 bytes per lane global -> LDS: source = sbase + voff (per lane), destination = LDS byte address lds_dst + 16 * lane (M0 carries the base, saved and restored)
__device__ __forceinline__ void pp_glds16(unsigned voff, const char* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// Epilogue operands (bias, the saved activation derivative) are fetched by inline-asm loads issued in the LAST stage's load phase and waited for by hand: a load the
// compiler knows about would get ITS wait - a vmcnt(0) that also drains the LDS-DMA stages in flight (the compiler cannot count asm loads).  The destination registers
// must not be touched before the wait: the issuing block is straight-line code up to pp_landed() (no branch, hence no phi copy, in between).
typedef unsigned int pp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pp_gload16(pp_u32x4& dst, const char* sbase, unsigned voff) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory"); }
__device__ __forceinline__ void pp_gload4(float& dst, const char* sbase, unsigned voff) { asm volatile("global_load_dword %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory"); }
__device__ __forceinline__ void pp_landed(pp_u32x4& r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ void pp_landed(float& r) { asm volatile("" : "+v"(r)); }

// v_mfma_f32_16x16x32 (one 128-byte row piece of a split tensor = one K step): under the package power cap this shape sustains 12 - 15 % more FLOP/s than
// 32x32x16 at the same cycles per FLOP (MI355X_MICROARCH.md 'DVFS give-back' (7); on this loop: 87.2 -> 83.3 us for the fc1 shape, profiles/r06_pp_proto_v3_mfma_shape.txt)
template <typename T> struct PpMma;
template <> struct PpMma<sbf16> { static __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };
template <> struct PpMma<bf16> { static __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };
template <> struct PpMma<f16> { static __device__ __forceinline__ f32x4 mma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };

// four f32 -> four 16-bit elements (8 bytes); split tensors: the hi parts and the lo parts.  TWO-element conversions (one packed instruction each: v_cvt_pk_bf16_f32 /
// v_cvt_pk_f16_f32) - a four-element convertvector is lowered element by element (128 instead of 64 conversions per wave and tile in the first build)
template <typename E> __device__ __forceinline__ unsigned pp_pk2(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef E e2 __attribute__((ext_vector_type(2)));
    const f2 x = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, e2));
}
template <typename E> __device__ __forceinline__ uint2 pp_pack4(f32x4 v) { return make_uint2(pp_pk2<E>(v[0], v[1]), pp_pk2<E>(v[2], v[3])); }
template <typename E> __device__ __forceinline__ void pp_split4(f32x4 v, uint2& hi, uint2& lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef E e2 __attribute__((ext_vector_type(2)));
    const f2 x0 = {v[0], v[1]}, x1 = {v[2], v[3]};
    const e2 h0 = __builtin_convertvector(x0, e2), h1 = __builtin_convertvector(x1, e2);
    const e2 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f2), e2), l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f2), e2);
    hi = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    lo = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}

// T = sbf16 | bf16 | f16
template <typename T, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_nt_pp_kernel(GemmP p, int ntiles, int ntall) {
    constexpr bool SPLIT = is_split<T>::value;
    constexpr int NMF = SPLIT ? 48 : 32;                       // MFMAs per wave and stage (16 cycles each)
    constexpr int KPS = SPLIT ? 32 : 64;                       // logical k per stage
    typedef typename Vec8<T>::type frag_t;
    typedef typename Vec4<T>::elem E16;
    typedef typename act_grad_type<T>::type AX;                // saved activation derivative: fp16 for split tensors, T otherwise (2 bytes either way)
    typedef pp_u32x4 u32x4;
    constexpr bool ACT = EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU;
    constexpr bool HAS_BIAS = EPI == EPI_BIAS || ACT || EPI == EPI_BIAS_X3F16;
    // Accumulator layout.  SWAP (every epilogue but x act'): MFMA operands swapped (W fragment first), so a lane owns ONE row (lane & 15) of a 16 x 16 tile and FOUR
    // CONSECUTIVE columns 4 (lane >> 4) .. + 3 - the bias enters as the accumulators' initial value (4 registers per column tile, fetched once per tile), the 16-bit
    // outputs leave the lane as 8-byte pieces (one ds_write_b64 per part instead of eight ds_write_b16).  Not swapped (x act' + column sums): a lane owns one COLUMN
    // and rows 4 (lane >> 4) .. + 3, so the column sums are in-lane adds + two shuffles.
    constexpr bool SWAP = EPI != EPI_GELU_BWD;
    // stores of one wave's epilogue that the counted waits of the next tile's first load phase ASSUME (an undercount is safe: the wait only gets stricter): those of a
    // SHORT tile (three row fragments per wave, below) - a tall one has a third more, the oldest of which have long left by then
    constexpr int NSTORE = SPLIT ? 12 : 6;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3, wm = wq >> 1, wn = wq & 1;
    const int ntn = p.N / PP_BN;
    const int nst = p.K / KPS;                                 // stages per tile
    const unsigned lda_b = (unsigned)p.lda * 2u, ldw_b = (unsigned)p.ldw * 2u;      // bytes per operand row
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    const int G = gridDim.x;
    const int my = xcd_remap(blockIdx.x, G);                   // each XCD walks a contiguous run of the tile order (n fastest: its workgroups share the A rows in L2)
    const int ntl = my < ntiles ? (ntiles - my + G - 1) / G : 0;
    if (ntl == 0) return;
    // MIXED tile heights (round 6, like gemm_rowp's): the first `ntall` row tiles are 256 rows (four 16-row fragments per wave), the others 192 (three) - the launcher
    // picks the mix that evens out the persistent workgroups' loads (qkv at M = 25,216: 891 tall tiles are 3.5 rounds of 256 workgroups, i.e. four; 57 tall + 56 short row
    // tiles are 14 of 16 units for every workgroup).  A short tile keeps the LDS layout of a tall one - each wave's 64-row block holds its 48 rows, the last 16 rows of the
    // block repeat row 47 (same number of LDS-DMA pieces: the counted waits do not change) - and skips the fourth fragment's reads, MFMAs and stores.
    auto tile_rt = [&](int tile) { return tile / ntn; };
    auto tile_m0 = [&](int rt) { return rt < ntall ? rt * PP_BM : ntall * PP_BM + (rt - ntall) * (3 * PP_BM / 4); };
    const char* gA = (const char*)p.A;
    const char* gW = (const char*)p.W;

    // ---- LDS-DMA addressing: piece q of an operand = rows 8 q .. 8 q + 7; A pieces wave + 8 i, W pieces wave + 8 i.  Lane l: row l >> 3 of the piece, LDS chunk l & 7,
    // which holds source chunk (l & 7) ^ swizzle(row)
    unsigned va[PP_NDA], vw[PP_NDW];
    int ld_t = 0, ld_s = 0, ld_slot = 0;                       // position of the DMA stream: tile (of this workgroup), stage, ring slot
    auto set_tile_offsets = [&](int t) {
        int ln = tid & 63;                                     // (laundered: the per-lane terms are recomputed once per tile instead of living in registers across the loops)
        asm volatile("" : "+v"(ln));
        const int tile = my + t * G;
        const int rt = tile_rt(tile), m0 = tile_m0(rt), n0 = (tile % ntn) * PP_BN;
        const bool tall = rt < ntall;
#pragma unroll
        for (int i = 0; i < PP_NDA; ++i) {
            const int row = 8 * (wave + 8 * i) + (ln >> 3);
            const int rb = row & 63;
            int gm = tall ? m0 + row : m0 + (row >> 6) * 48 + (rb < 48 ? rb : 47);      // (short tile: 48 rows per 64-row block, its tail repeats the block's last row)
            gm = gm < p.M ? gm : p.M - 1;                      // rows past the end replicate row M - 1 (finite data; their outputs are identical duplicates)
            va[i] = (unsigned)gm * lda_b + 16u * (unsigned)((ln & 7) ^ ((row >> 1) & 7));
        }
#pragma unroll
        for (int i = 0; i < PP_NDW; ++i) {
            const int row = 8 * (wave + 8 * i) + (ln >> 3);
            vw[i] = (unsigned)(n0 + row) * ldw_b + 16u * (unsigned)((ln & 7) ^ ((row >> 1) & 7));
        }
    };
    set_tile_offsets(0);
    // the six pieces of a stage: d = 0 .. 3 A, 4 .. 5 W.  NC of them can go out inside the COMPUTE phase, between its MFMAs (A/B builds -DPP_NC_PLAIN=n, plain 16-bit
    // types: a stage has a third fewer MFMA cycles than a split one and the load phase with all six pieces is the longer one - three pieces moved changed NOTHING in the
    // 16-bit configs: tile class 26.3 / 39.1 / 40.7 us either way, profiles/r06_pp_step_ab.txt 5; the split prototype had said the same)
    constexpr int NC = PP_NC_PLAIN > 0 && !SPLIT ? PP_NC_PLAIN : 0;
    unsigned dma_soff = 0, dma_dst = 0;
    auto dma_piece = [&](int d) __attribute__((always_inline)) {
        if (d < PP_NDA) pp_glds16(va[d], gA + dma_soff, dma_dst + d * 8 * 1024);
        else pp_glds16(vw[d - PP_NDA], gW + dma_soff, dma_dst + PP_A + (d - PP_NDA) * 8 * 1024);
    };
    auto dma_begin = [&]() __attribute__((always_inline)) {      // the pieces of the load phase
        dma_soff = (unsigned)ld_s * PP_SROW;
        dma_dst = lbase + (unsigned)ld_slot * PP_SLOT + (unsigned)wave * 1024u;
#pragma unroll
        for (int d = 0; d < PP_NDMA - NC; ++d) dma_piece(d);
    };
    auto dma_end = [&]() __attribute__((always_inline)) {
        ld_slot = ld_slot == PP_NSLOT - 1 ? 0 : ld_slot + 1;
        if (++ld_s == nst) {           // next tile (past the end: the last tile again - harmless refills of free slots, and the waits keep their counts)
            ld_s = 0;
            if (ld_t + 1 < ntl) { ++ld_t; set_tile_offsets(ld_t); }
        }
    };
    auto dma_stage = [&]() {           // a whole stage at once (prologue)
        dma_begin();
#pragma unroll
        for (int d = PP_NDMA - NC; d < PP_NDMA; ++d) dma_piece(d);
        dma_end();
    };
    // fragment addresses inside a slot (constant per lane): 16-row tile i of an operand, lane = (row lane & 15, 16-byte chunk lane >> 4); chunk + 4 (address ^ 64) is
    // the lo part of a split tensor / the second K step of a plain one
    unsigned fa[4], fw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = grp * 128 + wm * 64 + i * 16 + l15, rw = wn * 64 + i * 16 + l15;
        fa[i] = (unsigned)(ra * PP_SROW + 16 * (q4 ^ ((ra >> 1) & 7)));
        fw[i] = (unsigned)(PP_A + rw * PP_SROW + 16 * (q4 ^ ((rw >> 1) & 7)));
    }
    char* patch = lds + PP_RING + wave * PP_PATCH;

    // epilogue operands fetched by asm loads (E1 per lane and tile): the NEXT tile's bias (SWAP: 4 columns per column tile, the accumulators' initial value) in the
    // last load phase of a tile, or this tile's act' rows
    constexpr int E1 = EPI == EPI_GELU_BWD ? 8 : (HAS_BIAS ? 4 : 0);
    constexpr int KEEP = PP_NDMA * (PP_AHEAD - 1) - NC;         // pieces of the stage just (partly) issued that may stay in flight at the end of a load phase
    u32x4 bn[4];                                               // bias of the tile about to start, as loaded
    u32x4 ax[4][2];
    auto bias_loads = [&](int t) __attribute__((always_inline)) {
        const int tile = my + (t < ntl ? t : ntl - 1) * G;
        const unsigned nq = (unsigned)((tile % ntn) * PP_BN + wn * 64);
        const char* bsrc = p.bias ? (const char*)p.bias : (const char*)p.W;     // (no bias: loads from valid memory whose results are discarded - the counts stay fixed)
        int lq = tid >> 4 & 3;
        asm volatile("" : "+v"(lq));
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) pp_gload16(bn[jt], bsrc, (nq + jt * 16 + 4 * lq) * 4u);
    };
    if constexpr (SWAP && HAS_BIAS) bias_loads(0);
#pragma unroll
    for (int d = 0; d < PP_AHEAD; ++d) dma_stage();
    pp_wait_vm<PP_NDMA*(PP_AHEAD - 1)>();
    if constexpr (SWAP && HAS_BIAS) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) pp_landed(bn[jt]);
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                // group 1 runs half a stage behind
    int slot = 0;
    frag_t af[2][4], bf[2][4];                                 // [part / K step][16-row tile]
    for (int t = 0; t < ntl; ++t) {
      const int tile = my + t * G;
      // one tile with MI row fragments per wave (4: 256 rows, 3: 192): main loop + epilogue, instantiated for both heights
      auto run_tile = [&](auto mi_c) __attribute__((always_inline)) {
        constexpr int MI = decltype(mi_c)::value;
        constexpr int NMFI = NMF / 4 * MI;                     // MFMAs per wave and stage of this height
        f32x4 acc[4][4];                                       // [row tile][column tile]
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SWAP && HAS_BIAS) acc[i][j] = p.bias ? __builtin_bit_cast(f32x4, bn[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
                else acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        const int mw = tile_m0(tile_rt(tile)) + (grp * 2 + wm) * (16 * MI);     // first row / first LOGICAL column of this wave's (16 MI) x 64 quadrant
        const int nw = (tile % ntn) * PP_BN + wn * 64;
        auto stage = [&](auto last_c, bool behind_epilogue) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(last_c)::value;
            // ---- load phase (the other group computes)
            const char* sl = lds + slot * PP_SLOT;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < MI) af[c][i] = *(const frag_t*)(sl + (fa[i] ^ (64u * c)));
                    bf[c][i] = *(const frag_t*)(sl + (fw[i] ^ (64u * c)));
                }
            if constexpr (LAST && EPI == EPI_GELU_BWD) {
                // aux = act'(pre-activation) saved by the forward, [M][N] of AX: per 16-row tile 64 columns = 128 B per row, 16 bytes per lane
                int ln = tid & 63;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        int m = mw + (u < MI ? u : MI - 1) * 16 + 8 * q + (ln >> 3);      // (a short tile repeats its last fragment's rows: the counts stay fixed)
                        m = m < p.M ? m : p.M - 1;
                        pp_gload16(ax[u][q], (const char*)p.aux, (unsigned)m * (unsigned)(p.ldaux * (long)sizeof(AX)) + (unsigned)(nw * (int)sizeof(AX) + 16 * (ln & 7)));
                    }
            } else if constexpr (LAST && SWAP && HAS_BIAS) {
                bias_loads(t + 1);
            }
            dma_begin();
            // the NEXT stage must have landed before the next load phase.  This wave's queue, oldest first: [stage + 1][epilogue operands, last stage only][stage + 2, just
            // issued] - and behind an epilogue its stores sit between the two stages: counted, so that they stay in flight until the stage issued after them is needed
            if constexpr (LAST) pp_wait_vm<KEEP + E1>();
            else if (behind_epilogue) pp_wait_vm<KEEP + NSTORE>();
            else pp_wait_vm<KEEP>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads are retired: the slot may be refilled once the barrier is passed
            __builtin_amdgcn_s_barrier();
            // ---- compute phase: split tensors  acc += a_lo w_hi + a_hi w_lo + a_hi w_hi  per 16 x 16 tile; plain ones the two K steps
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int tt = 0; tt < NMFI; ++tt) {
                const int term = tt / (4 * MI), i = (tt >> 2) % MI, j = tt & 3;
                const frag_t a = af[SPLIT ? (term == 0 ? 1 : 0) : term][i], w = bf[SPLIT ? (term == 1 ? 1 : 0) : term][j];
                acc[i][j] = SWAP ? PpMma<T>::mma(w, a, acc[i][j]) : PpMma<T>::mma(a, w, acc[i][j]);
                if constexpr (NC > 0) {
                    constexpr int GAP = NMFI / (NC + 1);
                    if ((tt + 1) % GAP == 0 && (tt + 1) / GAP <= NC) { dma_piece(PP_NDMA - NC + (tt + 1) / GAP - 1); __builtin_amdgcn_sched_barrier(0); }
                }
            }
            dma_end();
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            slot = slot == PP_NSLOT - 1 ? 0 : slot + 1;
        };
        for (int s = 0; s < nst - 1; ++s) stage(std::false_type(), t > 0 && s == 0);
        stage(std::true_type(), false);                        // (nst >= 2: the last stage is never the first behind an epilogue)
        // both groups run their epilogues at the SAME time (two waves per SIMD share the vector pipe: the epilogue is bound by vector issue, and one wave alone leaves
        // most issue slots empty) instead of one behind the other: group 0 waits here for group 1's last compute phase, group 1 waits behind its epilogue for group 0's
        // next load phase.  Same box, in the step: tile class 82.3 -> 77.0 us (profiles/r06_pp_step_ab.txt)
        if (grp == 0) __builtin_amdgcn_s_barrier();

        // ---- epilogue of this wave's 64 x 64 quadrant (no workgroup barrier inside)
        if constexpr (E1 > 0) {                                 // the epilogue operands have landed (the stage issued behind them stays in flight)
            pp_wait_vm<PP_NDMA*(PP_AHEAD - 1)>();
            if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { pp_landed(ax[u][0]); pp_landed(ax[u][1]); }
            } else {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) pp_landed(bn[jt]);
            }
        }
        // the 16 x 128-byte patch -> global rows row0 .. row0 + 15 (clamped: duplicates of row M - 1 carry identical bytes), 16 bytes per lane.  No wait between the patch
        // writes and these reads, nor before the next unit's writes: the LDS executes one wave's operations in order (the compiler waits for the read DATA only)
        auto flush = [&](char* gcol, long ld_bytes, int row0, auto stream_c) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = 8 * q + (lane >> 3);
                // (SWAP: the 16-byte pieces of patch row r sit XOR-swizzled by r & 7 - the lanes of an 8-byte patch write are 16 ROWS of one column slot, which
                // without the swizzle all land on the same two banks: a 16-way conflict that cost the fc1 epilogue 18 us per launch)
                const u32x4 v = *(const u32x4*)(patch + row * 128 + 16 * (SWAP ? (lane & 7) ^ (lane >> 3) : (lane & 7)));
                int m = row0 + row;
                m = m < p.M ? m : p.M - 1;
                u32x4* gp = (u32x4*)(gcol + (long)m * ld_bytes + 16 * (lane & 7));
                if constexpr (decltype(stream_c)::value) store16_stream(gp, v);      // system-scope streaming store: no write-allocate fetch (common.cuh)
                else *gp = v;                                                      // qkv: the attention core reads it next - keep the lines in cache (gemm.hip, round 5)
            }
        };
        void* out = ACT ? p.out1 : p.out0;
        const long ldo_b = (ACT ? p.ldo1 : p.ldo0) * (long)sizeof(T);
        constexpr bool STREAM = !(EPI == EPI_BIAS || EPI == EPI_BIAS_X3F16);
        if constexpr (SWAP) {
            // this lane's row of the patch; 8-byte slot s (= 4 x column tile + lane >> 4) of the row sits at s ^ 2 (row & 7): pieces of 16 bytes stay whole
            char* prow_ = patch + l15 * 128;
            auto pslot = [&](int s4) -> uint2* { return (uint2*)(prow_ + 8 * ((4 * s4 + q4) ^ (2 * (l15 & 7)))); };
            const bool want_grad = p.out0 != nullptr;           // GELU / ReLU epilogues: no-grad forwards pass out0 = NULL and skip the derivative
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row0 = mw + i * 16;
                if constexpr (ACT) {
                    // out0 = act'(pre) in AX (the four column tiles of the quadrant: 64 x 2 B = one line per row), the activation replaces the accumulators
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 g, d;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[i][j][r];
                            if constexpr (EPI == EPI_BIAS_GELU) {
                                float gg, dd;
                                gelu_both_t<T>(v, gg, dd);      // one erf / exp evaluation for both outputs
                                g[r] = gg; d[r] = dd;
                            } else {                            // fuseattention.py:69: nn.ReLU
                                g[r] = fmaxf(v, 0.f); d[r] = v > 0.f ? 1.f : 0.f;
                            }
                        }
                        acc[i][j] = g;
                        if (want_grad) *pslot(j) = pp_pack4<AX>(d);
                    }
                    if (want_grad) flush((char*)p.out0 + (long)nw * (long)sizeof(AX), p.ldo0 * (long)sizeof(AX), row0, std::true_type());
                }
                if constexpr (SPLIT) {
                    // two column tiles = 32 columns = one 128-byte line per row: [hi x 32 | lo x 32]
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            uint2 hi, lo;
                            if constexpr (EPI == EPI_BIAS_X3F16) pp_split4<f16>(acc[i][2 * jp + jj], hi, lo);     // the attention core's operand format
                            else pp_split4<E16>(acc[i][2 * jp + jj], hi, lo);
                            *pslot(jj) = hi;
                            *pslot(2 + jj) = lo;
                        }
                        flush((char*)out + (long)(nw + jp * 32) * 4, ldo_b, row0, std::integral_constant<bool, STREAM>());
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *pslot(j) = pp_pack4<E16>(acc[i][j]);
                    flush((char*)out + (long)nw * 2, ldo_b, row0, std::integral_constant<bool, STREAM>());
                }
            }
        } else {
            // x act' + column sums: a lane owns column 16 j + (lane & 15) and rows 4 (lane >> 4) .. + 3 of every tile.  The aux rows (in registers since the last load
            // phase) go through the patch, every lane multiplies its accumulators in place; the column sums of the product (the bias gradient of the Linear in front)
            // leave as one row of partials per 64-row block of the tile (p.cpart: [4 x row tile + block][N], plain stores, added in a fixed order by colpart_reduce -
            // the same bits on every run, round 6) or, without a partial buffer, as float atomics
            float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int q = 0; q < 2; ++q) *(u32x4*)(patch + (8 * q + (lane >> 3)) * 128 + 16 * (lane & 7)) = ax[i][q];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[i][j][r] * load_elem<AX>((const AX*)(patch + (4 * q4 + r) * 128), j * 16 + l15);
                        acc[i][j][r] = v;
                        csum[j] += mw + i * 16 + 4 * q4 + r < p.M ? v : 0.f;
                    }
            }
            if (p.cs0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float c = csum[j] + __shfl_xor(csum[j], 16, 64);
                    c += __shfl_xor(c, 32, 64);
                    if (lane < 16) {
                        if (p.cpart) p.cpart[((long)(tile / ntn) * 4 + grp * 2 + wm) * p.N + nw + j * 16 + lane] = c;
                        else atomicAdd(p.cs0 + nw + j * 16 + lane, c);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row0 = mw + i * 16;
                if constexpr (SPLIT) {
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                            for (int r = 0; r < 4; r += 2)
                                store_elem_pair<T>((T*)(patch + (4 * q4 + r) * 128), (T*)(patch + (4 * q4 + r + 1) * 128), jj * 16 + l15, acc[i][2 * jp + jj][r],
                                                   acc[i][2 * jp + jj][r + 1]);
                        flush((char*)out + (long)(nw + jp * 32) * 4, ldo_b, row0, std::integral_constant<bool, STREAM>());
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; r += 2)
                            store_elem_pair<T>((T*)(patch + (4 * q4 + r) * 128), (T*)(patch + (4 * q4 + r + 1) * 128), j * 16 + l15, acc[i][j][r], acc[i][j][r + 1]);
                    flush((char*)out + (long)nw * 2, ldo_b, row0, std::integral_constant<bool, STREAM>());
                }
            }
        }
        if (grp == 1) __builtin_amdgcn_s_barrier();
      };
      if (tile_rt(tile) < ntall) run_tile(std::integral_constant<int, 4>());
      else run_tile(std::integral_constant<int, 3>());
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    pp_wait_vm<0>();
}

// The mix of tall (256-row) and short (192-row) row tiles of a launch: workgroup w of `cus` walks tiles w, w + cus, ... of the order [row tile][column tile], tall row
// tiles first; a tall tile costs 4 units, a short one 3.4 (measured: three of four fragments' MFMAs and stores, but the same LDS-DMA pieces, barriers and tile set-up).  Picks the number of tall row tiles with
// the smallest maximum load (ties: more tall tiles = fewer tiles).  MFVIT_PP_MIX=0: tall tiles only (round-6 first form).
struct PpPlan { int ntall, nrt; double load, ideal; };
static PpPlan pp_plan(int M, int ntn, int cus) {
    static int sw = INT_MIN;
    const bool mix = env_switch("MFVIT_PP_MIX", 1, sw) != 0;
    const int nt_all = (M + PP_BM - 1) / PP_BM;
    // (the search is ~nt_all x cus steps: cached per thread for the last shapes - an encoder step asks for the same four or five over and over)
    struct Key { int M, ntn, cus, mix; PpPlan plan; };
    static thread_local Key cache[8];
    static thread_local int cache_n = 0, cache_next = 0;
    for (int i = 0; i < cache_n; ++i)
        if (cache[i].M == M && cache[i].ntn == ntn && cache[i].cus == cus && cache[i].mix == (int)mix) return cache[i].plan;
    PpPlan best{nt_all, nt_all, 1e30, 0.0};
    for (int na = nt_all; na >= (mix ? 0 : nt_all); --na) {
        const int rest = M - na * PP_BM;
        const int nb = rest > 0 ? (rest + 191) / 192 : 0;
        if (na < nt_all && nb == 0) continue;
        const long ntile = (long)(na + nb) * ntn, ntall_t = (long)na * ntn;
        double worst = 0.0;
        for (int w = 0; w < cus && w < ntile; ++w) {
            // tiles w, w + cus, ...: those below ntall_t are tall
            const long n_all = (ntile - w + cus - 1) / cus;
            const long n_tall = ntall_t > w ? (ntall_t - w + cus - 1) / cus : 0;
            const double l = 4.0 * n_tall + 3.4 * (n_all - n_tall);
            worst = l > worst ? l : worst;
        }
        if (worst < best.load - 1e-9) best = PpPlan{na, na + nb, worst, 0.0};
    }
    best.ideal = (double)M / 64.0 * ntn / cus;
    cache[cache_next] = Key{M, ntn, cus, (int)mix, best};
    cache_next = (cache_next + 1) & 7;
    if (cache_n < 8) ++cache_n;
    return best;
}
template <typename T, int EPI> int launch_pp(const GemmP& p, hipStream_t st) {
    const PpPlan plan = pp_plan(p.M, p.N / PP_BN, device_cus());
    const int ntiles = (p.N / PP_BN) * plan.nrt;
    int grid = device_cus();
    if (grid > ntiles) grid = ntiles;
    static PerDeviceOnce attr_set;
    if (attr_set.first()) (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
    ProfScope ps(PROF_GEMM_TILE, 2.0 * p.M * p.N * p.K, 0, st);
    MFVIT_LAUNCH((gemm_nt_pp_kernel<T, EPI>), dim3(grid), dim3(512), PP_LDS, st, p, ntiles, plan.ntall);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
template <typename T> int pp_by_epi(int epi, const GemmP& p, hipStream_t st) {
    switch (epi) {
        case EPI_BIAS: return launch_pp<T, EPI_BIAS>(p, st);
        case EPI_BIAS_GELU: return launch_pp<T, EPI_BIAS_GELU>(p, st);
        case EPI_GELU_BWD: return launch_pp<T, EPI_GELU_BWD>(p, st);
        case EPI_NONE: return launch_pp<T, EPI_NONE>(p, st);
        case EPI_BIAS_RELU: return launch_pp<T, EPI_BIAS_RELU>(p, st);
        case EPI_BIAS_X3F16:
            if constexpr (is_split<T>::value) return launch_pp<T, EPI_BIAS_X3F16>(p, st);
            return MFVIT_EINVAL;
    }
    return MFVIT_EINVAL;
}

}  // namespace

// MFVIT_PP: 1 (default) the ping-pong kernel wherever it pays, 0 never (the round-5 tile kernel everywhere), 2 wherever it CAN run (tests); MFVIT_PP_MINROWS: smallest M it takes
bool gemm_nt_pp_supported(int dtype, int epi, const GemmP& p) {
    static int c_on = INT_MIN, c_min = INT_MIN;
    const int on = env_switch("MFVIT_PP", 1, c_on);            // 2: wherever the kernel CAN run (tests: small shapes, few tiles)
    if (!on) return false;
    if (dtype != MFVIT_BF16X3 && dtype != MFVIT_BF16 && dtype != MFVIT_F16) return false;
    if (epi == EPI_BIAS_X3F16 && dtype != MFVIT_BF16X3) return false;
    const int kps = dtype == MFVIT_BF16X3 ? 32 : 64;
    if (p.nb > 1 || (on != 2 && p.M < env_switch("MFVIT_PP_MINROWS", 2048, c_min)) || p.M < 1 || p.N % PP_BN || p.K % kps || p.K < 2 * kps) return false;
    if (p.omax) return false;                                   // per-(image, head) output maxima: the round-5 kernel's epilogue (proj data gradient)
    if (on != 2) {   // enough tiles to fill the persistent grid's rounds: 256 x 128 tiles on one workgroup per CU quantise coarsely (proj data gradient at the bench shape:
        // 297 tiles = 2 rounds at 58 % - 35.3 us against 31.7 us of the 128 x 128 kernel on two workgroups per CU; qkv 891 tiles = 4 rounds at 87 %)
        // (with mixed tile heights: the ideal load per workgroup against the plan's largest)
        const int cus = device_cus();
        const PpPlan plan = pp_plan(p.M, p.N / PP_BN, cus);
        if ((long)plan.nrt * (p.N / PP_BN) < cus || plan.ideal < 0.75 * plan.load) return false;
    }
    if ((p.lda * 2) % 16 || (p.ldw * 2) % 16) return false;
    // 32-bit byte offsets inside both operands
    if ((unsigned long long)(p.M - 1) * p.lda * 2 + 4096 >= (1ull << 32) || (unsigned long long)(p.N - 1) * p.ldw * 2 + 4096 >= (1ull << 32)) return false;
    if (epi == EPI_GELU_BWD && (!p.aux || (p.ldaux * 2) % 16 || (unsigned long long)(p.M - 1) * p.ldaux * 2 + (unsigned long long)p.N * 2 + 4096 >= (1ull << 32))) return false;
    return true;
}
int gemm_nt_pp(int dtype, int epi, const GemmP& p, hipStream_t st) {
    int rc = MFVIT_EINVAL;
    switch (dtype) {
        case MFVIT_BF16X3: rc = pp_by_epi<sbf16>(epi, p, st); break;
        case MFVIT_BF16: rc = pp_by_epi<bf16>(epi, p, st); break;
        case MFVIT_F16: rc = pp_by_epi<f16>(epi, p, st); break;
    }
    // column-sum partials of the x act' epilogue: [4 x row tiles][N] (p.cpart: room for 4 x ceil(M / 192) rows of N floats) -> cs0, fixed order
    if (rc == MFVIT_OK && epi == EPI_GELU_BWD && p.cs0 && p.cpart)
        return colpart_reduce(p.cpart, 4 * pp_plan(p.M, p.N / PP_BN, device_cus()).nrt, p.N, 1, p.cs0, nullptr, nullptr, st);
    return rc;
}

}  // namespace mfvit
