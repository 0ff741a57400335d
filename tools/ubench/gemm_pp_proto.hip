// Prototype of the round-6 tile GEMM main loop ("ping-pong"): C[M][N] = A[M][K] W[N][K]^T, split-bf16 operands ([hi x 32 | lo x 32] groups, 3 MFMAs per product),
// f32 out - the same problem and packing as tools/ubench/gemm_format_proto.hip (its `x3` kernel = the structure of the shipping tile kernel: 128 x 128 tile,
// 4 waves, register-staged double buffer, one barrier per K tile, two workgroups per CU), so the two binaries compare main-loop STRUCTURES on one box.
//
// Structure here (MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md 5 "8-phase template"):
//   * ONE persistent 512-thread workgroup per CU walks 256 x 128 output tiles; waves 0-3 (group 0) own rows 0..127 of the tile, waves 4-7 (group 1) rows 128..255,
//     each wave a 64 x 64 quadrant - every SIMD hosts one wave of each group;
//   * the groups run HALF A STAGE apart: while one group issues its 12 MFMAs of a stage (s_setprio 1), the other reads its fragments of the next stage from LDS and
//     issues its LDS-DMA pieces - matrix work beside memory work on every SIMD, separated by s_barrier (two per stage);
//   * operands go global -> LDS by LDS-DMA (no VGPR round trip, no ds_write) into a 5-slot ring of HALF k groups (a stage = 16 logical k = [hi x 16 | lo x 16] = 64 B per
//     row: A 16 KB + W 8 KB), three stages in flight with counted vmcnt; the stream of stages runs across tile boundaries (the next tile's first stages are in flight
//     during the epilogue), and the epilogue's stores stay in flight behind counted waits;
//   * epilogue per wave through a PRIVATE 4.5 KB LDS patch (no workgroup barrier): 32 x 32 accumulator tile -> rows -> 16-byte coalesced stores.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_pp_proto.hip -o tools/ubench/gemm_pp_proto && tools/ubench/gemm_pp_proto [M N K]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 256, BN = 128;
// FULL = false: a stage is HALF a k group (64 B per row: [hi x 16 | lo x 16]), 12 MFMAs per wave and phase, 5-slot ring, 3 stages ahead (version 1)
// FULL = true : a stage is a whole k group (128 B per row), 24 MFMAs per wave and phase (half as many barriers), 3-slot ring, 2 stages ahead
template <bool FULL> struct Geo {
    static constexpr int SROW = FULL ? 128 : 64;                       // bytes per row and stage
    static constexpr int A_BYTES = BM * SROW, W_BYTES = BN * SROW;
    static constexpr int SLOT = A_BYTES + W_BYTES;                     // 24 / 48 KB
    static constexpr int NSLOT = FULL ? 3 : 5, AHEAD = FULL ? 2 : 3;
    static constexpr int RING = NSLOT * SLOT;                          // 120 / 144 KB
    static constexpr int NDMA = SLOT / 1024 / 8;                       // LDS-DMA instructions per wave and stage: 3 / 6
    static constexpr int KS = FULL ? 2 : 1;                            // 16-wide k steps per stage
    static constexpr int EPI_ROWS = FULL ? 16 : 32;                    // rows of the per-wave epilogue patch (128 B each)
    static constexpr int LDS_TOTAL = RING + 8 * EPI_ROWS * 128;        // 152 / 160 KB
};
constexpr int NSTORE = 16;                               // epilogue stores per wave and tile

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void glds16(unsigned voff, const char* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// ABL (timing only): 1 no MFMAs, 2 no LDS-DMA after the prologue, 4 no epilogue, 8 no fragment reads after the first stage
// NC: LDS-DMA pieces of a stage issued inside the COMPUTE phase (between its MFMAs) instead of the load phase
// PRIO: 1 s_setprio 1 around the MFMAs (default), 0 none, 2 around the load phase, 3 static: waves 4 - 7 at priority 1;  DFIRST: LDS-DMA pieces before the fragment reads
// M16 (whole-group stages only): v_mfma_f32_16x16x32_bf16 (one k group = one MFMA K; 48 per stage at 16 cycles) instead of 32x32x16 (24 at 32 cycles)
template <bool FULL, int ABL, int NC = 0, int PRIO = 1, bool DFIRST = false, bool M16 = false>
__global__ __launch_bounds__(512, 1) void gemm_pp(const char* __restrict__ A, const char* __restrict__ W, float* __restrict__ C, int M, int N, int K, int ntiles) {
    typedef Geo<FULL> G_;
    constexpr int SROW = G_::SROW, A_BYTES = G_::A_BYTES, SLOT = G_::SLOT, NSLOT = G_::NSLOT, AHEAD = G_::AHEAD, NDMA = G_::NDMA, KS = G_::KS;
    constexpr int NDA = NDMA * 2 / 3, NDW = NDMA / 3;      // A / W pieces per wave and stage (2 : 1)
    constexpr int RPP = 1024 / SROW;                       // rows per LDS-DMA piece: 16 / 8
    constexpr int CPR = SROW / 16;                         // 16-byte chunks per row: 4 / 8
    constexpr int NMF = 12 * KS;                           // MFMAs per wave and stage
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3, wm = wq >> 1, wn = wq & 1;
    const int ntn = N / BN;
    const unsigned ldb = (unsigned)K * 4u;               // bytes per operand row
    const int nst = FULL ? K / 32 : K / 16;              // stages per tile
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    const int G = gridDim.x;
    const int my = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;      // each XCD walks a contiguous run of the tile order (G % 8 == 0)
    const int ntl = my < ntiles ? (ntiles - my + G - 1) / G : 0;
    if (ntl == 0) return;
    auto swz = [](int row) { return FULL ? (row >> 1) & 7 : (row >> 2) & 3; };     // 16 rows of a ds_read_b128 lane group -> 16 distinct slots of the bank row

    // ---- LDS-DMA: a piece = 1 KB = RPP rows; A pieces wave + 8 i, W pieces wave + 8 i.  Lane l: row l / CPR of the piece, LDS chunk l % CPR holding source chunk
    // (l % CPR) ^ swz(row).  Source chunks of a HALF stage: 0, 1 = hi elements, 2, 3 = lo elements (32 B each, 64 B apart in the k group); of a whole k group: 16 c.
    unsigned va[NDA], vw[NDW];
    int ld_t = 0, ld_s = 0, ld_slot = 0;                 // the DMA stream's position: tile (of this workgroup), stage, ring slot
    auto src_chunk = [&](int row) {
        const int c = (lane % CPR) ^ swz(row);
        return (unsigned)(FULL ? 16 * c : (c >> 1) * 64 + (c & 1) * 16);
    };
    auto set_tile_offsets = [&](int t) {
        const int tile = my + t * G;
        const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
#pragma unroll
        for (int i = 0; i < NDA; ++i) {
            const int row = RPP * (wave + 8 * i) + lane / CPR;
            int gm = m0 + row;
            gm = gm < M ? gm : M - 1;
            va[i] = (unsigned)gm * ldb + src_chunk(row);
        }
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            const int row = RPP * (wave + 8 * i) + lane / CPR;
            vw[i] = (unsigned)(n0 + row) * ldb + src_chunk(row);
        }
    };
    set_tile_offsets(0);
    unsigned dma_soff = 0, dma_dst = 0;
    auto dma_begin = [&]() {                              // scalar state of the stage the DMA stream is at
        dma_soff = FULL ? (unsigned)ld_s * 128u : (unsigned)(ld_s >> 1) * 128u + (unsigned)(ld_s & 1) * 32u;
        dma_dst = lbase + (unsigned)ld_slot * SLOT + (unsigned)wave * 1024u;
    };
    auto dma_piece = [&](int d) __attribute__((always_inline)) {      // piece d of the current stage (compile-time d)
        if (ABL & 2) return;
        if (d < NDA) glds16(va[d], A + dma_soff, dma_dst + d * 8 * 1024);
        else glds16(vw[d - NDA], W + dma_soff, dma_dst + A_BYTES + (d - NDA) * 8 * 1024);
    };
    auto dma_end = [&]() {
        ld_slot = ld_slot == NSLOT - 1 ? 0 : ld_slot + 1;
        if (++ld_s == nst) {                             // next tile (past the end: the last tile again - harmless refills of free slots, the waits keep their counts)
            ld_s = 0;
            if (ld_t + 1 < ntl) { ++ld_t; set_tile_offsets(ld_t); }
        }
    };
    // fragment addresses inside a slot (constant per lane): hi fragment of k step 0; k step 1 at ^ 32, the lo parts at ^ 64 (FULL) / ^ 32 (half stages)
    unsigned fa[2], fw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = grp * 128 + (wm * 2 + i) * 32 + (lane & 31), rw = (wn * 2 + i) * 32 + (lane & 31);
        fa[i] = (unsigned)(ra * SROW + 16 * (h ^ swz(ra)));
        fw[i] = (unsigned)(A_BYTES + rw * SROW + 16 * (h ^ swz(rw)));
    }
    // 16 x 16 x 32: lane = (row lane & 15, k quarter lane >> 4); hi part = chunk kq, lo part = chunk 4 + kq (address ^ 64); four 16-row tiles per wave and operand
    unsigned ga[4], gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = grp * 128 + wm * 64 + i * 16 + (lane & 15), rw = wn * 64 + i * 16 + (lane & 15);
        ga[i] = (unsigned)(ra * SROW + 16 * ((lane >> 4) ^ swz(ra)));
        gw[i] = (unsigned)(A_BYTES + rw * SROW + 16 * ((lane >> 4) ^ swz(rw)));
    }
    bf16x8 a16[2][4], b16[2][4];
    f32x4 c16[4][4];
    constexpr unsigned LO = FULL ? 64u : 32u;
    char* epi = lds + G_::RING + wave * G_::EPI_ROWS * 128;

    if (ABL & 2) {
        const unsigned dst = lbase + (unsigned)wave * 1024u;
#pragma unroll
        for (int d = 0; d < NDA; ++d) glds16(va[d], A, dst + d * 8 * 1024);
#pragma unroll
        for (int d = 0; d < NDW; ++d) glds16(vw[d], W, dst + A_BYTES + d * 8 * 1024);
    }
#pragma unroll
    for (int d = 0; d < AHEAD; ++d) {
        dma_begin();
#pragma unroll
        for (int q = 0; q < NDMA; ++q) dma_piece(q);
        dma_end();
    }
    if (ABL & 2) wait_vm<0>(); else wait_vm<NDMA*(AHEAD - 1)>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs half a stage behind
    int slot = 0;
    if (PRIO == 3 && grp == 1) __builtin_amdgcn_s_setprio(1);
    bf16x8 ah[KS][2], al[KS][2], bh[KS][2], bl[KS][2];
    for (int t = 0; t < ntl; ++t) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (M16) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int s = 0; s < nst; ++s) {
            // ---- load phase (the other group computes)
            const char* sl = lds + slot * SLOT;
            if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
            if (DFIRST) {
                dma_begin();
#pragma unroll
                for (int q = 0; q < NDMA - NC; ++q) dma_piece(q);
            }
            if (M16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a16[0][i] = *(const bf16x8*)(sl + ga[i]);
                    a16[1][i] = *(const bf16x8*)(sl + (ga[i] ^ 64u));
                    b16[0][i] = *(const bf16x8*)(sl + gw[i]);
                    b16[1][i] = *(const bf16x8*)(sl + (gw[i] ^ 64u));
                }
            } else if (!(ABL & 8) || (t == 0 && s == 0)) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        ah[ks][i] = *(const bf16x8*)(sl + (fa[i] ^ (32u * ks)));
                        al[ks][i] = *(const bf16x8*)(sl + (fa[i] ^ (32u * ks) ^ LO));
                        bh[ks][i] = *(const bf16x8*)(sl + (fw[i] ^ (32u * ks)));
                        bl[ks][i] = *(const bf16x8*)(sl + (fw[i] ^ (32u * ks) ^ LO));
                    }
            }
            if (!DFIRST) {
                dma_begin();
#pragma unroll
                for (int q = 0; q < NDMA - NC; ++q) dma_piece(q);
            }
            // the NEXT stage must have landed before the next load phase.  Queue of this wave, oldest first: [stage +1 .. +AHEAD-1, NDMA pieces each][stage +AHEAD: the
            // NDMA - NC pieces just issued] - and behind an epilogue its NSTORE stores sit between the stages issued before and after it: counted, so that they stay
            // in flight until a stage issued after them is needed
            constexpr int KEEP = NDMA * (AHEAD - 1) - NC;
            if (ABL & 2) {}
            else if ((ABL & 4) || t == 0 || s >= AHEAD - 1) wait_vm<KEEP>();
            else wait_vm<KEEP + NSTORE>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's fragment reads are done: the slot may be refilled once the barrier is passed
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- compute phase
            if (M16) {
                if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int tt = 0; tt < 48; ++tt) {
                    const int term = tt >> 4, i = (tt >> 2) & 3, j = tt & 3;
                    c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[term == 0 ? 1 : 0][i], b16[term == 1 ? 1 : 0][j], c16[i][j], 0, 0, 0);
                }
                if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            } else if (!(ABL & 1)) {
                if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int tt = 0; tt < NMF; ++tt) {
                    const int ks = tt / 12, term = (tt % 12) >> 2, i = (tt >> 1) & 1, j = tt & 1;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 0 ? al[ks][i] : ah[ks][i], term == 1 ? bl[ks][j] : bh[ks][j], acc[i][j], 0, 0, 0);
                    if (NC > 0) {
                        constexpr int GAP = NMF / (NC > 0 ? NC + 1 : 1);
                        if ((tt + 1) % GAP == 0 && (tt + 1) / GAP <= NC) { dma_piece(NDMA - NC + (tt + 1) / GAP - 1); __builtin_amdgcn_sched_barrier(0); }
                    }
                }
                if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            } else {
#pragma unroll
                for (int q = NDMA - NC; q < NDMA; ++q) dma_piece(q);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const i32x4 u = __builtin_bit_cast(i32x4, al[ks][i]) ^ __builtin_bit_cast(i32x4, bh[ks][j]) ^ __builtin_bit_cast(i32x4, ah[ks][i]) ^ __builtin_bit_cast(i32x4, bl[ks][j]);
                            acc[i][j][0] += __int_as_float((u[0] ^ u[1] ^ u[2] ^ u[3]) & 0x3fffffff);
                        }
            }
            dma_end();
            __builtin_amdgcn_s_barrier();
            slot = slot == NSLOT - 1 ? 0 : slot + 1;
        }
        // ---- epilogue (no workgroup barrier inside: the other group goes on with its phases); EPI_ROWS rows of a 32 x 32 accumulator tile at a time
        if (M16 && !(ABL & 4)) {
            // C / D map of 16 x 16 x 32: column lane & 15, rows 4 (lane >> 4) + register.  Unit = 16 rows x 32 columns (two column tiles) through the 16 x 128-byte patch
            const int tile = my + t * G;
            const int m0 = (tile / ntn) * BM + grp * 128 + wm * 64, n0 = (tile % ntn) * BN + wn * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) *(float*)(epi + (4 * (lane >> 4) + r) * 128 + 4 * (jj * 16 + (lane & 15))) = c16[i][2 * jp + jj][r];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int row = 8 * q + (lane >> 3);
                        const f32x4 v = *(const f32x4*)(epi + row * 128 + 16 * (lane & 7));
                        int m = m0 + i * 16 + row;
                        m = m < M ? m : M - 1;
                        *(f32x4*)(C + (long)m * N + n0 + jp * 32 + 4 * (lane & 7)) = v;
                    }
                }
        }
        if (!M16 && !(ABL & 4)) {
            const int tile = my + t * G;
            const int m0 = (tile / ntn) * BM + grp * 128 + wm * 64, n0 = (tile % ntn) * BN + wn * 64;
            constexpr int ER = G_::EPI_ROWS, NP = 32 / ER;                  // patch rows, passes per accumulator tile
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pass = 0; pass < NP; ++pass) {
#pragma unroll
                        for (int r = pass * (16 / NP); r < (pass + 1) * (16 / NP); ++r) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * h - pass * ER;
                            *(float*)(epi + row * 128 + 4 * (lane & 31)) = acc[i][j][r];
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int q = 0; q < ER / 8; ++q) {
                            const int row = 8 * q + (lane >> 3);
                            const f32x4 v = *(const f32x4*)(epi + row * 128 + 16 * (lane & 7));
                            int m = m0 + i * 32 + pass * ER + row;
                            m = m < M ? m : M - 1;      // rows past the end replicate row M - 1 (clamped A rows): identical duplicates, and the store COUNT stays fixed for the counted waits
                            *(f32x4*)(C + (long)m * N + n0 + j * 32 + 4 * (lane & 7)) = v;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
        }
        if (ABL & 4) {                                   // timing builds without the epilogue: keep the accumulators alive
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
            if (M16) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum += c16[i][j][0] + c16[i][j][1] + c16[i][j][2] + c16[i][j][3];
            }
            if (sum == 12345.678f) C[tid] = sum;
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    wait_vm<0>();
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// Second experiment (round 6, after the first production kernel): can the EPILOGUE be hidden?  gemm_pp2 = the same main loop with
//   HEAVY   : a GELU on every output element (the vector load of the real fc1 epilogue: ~17 instructions per element)
//   EPIMODE : 0 both groups run the epilogue at the tile end, at the same time (what ships);
//             2 DEFERRED: the accumulators of tile t move to a second register set and its epilogue runs in 16 pieces (4 registers of one 32 x 32 tile: 8 rows x 32
//               columns through a 1 KB patch, one 16-byte store per lane) inside the COMPUTE phases of the first 16 stages of tile t + 1, interleaved with their
//               MFMAs (sched_group_barrier).  Half-k-group stages only: whole-stage fragments (64 registers) + two accumulator sets (128) do not fit 256.
__device__ __forceinline__ float gelu_heavy(float x) {
    const float z = fabsf(x) * 0.70710678f, t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f)), e = __builtin_amdgcn_exp2f(x * x * -0.72134752f);
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f); pl = fmaf(pl, t, -0.284496736f); pl = fmaf(pl, t, 0.254829592f);
    const float cdf = 0.5f * (1.0f + copysignf(fmaf(-pl * t, e, 1.0f), x));
    return fmaf(x * 0.3989422804f, e, cdf) * 1e-30f + x * cdf;      // (gelu' rides along at no weight: both are computed, as in the real epilogue)
}
template <bool FULL, int EPIMODE, bool HEAVY>
__global__ __launch_bounds__(512, 1) void gemm_pp2(const char* __restrict__ A, const char* __restrict__ W, float* __restrict__ C, int M, int N, int K, int ntiles) {
    typedef Geo<FULL> G_;
    constexpr int SROW = G_::SROW, A_BYTES = G_::A_BYTES, SLOT = G_::SLOT, NSLOT = G_::NSLOT, AHEAD = G_::AHEAD, NDMA = G_::NDMA, KS = G_::KS;
    constexpr int NDA = NDMA * 2 / 3, NDW = NDMA / 3, RPP = 1024 / SROW, CPR = SROW / 16, NMF = 12 * KS;
    static_assert(EPIMODE == 0 || !FULL, "deferred epilogue: half-group stages");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3, wm = wq >> 1, wn = wq & 1;
    const int ntn = N / BN;
    const unsigned ldb = (unsigned)K * 4u;
    const int nst = FULL ? K / 32 : K / 16;
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    const int G = gridDim.x;
    const int my = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;
    const int ntl = my < ntiles ? (ntiles - my + G - 1) / G : 0;
    if (ntl == 0) return;
    auto swz = [](int row) { return FULL ? (row >> 1) & 7 : (row >> 2) & 3; };
    unsigned va[NDA], vw[NDW];
    int ld_t = 0, ld_s = 0, ld_slot = 0;
    auto set_tile_offsets = [&](int t) {
        int ln = tid & 63;
        asm volatile("" : "+v"(ln));
        const int tile = my + t * G;
        const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
#pragma unroll
        for (int i = 0; i < NDA; ++i) {
            const int row = RPP * (wave + 8 * i) + ln / CPR, c = (ln % CPR) ^ swz(row);
            int gm = m0 + row;
            gm = gm < M ? gm : M - 1;
            va[i] = (unsigned)gm * ldb + (unsigned)(FULL ? 16 * c : (c >> 1) * 64 + (c & 1) * 16);
        }
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            const int row = RPP * (wave + 8 * i) + ln / CPR, c = (ln % CPR) ^ swz(row);
            vw[i] = (unsigned)(n0 + row) * ldb + (unsigned)(FULL ? 16 * c : (c >> 1) * 64 + (c & 1) * 16);
        }
    };
    set_tile_offsets(0);
    auto dma_stage = [&]() {
        const unsigned soff = FULL ? (unsigned)ld_s * 128u : (unsigned)(ld_s >> 1) * 128u + (unsigned)(ld_s & 1) * 32u;
        const unsigned dst = lbase + (unsigned)ld_slot * SLOT + (unsigned)wave * 1024u;
#pragma unroll
        for (int d = 0; d < NDA; ++d) glds16(va[d], A + soff, dst + d * 8 * 1024);
#pragma unroll
        for (int d = 0; d < NDW; ++d) glds16(vw[d], W + soff, dst + A_BYTES + d * 8 * 1024);
        ld_slot = ld_slot == NSLOT - 1 ? 0 : ld_slot + 1;
        if (++ld_s == nst) { ld_s = 0; if (ld_t + 1 < ntl) { ++ld_t; set_tile_offsets(ld_t); } }
    };
    unsigned fa[2], fw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = grp * 128 + (wm * 2 + i) * 32 + (lane & 31), rw = (wn * 2 + i) * 32 + (lane & 31);
        fa[i] = (unsigned)(ra * SROW + 16 * (h ^ swz(ra)));
        fw[i] = (unsigned)(A_BYTES + rw * SROW + 16 * (h ^ swz(rw)));
    }
    constexpr unsigned LO = FULL ? 64u : 32u;
    char* patch = lds + G_::RING + wave * 1024;
#pragma unroll
    for (int d = 0; d < AHEAD; ++d) dma_stage();
    wait_vm<NDMA*(AHEAD - 1)>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    int slot = 0;
    bf16x8 ah[KS][2], al[KS][2], bh[KS][2], bl[KS][2];
    f32x16 acc[2][2], accp[2][2];
    int mp = 0, np = 0;                                    // quadrant origin of the tile whose epilogue is pending
    // one piece: registers 4 q .. 4 q + 3 of tile (i, j) of `src` = 8 rows x 32 columns
    auto piece = [&](auto kc, f32x16 (&src)[2][2], int m0q, int n0q) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, i = k >> 3, j = (k >> 2) & 1, q = k & 3;
        int ln = tid & 63;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = src[i][j][4 * q + r];
            if (HEAVY) v = gelu_heavy(v);
            *(float*)(patch + (r + 4 * (ln >> 5)) * 128 + 4 * (ln & 31)) = v;
        }
        const int row = ln >> 3;
        const f32x4 v4 = *(const f32x4*)(patch + row * 128 + 16 * (ln & 7));
        int m = m0q + i * 32 + 8 * q + row;
        m = m < M ? m : M - 1;
        *(f32x4*)(C + (long)m * N + n0q + j * 32 + 4 * (ln & 7)) = v4;
    };
    auto all_pieces = [&](f32x16 (&src)[2][2], int m0q, int n0q) __attribute__((always_inline)) {
        piece(std::integral_constant<int, 0>(), src, m0q, n0q); piece(std::integral_constant<int, 1>(), src, m0q, n0q); piece(std::integral_constant<int, 2>(), src, m0q, n0q);
        piece(std::integral_constant<int, 3>(), src, m0q, n0q); piece(std::integral_constant<int, 4>(), src, m0q, n0q); piece(std::integral_constant<int, 5>(), src, m0q, n0q);
        piece(std::integral_constant<int, 6>(), src, m0q, n0q); piece(std::integral_constant<int, 7>(), src, m0q, n0q); piece(std::integral_constant<int, 8>(), src, m0q, n0q);
        piece(std::integral_constant<int, 9>(), src, m0q, n0q); piece(std::integral_constant<int, 10>(), src, m0q, n0q); piece(std::integral_constant<int, 11>(), src, m0q, n0q);
        piece(std::integral_constant<int, 12>(), src, m0q, n0q); piece(std::integral_constant<int, 13>(), src, m0q, n0q); piece(std::integral_constant<int, 14>(), src, m0q, n0q);
        piece(std::integral_constant<int, 15>(), src, m0q, n0q);
    };
    constexpr int KEEP = NDMA * (AHEAD - 1);
    for (int t = 0; t < ntl; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int tile = my + t * G;
        const int m0q = (tile / ntn) * BM + grp * 128 + wm * 64, n0q = (tile % ntn) * BN + wn * 64;
        for (int s = 0; s < nst; ++s) {
            const char* sl = lds + slot * SLOT;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[ks][i] = *(const bf16x8*)(sl + (fa[i] ^ (32u * ks)));
                    al[ks][i] = *(const bf16x8*)(sl + (fa[i] ^ (32u * ks) ^ LO));
                    bh[ks][i] = *(const bf16x8*)(sl + (fw[i] ^ (32u * ks)));
                    bl[ks][i] = *(const bf16x8*)(sl + (fw[i] ^ (32u * ks) ^ LO));
                }
            dma_stage();
            if (EPIMODE == 2) {
                // queue, oldest first: [stage +1][store of piece s - 2][stage +2][store of piece s - 1][stage +3, just issued]   (AHEAD = 3, one store per piece)
                const int st = t == 0 ? 0 : (s >= 1 && s <= 16 ? 1 : 0) + (s >= 2 && s <= 17 ? 1 : 0);
                if (st == 2) wait_vm<KEEP + 2>(); else if (st == 1) wait_vm<KEEP + 1>(); else wait_vm<KEEP>();
            } else {
                if (t == 0 || s >= AHEAD - 1) wait_vm<KEEP>(); else wait_vm<KEEP + 16>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int tt = 0; tt < NMF; ++tt) {
                    const int ks = tt / 12, term = (tt % 12) >> 2, i = (tt >> 1) & 1, j = tt & 1;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 0 ? al[ks][i] : ah[ks][i], term == 1 ? bl[ks][j] : bh[ks][j], acc[i][j], 0, 0, 0);
                }
            };
            if (EPIMODE == 2 && t > 0 && s < 16) {
                auto both = [&](auto kc) __attribute__((always_inline)) {
                    mfmas();
                    piece(kc, accp, mp, np);
#pragma unroll
                    for (int n = 0; n < NMF; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, HEAVY ? 7 : 2, 0);
                    }
                };
                switch (s) {
                    case 0: both(std::integral_constant<int, 0>()); break;   case 1: both(std::integral_constant<int, 1>()); break;
                    case 2: both(std::integral_constant<int, 2>()); break;   case 3: both(std::integral_constant<int, 3>()); break;
                    case 4: both(std::integral_constant<int, 4>()); break;   case 5: both(std::integral_constant<int, 5>()); break;
                    case 6: both(std::integral_constant<int, 6>()); break;   case 7: both(std::integral_constant<int, 7>()); break;
                    case 8: both(std::integral_constant<int, 8>()); break;   case 9: both(std::integral_constant<int, 9>()); break;
                    case 10: both(std::integral_constant<int, 10>()); break; case 11: both(std::integral_constant<int, 11>()); break;
                    case 12: both(std::integral_constant<int, 12>()); break; case 13: both(std::integral_constant<int, 13>()); break;
                    case 14: both(std::integral_constant<int, 14>()); break; default: both(std::integral_constant<int, 15>()); break;
                }
            } else {
                __builtin_amdgcn_s_setprio(1);
                mfmas();
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_s_barrier();
            slot = slot == NSLOT - 1 ? 0 : slot + 1;
        }
        if (EPIMODE == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accp[i][j] = acc[i][j];
            mp = m0q; np = n0q;
        } else {
            if (grp == 0) __builtin_amdgcn_s_barrier();      // both groups' epilogues at the same time
            all_pieces(acc, m0q, n0q);
            if (grp == 1) __builtin_amdgcn_s_barrier();
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    if (EPIMODE == 2) all_pieces(accp, mp, np);
    wait_vm<0>();
}

static uint16_t f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 25216, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 384;
    if (N % 128 || K % 32) { printf("N %% 128 and K %% 32 required\n"); return 1; }
    std::vector<float> a((size_t)M * K), w((size_t)N * K);
    unsigned x = 1234567u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (float)((x >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : a) { float s = rnd() + rnd() + rnd() + rnd(); v = 1.7f * s; }
    for (size_t i = 0; i < a.size(); i += 97) a[i] *= 6.f;
    for (auto& v : w) { float s = rnd() + rnd() + rnd() + rnd(); v = 0.09f * s; }
    auto pack = [&](const std::vector<float>& src, int rows) {
        std::vector<char> out((size_t)rows * K * 4);
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < K; ++k) {
                const float v = src[(size_t)r * K + k];
                const uint16_t hi = f2bf(v), lo = f2bf(v - bf2f(hi));
                char* g = out.data() + (size_t)r * K * 4 + (k / 32) * 128;
                memcpy(g + 2 * (k % 32), &hi, 2);
                memcpy(g + 64 + 2 * (k % 32), &lo, 2);
            }
        return out;
    };
    std::vector<char> pa = pack(a, M), pw = pack(w, N);
    char *dA, *dW; float* dC;
    (void)hipMalloc(&dA, pa.size()); (void)hipMalloc(&dW, pw.size()); (void)hipMalloc(&dC, (size_t)M * N * 4);
    (void)hipMemcpy(dA, pa.data(), pa.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dW, pw.data(), pw.size(), hipMemcpyHostToDevice);
    (void)hipMemset(dC, 0, (size_t)M * N * 4);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int ntiles = ((M + BM - 1) / BM) * (N / BN);
    int grid = prop.multiProcessorCount / 8 * 8;
    if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> c((size_t)M * N);
    auto timeit = [&](auto kern, int ldsb, const char* what, bool check) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        if (check) (void)hipMemset(dC, 0, (size_t)M * N * 4);
        float best = 1e30f;
        for (int rep = 0; rep < 25; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ldsb, 0, dA, dW, dC, M, N, K, ntiles);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 5 && ms < best) best = ms;
        }
        hipError_t err = hipGetLastError();
        printf("pp   %-72s %8.1f us  %7.1f algorithmic TFLOP/s%s", what, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12, err == hipSuccess ? "" : "  (LAUNCH ERROR)");
        if (check) {                // accuracy on 64 sampled rows against float64 (+ the last rows of the partial M tile)
            (void)hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
            double emax = 0, ymax = 0;
            for (int s = 0; s < 96; ++s) {
                const int m = s < 64 ? (int)(((long)s * 7919) % M) : M - 1 - (s - 64);
                for (int n = 0; n < N; ++n) {
                    double y = 0;
                    for (int k = 0; k < K; ++k) y += (double)a[(size_t)m * K + k] * (double)w[(size_t)n * K + k];
                    const double d = fabs((double)c[(size_t)m * N + n] - y);
                    emax = d > emax ? d : emax; ymax = fabs(y) > ymax ? fabs(y) : ymax;
                }
            }
            printf("   max-rel vs float64 %.2e (%s)", emax / ymax, emax / ymax < 5e-5 ? "OK" : "WRONG");
        }
        printf("\n");
        return best;
    };
    printf("M %d N %d K %d: %d tiles of 256 x 128 on %d workgroups\n", M, N, K, ntiles, grid);
    if (argc > 4 && argv[4][0] == 'e') {           // the epilogue experiment only
        constexpr int LH2 = Geo<false>::LDS_TOTAL, LF2 = Geo<true>::LDS_TOTAL;
        for (int rep = 0; rep < 2; ++rep) {
            timeit(gemm_pp2<true, 0, false>, LF2, "whole-group stages, joint epilogue, plain stores", rep == 0);
            timeit(gemm_pp2<true, 0, true>, LF2, "whole-group stages, joint epilogue, GELU on every element", false);
            timeit(gemm_pp2<false, 0, false>, LH2, "half-group stages, joint epilogue, plain stores", rep == 0);
            timeit(gemm_pp2<false, 0, true>, LH2, "half-group stages, joint epilogue, GELU on every element", false);
            timeit(gemm_pp2<false, 2, false>, LH2, "half-group stages, DEFERRED epilogue (16 pieces in the next tile's compute phases), plain", rep == 0);
            timeit(gemm_pp2<false, 2, true>, LH2, "half-group stages, DEFERRED epilogue, GELU on every element", false);
        }
        return 0;
    }
    constexpr int LH = Geo<false>::LDS_TOTAL, LF = Geo<true>::LDS_TOTAL;
    timeit(gemm_pp<false, 0>, LH, "half-group stages (12 MFMAs per phase, 5 slots): full kernel", true);
    timeit(gemm_pp<true, 0>, LF, "whole-group stages (24 MFMAs per phase, 3 slots): full kernel", true);
    timeit(gemm_pp<true, 0, 0, 1, false, true>, LF, "whole-group stages, 16 x 16 x 32 MFMAs (48 per phase)", true);
    timeit(gemm_pp<true, 4, 0, 1, false, true>, LF, "whole-group stages, 16 x 16 x 32 MFMAs: no epilogue", false);
    timeit(gemm_pp<true, 4>, LF, "whole-group stages, 32 x 32 x 16 MFMAs: no epilogue", false);
    timeit(gemm_pp<true, 0, 0, 1, false, true>, LF, "whole-group stages, 16 x 16 x 32 MFMAs again", false);
    timeit(gemm_pp<true, 0>, LF, "whole-group stages, 32 x 32 x 16 again", false);
    timeit(gemm_pp<true, 0, 0, 0>, LF, "whole-group stages, no s_setprio", true);
    timeit(gemm_pp<true, 0, 0, 2>, LF, "whole-group stages, s_setprio 1 around the LOAD phase", true);
    timeit(gemm_pp<true, 0, 0, 3>, LF, "whole-group stages, static priority 1 for waves 4 - 7", true);
    timeit(gemm_pp<true, 0, 0, 1, true>, LF, "whole-group stages, LDS-DMA pieces before the fragment reads", true);
    timeit(gemm_pp<true, 0, 0, 0, true>, LF, "whole-group stages, LDS-DMA first, no s_setprio", true);
    timeit(gemm_pp<true, 0>, LF, "whole-group stages: full kernel again", false);
    if (argc > 4) {
        timeit(gemm_pp<true, 1>, LF, "whole-group: no MFMAs (LDS-DMA + fragment reads + epilogue)", false);
        timeit(gemm_pp<true, 4>, LF, "whole-group: no epilogue (main loop only)", false);
        timeit(gemm_pp<true, 1 | 4>, LF, "whole-group: no MFMAs, no epilogue (LDS-DMA + fragment reads)", false);
        timeit(gemm_pp<true, 2 | 4>, LF, "whole-group: no LDS-DMA, no epilogue (fragment reads + MFMAs)", false);
        timeit(gemm_pp<true, 2 | 4 | 8>, LF, "whole-group: MFMAs only (+ barriers)", false);
        timeit(gemm_pp<true, 2>, LF, "whole-group: no LDS-DMA (fragment reads + MFMAs + epilogue)", false);
        timeit(gemm_pp<true, 4, 2>, LF, "whole-group, 2 pieces in the compute phase: no epilogue", false);
        timeit(gemm_pp<true, 0>, LF, "whole-group: full kernel again", false);
    }
    return 0;
}
