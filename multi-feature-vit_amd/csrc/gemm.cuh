// MFMA GEMM building blocks for gfx950 (CDNA4): LDS tile images, fragment loaders, main loops.
//
// Two operand images per element type T:
//   * "K-contiguous" operands (activations [M][K], weights [N][K]):
//       bf16: LDS image [row][BK] with 16-B chunks XOR-swizzled so that the ds_read_b128 fragment reads of
//             16 consecutive rows hit 16 distinct 16-B slots of the 256-B bank row (conflict-free);
//       f32 : LDS image [k][row + pad] (k-major) read with ds_read_b32, one float per lane, as the
//             v_mfma_f32_32x32x2_f32 operand map wants (A[i = lane&31][k = lane>>5]).
//   * "K-strided" operands (wgrad: dY [m][n] and X [m][k], reduction index m is the slow axis):
//       bf16: LDS image [m][row + 32] copied as is (rows 16-B aligned, 320-B pitch) and read with the
//             transposing ds_read_b64_tr_b16 (two per fragment);
//       f32 : LDS image [m][row + 4], again one ds_read_b32 per fragment.
// MFMA shapes: v_mfma_f32_32x32x16_bf16 (8 bf16 per lane per operand) and v_mfma_f32_32x32x2_f32.
// C/D map (both): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma once
#include "common.cuh"
#ifndef MFVIT_NT_PIPE
#define MFVIT_NT_PIPE 1
#endif
#ifndef MFVIT_TN_PIPE
#define MFVIT_TN_PIPE 1
#endif

#include <type_traits>

namespace mfvit {

template <typename T> struct MmaTraits;
template <> struct MmaTraits<bf16> {
    typedef bf16x8 frag_t;
    static constexpr int KSTEP = 16;  // k elements consumed per MFMA
    static __device__ __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct MmaTraits<sbf16> : MmaTraits<bf16> {};   // split tensors: the same MFMA on the hi / lo parts
template <> struct MmaTraits<f16> {
    typedef f16x8 frag_t;
    static constexpr int KSTEP = 16;
    static __device__ __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct MmaTraits<float> {
    typedef float frag_t;
    static constexpr int KSTEP = 2;
    static __device__ __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------------------------------
// K-contiguous tile image: R rows, BKB bytes of k per row (64 or 128).
template <typename T, int R, int BKB> struct KTile;

template <typename T, int R, int BKB> struct KTile16 {
    static constexpr int CPR = BKB / 16;            // 16-B chunks per row
    static constexpr int RPW = 16 / CPR;            // rows per 256-B bank row
    static constexpr int BYTES = R * BKB;
    static constexpr int KSTEPS = BKB / 32;         // 16 bf16 = 32 B per MFMA step
    static __device__ __forceinline__ int off(int row, int c) { return row * BKB + 16 * (c ^ ((row / RPW) % CPR)); }
    static __device__ __forceinline__ void put(char* t, int row, int c, uint4 v) { *(uint4*)(t + off(row, c)) = v; }
    // fragment for MFMA step s: lane holds row (rowbase + lane&31), k = 16 s + 8 (lane>>5) .. +8
    static __device__ __forceinline__ typename Vec8<T>::type frag(const char* t, int rowbase, int s, int lane) {
        return *(const typename Vec8<T>::type*)(t + off(rowbase + (lane & 31), 2 * s + (lane >> 5)));
    }
};
template <int R, int BKB> struct KTile<bf16, R, BKB> : KTile16<bf16, R, BKB> {};
template <int R, int BKB> struct KTile<f16, R, BKB> : KTile16<f16, R, BKB> {};
// split tensors: with BKB = 128 a row holds one 32-wide k group, [hi x 32 | lo x 32] = MFMA steps 0, 1 (hi) and 2, 3 (lo)
template <int R, int BKB> struct KTile<sbf16, R, BKB> : KTile16<sbf16, R, BKB> {};

template <int R, int BKB> struct KTile<float, R, BKB> {
    static constexpr int CPR = BKB / 16;
    static constexpr int BK = BKB / 4;
    static constexpr int PAD = (CPR == 4) ? 2 : 1;  // makes the 4 x ds_write_b32 of a staged chunk conflict-free
    static constexpr int LDR = R + PAD;
    static constexpr int BYTES = BK * LDR * 4;
    static constexpr int KSTEPS = BK / 2;
    static __device__ __forceinline__ void put(char* t, int row, int c, uint4 v) {
        float* f = (float*)t;
        const int k = 4 * c;
        f[(k + 0) * LDR + row] = __uint_as_float(v.x);
        f[(k + 1) * LDR + row] = __uint_as_float(v.y);
        f[(k + 2) * LDR + row] = __uint_as_float(v.z);
        f[(k + 3) * LDR + row] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ float frag(const char* t, int rowbase, int s, int lane) {
        return ((const float*)t)[(2 * s + (lane >> 5)) * LDR + rowbase + (lane & 31)];
    }
};

// Register-staged global -> LDS copy of a K-contiguous tile (R rows x BKB bytes) by NT threads.
// HALF (split tensors, BKB = 64): the tile is HALF a 32-wide k group - storage elements [hi x 16 | lo x 16] of group k0 / 64, half
// (k0 / 32) & 1 - so that a double-buffered stage is 64 B per row and two workgroups fit a CU (row-complete kernels); the LDS row is then
// MFMA step 0 = hi, step 1 = lo.  k0 counts storage elements in steps of 32 as for any other tile.
template <typename T, int R, int BKB, int NT, bool HALF = false> struct KStage {
    typedef KTile<T, R, BKB> Tile;
    static constexpr int CPR = BKB / 16;
    static constexpr int TOTAL = R * CPR;
    static constexpr int NCH = (TOTAL + NT - 1) / NT;
    static constexpr int EPC = 16 / sizeof(T);  // elements per chunk
    static_assert(TOTAL % NT == 0 || TOTAL < NT, "tile chunks must divide over the block (or fit in one pass)");
    static_assert(!HALF || (BKB == 64 && is_split<T>::value), "half-group tiles are a split-tensor layout");
    uint4 reg[NCH];
    // rows >= rmax are clamped (their products are never stored)
    __device__ __forceinline__ void load(const T* base, long ld, int row0, int rmax, int k0, int tid) {
        const int kb = HALF ? (k0 & ~63) + ((k0 >> 5) & 1) * 16 : k0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int q = tid + i * NT, row = q / CPR, c = q % CPR;
            if (TOTAL < NT && q >= TOTAL) continue;
            int gr = row0 + row;
            gr = gr < rmax ? gr : rmax - 1;
            const int co = HALF ? (c >> 1) * 32 + (c & 1) * 8 : c * EPC;
            reg[i] = *(const uint4*)(base + (long)gr * ld + kb + co);
        }
    }
    __device__ __forceinline__ void store(char* t, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int q = tid + i * NT, row = q / CPR, c = q % CPR;
            if (TOTAL < NT && q >= TOTAL) continue;
            Tile::put(t, row, c, reg[i]);
        }
    }
};

// ---------------------------------------------------------------------------------------------------
// K-strided tile image (wgrad operands): KR reduction rows x R "row" elements, global layout [m][row].
template <typename T, int R, int KR> struct STile;

template <typename T, int R, int KR> struct STile16 {
    static constexpr int LD = R + 32;               // 320-B pitch for R = 128: the 4 rows of a tr-read block land
    static constexpr int BYTES = KR * LD * 2;       // in 4 disjoint 64-B windows of the 256-B bank row
    static constexpr int KSTEPS = KR / 16;
    // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(row) block; lane 4q+p supplies &img[k0+q][row0+4p],
    // lane i receives column i, element q = row k0+q.  Two of them give k = 8h .. 8h+7 for row (lane&31).
    static __device__ __forceinline__ typename Vec8<T>::type frag(const char* t, int rowbase, int s, int lane) {
        const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
        const int k0 = 16 * s + 8 * h + q;
        const int col = rowbase + 16 * g1 + 4 * p;
        typedef __attribute__((address_space(3))) s16x4* lptr;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(t + ((k0)*LD + col) * 2));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(t + ((k0 + 4) * LD + col) * 2));
        union { struct { s16x4 a, b; } s; typename Vec8<T>::type v; } u;
        u.s.a = lo;
        u.s.b = hi;
        return u.v;
    }
};
template <int R, int KR> struct STile<bf16, R, KR> : STile16<bf16, R, KR> {};
template <int R, int KR> struct STile<f16, R, KR> : STile16<f16, R, KR> {};
template <int R, int KR> struct STile<sbf16, R, KR> : STile16<sbf16, R, KR> {};
template <int R, int KR> struct STile<float, R, KR> {
    static constexpr int LD = R + 4;
    static constexpr int BYTES = KR * LD * 4;
    static constexpr int KSTEPS = KR / 2;
    static __device__ __forceinline__ float frag(const char* t, int rowbase, int s, int lane) {
        return ((const float*)t)[(2 * s + (lane >> 5)) * LD + rowbase + (lane & 31)];
    }
};

// Register-staged copy of a K-strided tile: KR rows (m) x R elements, straight 16-B copies.
template <typename T, int R, int KR, int NT> struct SStage {
    typedef STile<T, R, KR> Tile;
    static constexpr int EPC = 16 / sizeof(T);
    static constexpr int CPRW = R / EPC;  // chunks per m-row
    static constexpr int NCH = KR * CPRW / NT;
    static_assert(KR * CPRW % NT == 0, "tile chunks must divide over the block");
    uint4 reg[NCH];
    // m rows >= mmax are zero-filled (they would otherwise add into the reduction)
    // BRANCH-FREE (a load in its own basic block gets a vmcnt(0) behind it): out-of-range rows are clamped for the load and
    // zeroed by a select.  REMAP: row m -> (m / rin) * rout + m % rin + roff  (patch rows -> token rows), compile-time switch.
    // CHECK = false: the whole stage is in range (every stage but possibly the last of a split) - no compare, no select.
    template <bool REMAP, bool CHECK>
    __device__ __forceinline__ void load(const T* base, long ld, int m0, int mmax, int c0, int tid, int rin = 0, int rout = 0,
                                         int roff = 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int q = tid + i * NT, m = q / CPRW, c = q % CPRW;
            const int mm = m0 + m;
            const bool ok = !CHECK || mm < mmax;
            const int mc = ok ? mm : mmax - 1;
            long gr = mc;
            if (REMAP) gr = (long)(mc / rin) * rout + (mc % rin) + roff;
            const uint4 v = *(const uint4*)(base + gr * ld + c0 + c * EPC);
            reg[i] = ok ? v : make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(char* t, int tid) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int q = tid + i * NT, m = q / CPRW, c = q % CPRW;
            *(uint4*)(t + ((long)m * Tile::LD + c * EPC) * sizeof(T)) = reg[i];
        }
    }
};

// ---------------------------------------------------------------------------------------------------
// Parameter block shared by every GEMM kernel (plain C layout; filled by the C ABI entry points).
struct GemmP {
    const void* A; long lda;
    const void* W; long ldw;
    int M, N, K;
    const float* bias;
    void* out0; long ldo0;
    void* out1; long ldo1;
    const void* aux; long ldaux;
    const float* res; long ldres;
    const void* res_t; long ldres_t;   // LayerNorm-backward row kernels: the residual gradient in the operand type T instead of f32 (see vit.hip)
    int res_mod, res_off;
    int orow_in, orow_out, orow_off;
    const float* gamma; const float* beta; float eps;
    float* mean; float* rstd;
    float* cs0; float* cs1; float* cs2;
    float* cpart;      // optional scratch for per-workgroup column-sum partials (then reduced by colpart_reduce): avoids
                       // hundreds of workgroups hammering the same few hundred addresses with float atomics (14x slower)
    int y_f32;
    int splits;        // wgrad: number of m-splits (grid.y); gemm_rowp: number of K splits of a row tile (set by its launcher)
    float* kpart;      // gemm_rowp with K splits: scratch for the partial accumulator tiles ([workgroup][MF * 12 * 512] floats), see gemm_rowp.hip
    unsigned* kcnt;    // ... and one arrival counter per row tile (zero between launches: the last arrival resets it)
    int rows_per_wg;   // row kernels: token rows owned by one workgroup (<= its tile height; 0 = the tile height), see launch_row
    unsigned* omax;    // tile kernel, EPI_NONE: largest |out| per (image, head) as f32 bits, [M / omax_rows][N / omax_hd], by atomicMax onto a zeroed buffer;
    int omax_rows, omax_hd;   // token rows per image (>= 64), columns per head (32 or 64) - the proj data gradient: the attention backward's dO scale
    // two-level batch over blockIdx.z = zo * nbi + zi (element offsets; bias/out0 only)
    int nb, nbi;
    long sAo, sAi, sWo, sWi, sOo, sOi, sBo, sBi;
};

// apply the batch offsets of blockIdx.z to a by-value copy of the parameter block
template <typename T> __device__ __forceinline__ void apply_batch(GemmP& p, int out_elem_bytes) {
    if (p.nb <= 1) return;
    const int z = blockIdx.z, zo = z / p.nbi, zi = z % p.nbi;
    p.A = (const T*)p.A + zo * p.sAo + zi * p.sAi;
    p.W = (const T*)p.W + zo * p.sWo + zi * p.sWi;
    p.out0 = (char*)p.out0 + (zo * p.sOo + zi * p.sOi) * out_elem_bytes;
    if (p.bias) p.bias += zo * p.sBo + zi * p.sBi;
}

__device__ __forceinline__ int out_row(const GemmP& p, int m) {
    return p.orow_in ? (m / p.orow_in) * p.orow_out + (m % p.orow_in) + p.orow_off : m;
}

// C/D register -> row within a 32x32 tile
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------------
// NT main loop:  acc[TM][TN] (32x32 tiles) += A[m0.., :] * W[n0.., :]^T, double-buffered LDS, one barrier
// per K tile, next tile's global loads issued before the MFMAs of the current one (write after).
// (Tried in round 2 for the row-complete kernels, which run ONE workgroup per CU, and dropped: prefetching the activation tile two K
// tiles ahead through a second register set.  Measured inside the step, split bf16: proj / fc2 + LN 118.8 -> 147 us, LN-backward
// 156 -> 190 us - the extra live registers and the longer dependency chain cost more than the HBM latency they were meant to hide.)
template <typename T, int BM, int BN, int BKB, int WM, int WN> struct NtLoop {
    static constexpr bool PIPE = MFVIT_NT_PIPE;
    static constexpr int NT = WM * WN * 64;
    static constexpr int BK = BKB / (int)sizeof(T);
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    typedef KTile<T, BM, BKB> TA;
    typedef KTile<T, BN, BKB> TB;
    static constexpr int STAGE_BYTES = TA::BYTES + TB::BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr bool SPLIT = is_split<T>::value;
    static constexpr bool HALF = SPLIT && BKB == 64;   // half a k group per tile, see KStage
    // (Tried for the two-workgroups-per-CU row kernels and dropped: the activation chunk - ONE 16-byte load per thread and tile - fetched
    // three tiles ahead through a rotating register queue, weight loads issued first so that the in-order vmcnt does not wait for the
    // youngest activation load.  hipcc's waitcnt insertion treats the loop-carried reuse of a queue register as a hazard and puts a
    // vmcnt(0..2) in front of every deep load, i.e. right behind the six weight loads of the tile - slower than the plain loop.)
    static_assert(!SPLIT || BKB == 128 || BKB == 64, "a split K tile is one 32-wide k group ([hi x 32 | lo x 32] = 128 bytes per row) or half of one");

    // the MFMAs of one K tile held in LDS (ta: A rows, tb: W rows)
    static __device__ __forceinline__ void compute(const char* ta, const char* tb, int wm, int wn, int lane, f32x16 (&acc)[TM][TN]) {
        if constexpr (HALF) {
            // one 16-wide k step per tile: MFMA step 0 of the row = hi, step 1 = lo
            typename MmaTraits<T>::frag_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) al[i] = TA::frag(ta, (wm * TM + i) * 32, 1, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bh[j] = TB::frag(tb, (wn * TN + j) * 32, 0, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i) ah[i] = TA::frag(ta, (wm * TM + i) * 32, 0, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bl[j] = TB::frag(tb, (wn * TN + j) * 32, 1, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MmaTraits<T>::mma(al[i], bh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MmaTraits<T>::mma(ah[i], bl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MmaTraits<T>::mma(ah[i], bh[j], acc[i][j]);
        } else if constexpr (SPLIT) {
            compute_hooked<false>(ta, tb, wm, wn, lane, acc, [](int) {});
        } else {
            compute_plain(ta, tb, wm, wn, lane, acc);
        }
    }
    // split bf16: the same MFMAs with `side(t)` called behind MFMA number t (0 .. 6 TM TN - 1): the interleaved main loop hangs its global
    // loads and LDS stores there, one at a time
    template <bool HOOKED, typename F>
    static __device__ __forceinline__ void compute_hooked(const char* ta, const char* tb, int wm, int wn, int lane, f32x16 (&acc)[TM][TN], F side) {
        {
            // split product: per 16-wide k step  acc += a_lo b_hi + a_hi b_lo + a_hi b_hi  (3 MFMAs from 4 fragments; the fragments
            // of step 1 are read while the 12 - 18 MFMAs of step 0 run).  MFMA steps 0, 1 of the row are the hi parts, 2, 3 the lo
            // parts (KTile<sbf16>).
            typename MmaTraits<T>::frag_t ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[0][i] = TA::frag(ta, (wm * TM + i) * 32, 0, lane);
                al[0][i] = TA::frag(ta, (wm * TM + i) * 32, 2, lane);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[0][j] = TB::frag(tb, (wn * TN + j) * 32, 0, lane);
                bl[0][j] = TB::frag(tb, (wn * TN + j) * 32, 2, lane);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        ah[1][i] = TA::frag(ta, (wm * TM + i) * 32, 1, lane);
                        al[1][i] = TA::frag(ta, (wm * TM + i) * 32, 3, lane);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bh[1][j] = TB::frag(tb, (wn * TN + j) * 32, 1, lane);
                        bl[1][j] = TB::frag(tb, (wn * TN + j) * 32, 3, lane);
                    }
                }
                if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 3 * TM * TN; ++t) {
                    const int term = t / (TM * TN), i = (t % (TM * TN)) / TN, j = t % TN;
                    acc[i][j] = MmaTraits<T>::mma(term == 0 ? al[s][i] : ah[s][i], term == 1 ? bl[s][j] : bh[s][j], acc[i][j]);
                    side(s * 3 * TM * TN + t);
                    if (HOOKED) __builtin_amdgcn_sched_barrier(0);
                }
                if (PIPE) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    static __device__ __forceinline__ void compute_plain(const char* ta, const char* tb, int wm, int wn, int lane, f32x16 (&acc)[TM][TN]) {
        {
            // fragments double-buffered by hand: the LDS reads of sub-step s+1 are issued before the MFMAs of sub-step s (left to
            // itself the compiler waits right behind each read, which at 1-2 waves per SIMD exposes the LDS latency every sub-step)
            typename MmaTraits<T>::frag_t a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = TA::frag(ta, (wm * TM + i) * 32, 0, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = TB::frag(tb, (wn * TN + j) * 32, 0, lane);
#pragma unroll
            for (int s = 0; s < TA::KSTEPS; ++s) {
                if (s + 1 < TA::KSTEPS) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[(s + 1) & 1][i] = TA::frag(ta, (wm * TM + i) * 32, s + 1, lane);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[(s + 1) & 1][j] = TB::frag(tb, (wn * TN + j) * 32, s + 1, lane);
                }
                if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MmaTraits<T>::mma(a[s & 1][i], b[s & 1][j], acc[i][j]);
                if (PIPE) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    static __device__ __forceinline__ void run(const GemmP& p, int m0, int n0, char* lds, f32x16 (&acc)[TM][TN]) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const T* A = (const T*)p.A;
        const T* W = (const T*)p.W;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int nk = p.K * elems_per<T>::value / BK;      // K tiles over the STORAGE width
        {
            KStage<T, BM, BKB, NT, HALF> sa;
            KStage<T, BN, BKB, NT, HALF> sb;
            sa.load(A, p.lda, m0, p.M, 0, tid);
            sb.load(W, p.ldw, n0, p.N, 0, tid);
            sa.store(lds, tid);
            sb.store(lds + TA::BYTES, tid);
            __syncthreads();
            int cur = 0;
            for (int kt = 0; kt < nk; ++kt) {
                const char* ta = lds + cur * STAGE_BYTES;
                const char* tb = ta + TA::BYTES;
                if (kt + 1 < nk) {
                    sa.load(A, p.lda, m0, p.M, (kt + 1) * BK, tid);
                    sb.load(W, p.ldw, n0, p.N, (kt + 1) * BK, tid);
                }
                compute(ta, tb, wm, wn, lane, acc);
                if (kt + 1 < nk) {
                    char* na = lds + (cur ^ 1) * STAGE_BYTES;
                    sa.store(na, tid);
                    sb.store(na + TA::BYTES, tid);
                }
                __syncthreads();
                cur ^= 1;
            }
        }
    }
};


// ---------------------------------------------------------------------------------------------------
// NT main loop with the register-staged tiles fetched D K-tiles ahead (16-bit element types).
//
// Why: per K tile a wave has 0.3 - 0.5 us of MFMAs, a global load under load takes 1.5 - 2 us, and NtLoop keeps ONE tile per
// workgroup in flight (issued at the top of an iteration, written to LDS at its end): with two workgroups per CU that is 64 KB in
// flight per CU where bandwidth x latency wants ~200 KB - the measured 26 - 34 % MFMA-busy of the tile GEMMs.  The VGPR file
// (512 KB per CU) is the only place to land more: D register sets per thread, tile kt + D issued at the top of iteration kt.
// How: the loads are inline asm (`global_load_dwordx4 v, voff, s[base]`: one scalar base per operand and tile, constant 32-bit
// per-thread offsets - no address VALU in the loop) and the waits are explicit counted `s_waitcnt vmcnt`, so the compiler neither
// sinks the loads into the MFMAs nor puts its own conservative vmcnt(0) on the loop-carried register reuse (what defeated the
// C++-level attempts at a deeper pipeline, see NtLoop).  The K loop is unrolled D times: static register-set indices.
template <typename T, int BM, int BN, int BKB, int WM, int WN, int D, bool IL = false> struct NtLoopDeep {
    typedef NtLoop<T, BM, BN, BKB, WM, WN> Base;
    static constexpr int NT = Base::NT, BK = Base::BK, TM = Base::TM, TN = Base::TN;
    typedef typename Base::TA TA;
    typedef typename Base::TB TB;
    static constexpr int STAGE_BYTES = Base::STAGE_BYTES, LDS_BYTES = Base::LDS_BYTES;
    static constexpr bool HALF = Base::HALF;
    static constexpr int CPR = BKB / 16;
    static constexpr int NA = BM * CPR / NT, NB = BN * CPR / NT, NL = NA + NB;      // 16-byte loads per thread and tile
    static_assert(sizeof(T) == 2 && BM * CPR % NT == 0 && BN * CPR % NT == 0, "16-bit tiles whose chunks divide over the block");
    static_assert(NL * (D - 1) <= 63, "vmcnt is a 6-bit counter");
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

    static __device__ __forceinline__ void gload(u32x4& dst, const char* sbase, unsigned voff) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase));
    }
    // wait until at most N of this thread's loads are outstanding, then hand the registers of one set back to the compiler
    template <int N> static __device__ __forceinline__ void wait_set(u32x4 (&r)[NL]) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
#pragma unroll
        for (int i = 0; i < NL; ++i) asm volatile("" : "+v"(r[i]));
    }
    // byte offset of K tile kt inside an operand row
    static __device__ __forceinline__ long tile_off(int kt) {
        const int k0 = kt * BK;
        return (long)(HALF ? (k0 & ~63) + ((k0 >> 5) & 1) * 16 : k0) * (long)sizeof(T);
    }

    // requires: K tiles a multiple of D (>= D), (rows - 1) * ld * 2 + 128 < 2^32 for both operands (checked by the launcher)
    static __device__ __forceinline__ void run(const GemmP& p, int m0, int n0, char* lds, f32x16 (&acc)[TM][TN]) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int nk = p.K * elems_per<T>::value / BK;
        unsigned va[NA], vb[NB];
        int ra[NA], rb[NB];        // LDS (row, chunk) of every staged chunk: row in the low bits is implied by q
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + i * NT, row = q / CPR, c = q % CPR;
            int gr = m0 + row;
            gr = gr < p.M ? gr : p.M - 1;
            const int co = HALF ? (c >> 1) * 32 + (c & 1) * 8 : c * 8;
            va[i] = (unsigned)gr * (unsigned)p.lda * 2u + (unsigned)co * 2u;
            ra[i] = TA::off(row, c);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = tid + i * NT, row = q / CPR, c = q % CPR;
            int gr = n0 + row;
            gr = gr < p.N ? gr : p.N - 1;
            const int co = HALF ? (c >> 1) * 32 + (c & 1) * 8 : c * 8;
            vb[i] = (unsigned)gr * (unsigned)p.ldw * 2u + (unsigned)co * 2u;
            rb[i] = TA::BYTES + TB::off(row, c);
        }
        const char* gA = (const char*)p.A;
        const char* gW = (const char*)p.W;
        u32x4 st[D][NL];
        auto issue = [&](u32x4 (&r)[NL], int kt) {
            const long off = tile_off(kt);
#pragma unroll
            for (int i = 0; i < NB; ++i) gload(r[NA + i], gW + off, vb[i]);
#pragma unroll
            for (int i = 0; i < NA; ++i) gload(r[i], gA + off, va[i]);
        };
        auto put = [&](const u32x4 (&r)[NL], char* stage) {
#pragma unroll
            for (int i = 0; i < NA; ++i) *(u32x4*)(stage + ra[i]) = r[i];
#pragma unroll
            for (int i = 0; i < NB; ++i) *(u32x4*)(stage + rb[i]) = r[NA + i];
        };
#pragma unroll
        for (int d = 0; d < D; ++d) issue(st[d], d);
        wait_set<NL*(D - 1)>(st[0]);
        put(st[0], lds);
        __syncthreads();
        int cur = 0;
        // steady state: tile kt + D exists.  Every wait is ONE straight-line statement with a constant count (with a branch per count
        // the compiler merges the register sets through copies placed BEFORE the wait, i.e. it reads registers still in flight).
        int kt0 = 0;
        for (; kt0 + D < nk; kt0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const char* ta = lds + cur * STAGE_BYTES;
                if constexpr (IL && Base::SPLIT && !HALF && D == 2) {
                    // interleaved form: the loads of tile kt + 2 go out one behind each of the first MFMAs, then the wait for tile kt + 1
                    // (requested a whole iteration ago), then its LDS stores one behind each of the later MFMAs - instead of a burst of
                    // loads in front of the MFMAs and a burst of ds_write_b128 (13 LDS-path cycles each) behind them, during which this
                    // workgroup's matrix pipe share idles
                    constexpr int NMF = 6 * Base::TM * Base::TN;     // MFMAs per K tile and wave
                    static_assert(NL <= NMF / 2, "one load / one store per MFMA gap");
                    char* nxt = lds + (cur ^ 1) * STAGE_BYTES;
                    const long off = tile_off(kt0 + d + D);
                    Base::template compute_hooked<true>(ta, ta + TA::BYTES, wm, wn, lane, acc, [&](int t) __attribute__((always_inline)) {
                        if (t < NL) {
                            if (t < NB) gload(st[d][NA + t], gW + off, vb[t]);
                            else gload(st[d][t - NB], gA + off, va[t - NB]);
                        }
                        if (t == NMF / 2 - 1) wait_set<NL*(D - 1)>(st[(d + 1) % D]);
                        if (t >= NMF / 2 && t < NMF / 2 + NL) {
                            const int i = t - NMF / 2;
                            if (i < NA) *(u32x4*)(nxt + ra[i]) = st[(d + 1) % D][i];
                            else *(u32x4*)(nxt + rb[i - NA]) = st[(d + 1) % D][i];
                        }
                    });
                    __syncthreads();
                    cur ^= 1;
                } else {
                issue(st[d], kt0 + d + D);                   // set d is free: tile kt0 + d went to LDS one iteration ago
                __builtin_amdgcn_sched_barrier(0);
                Base::compute(ta, ta + TA::BYTES, wm, wn, lane, acc);
                wait_set<NL*(D - 1)>(st[(d + 1) % D]);      // outstanding, in order: tiles kt + 1 .. kt + D
                put(st[(d + 1) % D], lds + (cur ^ 1) * STAGE_BYTES);
                __syncthreads();
                cur ^= 1;
                }
            }
        }
        // the last D tiles (nk % D == 0: tile nk - D + d sits in set d): nothing left to issue, the counts run down
        auto tail = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            const char* ta = lds + cur * STAGE_BYTES;
            Base::compute(ta, ta + TA::BYTES, wm, wn, lane, acc);
            if constexpr (d + 1 < D) {
                wait_set<NL*(D - 2 - d)>(st[d + 1]);
                put(st[d + 1], lds + (cur ^ 1) * STAGE_BYTES);
            }
            __syncthreads();
            cur ^= 1;
        };
        tail(std::integral_constant<int, 0>());
        if constexpr (D > 1) tail(std::integral_constant<int, 1>());
        if constexpr (D > 2) tail(std::integral_constant<int, 2>());
        static_assert(D <= 3, "tail written out for D <= 3");
    }
};

}  // namespace mfvit
