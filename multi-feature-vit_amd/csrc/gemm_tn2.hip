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
//   * a split's last stage may be partial: its rows past the end are fetched clamped (finite data) and the A-operand rows are
//     zeroed in LDS before use, so they add nothing.
#include <type_traits>
#include "gemm.cuh"
#include "kernels.h"
#include "prof.h"

namespace mfvit {

namespace {

constexpr int T2_BN = 128, T2_BK = 128, T2_KR = 64, T2_NS = 3;
constexpr int T2_TILE = T2_KR * 256;            // 16 KB per operand and stage
constexpr int T2_STAGE = 2 * T2_TILE;           // 32 KB
constexpr int T2_LPS = 8;                       // LDS-DMA instructions per wave and stage (4 per operand)

template <int N> __device__ __forceinline__ void t2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void t2_glds16(const void* gsrc, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_off)
                 : "memory");
}

// transposed fragment of the swizzled image: lane holds column (rowbase + lane & 31), elements k = 16 s + 8 (lane >> 5) .. +8
template <typename T> __device__ __forceinline__ typename Vec8<T>::type t2_frag(const char* t, int rowbase, int s, int lane) {
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
    const int k0 = 16 * s + 8 * h + q;                       // (k0 & 3) == q, also for k0 + 4
    const int col = rowbase + 16 * g1 + 4 * pp;
    const char* a = t + k0 * 256 + (((col >> 3) ^ (4 * q)) << 4) + (col & 7) * 2;
    typedef __attribute__((address_space(3))) s16x4* lptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a + 4 * 256));
    union { struct { s16x4 a, b; } s; typename Vec8<T>::type v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

// T = bf16 | f16 | sbf16.  Split tensors (sbf16): the tile grid runs over the STORAGE columns; a wave's four accumulators are the
// hi*hi, hi*lo, lo*hi (lo*lo skipped) blocks of one 32 x 32 logical tile and are summed in the epilogue (see gemm_tn_kernel).
template <typename T, bool CS>
__global__ __launch_bounds__(256) void gemm_tn_glds_kernel(GemmP p) {
    constexpr bool SPLIT = is_split<T>::value;
    constexpr int EP = elems_per<T>::value;
    typedef typename Vec8<T>::type frag_t;
    typedef typename Vec4<T>::elem E16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntk = p.K * EP / T2_BK;
    const int tiles = ntk * (p.N * EP / T2_BN);
    const int lin = xcd_remap(blockIdx.x, gridDim.x);        // each XCD gets a contiguous run of the split-major block order
    const int split = lin / tiles, tile = lin % tiles;
    const int n0 = (tile / ntk) * T2_BN, k0 = (tile % ntk) * T2_BK;
    int chunk = (p.M + p.splits - 1) / p.splits;
    chunk = (chunk + T2_KR - 1) / T2_KR * T2_KR;
    const int mbeg = split * chunk;
    const int mend = min(p.M, mbeg + chunk);
    if (mbeg >= mend) return;
    const int nst = (mend - mbeg + T2_KR - 1) / T2_KR;
    const E16* A = (const E16*)p.A;
    const E16* X = (const E16*)p.W;

    // LDS-DMA g = wave + 4 i (i = 0..3) of an operand fills rows 4 g .. 4 g + 3: lane -> row 4 g + (lane >> 4), chunk position
    // lane & 15, which receives source chunk (lane & 15) ^ (4 * (row & 3)) = (lane & 15) ^ (4 * (lane >> 4))
    const int lrow = 4 * wave + (lane >> 4);                 // + 16 i
    const int lch = (lane & 15) ^ (4 * (lane >> 4));
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)wave * 1024u);
    auto issue = [&](int st, int slot) {
        const unsigned sa = lbase + (unsigned)slot * T2_STAGE, sb = sa + T2_TILE;
        const int mrow = mbeg + st * T2_KR + lrow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = mrow + 16 * i;
            m = m < p.M ? m : p.M - 1;                        // never read past the tensor (clamped rows are zeroed below)
            t2_glds16(A + (long)m * p.lda + n0 + lch * 8, sa + i * 4096);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = mrow + 16 * i;
            m = m < p.M ? m : p.M - 1;
            t2_glds16(X + (long)m * p.ldw + k0 + lch * 8, sb + i * 4096);
        }
    };

    f32x16 acc[2][2], bacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[i][r] = 0.f;
    }
    frag_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (E16)1.0f;
    const bool do_cs = CS && k0 == 0 && wn == 0;             // column sums of A (= bias gradient): k-tile 0, wn 0 waves

    // the bias-column-sum MFMAs are selected ONCE per wave (template flag), not per k-step: a branch inside the hot loop splits it
    // into basic blocks and serialises the ds_read -> MFMA pipeline (measured: 72 vs 52 us on the kernels that carry a bias sum)
    auto main_loop = [&](auto cs_tag) {
        constexpr bool WITH_CS = decltype(cs_tag)::value;
        issue(0, 0);
        if (nst > 1) issue(1, 1);
        int slot = 0;
        for (int st = 0; st < nst; ++st) {
            if (st + 1 < nst) t2_wait_vm<T2_LPS>();              // stage st landed; stage st+1 may stay in flight
            else t2_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (st + 2 < nst) issue(st + 2, slot == 0 ? 2 : slot - 1);      // (st + 2) % 3: read last in step st-1, free since the barrier
            char* ta = lds + slot * T2_STAGE;
            const char* tb = ta + T2_TILE;
            const int valid = mend - mbeg - st * T2_KR;          // rows of this stage inside the split
            if (valid < T2_KR) {                                 // partial last stage: zero the A rows past the end (block-uniform branch)
                for (int q = tid; q < (T2_KR - valid) * 16; q += 256)
                    *(uint4*)(ta + (valid + (q >> 4)) * 256 + (q & 15) * 16) = make_uint4(0, 0, 0, 0);
                __syncthreads();
            }
            frag_t a[2][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[0][i] = t2_frag<T>(ta, (wm * 2 + i) * 32, 0, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[0][j] = t2_frag<T>(tb, (wn * 2 + j) * 32, 0, lane);
#pragma unroll
            for (int s = 0; s < T2_KR / 16; ++s) {
                if (s + 1 < T2_KR / 16) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[(s + 1) & 1][i] = t2_frag<T>(ta, (wm * 2 + i) * 32, s + 1, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[(s + 1) & 1][j] = t2_frag<T>(tb, (wn * 2 + j) * 32, s + 1, lane);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (!(SPLIT && i == 1 && j == 1)) acc[i][j] = MmaTraits<T>::mma(a[s & 1][i], b[s & 1][j], acc[i][j]);
                if constexpr (WITH_CS) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) bacc[i] = MmaTraits<T>::mma(a[s & 1][i], ones, bacc[i]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot are done before it reaches the next barrier
            slot = slot == T2_NS - 1 ? 0 : slot + 1;
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
    constexpr int NI = SPLIT ? 1 : 2;
    const int nw = SPLIT ? n0 / 2 + wm * 32 : n0 + wm * 64, kw = SPLIT ? k0 / 2 + wn * 32 : k0 + wn * 64;   // LOGICAL origin of this wave's tile
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
    if (CS) {
        if (do_cs && (lane & 31) == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) atomicAdd(p.cs0 + nw + i * 32 + acc_row(r, lane), bacc[i][r]);
        }
    }
}

}  // namespace

static int t2_ep(int dtype) { return dtype == MFVIT_BF16X3 ? 2 : 1; }
bool gemm_tn_glds_supported(int dtype, const GemmP& p) {
    static const int on = [] { const char* e = getenv("MFVIT_TN_GLDS"); return e ? atoi(e) : 1; }();
    if (!on || (dtype != MFVIT_BF16 && dtype != MFVIT_BF16X3 && dtype != MFVIT_F16) || p.nb > 1 || p.orow_in || p.cpart) return false;
    const int ep = t2_ep(dtype);
    if (p.N * ep % T2_BN || p.K * ep % T2_BK || p.M < 4096) return false;
    if (p.lda % 8 || p.ldw % 8) return false;
    return true;
}

template <typename T> static int launch_t2(GemmP p, hipStream_t st) {
    constexpr int EP = elems_per<T>::value;
    const int tiles = (p.N * EP / T2_BN) * (p.K * EP / T2_BK);
    if (p.splits <= 0) {
        static const int target = [] { const char* e = getenv("MFVIT_TN2_TARGET"); return e ? atoi(e) : 256; }();
        // one workgroup per CU (96 KB of LDS): tiles x splits <= 256 when the tiles allow it; more tiles than CUs (split tensors:
        // 4 x the storage tiles) run in whole rounds of one split each
        int s = target / tiles;
        const int maxs = (p.M + 4 * T2_KR - 1) / (4 * T2_KR);
        if (EP == 2 && 2 * tiles > target) {
            // more than half a round of tiles: the smallest split count (<= 8) whose grid fills >= 90 % of its rounds of 256 CUs
            double best = 0.0;
            for (int c = 1; c <= 8 && c <= maxs; ++c) {
                const int g = tiles * c, rounds = (g + target - 1) / target;
                const double eff = (double)g / ((double)rounds * target);
                if (eff > best + 1e-9) { best = eff; s = c; }
                if (eff >= 0.9) break;
            }
        }
        p.splits = s < 1 ? 1 : (s > maxs ? maxs : s);
    }
    constexpr int bytes = T2_NS * T2_STAGE;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr = true;
    }
    ProfScope ps(PROF_GEMM_TN, 2.0 * p.M * p.N * p.K, 0, st);
    if (p.cs0)
        MFVIT_LAUNCH((gemm_tn_glds_kernel<T, true>), dim3(tiles * p.splits), dim3(256), bytes, st, p);
    else
        MFVIT_LAUNCH((gemm_tn_glds_kernel<T, false>), dim3(tiles * p.splits), dim3(256), bytes, st, p);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

int gemm_tn_glds(int dtype, GemmP p, hipStream_t st) {
    if (dtype == MFVIT_BF16) return launch_t2<bf16>(p, st);
    if (dtype == MFVIT_BF16X3) return launch_t2<sbf16>(p, st);
    if (dtype == MFVIT_F16) return launch_t2<f16>(p, st);
    return MFVIT_EINVAL;
}

}  // namespace mfvit
