// Persistent bf16 / split-bf16 NT GEMM for the short-K (K = 384) linears of the encoder:  out[M][N] = epi(A[M][K] W[N][K]^T).
// (split bf16, T = sbf16: 128-byte stage rows = one [hi x 32 | lo x 32] k group; per 16-wide k step 12 fragments feed 24 MFMAs per
// wave - a_hi b_lo + a_lo b_hi + a_hi b_hi for each of the 4 x 2 tiles - so a 48 KB stage carries 3 x the MFMA work of the bf16 one and
// the LDS-DMA issue / epilogue shares that bound the bf16 kernel shrink accordingly; outputs leave as hi / lo pairs in the I32 layout.)
//
// Why a second tile kernel: at K = 384 a 128x128 tile's main loop is only six K-steps, so pipeline fill, the epilogue and the
// L2 re-fetch of both operands (1/64 B per flop) dominate; with 64x64 per wave the LDS fragment traffic equals the MFMA time.
//   * 256 x 128 workgroup tile, 4 waves as 2 x 2, 128 x 64 per wave (4 x 2 MFMA tiles of 32x32x16): 6 KB of fragment reads per
//     8 MFMAs (75 % of the LDS rate instead of 100 %), operand traffic 1/85 B per flop.
//   * one workgroup per CU, persistent over its tiles; the K-steps of ALL its tiles form one flat pipeline: a 3-slot LDS ring
//     (3 x 48 KB) filled by LDS-DMA (global_load_lds_dwordx4) two steps ahead, so the first stages of tile t+1 are in flight
//     while tile t finishes and runs its epilogue - no per-tile fill bubble.
//   * MFMA operands swapped (W fragment as A, activation fragment as B): the accumulator holds out^T, i.e. a lane owns ONE
//     output row and 4 consecutive columns per register quad -> bias comes in as float4, outputs leave as packed bf16x4.
//   * tiles are dealt so that the 32 workgroups of an XCD work on neighbouring tiles of the n-fastest order at the same time
//     (they share activation rows in that XCD's L2).
// Synchronisation per K-step: counted s_waitcnt vmcnt (never 0 while a younger stage is in flight) -> raw s_barrier -> issue
// the stage two steps ahead into the slot every wave has just finished reading -> MFMAs.  vmcnt retires in order, so
// "at most LPS operations outstanding" proves the stage before the youngest one has landed, whatever the epilogue stored
// in between (cdna_hip_programming.md 5, "Pipelining across barriers").
#include "gemm.cuh"
#include "kernels.h"
#include "prof.h"

namespace mfvit {

namespace {

// staging accesses of the epilogue write packed bf16x4 and read 16-byte chunks of the same LDS bytes: both through may_alias types
// (type-based alias analysis would otherwise let the compiler reorder the differently typed stores and loads)
typedef uint4 __attribute__((may_alias)) stg_u4;
typedef bf16x4 __attribute__((may_alias)) stg_b4;

constexpr int PBM = 256, PBN = 128, PNS = 3;
constexpr int PTM = 4, PTN = 2;
constexpr int PMAXN = 1536;                              // bias vector kept in LDS

// BKB = bytes of K per LDS row: 128 (BK 64, 48 KB stages, one workgroup per CU) or 64 (BK 32, 24 KB stages, two per CU)
template <int BKB> struct PCfg {
    typedef KTile<bf16, PBM, BKB> TA;
    typedef KTile<bf16, PBN, BKB> TB;
    static constexpr int BK = BKB / 2;
    static constexpr int CPR = BKB / 16;                              // 16-B chunks per row
    static constexpr int STAGE = TA::BYTES + TB::BYTES;
    static constexpr int LA = PBM * CPR / 256, LB = PBN * CPR / 256;  // LDS-DMA instructions per thread per stage
    static constexpr int LPS = LA + LB;
    static constexpr int RPI = 256 / CPR;                             // tile rows covered by one LDS-DMA of the block
    static constexpr int KS = TA::KSTEPS;
    static constexpr int LDS_BYTES = PNS * STAGE + PMAXN * 4;
};

// vmcnt is a 6-bit counter: a bound above 63 is clamped (waiting for fewer outstanding operations is stricter, never wrong)
template <int N> __device__ __forceinline__ void wait_vm_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory"); }

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_off)
                 : "memory");
}

template <typename T, int EPI, int BKB>
__global__ __launch_bounds__(256, BKB == 64 ? 2 : 1) void gemm_nt_pers_kernel(GemmP p, int ntm, int ntn) {
    constexpr bool SPLIT = is_split<T>::value;
    constexpr int EP = elems_per<T>::value;
    static_assert(!SPLIT || BKB == 128, "a split stage row is one whole [hi | lo] k group");
#ifdef MFVIT_PERS_DBG
    constexpr int dbg = MFVIT_PERS_DBG;   // build-time experiment switch: 1 no epilogue, 2 no MFMAs, 4 no in-loop LDS-DMA
#else
    constexpr int dbg = 0;
#endif
    typedef PCfg<BKB> C;
    typedef typename C::TA TA;
    typedef typename C::TB TB;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x, ntiles = ntm * ntn;
    // chunk-local slot of this workgroup: the workgroups of one XCD (blockIdx & 7) get consecutive tiles of every round
    const int cslot = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
    if (cslot >= ntiles) return;
    const int nloc = (ntiles - cslot + G - 1) / G;
    const int nk = p.K * EP / C::BK;          // K steps over the STORAGE width
    const int S = nloc * nk;

    // ---- issue side state: byte offsets of this thread's 16-B chunks (row of the tile, swizzled chunk), per tile
    const int crow = tid / C::CPR;                                            // + RPI i
    const int cchunk = (tid % C::CPR) ^ ((crow / TA::RPW) % C::CPR);          // source chunk landing on LDS chunk position tid % CPR
    static_assert((C::RPI / TA::RPW) % C::CPR == 0, "row step of successive LDS-DMAs must not change the swizzle");
    unsigned offA[C::LA], offB[C::LB];
    int it_tile = cslot, it_kt = 0;
    auto set_tile_offsets = [&](int tile) {
        const int m0 = (tile / ntn) * PBM, n0 = (tile % ntn) * PBN;
#pragma unroll
        for (int i = 0; i < C::LA; ++i) {
            int gr = m0 + crow + C::RPI * i;
            gr = gr < p.M ? gr : p.M - 1;
            offA[i] = (unsigned)gr * (unsigned)(p.lda * 2) + cchunk * 16;
        }
#pragma unroll
        for (int i = 0; i < C::LB; ++i) offB[i] = (unsigned)(n0 + crow + C::RPI * i) * (unsigned)(p.ldw * 2) + cchunk * 16;
    };
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)wave * 1024u);
    auto issue = [&](int slot) {
        if (it_kt == 0) set_tile_offsets(it_tile);
        const unsigned sa = lbase + (unsigned)slot * C::STAGE, sb = sa + TA::BYTES;
        const unsigned kb = (unsigned)it_kt * BKB;
#pragma unroll
        for (int i = 0; i < C::LA; ++i) glds16((const char*)p.A + (offA[i] + kb), sa + i * 4096);
#pragma unroll
        for (int i = 0; i < C::LB; ++i) glds16((const char*)p.W + (offB[i] + kb), sb + i * 4096);
        if (++it_kt == nk) { it_kt = 0; it_tile += G; }
    };

    // bias vector -> LDS once (read back with ds_read in the epilogues: no vector-memory loads inside the pipelined loop,
    // whose compiler-inserted vmcnt(0) waits would drain the in-flight LDS-DMA stages)
    float* lbias = (float*)(lds + PNS * C::STAGE);
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU)
        for (int n = tid; n < p.N; n += 256) lbias[n] = p.bias ? p.bias[n] : 0.f;

    // fragment registers, double-buffered by hand: the reads of sub-step ks+1 are issued BEFORE the MFMAs of sub-step ks
    // (one wave per SIMD cannot hide LDS latency any other way; left to itself the compiler waits right behind each read).
    // NSUB sub-steps per stage: bf16 one per 16-wide MFMA step of the row (KS); split: the 2 steps of the hi half, each with its lo
    // twin (MFMA steps u and u + 2 of the row), 3 MFMAs per tile.
    constexpr int NSUB = SPLIT ? 2 : C::KS;
    static_assert(NSUB % 2 == 0, "fragment double buffer assumes an even number of sub-steps");
    constexpr int NP = SPLIT ? 2 : 1;                   // fragment parts (hi, lo)
    bf16x8 fa[2][NP][PTM], fb[2][NP][PTN];
    auto load_frags = [&](int slot, int ks, bf16x8 (&a)[NP][PTM], bf16x8 (&b)[NP][PTN]) {
        if constexpr ((dbg & 16) != 0) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int j = 0; j < PTN; ++j) asm volatile("" : "+v"(b[q][j]));
#pragma unroll
                for (int i = 0; i < PTM; ++i) asm volatile("" : "+v"(a[q][i]));
            }
            return;
        }
        const char* ta = lds + slot * C::STAGE;
        const char* tb = ta + TA::BYTES;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int j = 0; j < PTN; ++j) b[q][j] = TB::frag(tb, (wn * PTN + j) * 32, ks + 2 * q, lane);
#pragma unroll
            for (int i = 0; i < PTM; ++i) a[q][i] = TA::frag(ta, (wm * PTM + i) * 32, ks + 2 * q, lane);
        }
    };

    // Two workgroups share a CU and do identical work: started together they would run their main loops and their epilogues
    // in lockstep.  The second wave of workgroups starts half a tile late so that one's epilogue overlaps the other's MFMAs.
    if (p.y_f32 > 0 && (int)blockIdx.x >= G / 2)
        for (int z = 0; z < p.y_f32; ++z) __builtin_amdgcn_s_sleep(16);

    issue(0);
    if (S > 1) issue(1);
    if (S > 2) issue(2);
    if (S > 2) wait_vm_le<2 * C::LPS>();
    else if (S > 1) wait_vm_le<C::LPS>();
    else wait_vm_le<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    load_frags(0, 0, fa[0], fb[0]);

#ifdef MFVIT_PERS_TRACE
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_readcyclecounter();
#define TICK(i) do { const long long t__ = __builtin_readcyclecounter(); tacc[i] += t__ - tlast; tlast = t__; } while (0)
#else
#define TICK(i) do { } while (0)
#endif
    constexpr int NST = (EPI == EPI_BIAS_GELU ? 2 : 1) * PTM * 4 * EP;   // global stores per lane and epilogue
    int slot = 0, s = 0, since = 2;                                  // since = steps since the last epilogue
    int tile = cslot;
    for (int t = 0; t < nloc; ++t, tile += G) {
        // the accumulators live for ONE tile (zeroed here, consumed by the epilogue below): no loop-carried copies of 128 registers
        f32x16 acc[PTM][PTN];
#pragma unroll
        for (int i = 0; i < PTM; ++i)
#pragma unroll
            for (int j = 0; j < PTN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        auto mfmas = [&](const bf16x8 (&a)[NP][PTM], const bf16x8 (&b)[NP][PTN]) {
            if constexpr ((dbg & 2) != 0) return;
            if constexpr (SPLIT) {   // same term order as NtLoop::compute (lo x hi, hi x lo, hi x hi): bit-identical accumulators
#pragma unroll
                for (int i = 0; i < PTM; ++i)
#pragma unroll
                    for (int j = 0; j < PTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[1][i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < PTM; ++i)
#pragma unroll
                    for (int j = 0; j < PTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][j], a[0][i], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < PTM; ++i)
#pragma unroll
                for (int j = 0; j < PTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][j], a[0][i], acc[i][j], 0, 0, 0);
        };
        auto front = [&]() {   // all sub-steps but the last: prefetch the next fragments, then the MFMAs of this one
#pragma unroll
            for (int ks = 0; ks < NSUB - 1; ++ks) {
                load_frags(slot, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(fa[ks & 1], fb[ks & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // Every LDS read of this slot has been issued: retire them, make sure stage s+1 has landed, meet the other waves.
        // Younger operations that may stay in flight: stage s+2 (if there is one) and, in the step right after an epilogue,
        // that epilogue's NST stores (issued after stage s+1, before stage s+2).  vmcnt retires in order.
        auto sync = [&]() {
            const bool young = s + 2 < S;
            TICK(0);
            if (young && since == 0) wait_vm_le<C::LPS + NST>();
            else if (young) wait_vm_le<C::LPS>();
            else if (since == 0) wait_vm_le<NST>();
            else wait_vm_le<0>();
            TICK(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TICK(2);
            if constexpr (!(dbg & 8)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            TICK(3);
        };
        for (int kt = 0; kt < nk - 1; ++kt) {
            front();
            sync();
            const int nslot = slot == PNS - 1 ? 0 : slot + 1;
            if (s + 3 < S && !(dbg & 4)) issue(slot);        // refill the slot every wave has finished reading, three steps ahead
            load_frags(nslot, 0, fa[0], fb[0]);              // first fragments of the next step, hidden behind the last MFMAs
            __builtin_amdgcn_sched_barrier(0);
            TICK(4);
            mfmas(fa[(NSUB - 1) & 1], fb[(NSUB - 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            TICK(5);
            slot = nslot;
            ++since;
            ++s;
        }
        // ---- last step of the tile: this wave's own pieces of the free slot first serve as the epilogue's staging buffer
        front();
        sync();
        TICK(4);
        mfmas(fa[(NSUB - 1) & 1], fb[(NSUB - 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        TICK(5);
        since = 0;
        {
            // acc[i][j][r] = out[m][n], m = m0 + wm 128 + 32 i + (lane & 31), n = n0 + wn 64 + 32 j + 8 (r >> 2) + 4 (lane >> 5) + (r & 3).
            // Staged per wave through LDS (32 rows x 128 B in this wave's own four 1-KiB LDS-DMA pieces of the free slot, 16-B
            // chunks XOR-swizzled by (row >> 1) & 7) so that the tile leaves as full 128-B rows: 16-B stores, 8 lanes per row.
            const int m0 = (tile / ntn) * PBM, n0 = (tile % ntn) * PBN;
            if constexpr ((dbg & 1) != 0) {
#pragma unroll
                for (int i = 0; i < PTM; ++i)
#pragma unroll
                    for (int j = 0; j < PTN; ++j) asm volatile("" ::"v"(acc[i][j]));
            } else {
                char* stg = lds + slot * C::STAGE + wave * 1024;
                const int mrow = lane & 31, h = lane >> 5;
                const int nb = n0 + wn * 64 + 4 * h;
                if constexpr (SPLIT) {
                    // split outputs: a row of this wave's 32 x 64 logical tile is 2 groups x [hi x 32 | lo x 32] = 256 B = 16 chunks of 16 B
                    // (chunk 8 j + g: hi of columns 32 j + 8 g .. + 8, chunk 8 j + 4 + g: their lo parts), staged in this wave's own EIGHT
                    // 1-KiB LDS-DMA pieces of the free slot's A region (4 rows per piece), chunk position XOR (row & 15).
                    char* wrow = stg + (mrow >> 2) * 4096 + (mrow & 3) * 256 + 8 * h;
                    const int wsw = mrow & 15;
                    auto flush = [&](int i, void* out, long ldo) {
                        asm volatile("" ::: "memory");   // lanes exchange data through LDS inside one wave: compiler fence (see the bf16 path)
#pragma unroll
                        for (int q8 = 0; q8 < 8; ++q8) {
                            const int q = q8 * 64 + lane, row = q >> 4, ch = q & 15;
                            const uint4 v = *(const stg_u4*)(stg + (row >> 2) * 4096 + (row & 3) * 256 + 16 * (ch ^ (row & 15)));
                            int m = m0 + wm * 128 + 32 * i + row;
                            m = m < p.M ? m : p.M - 1;
                            *(uint4*)((bf16*)out + (long)m * ldo + (n0 + wn * 64) * 2 + 8 * ch) = v;
                        }
                    };
#pragma unroll
                    for (int i = 0; i < PTM; ++i) {
                        bf16x4 second[PTN][4], second_lo[PTN][4];
#pragma unroll
                        for (int j = 0; j < PTN; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const float4 bq = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? *(const float4*)(lbias + nb + 32 * j + 8 * g)
                                                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                                float v[4] = {acc[i][j][4 * g] + bq.x, acc[i][j][4 * g + 1] + bq.y, acc[i][j][4 * g + 2] + bq.z,
                                              acc[i][j][4 * g + 3] + bq.w};
                                bf16x4 w0, w1;
                                if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        float cdf, ex;
                                        gelu_parts_fast(v[e], cdf, ex);
                                        bf16 hh, ll;
                                        split2(fmaf(v[e] * 0.39894228040143267794f, ex, cdf), hh, ll);             // gelu'(pre)
                                        w0[e] = hh; w1[e] = ll;
                                        split2(v[e] * cdf, hh, ll);                                                // gelu(pre)
                                        second[j][g][e] = hh; second_lo[j][g][e] = ll;
                                    }
                                } else if (EPI == EPI_GELU_BWD) {
                                    int m = m0 + wm * 128 + 32 * i + mrow;
                                    m = m < p.M ? m : p.M - 1;
                                    const bf16* ap = (const bf16*)p.aux + (long)m * p.ldaux + (n0 + wn * 64 + 32 * j) * 2 + 8 * g + 4 * h;
                                    const bf16x4 ah = *(const bf16x4*)ap, al = *(const bf16x4*)(ap + 32);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        bf16 hh, ll;
                                        split2(v[e] * ((float)ah[e] + (float)al[e]), hh, ll);
                                        w0[e] = hh; w1[e] = ll;
                                    }
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        bf16 hh, ll;
                                        split2(v[e], hh, ll);
                                        w0[e] = hh; w1[e] = ll;
                                    }
                                }
                                *(stg_b4*)(wrow + 16 * ((8 * j + g) ^ wsw)) = w0;
                                *(stg_b4*)(wrow + 16 * ((8 * j + 4 + g) ^ wsw)) = w1;
                            }
                        flush(i, p.out0, p.ldo0);
                        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                            for (int j = 0; j < PTN; ++j)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    *(stg_b4*)(wrow + 16 * ((8 * j + g) ^ wsw)) = second[j][g];
                                    *(stg_b4*)(wrow + 16 * ((8 * j + 4 + g) ^ wsw)) = second_lo[j][g];
                                }
                            flush(i, p.out1, p.ldo1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                char* wrow = stg + (mrow >> 3) * 4096 + (mrow & 7) * 128 + 8 * h;
                const int wsw = (mrow >> 1) & 7;
                auto flush = [&](int i, void* out, long ldo) {   // staged 32 x 64 tile -> global, 4 x 16 B per lane
                    // lanes exchange data through LDS inside ONE wave (hardware executes a wave's LDS instructions in order, no barrier
                    // needed) - but the compiler reasons per thread: without this fence it forwards a lane's earlier load of the same
                    // address past the OTHER lanes' stores (observed: the reads sunk under the writers' exec mask)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int q = q4 * 64 + lane, row = q >> 3, ch = q & 7;
                        const uint4 v = *(const stg_u4*)(stg + (row >> 3) * 4096 + (row & 7) * 128 + 16 * (ch ^ ((row >> 1) & 7)));
                        int m = m0 + wm * 128 + 32 * i + row;
                        m = m < p.M ? m : p.M - 1;   // rows past M replicate row M-1 exactly (clamped A loads): identical duplicate
                        *(uint4*)((bf16*)out + (long)m * ldo + n0 + wn * 64 + 8 * ch) = v;   // stores, data-independent store count
                    }
                };
#pragma unroll
                for (int i = 0; i < PTM; ++i) {
                    bf16x4 second[PTN][4];
#pragma unroll
                    for (int j = 0; j < PTN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 bq = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? *(const float4*)(lbias + nb + 32 * j + 8 * g)
                                                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                            float v[4] = {acc[i][j][4 * g] + bq.x, acc[i][j][4 * g + 1] + bq.y, acc[i][j][4 * g + 2] + bq.z,
                                          acc[i][j][4 * g + 3] + bq.w};
                            bf16x4 w0;
                            if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float cdf, ex;
                                    gelu_parts_fast(v[e], cdf, ex);
                                    w0[e] = (bf16)fmaf(v[e] * 0.39894228040143267794f, ex, cdf);   // gelu'(pre)
                                    second[j][g][e] = (bf16)(v[e] * cdf);                          // gelu(pre)
                                }
                            } else if (EPI == EPI_GELU_BWD) {
                                int m = m0 + wm * 128 + 32 * i + mrow;
                                m = m < p.M ? m : p.M - 1;
                                const bf16x4 ax = *(const bf16x4*)((const bf16*)p.aux + (long)m * p.ldaux + nb + 32 * j + 8 * g);
#pragma unroll
                                for (int e = 0; e < 4; ++e) w0[e] = (bf16)(v[e] * (float)ax[e]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) w0[e] = (bf16)v[e];
                            }
                            *(stg_b4*)(wrow + 16 * ((4 * j + g) ^ wsw)) = w0;
                        }
                    flush(i, p.out0, p.ldo0);
                    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                        for (int j = 0; j < PTN; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) *(stg_b4*)(wrow + 16 * ((4 * j + g) ^ wsw)) = second[j][g];
                        flush(i, p.out1, p.ldo1);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the four row groups apart: interleaved, their temporaries spill
                }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's staging reads are done: its pieces may be refilled
        __builtin_amdgcn_sched_barrier(0);
        const int nslot = slot == PNS - 1 ? 0 : slot + 1;
        TICK(6);
        if (s + 3 < S && !(dbg & 4)) issue(slot);
        if (s + 1 < S) load_frags(nslot, 0, fa[0], fb[0]);
        slot = nslot;
        ++s;
        TICK(7);
    }
#ifdef MFVIT_PERS_TRACE
    // per-workgroup phase totals (wave 0, lane 0) -> p.res[block][8] (cycles): 0 front MFMAs, 1 vmcnt wait, 2 lgkm wait, 3 barrier,
    // 4 issue + fragment prefetch, 5 last MFMAs, 6 epilogue, 7 refill after epilogue
    if (threadIdx.x == 0 && p.res)
        for (int i = 0; i < 8; ++i) ((float*)p.res)[blockIdx.x * 8 + i] = (float)tacc[i];
#endif
}

template <typename T, int EPI, int BKB> int launch_pers_v(const GemmP& p, hipStream_t st, int wgs_per_cu) {
    const int ntm = (p.M + PBM - 1) / PBM, ntn = p.N / PBN;
    const int ntiles = ntm * ntn;
    const int cap = 256 * wgs_per_cu;
    const int G = ntiles < cap ? ntiles : cap;
    constexpr int bytes = PCfg<BKB>::LDS_BYTES;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_pers_kernel<T, EPI, BKB>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr = true;
    }
    ProfScope ps(PROF_GEMM_TILE, 2.0 * p.M * p.N * p.K, 0, st);
    GemmP q = p;
    static const int stagger = [] { const char* e = getenv("MFVIT_PERS_STAGGER"); return e ? atoi(e) : 0; }();
    q.y_f32 = wgs_per_cu > 1 ? stagger : 0;   // field reused as the stagger length (x 1024 cycles)
#ifdef MFVIT_PERS_TRACE
    if (EPI != EPI_BIAS_GELU) q.res = (const float*)q.out1;   // trace build: the unused second output receives [grid][8] cycle totals
#endif
    MFVIT_LAUNCH((gemm_nt_pers_kernel<T, EPI, BKB>), dim3(G), dim3(256), bytes, st, q, ntm, ntn);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
template <int EPI> int launch_pers(int dtype, const GemmP& p, hipStream_t st) {
    if (dtype == MFVIT_BF16X3) return launch_pers_v<sbf16, EPI, 128>(p, st, 1);      // split: 128-byte rows (one k group), one workgroup per CU
    static const int bk = [] { const char* e = getenv("MFVIT_PERS_BK"); return e ? atoi(e) : 32; }();
    if (bk == 64 && p.K % 64 == 0) return launch_pers_v<bf16, EPI, 128>(p, st, 1);
    return launch_pers_v<bf16, EPI, 64>(p, st, 2);
}

}  // namespace

// bf16: opt-in (MFVIT_PERS=1, or the mfvit_linear_fwd_persistent entry point): measured on MI355X it ties the 128x128 kernel on an
// isolated launch (qkv 38.0 vs 39.6 us, fc1+GELU 82 vs 80 us) and loses ~3 % inside the four-stream training step, where the
// many small workgroups of the 128x128 kernel interleave better with the co-scheduled kernels of the other streams.
// split bf16: MFVIT_PERS_SPLIT (see gemm_nt_pers_supported).
bool gemm_nt_pers_supported(int dtype, int epi, const GemmP& p, bool force) {
    static const int on = [] { const char* e = getenv("MFVIT_PERS"); return e ? atoi(e) : 0; }();
    static const int on_split = [] { const char* e = getenv("MFVIT_PERS_SPLIT"); return e ? atoi(e) : 0; }();
    if (dtype != MFVIT_BF16 && dtype != MFVIT_BF16X3) return false;
    const int ep = dtype == MFVIT_BF16X3 ? 2 : 1;
    if ((!(ep == 2 ? on_split : on) && !force) || p.nb > 1 || p.M < 1024) return false;
    if (p.N % PBN || p.N > PMAXN || (ep == 2 ? p.K % 32 : p.K % 32) || p.K < 128) return false;   // split: whole 32-wide k groups
    if (epi == EPI_BIAS_RELU) return false;                    // ReLU MLP (TransFuser-GPT): 128x128 kernel only
    if (epi == EPI_BIAS_GELU && !p.out0) return false;         // no-grad forward without the saved derivative: 128x128 kernel only
    if (ep == 2 && (epi == EPI_BIAS_GELU || epi == EPI_GELU_BWD)) return false;   // split tensors keep gelu' as plain fp16 (gemm.hip)
    if (epi == EPI_GELU_BWD && p.cs0) return false;            // column sums stay with the 128x128 kernel
    if ((long)p.M * p.lda * 2 >= (1L << 32) || (long)p.N * p.ldw * 2 >= (1L << 32)) return false;   // 32-bit byte offsets
    if (p.lda % 8 || p.ldw % 8 || p.ldo0 % 8 || (p.out1 && p.ldo1 % 8) || (p.aux && p.ldaux % 8)) return false;
    return true;
}

int gemm_nt_pers(int dtype, int epi, const GemmP& p, hipStream_t st) {
    switch (epi) {
        case EPI_BIAS: return launch_pers<EPI_BIAS>(dtype, p, st);
        case EPI_BIAS_GELU: return launch_pers<EPI_BIAS_GELU>(dtype, p, st);
        case EPI_GELU_BWD: return launch_pers<EPI_GELU_BWD>(dtype, p, st);
        case EPI_NONE: return launch_pers<EPI_NONE>(dtype, p, st);
    }
    return MFVIT_EINVAL;
}

}  // namespace mfvit
