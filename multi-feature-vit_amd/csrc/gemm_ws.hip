// Warp-specialised persistent bf16 NT GEMM:  out[M][N] = epi(A[M][K] W[N][K]^T),  the short-K linears of the encoder.
//
// Evidence (cycle-counter trace of gemm_pers.hip, tools/pers_trace.py): a wave that both computes and loads spends 22-31 % of its
// life BLOCKED ISSUING its global_load_lds instructions (the vector-memory queue is full because the kernel is bound by L2 -> LDS
// bandwidth) and another 21-24 % issuing the epilogue's stores - time in which that SIMD's matrix core idles.  Here the two jobs
// are split:
//   waves 0-3  consumers: LDS fragment reads + MFMAs + the epilogue arithmetic; they never issue a vector-memory load
//   waves 4-7  producers (one per SIMD, next to a consumer): all LDS-DMA of the 3-slot ring (48 KB stages, BK = 64) - their
//              issue stalls cost nothing, the consumer on the same SIMD keeps the matrix core busy
// One workgroup (512 threads) per CU, persistent over its tiles (256 x 128, 128 x 64 per consumer wave, MFMA operands swapped so a
// lane owns one output row), K-steps of all its tiles in one flat pipeline, ONE s_barrier per K-step shared by both roles:
//   consumer step s : MFMAs of sub-steps 0..KS-2 (fragment reads one sub-step ahead) | lgkmcnt(0) | BARRIER | first fragments of
//                     step s+1 | last MFMAs | (tile end: epilogue)
//   producer step s : vmcnt(<= one stage) = stage s+1 landed | BARRIER | refill the slot of step s with stage s+3
// The barrier orders "every consumer finished reading slot s" before the refill and "stage s+1 landed" before its first read.
// Epilogue: per consumer wave through a private 2 KB LDS buffer (16 rows x 128 B, XOR-swizzled) -> full 128-B rows, 16-B stores.
#include "gemm.cuh"
#include "kernels.h"
#include "prof.h"

namespace mfvit {

namespace {

// staging accesses of the epilogue write packed bf16x4 and read 16-byte chunks of the same LDS bytes: both through may_alias types
// (type-based alias analysis would otherwise let the compiler reorder the differently typed stores and loads)
typedef uint4 __attribute__((may_alias)) stg_u4;
typedef bf16x4 __attribute__((may_alias)) stg_b4;

constexpr int WBM = 256, WBN = 128, WNS = 3, WBKB = 128, WBK = 64;
constexpr int WTM = 4, WTN = 2;
constexpr int WMAXN = 1536;
typedef KTile<bf16, WBM, WBKB> WTA;
typedef KTile<bf16, WBN, WBKB> WTB;
constexpr int WSTAGE = WTA::BYTES + WTB::BYTES;                 // 48 KB
constexpr int WGLDS = WSTAGE / 1024;                            // 48 LDS-DMA wave-instructions per stage (1 KB each)
constexpr int WPER = WGLDS / 4;                                 // 12 per producer wave
constexpr int WSTG = 2048;                                      // epilogue staging bytes per consumer wave
constexpr int WLDS = WNS * WSTAGE + WMAXN * 4 + 4 * WSTG;       // 147456 + 6144 + 8192 = 161792 <= 160 KB

template <int N> __device__ __forceinline__ void ws_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void ws_glds16(const void* gsrc, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_off)
                 : "memory");
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_ws_kernel(GemmP p, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, ntiles = ntm * ntn;
    const int cslot = (G & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
    if (cslot >= ntiles) return;
    const int nloc = (ntiles - cslot + G - 1) / G;
    const int nk = p.K / WBK;
    const int S = nloc * nk;
    float* lbias = (float*)(lds + WNS * WSTAGE);
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU)
        for (int n = tid; n < p.N; n += 512) lbias[n] = p.bias ? p.bias[n] : 0.f;

    // The epilogue of a tile writes at the full HBM rate while the main loop hardly touches HBM: with every CU in the same phase the
    // writes come in bursts.  Half of the workgroups start a few K-steps late so that their epilogues fall into the others' loops.
    if (p.y_f32 > 0 && ((blockIdx.x >> 3) & 1))
        for (int z = 0; z < p.y_f32; ++z) __builtin_amdgcn_s_sleep(32);

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ producer
        const int pw = __builtin_amdgcn_readfirstlane(wave - 4);     // wave-uniform: the LDS-DMA destination goes through M0 (an SGPR)
        // LDS-DMA g (0..47) of a stage fills bytes [g KB, g KB + 1 KB): chunk q = 64 g + lane -> tile row q >> 3 (rows 0..255 = A,
        // 256..383 = W), LDS chunk position q & 7, which receives source chunk (q & 7) ^ ((row >> 1) & 7) (KTile swizzle)
        unsigned off[WPER];
        const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
        int it_tile = cslot, it_kt = 0;
        auto set_tile = [&](int tile) {
            const int m0 = (tile / ntn) * WBM, n0 = (tile % ntn) * WBN;
#pragma unroll
            for (int i = 0; i < WPER; ++i) {
                const int g = pw + 4 * i, q = g * 64 + lane, row = q >> 3, ch = (q & 7) ^ ((row >> 1) & 7);
                if (g < WTA::BYTES / 1024) {
                    int gr = m0 + row;
                    gr = gr < p.M ? gr : p.M - 1;
                    off[i] = (unsigned)gr * (unsigned)(p.lda * 2) + ch * 16;
                } else {
                    off[i] = (unsigned)(n0 + row - WBM) * (unsigned)(p.ldw * 2) + ch * 16;
                }
            }
        };
        auto issue = [&](int slot) {
            if (it_kt == 0) set_tile(it_tile);
            const unsigned sb = lbase + (unsigned)slot * WSTAGE;
            const unsigned kb = (unsigned)it_kt * WBKB;
#pragma unroll
            for (int i = 0; i < WPER; ++i) {
                const int g = pw + 4 * i;
                const char* base = g < WTA::BYTES / 1024 ? (const char*)p.A : (const char*)p.W;
                ws_glds16(base + (off[i] + kb), __builtin_amdgcn_readfirstlane(sb + g * 1024));
            }
            if (++it_kt == nk) { it_kt = 0; it_tile += G; }
        };
        issue(0);
        if (S > 1) issue(1);
        if (S > 2) issue(2);
        if (S > 2) ws_wait_vm<2 * WPER>();
        else if (S > 1) ws_wait_vm<WPER>();
        else ws_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                                    // stage 0 landed (prologue barrier)
        int slot = 0;
        for (int s = 0; s < S; ++s) {
            if (s + 2 < S) ws_wait_vm<WPER>();                           // stage s+1 landed (stage s+2 may stay in flight)
            else ws_wait_vm<0>();
            __builtin_amdgcn_s_barrier();                                // step barrier: slot of step s is free from here
            __builtin_amdgcn_sched_barrier(0);
            if (s + 3 < S) issue(slot);
            slot = slot == WNS - 1 ? 0 : slot + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumer
    const int wm = wave >> 1, wn = wave & 1;
    bf16x8 fa[2][WTM], fb[2][WTN];
    auto load_frags = [&](int slot, int ks, bf16x8 (&a)[WTM], bf16x8 (&b)[WTN]) {
        const char* ta = lds + slot * WSTAGE;
        const char* tb = ta + WTA::BYTES;
#pragma unroll
        for (int j = 0; j < WTN; ++j) b[j] = WTB::frag(tb, (wn * WTN + j) * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < WTM; ++i) a[i] = WTA::frag(ta, (wm * WTM + i) * 32, ks, lane);
    };
    __builtin_amdgcn_s_barrier();                                        // prologue barrier (stage 0 landed)
    __builtin_amdgcn_sched_barrier(0);
    load_frags(0, 0, fa[0], fb[0]);
    char* stg = lds + WNS * WSTAGE + WMAXN * 4 + wave * WSTG;
    int slot = 0, s = 0, tile = cslot;
    for (int t = 0; t < nloc; ++t, tile += G) {
        f32x16 acc[WTM][WTN];
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        auto mfmas = [&](const bf16x8 (&a)[WTM], const bf16x8 (&b)[WTN]) {
#ifdef MFVIT_WS_DBG
            if constexpr ((MFVIT_WS_DBG & 2) != 0) return;   // build-time experiment switches: 1 no epilogue, 2 no MFMAs
#endif
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        };
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int ks = 0; ks < WTA::KSTEPS - 1; ++ks) {
                load_frags(slot, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(fa[ks & 1], fb[ks & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // every read of this slot is done
            __builtin_amdgcn_s_barrier();                                // step barrier
            __builtin_amdgcn_sched_barrier(0);
            const int nslot = slot == WNS - 1 ? 0 : slot + 1;
            if (s + 1 < S) load_frags(nslot, 0, fa[0], fb[0]);           // hidden behind the last MFMAs / the epilogue
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[(WTA::KSTEPS - 1) & 1], fb[(WTA::KSTEPS - 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            slot = nslot;
            ++s;
        }
#ifdef MFVIT_WS_DBG
        if constexpr ((MFVIT_WS_DBG & 1) != 0) {
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) asm volatile("" ::"v"(acc[i][j]));
            continue;
        }
#endif
        // ---- epilogue: acc[i][j][r] = out[m][n], m = m0 + wm 128 + 32 i + (lane & 31), n = n0 + wn 64 + 32 j + 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
        const int m0 = (tile / ntn) * WBM, n0 = (tile % ntn) * WBN;
        const int mrow = lane & 31, h = lane >> 5;
        const int nb = n0 + wn * 64 + 4 * h;
        char* wrow = stg + (mrow & 15) * 128 + 8 * h;                    // 16-row staging: rows 0-15 first, then rows 16-31
        const int wsw = (mrow >> 1) & 7;
        auto flush = [&](int i, int half, void* out, long ldo) {         // staged 16 x 64 tile -> global, 2 x 16 B per lane
            // lanes exchange data through LDS inside ONE wave (hardware executes a wave's LDS instructions in order, no barrier
            // needed) - but the compiler reasons per thread: without this fence it forwards a lane's earlier load of the same
            // address past the OTHER lanes' stores (observed: the reads sunk under the writers' exec mask)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int q = q2 * 64 + lane, row = q >> 3, ch = q & 7;
                const uint4 v = *(const stg_u4*)(stg + row * 128 + 16 * (ch ^ ((row >> 1) & 7)));
                int m = m0 + wm * 128 + 32 * i + 16 * half + row;
                m = m < p.M ? m : p.M - 1;                               // rows past M replicate row M-1: identical duplicate stores
                *(uint4*)((bf16*)out + (long)m * ldo + n0 + wn * 64 + 8 * ch) = v;
            }
        };
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            bf16x4 first[WTN][4], second[WTN][4];
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bq = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? *(const float4*)(lbias + nb + 32 * j + 8 * g)
                                                                               : make_float4(0.f, 0.f, 0.f, 0.f);
                    float v[4] = {acc[i][j][4 * g] + bq.x, acc[i][j][4 * g + 1] + bq.y, acc[i][j][4 * g + 2] + bq.z,
                                  acc[i][j][4 * g + 3] + bq.w};
                    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float cdf, ex;
                            gelu_parts_fast(v[e], cdf, ex);
                            first[j][g][e] = (bf16)fmaf(v[e] * 0.39894228040143267794f, ex, cdf);   // gelu'(pre)
                            second[j][g][e] = (bf16)(v[e] * cdf);                                    // gelu(pre)
                        }
                    } else if (EPI == EPI_GELU_BWD) {
                        int m = m0 + wm * 128 + 32 * i + mrow;
                        m = m < p.M ? m : p.M - 1;
                        const bf16x4 ax = *(const bf16x4*)((const bf16*)p.aux + (long)m * p.ldaux + nb + 32 * j + 8 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) first[j][g][e] = (bf16)(v[e] * (float)ax[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) first[j][g][e] = (bf16)v[e];
                    }
                }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if ((mrow >> 4) == half) {
#pragma unroll
                    for (int j = 0; j < WTN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) *(stg_b4*)(wrow + 16 * ((4 * j + g) ^ wsw)) = first[j][g];
                }
                flush(i, half, p.out0, p.ldo0);
                if (EPI == EPI_BIAS_GELU) {
                    if ((mrow >> 4) == half) {
#pragma unroll
                        for (int j = 0; j < WTN; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) *(stg_b4*)(wrow + 16 * ((4 * j + g) ^ wsw)) = second[j][g];
                    }
                    flush(i, half, p.out1, p.ldo1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int EPI> int launch_ws(const GemmP& p, hipStream_t st) {
    const int ntm = (p.M + WBM - 1) / WBM, ntn = p.N / WBN;
    const int ntiles = ntm * ntn;
    const int G = ntiles < 256 ? ntiles : 256;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_ws_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS);
        attr = true;
    }
    ProfScope ps(PROF_GEMM_TILE, 2.0 * p.M * p.N * p.K, 0, st);
    GemmP q = p;
    static const int stagger = [] { const char* e = getenv("MFVIT_WS_STAGGER"); return e ? atoi(e) : 0; }();
    q.y_f32 = stagger;                                              // field reused: start delay of every other workgroup (x ~2 k cycles)
    MFVIT_LAUNCH((gemm_nt_ws_kernel<EPI>), dim3(G), dim3(512), WLDS, st, q, ntm, ntn);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

}  // namespace

bool gemm_nt_ws_supported(int dtype, int epi, const GemmP& p, bool force) {
    static const int on = [] { const char* e = getenv("MFVIT_WS"); return e ? atoi(e) : 0; }();
    if ((!on && !force) || dtype != MFVIT_BF16 || p.nb > 1 || p.M < 1024) return false;
    if (p.N % WBN || p.N > WMAXN || p.K % WBK || p.K < 2 * WBK) return false;
    if (epi == EPI_BIAS_RELU) return false;                    // ReLU MLP (TransFuser-GPT): 128x128 kernel only
    if (epi == EPI_BIAS_GELU && !p.out0) return false;         // no-grad forward without the saved derivative: 128x128 kernel only
    if (epi == EPI_GELU_BWD && p.cs0) return false;
    if ((long)p.M * p.lda * 2 >= (1L << 32) || (long)p.N * p.ldw * 2 >= (1L << 32)) return false;
    if (p.lda % 8 || p.ldw % 8 || p.ldo0 % 8 || (p.out1 && p.ldo1 % 8) || (p.aux && p.ldaux % 4)) return false;
    return true;
}

int gemm_nt_ws(int epi, const GemmP& p, hipStream_t st) {
    switch (epi) {
        case EPI_BIAS: return launch_ws<EPI_BIAS>(p, st);
        case EPI_BIAS_GELU: return launch_ws<EPI_BIAS_GELU>(p, st);
        case EPI_GELU_BWD: return launch_ws<EPI_GELU_BWD>(p, st);
        case EPI_NONE: return launch_ws<EPI_NONE>(p, st);
    }
    return MFVIT_EINVAL;
}

}  // namespace mfvit
