// MFMA multi-head self-attention core for bf16, head_dim 32 (ViT-S: 12 heads x 32) on gfx950.
//
// One workgroup (4 waves) per (image, head); K/V (forward) or Q/K/V/dO (backward) of that head live in LDS as bf16.
// T = 197 keys fit whole, so the forward does an exact softmax per 32-query tile over register-resident scores
// (chunks of up to 8 key tiles; longer sequences chain chunks with an online rescale).
//
// Operand orientation ("swapped" products, wave64 32x32x16 MFMA):
//   S^T[key][q]  = K_tile (A, rows = keys)  x  Q^T (B, cols = queries)     -> the query sits on the LANE: row max / row
//                                                                             sum are in-register + one cross-half shuffle
//   O^T[d][q]   += V^T (A, rows = d)        x  P^T (B)                       -> P^T is the S^T accumulator itself (registers
//                  8s..8s+7 = k-step s); V^T fragments come from the [key][d] LDS image with ds_read_b64_tr_b16 in the
//                  accumulator's k order (key = 16s + 8(j>>2) + 4h + (j&3)).
// head_dim 32 makes this kernel VALU(exp)-bound, not MFMA-bound (hard parts): per 32x32 tile 4 MFMAs (128
// cycles) vs ~16 x (fma + exp2 + max + add + cvt) per lane.
//
// Backward (flash-style recompute from the saved log-sum-exp, two phases, no atomics, deterministic):
//   phase A (wave = query tile): dS^T[key][q] -> dQ^T[d][q] += K^T x dS^T       (lse_q, D_q are per-lane scalars)
//   phase B (wave = key tile)  : P[q][key], dS[q][key] -> dV^T[d][key] += dO^T x P ; dK^T[d][key] += Q^T x dS
#include <stdlib.h>

#include "common.cuh"
#include "prof.h"

namespace mfvit {

namespace {

// Element types: bf16, f16 (same data movement, v_mfma_f32_32x32x16_f16) and sbf16 = SPLIT bf16 (MFVIT_BF16X3): a head's row piece
// is [hi x 32 | lo x 32] (128 B), every product of two tensors runs as three MFMAs (hi*hi + lo*hi + hi*lo) and P / dS are split
// in registers (hi = bf16(p), lo = bf16(p - hi)) before they feed the second product - f32-grade attention on the bf16 matrix core.
constexpr int HD = 32;
// LDS images: padded pitch RB + 16 (80 B, split 144 B): the natural-order b128 row reads are bank-conflict free, the transposed
// ds_read_b64_tr_b16 reads are 2-way conflicted (rows q and q + 2 of a 4-row block share banks).  Tried in round 2 and dropped: unpadded
// rows with the 16-B chunks XOR-swizzled by the row (chunk ^ ((row >> 2) & 3), split: ((row >> 1) & 1) << 2 | (row >> 2) & 3) make BOTH
// patterns conflict free, but the chunk position then no longer folds into the ds_read immediate offset: +14 ... +70 VGPRs of
// precomputed addresses and an address add per read - forward 55.4 -> 59.8 us, backward 158 -> 162 us (split bf16, B = 128): slower.

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

template <typename T> struct AttnT {
    static constexpr bool SP = is_split<T>::value;
    static constexpr int EP = SP ? 2 : 1;
    static constexpr int RB = 64 * EP;        // bytes of one head row piece
    static constexpr int RSB = RB + 16;       // padded LDS row pitch: b128 row reads of 16 consecutive rows are conflict-free
    typedef typename Vec8<T>::type frag_t;
    typedef typename Vec4<T>::type vec4_t;
    typedef typename Vec4<T>::elem E;
    static __device__ __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) { return MmaTraits_mma(a, b, c); }
};
__device__ __forceinline__ f32x16 MmaTraits_mma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 MmaTraits_mma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// natural-order fragment: row (rowbase + lane&31), elements d = 16 s + 8 (lane>>5) .. +8 of part `part` (0 = hi / plain, 1 = lo)
template <typename T> __device__ __forceinline__ typename Vec8<T>::type row_frag(const char* img, int pitch, int rowbase, int s, int lane, int part = 0) {
    return *(const typename Vec8<T>::type*)(img + (rowbase + (lane & 31)) * pitch + 64 * part + 32 * s + 16 * (lane >> 5));
}
// transposed fragment in ACCUMULATOR k order: lane holds column d = lane&31; element j = row (rowbase + 16 s + 8 (j>>2) + 4 h + (j&3))
template <typename T> __device__ __forceinline__ typename Vec8<T>::type tr_frag(const char* img, int pitch, int rowbase, int s, int lane, int part = 0) {
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const char* a = img + (rowbase + 16 * s + 4 * h + q) * pitch + 64 * part + (16 * g1 + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 8 * pitch));
    union { struct { s16x4 a, b; } s; typename Vec8<T>::type v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}
// registers 8 s .. 8 s + 7 of an accumulator as a B fragment (k step s); split: hi and lo parts
template <typename T> __device__ __forceinline__ void pack8(const f32x16& v, int s, typename Vec8<T>::type& hi, typename Vec8<T>::type& lo) {
    typedef typename Vec4<T>::elem E;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        E h0, h1, l0, l1;
        cvt_pair<E, is_split<T>::value>(v[8 * s + j], v[8 * s + j + 1], h0, h1, l0, l1);
        hi[j] = h0;
        hi[j + 1] = h1;
        if constexpr (is_split<T>::value) {
            lo[j] = l0;
            lo[j + 1] = l1;
        }
    }
}
// acc += A (x) B as one MFMA (plain) or three (split: a_hi b_hi + a_lo b_hi + a_hi b_lo)
template <typename T> __device__ __forceinline__ f32x16 mma3(typename Vec8<T>::type ah, typename Vec8<T>::type al, typename Vec8<T>::type bh,
                                                             typename Vec8<T>::type bl, f32x16 c) {
    if constexpr (is_split<T>::value) {
        c = MmaTraits_mma(al, bh, c);
        c = MmaTraits_mma(ah, bl, c);
    }
    return MmaTraits_mma(ah, bh, c);
}

// store an accumulator tile X^T[d][col] (col on the lane) as rows of a [.., HD] tensor of T: 4 (split: 8) x 8-byte stores per lane
template <typename T> __device__ __forceinline__ void store_tile_T(typename Vec4<T>::elem* row_ptr, const f32x16& acc, float mul, int lane) {
    typedef typename Vec4<T>::elem E;
    typedef typename Vec4<T>::type V4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        V4 o, l;
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            E h0, h1, l0, l1;
            cvt_pair<E, is_split<T>::value>(acc[4 * g + j] * mul, acc[4 * g + j + 1] * mul, h0, h1, l0, l1);
            o[j] = h0;
            o[j + 1] = h1;
            if constexpr (is_split<T>::value) {
                l[j] = l0;
                l[j + 1] = l1;
            }
        }
        *(V4*)(row_ptr + 8 * g + 4 * (lane >> 5)) = o;
        if constexpr (is_split<T>::value) *(V4*)(row_ptr + 32 + 8 * g + 4 * (lane >> 5)) = l;
    }
}

constexpr int NKC = 4;  // key tiles per register-resident chunk (4 x 16 = 64 score registers -> 2 waves per SIMD)

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const typename Vec4<T>::elem* __restrict__ qkv, typename Vec4<T>::elem* __restrict__ out,
                                                            float* __restrict__ lse, int Tn, int H, float scale) {
    typedef AttnT<T> A;
    typedef typename A::E E;
    typedef typename A::frag_t frag_t;
    constexpr int EP = A::EP, KP = A::RSB, VP = A::RB, CPR = A::RB / 16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Tpad = (Tn + 31) & ~31;
    char* Ks = lds;
    char* Vs = lds + Tpad * KP;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);  // the heads of one image share an XCD: their row pieces share L2 lines
    const int b = bid / H, h = bid % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long hs = (long)H * HD * EP;                  // storage elements of one of q / k / v per token
    const long rs = 3 * hs;
    const E* base = qkv + (long)b * Tn * rs + h * HD * EP;
    for (int q = threadIdx.x; q < Tpad * CPR; q += blockDim.x) {   // K and V pieces of a row loaded together (one HBM round trip, not two)
        const int t = q / CPR, cidx = q % CPR;
        uint4 vk = make_uint4(0, 0, 0, 0), vv = vk;
        if (t < Tn) {
            const E* rp = base + (long)t * rs + hs + 8 * cidx;
            vk = *(const uint4*)rp;
            vv = *(const uint4*)(rp + hs);
        }
        *(uint4*)(Ks + t * KP + 16 * cidx) = vk;
        *(uint4*)(Vs + t * VP + 16 * cidx) = vv;
    }
    __syncthreads();
    const float c = scale * 1.4426950408889634f;
    const int nt = Tpad >> 5;
    for (int qt = wave; qt < nt; qt += 4) {
        const int q = qt * 32 + (lane & 31);
        const int qc = q < Tn ? q : Tn - 1;
        frag_t qf[2], ql[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[s] = *(const frag_t*)(base + (long)qc * rs + 16 * s + 8 * (lane >> 5));
            if constexpr (A::SP) ql[s] = *(const frag_t*)(base + (long)qc * rs + 32 + 16 * s + 8 * (lane >> 5));
            else ql[s] = qf[s];
        }
        float m2 = -INFINITY, lsum = 0.f;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        for (int k0 = 0; k0 < nt; k0 += NKC) {
            const int n = nt - k0 < NKC ? nt - k0 : NKC;
            f32x16 sc[NKC];
#pragma unroll
            for (int t = 0; t < NKC; ++t) {
                if (t < n) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        sc[t] = mma3<T>(row_frag<T>(Ks, KP, (k0 + t) * 32, s, lane, 0), row_frag<T>(Ks, KP, (k0 + t) * 32, s, lane, A::SP ? 1 : 0),
                                        qf[s], ql[s], sc[t]);
                    if ((k0 + t + 1) * 32 > Tn) {  // last key tile: mask the zero-padded keys
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if ((k0 + t) * 32 + acc_row(r, lane) >= Tn) sc[t][r] = -INFINITY;
                    }
                }
            }
            float cm = -INFINITY;
#pragma unroll
            for (int t = 0; t < NKC; ++t)
                if (t < n) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) cm = fmaxf(cm, sc[t][r]);
                }
            cm = fmaxf(cm, __shfl_xor(cm, 32, 64)) * c;
            const float mn = fmaxf(m2, cm);
            const float alpha = exp2f(m2 - mn);
            m2 = mn;
            lsum *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
            for (int t = 0; t < NKC; ++t)
                if (t < n) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(sc[t][r], c, -m2));
                        sc[t][r] = p;
                        lsum += p;
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        frag_t ph, pl;
                        pack8<T>(sc[t], s, ph, pl);
                        if constexpr (!A::SP) pl = ph;
                        o = mma3<T>(tr_frag<T>(Vs, VP, (k0 + t) * 32, s, lane, 0), tr_frag<T>(Vs, VP, (k0 + t) * 32, s, lane, A::SP ? 1 : 0), ph, pl, o);
                    }
                }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        if (q < Tn) {
            store_tile_T<T>(out + (((long)b * Tn + q) * H * HD + h * HD) * EP, o, 1.0f / lsum, lane);
            if (lane < 32) lse[((long)b * H + h) * Tn + q] = (m2 + log2f(lsum)) * 0.6931471805599453f;
        }
    }
}

// One 8-wave workgroup per (image, head): Q, K, V, dO of the head are read from HBM exactly once into four LDS images
// (padded pitch: conflict-free row reads; unpadded when only that fits the 160 KB of LDS).  -lse/scale and
// -D = -rowsum(dO o O) enter the score MFMAs as accumulator initial values, so S - lse/scale and dP - D come out of the
// matrix core and the VALU work per element is mul, exp2, mul, cvt (+ the hi / lo split of P and dS for split tensors).
template <typename T, int PITCH>
__global__ __launch_bounds__(512) void attn_bwd_mfma_kernel(const typename Vec4<T>::elem* __restrict__ qkv, const typename Vec4<T>::elem* __restrict__ out,
                                                            const typename Vec4<T>::elem* __restrict__ dout, const float* __restrict__ lse,
                                                            typename Vec4<T>::elem* __restrict__ dqkv, int Tn, int H, float scale, int nbh) {
    typedef AttnT<T> A;
    typedef typename A::E E;
    typedef typename A::frag_t frag_t;
    constexpr int EP = A::EP, CPR = A::RB / 16, LO = A::SP ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Tpad = (Tn + 31) & ~31;
    char* Qs = lds;
    char* Ks = Qs + Tpad * PITCH;
    char* Vs = Ks + Tpad * PITCH;
    char* dOs = Vs + Tpad * PITCH;
    float* Ls = (float*)(dOs + Tpad * PITCH);
    float* Ds = Ls + Tpad;
    char* Os = (char*)(Ds + Tpad);                                     // O rows of the NEXT pair (unpadded, A::RB bytes each): only D needs them
    // PERSISTENT over the (image, head) pairs: the workgroup's four LDS images (129 KB for split bf16) allow one workgroup per CU, so a
    // workgroup per pair left every CU idle while its ~129 KB came in cold (measured: 6 rounds x ~6 us of the 148 us launch).  Now each
    // workgroup walks through its pairs and fetches the NEXT pair's row pieces into registers (5 x 16 B per chunk, 4 chunks per thread)
    // before it starts the two compute phases of the current one; they go to LDS when the phases are done.
    const int npair = nbh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long hs = (long)H * HD * EP, rs = 3 * hs, os = hs;
    constexpr int NCH = 4;                                             // 16-byte chunks per thread and image (Tpad * CPR <= 4 * 512)
    constexpr int RSTEP = 512 / CPR;                                   // rows between a thread's chunks
    const int t0 = threadIdx.x / CPR, cidx = threadIdx.x % CPR;
    uint4 pq[NCH], pk[NCH], pv[NCH], pd[NCH];
    float pl[NCH];
    auto fetch = [&](int vi) __attribute__((always_inline)) {
        const int bid = xcd_remap(vi, npair);                          // the heads of one image share an XCD: their row pieces share L2 lines
        const int b = bid / H, h = bid % H;
        const E* base = qkv + (long)b * Tn * rs + h * HD * EP;
        const E* obase = out + (long)b * Tn * os + h * HD * EP;
        const E* dobase = dout + (long)b * Tn * os + h * HD * EP;
        // chunk i of this thread: row t0 + RSTEP i, 16-byte piece cidx - ONE per-lane address per tensor, the rows of the other chunks are a
        // uniform stride away (per-chunk addresses were spilled, and every scratch reload waits for ALL loads in flight: the prefetch ran
        // one chunk at a time)
        const E* rp = base + (long)t0 * rs + 8 * cidx;
        const E* dp_ = dobase + (long)t0 * os + 8 * cidx;
        const float* lp = lse + ((long)b * H + h) * Tn + t0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            pq[i] = pk[i] = pv[i] = pd[i] = make_uint4(0, 0, 0, 0);
            pl[i] = 1e30f;                                             // (raw lse; no arithmetic on a loaded value here: it would wait for the loads)
            if (t0 + RSTEP * i < Tn) {                                 // (Tn <= Tpad <= RSTEP * NCH: checked by the launcher)
                pq[i] = *(const uint4*)(rp + (long)(RSTEP * i) * rs);
                pk[i] = *(const uint4*)(rp + (long)(RSTEP * i) * rs + hs);
                pv[i] = *(const uint4*)(rp + (long)(RSTEP * i) * rs + 2 * hs);
                pd[i] = *(const uint4*)(dp_ + (long)(RSTEP * i) * os);
                if (cidx == 0) pl[i] = lp[RSTEP * i];
            }
        }
        // the O rows go global -> LDS directly (LDS-DMA, lane-linear destination = the unpadded [row][piece] image): 16 registers per thread
        // less to carry through the compute phases - with them the split-bf16 variant spilled, and every scratch reload waits for ALL loads
        // in flight.  Rows past Tn re-fetch row Tn - 1 (finite; their D is never used).
        const unsigned lbase_o = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)Os + (unsigned)wave * 1024u);
        const unsigned long long ob = (unsigned long long)obase;       // uniform: scalar base + 32-bit per-lane byte offsets, derived HERE from a
        const unsigned ob_lo = __builtin_amdgcn_readfirstlane((unsigned)ob), ob_hi = __builtin_amdgcn_readfirstlane((unsigned)(ob >> 32));
        const char* sb = (const char*)(((unsigned long long)ob_hi << 32) | ob_lo);   // laundered lane id (carried across the loop they were spilled,
        int tl = t0, cl = cidx;                                                     // and a reload between the loads waits for all of them)
        asm volatile("" : "+v"(tl), "+v"(cl));
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int t = tl + RSTEP * i;
            t = t < Tn ? t : Tn - 1;
            const unsigned voff = ((unsigned)t * (unsigned)os + 8u * (unsigned)cl) * (unsigned)sizeof(E);
            // wave-exact guard: a wave covers 64 / CPR whole rows and Tpad is a multiple of 32, so a wave's 1 KB piece lies entirely inside or
            // entirely outside the Tpad * RB bytes of Os (a block-uniform guard let waves 4 - 7 of the last chunk write past the allocation)
            if (RSTEP * i + (__builtin_amdgcn_readfirstlane(wave) * 64) / CPR < Tpad) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(voff), "s"(sb), "s"(lbase_o + (unsigned)i * 8192u)
                             : "memory");
            }
        }
    };
    // registers -> LDS images; D = sum_d dO[d] O[d] per row (split: both parts of dO and O - lane cidx < 4 holds the hi parts of 8 d's, lane
    // cidx + 4 the lo parts of the same d's)
    float pD[NCH];
    // D of the fetched rows (lane cidx == 0 holds it).  Split tensors: lane cidx < 4 holds the hi parts of 8 d's, lane cidx ^ 4 the lo parts of the
    // same d's: every lane completes O = hi + lo with ONE exchange of its raw 16 bytes (4 cross-lane moves) and multiplies it with its own
    // part of dO; the sum over the row's 8 lanes adds the hi and lo parts of dO.
    auto reduce_D = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            union { uint4 u; frag_t f; } cd, co, cx;
            const int t = t0 + RSTEP * i;
            cd.u = pd[i];
            co.u = *(const uint4*)(Os + ((t < Tpad ? t : 0) * CPR + cidx) * 16);
            float of[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) of[j] = (float)co.f[j];
            if constexpr (A::SP) {                                      // the other part of the same 8 d's: piece cidx ^ 4 of the row
                cx.u = *(const uint4*)(Os + ((t < Tpad ? t : 0) * CPR + (cidx ^ 4)) * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) of[j] += (float)cx.f[j];
            }
            float D = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) D = fmaf((float)cd.f[j], of[j], D);
            D += __shfl_xor(D, 1, 64);
            D += __shfl_xor(D, 2, 64);
            if constexpr (A::SP) D += __shfl_xor(D, 4, 64);
            pD[i] = D;
        }
    };
    auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int t = t0 + RSTEP * i;
            if (t < Tpad) {
                *(uint4*)(Qs + t * PITCH + 16 * cidx) = pq[i];
                *(uint4*)(Ks + t * PITCH + 16 * cidx) = pk[i];
                *(uint4*)(Vs + t * PITCH + 16 * cidx) = pv[i];
                *(uint4*)(dOs + t * PITCH + 16 * cidx) = pd[i];
                if (cidx == 0) {
                    Ds[t] = -pD[i];
                    Ls[t] = t < Tn ? -pl[i] / scale : -1e30f;          // padded queries: p = exp2(-1e30 c) = 0
                }
            }
        }
    };
    fetch(blockIdx.x);
    for (int vi = blockIdx.x; vi < npair; vi += gridDim.x) {
    const int bid = xcd_remap(vi, npair);
    const int b = bid / H, h = bid % H;
    E* dbase = dqkv + (long)b * Tn * rs + h * HD * EP;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's O pieces have landed ...
    __syncthreads();                                                   // ... and everybody else's
    reduce_D();
    stage();
    __syncthreads();                                                   // (also: Os may be refilled)
    if (vi + (int)gridDim.x < npair) fetch(vi + gridDim.x);
    const float c = scale * 1.4426950408889634f;
    const int nt = Tpad >> 5;
    // ---------------- phase A: dQ, wave = query tile
    for (int qt = wave; qt < nt; qt += 8) {
        frag_t qf[2], ql[2], dof[2], dol[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[s] = row_frag<T>(Qs, PITCH, qt * 32, s, lane, 0);
            ql[s] = row_frag<T>(Qs, PITCH, qt * 32, s, lane, LO);
            dof[s] = row_frag<T>(dOs, PITCH, qt * 32, s, lane, 0);
            dol[s] = row_frag<T>(dOs, PITCH, qt * 32, s, lane, LO);
        }
        const float L = Ls[qt * 32 + (lane & 31)], Dq = Ds[qt * 32 + (lane & 31)];  // -lse/scale, -D
        f32x16 dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
        for (int kt = 0; kt < nt; ++kt) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = L; dp[r] = Dq; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                st = mma3<T>(row_frag<T>(Ks, PITCH, kt * 32, s, lane, 0), row_frag<T>(Ks, PITCH, kt * 32, s, lane, LO), qf[s], ql[s], st);
                dp = mma3<T>(row_frag<T>(Vs, PITCH, kt * 32, s, lane, 0), row_frag<T>(Vs, PITCH, kt * 32, s, lane, LO), dof[s], dol[s], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)  // dS^T / scale (padded keys: K rows are zero, so their dQ contribution vanishes)
                st[r] = __builtin_amdgcn_exp2f(st[r] * c) * dp[r];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                frag_t sh, sl;
                pack8<T>(st, s, sh, sl);
                if constexpr (!A::SP) sl = sh;
                dq = mma3<T>(tr_frag<T>(Ks, PITCH, kt * 32, s, lane, 0), tr_frag<T>(Ks, PITCH, kt * 32, s, lane, LO), sh, sl, dq);
            }
        }
        // (store addresses from a laundered lane id: carried across the persistent loop they were spilled, and the reload - placed right behind
        // the prefetch - waited for every load in flight)
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const int q = qt * 32 + (lane_s & 31);
        if (q < Tn) store_tile_T<T>(dbase + (long)q * rs, dq, scale, lane_s);
    }
    // ---------------- phase B: dK, dV, wave = key tile
    for (int kt = wave; kt < nt; kt += 8) {
        frag_t kf[2], kl[2], vf[2], vl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[s] = row_frag<T>(Ks, PITCH, kt * 32, s, lane, 0);
            kl[s] = row_frag<T>(Ks, PITCH, kt * 32, s, lane, LO);
            vf[s] = row_frag<T>(Vs, PITCH, kt * 32, s, lane, 0);
            vl[s] = row_frag<T>(Vs, PITCH, kt * 32, s, lane, LO);
        }
        f32x16 dk, dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[r] = dv[r] = 0.f;
        for (int qt = 0; qt < nt; ++qt) {
            f32x16 sm, dp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q0 = qt * 32 + 8 * g + 4 * (lane >> 5);
                const float4 L4 = *(const float4*)(Ls + q0);
                const float4 D4 = *(const float4*)(Ds + q0);
                sm[4 * g] = L4.x, sm[4 * g + 1] = L4.y, sm[4 * g + 2] = L4.z, sm[4 * g + 3] = L4.w;
                dp[4 * g] = D4.x, dp[4 * g + 1] = D4.y, dp[4 * g + 2] = D4.z, dp[4 * g + 3] = D4.w;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sm = mma3<T>(row_frag<T>(Qs, PITCH, qt * 32, s, lane, 0), row_frag<T>(Qs, PITCH, qt * 32, s, lane, LO), kf[s], kl[s], sm);   // S[q][key] - lse[q]/scale
                dp = mma3<T>(row_frag<T>(dOs, PITCH, qt * 32, s, lane, 0), row_frag<T>(dOs, PITCH, qt * 32, s, lane, LO), vf[s], vl[s], dp);  // dP[q][key] - D[q]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sm[r] = __builtin_amdgcn_exp2f(sm[r] * c);
                dp[r] *= sm[r];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                frag_t ph, pl, sh, sl;
                pack8<T>(sm, s, ph, pl);
                pack8<T>(dp, s, sh, sl);
                if constexpr (!A::SP) { pl = ph; sl = sh; }
                dv = mma3<T>(tr_frag<T>(dOs, PITCH, qt * 32, s, lane, 0), tr_frag<T>(dOs, PITCH, qt * 32, s, lane, LO), ph, pl, dv);
                dk = mma3<T>(tr_frag<T>(Qs, PITCH, qt * 32, s, lane, 0), tr_frag<T>(Qs, PITCH, qt * 32, s, lane, LO), sh, sl, dk);
            }
        }
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const int k = kt * 32 + (lane_s & 31);
        if (k < Tn) {
            store_tile_T<T>(dbase + (long)k * rs + hs, dk, scale, lane_s);
            store_tile_T<T>(dbase + (long)k * rs + 2 * hs, dv, 1.0f, lane_s);
        }
    }
    __syncthreads();                                               // every wave is done with the images before the next pair overwrites them
    }
}

// column sums of a [M][N] matrix of T (N logical columns, N % 8 == 0) into f32 out[N] (atomicAdd).  Used for d qkv.bias.
template <typename T>
__global__ __launch_bounds__(256) void colsum_t_kernel(const typename Vec4<T>::elem* __restrict__ x, long ld, float* __restrict__ out, int M, int N) {
    typedef typename Vec8<T>::type V8;
    const int r0 = blockIdx.x * 64, r1 = min(M, r0 + 64);
    for (int cch = threadIdx.x; cch < N / 8; cch += 256) {
        const int col = is_split<T>::value ? split_col(8 * cch) : 8 * cch;
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0.f;
        for (int r = r0; r < r1; ++r) {
            const V8 v = *(const V8*)(x + (long)r * ld + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += (float)v[j];
            if constexpr (is_split<T>::value) {
                const V8 l = *(const V8*)(x + (long)r * ld + col + 32);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += (float)l[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(out + 8 * cch + j, a[j]);
    }
}

template <typename T> int launch_fwd_t(const void* qkv, void* out, float* lse, int B, int Tn, int H, hipStream_t st) {
    typedef typename Vec4<T>::elem E;
    const int Tpad = (Tn + 31) & ~31;
    const int bytes = Tpad * (AttnT<T>::RSB + AttnT<T>::RB);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    ProfScope ps(PROF_ATTN_FWD, 4.0 * B * H * (double)Tn * Tn * HD, 0, st);
    MFVIT_LAUNCH((attn_fwd_mfma_kernel<T>), dim3(B * H), dim3(256), bytes, st, (const E*)qkv, (E*)out, lse, Tn, H, 1.0f / sqrtf((float)HD));
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
template <typename T> int launch_bwd_t(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn,
                                       int H, hipStream_t st) {
    typedef typename Vec4<T>::elem E;
    constexpr int RSB = AttnT<T>::RSB, RB = AttnT<T>::RB;
    const int Tpad = (Tn + 31) & ~31;
    const bool wide = 4 * Tpad * RSB + 2 * Tpad * 4 + Tpad * RB <= 160 * 1024;
    const int bytes = 4 * Tpad * (wide ? RSB : RB) + 2 * Tpad * 4 + Tpad * RB;      // Q, K, V, dO images, lse / D rows, the next pair's O rows
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<T, RSB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<T, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    {
        ProfScope ps(PROF_ATTN_BWD, 8.0 * B * H * (double)Tn * Tn * HD, 0, st);
        // one persistent workgroup per CU (the LDS images allow no second one): a multiple of 8 so that a workgroup's pairs stay on its XCD
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
            return n & ~7;
        }();
        const char* eg = getenv("MFVIT_ATTN_BWD_PERSIST");             // 0: one workgroup per pair (A/B switch, read at every launch)
        const int grid = (B * H < cus || (eg && atoi(eg) == 0)) ? B * H : cus;
        if (wide)
            MFVIT_LAUNCH((attn_bwd_mfma_kernel<T, RSB>), dim3(grid), dim3(512), bytes, st, (const E*)qkv, (const E*)out, (const E*)dout, lse,
                         (E*)dqkv, Tn, H, 1.0f / sqrtf((float)HD), B * H);
        else
            MFVIT_LAUNCH((attn_bwd_mfma_kernel<T, RB>), dim3(grid), dim3(512), bytes, st, (const E*)qkv, (const E*)out, (const E*)dout, lse,
                         (E*)dqkv, Tn, H, 1.0f / sqrtf((float)HD), B * H);
        MFVIT_CHECK_LAUNCH();
    }
    if (dbias) {
        const int M = B * Tn, N = 3 * H * HD;
        MFVIT_LAUNCH((colsum_t_kernel<T>), dim3((M + 63) / 64), dim3(256), 0, st, (const E*)dqkv, (long)N * AttnT<T>::EP, dbias, M, N);
        MFVIT_CHECK_LAUNCH();
    }
    return MFVIT_OK;
}

}  // namespace

bool attn_mfma_supported(int dtype, int Tn, int HDim, bool backward) {
    if ((dtype != MFVIT_BF16 && dtype != MFVIT_BF16X3 && dtype != MFVIT_F16) || HDim != HD || Tn < 1) return false;
    const int Tpad = (Tn + 31) & ~31;
    const int rb = dtype == MFVIT_BF16X3 ? 128 : 64;
    const int bytes = backward ? 4 * Tpad * rb + 2 * Tpad * 4 + Tpad * rb : Tpad * (rb + 16) + Tpad * rb;
    if (backward && Tpad * (rb / 16) > 4 * 512) return false;      // the backward keeps 4 chunks per thread and image in registers
    return bytes <= 160 * 1024;
}

// d qkv.bias = column sums of dqkv [M][N logical columns] (used by the tiled kernels, which do not fuse it)
int attn_colsum(int dtype, const void* dqkv, int M, int N, float* dbias, hipStream_t st) {
    const dim3 grid((M + 63) / 64);
    if (dtype == MFVIT_BF16) MFVIT_LAUNCH((colsum_t_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)dqkv, (long)N, dbias, M, N);
    else if (dtype == MFVIT_BF16X3) MFVIT_LAUNCH((colsum_t_kernel<sbf16>), grid, dim3(256), 0, st, (const bf16*)dqkv, (long)N * 2, dbias, M, N);
    else if (dtype == MFVIT_F16) MFVIT_LAUNCH((colsum_t_kernel<f16>), grid, dim3(256), 0, st, (const f16*)dqkv, (long)N, dbias, M, N);
    else return MFVIT_EINVAL;
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

int attn_fwd_mfma(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, hipStream_t st) {
    if (dtype == MFVIT_BF16) return launch_fwd_t<bf16>(qkv, out, lse, B, Tn, H, st);
    if (dtype == MFVIT_BF16X3) return launch_fwd_t<sbf16>(qkv, out, lse, B, Tn, H, st);
    if (dtype == MFVIT_F16) return launch_fwd_t<f16>(qkv, out, lse, B, Tn, H, st);
    return MFVIT_EINVAL;
}
int attn_bwd_mfma(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn, int H,
                  hipStream_t st) {
    if (dtype == MFVIT_BF16) return launch_bwd_t<bf16>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st);
    if (dtype == MFVIT_BF16X3) return launch_bwd_t<sbf16>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st);
    if (dtype == MFVIT_F16) return launch_bwd_t<f16>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st);
    return MFVIT_EINVAL;
}

}  // namespace mfvit
