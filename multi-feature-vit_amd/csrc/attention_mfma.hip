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
#include <limits.h>
#include <type_traits>
#include <stdlib.h>

#include "common.cuh"
#include "prof.h"

namespace mfvit {

namespace {

// Element types: bf16, f16 (same data movement, v_mfma_f32_32x32x16_f16) and sbf16 = SPLIT bf16 (MFVIT_BF16X3): a head's row piece
// is [hi x 32 | lo x 32] (128 B), every product of two tensors runs as three MFMAs (hi*hi + lo*hi + hi*lo) and P / dS are split
// in registers (hi = bf16(p), lo = bf16(p - hi)) before they feed the second product - f32-grade attention on the bf16 matrix core.
//
// sf16 (MFVIT_X3F16, round 5; what the encoder runs in bf16x3 mode): q, k, v arrive as SPLIT FP16 (the qkv GEMM's epilogue writes them so,
// common.cuh) and every MFMA of the kernel is v_mfma_f32_32x32x16_f16; out, dout and dqkv stay split bf16 (the neighbouring GEMMs' format).
// Why: the kernels are VALU-bound, and a third of that VALU work was the hi / lo split of P and dS in bf16 - convert, shift, mask, subtract,
// convert: 2.5 instructions per element.  In fp16 the split is v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16: 1.5 per element at the same 22 bits (P <= 2^14
// by the lazy-maximum threshold, the absolute floor 2^-25 sits far below the row's largest probability) - or 0.5 per element and two MFMAs
// instead of three where ONE fp16 part (11 bits) is enough (template parameter NP; measured precision: DESIGN.md 5, round 5).
// dO has gradient scale (1e-6 ... 1e-2: below fp16's normal range), so the backward converts it on chip to split fp16 times a power of two
// chosen per (image, head) from its largest element (|dO| 2^E in [4, 8)): dP, dS, dQ, dK, dV all carry 2^E, the final multipliers take it out.
constexpr int HD = 32;
// LDS images: padded pitch RB + 16 (80 B, split 144 B): the natural-order b128 row reads are bank-conflict free, the transposed
// ds_read_b64_tr_b16 reads are 2-way conflicted (rows q and q + 2 of a 4-row block share banks).  Tried in round 2 and dropped: unpadded
// rows with the 16-B chunks XOR-swizzled by the row (chunk ^ ((row >> 2) & 3), split: ((row >> 1) & 1) << 2 | (row >> 2) & 3) make BOTH
// patterns conflict free, but the chunk position then no longer folds into the ds_read immediate offset: +14 ... +70 VGPRs of
// precomputed addresses and an address add per read - forward 55.4 -> 59.8 us, backward 158 -> 162 us (split bf16, B = 128): slower.

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

template <typename T> struct AttnT {
    static constexpr bool SP = is_split<T>::value;
    static constexpr bool X = std::is_same<T, sf16>::value;   // split FP16 q / k / v (out, dout, dqkv: split bf16)
    static constexpr int EP = SP ? 2 : 1;
    static constexpr int RB = 64 * EP;        // bytes of one head row piece
    static constexpr int RSB = RB + 16;       // padded LDS row pitch: b128 row reads of 16 consecutive rows are conflict-free
    typedef typename Vec8<T>::type frag_t;
    typedef typename Vec4<T>::type vec4_t;
    typedef typename Vec4<T>::elem E;         // element of qkv
    typedef typename std::conditional<X, sbf16, T>::type OT;   // tensor type of out / dout / dqkv
    typedef typename Vec4<OT>::elem OE;
    typedef typename Vec8<OT>::type ofrag_t;
    static constexpr float LAZY = X ? 14.f : 24.f;             // lazy-maximum threshold (log2): P <= 2^LAZY must fit the operand type
    static __device__ __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) { return MmaTraits_mma(a, b, c); }
};
// parts of P / dS that feed the second products: plain types 1, split bf16 2 (hi + lo), split fp16 NPX (1: hi only = 11 bits; 2: hi + lo)
template <typename T, int NPX> struct PParts { static constexpr int value = AttnT<T>::X ? NPX : (AttnT<T>::SP ? 2 : 1); };
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
// split fp16: one v_cvt_pk_f16_f32 per pair for the hi parts, v_fma_mixlo / mixhi_f16 (lo = f16(x - f32(hi)), the f16 source read in place) for
// the lo parts; NP == 1: hi only.  SUM (NP == 1): the row sum of the ROUNDED probabilities (v_dot2c_f32_f16 against (1, 1)) - the normaliser
// then belongs to the same weights as the numerator.
template <int NP, bool SUM = false> __device__ __forceinline__ void pack8x(const f32x16& v, int s, f16x8& hi, f16x8& lo, float* sum = nullptr) {
    union { f16x8 f; unsigned u[4]; } h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v[8 * s + 2 * j], b = v[8 * s + 2 * j + 1];
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h.u[j]) : "v"(a), "v"(b));
        if constexpr (NP == 2) {
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l.u[j]) : "v"(h.u[j]), "v"(a));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l.u[j]) : "v"(h.u[j]), "v"(b));
        }
        if constexpr (SUM) asm("v_dot2c_f32_f16 %0, 0x3c003c00, %1" : "+v"(*sum) : "v"(h.u[j]));
    }
    hi = h.f;
    if constexpr (NP == 2) lo = l.f;
    else lo = h.f;
}
template <typename T, int NP = 2> __device__ __forceinline__ void pack8(const f32x16& v, int s, typename Vec8<T>::type& hi, typename Vec8<T>::type& lo) {
    typedef typename Vec4<T>::elem E;
    if constexpr (AttnT<T>::X) {
        pack8x<NP>(v, s, hi, lo);
        return;
    }
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
// acc += A (x) B with B = P or dS in NP parts (1: b_lo is not used - two MFMAs for a split A, 11 bits of B)
template <typename T, int NP> __device__ __forceinline__ f32x16 mmap(typename Vec8<T>::type ah, typename Vec8<T>::type al, typename Vec8<T>::type bh,
                                                                     typename Vec8<T>::type bl, f32x16 c) {
    if constexpr (is_split<T>::value) {
        c = MmaTraits_mma(al, bh, c);
        if constexpr (NP == 2) c = MmaTraits_mma(ah, bl, c);
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

template <typename T, int NPX>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const typename Vec4<T>::elem* __restrict__ qkv, typename AttnT<T>::OE* __restrict__ out,
                                                            float* __restrict__ lse, int Tn, int H, float scale) {
    typedef AttnT<T> A;
    typedef typename A::E E;
    typedef typename A::OT OT;
    typedef typename A::frag_t frag_t;
    constexpr int EP = A::EP, KP = A::RSB, VP = A::RB, CPR = A::RB / 16, NP = PParts<T, NPX>::value;
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
                        if constexpr (!(A::X && NP == 1)) lsum += p;
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        frag_t ph, pl;
                        if constexpr (A::X && NP == 1) pack8x<1, true>(sc[t], s, ph, pl, &lsum);
                        else pack8<T, NP>(sc[t], s, ph, pl);
                        if constexpr (!A::SP) pl = ph;
                        o = mmap<T, NP>(tr_frag<T>(Vs, VP, (k0 + t) * 32, s, lane, 0), tr_frag<T>(Vs, VP, (k0 + t) * 32, s, lane, A::SP ? 1 : 0), ph, pl, o);
                    }
                }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        if (q < Tn) {
            store_tile_T<OT>(out + (((long)b * Tn + q) * H * HD + h * HD) * EP, o, 1.0f / lsum, lane);
            if (lane < 32) lse[((long)b * H + h) * Tn + q] = (m2 + log2f(lsum)) * 0.6931471805599453f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Forward, PERSISTENT and pair-synchronous with a producer wave (round 4).
//
// What the cycle stamps of the first persistent attempts showed (tools/attn_stamps.py, profiles/r04_attn_fwd_stamps.txt):
//   * the kernel above is not short of bandwidth: a wave needs ~2,400 cycles per 32 x 32 tile step whose ~170 vector and 12 matrix
//     instructions issue in ~800 - it waits for LDS fragments it has just requested, and the two waves of a SIMD sit in the same phase
//     (both in the score MFMAs, then both in the softmax), so the matrix and the vector pipe take turns instead of running together;
//   * the row's output stores (8-byte pieces at a 1.5 KB row stride: 32 partial lines per instruction) hold the CU's vector-memory path
//     for ~3,000 cycles and the NEXT loads queue behind them for as long again;
//   * LDS-DMA issued in a burst by the computing waves blocks them until the memory path has taken every piece.
// Hence this structure - one 8-wave workgroup per CU walking through its (image, head) pairs:
//   * waves 0 .. 6 each own one 32-query ROW TILE of the current pair (T = 197: exactly 7); wave 7 is the PRODUCER: it streams the next
//     pair's K and V into the other LDS slot by LDS-DMA (global_load_lds_dwordx4, paced), waits for them, and meets the others at the one
//     barrier per pair.  Nobody else ever waits for a K / V byte, and a blocked DMA issue blocks nothing but the producer.
//   * per key tile ONE software-pipelined step, hand-ordered (sched_barrier after every group): the K fragments of tile t + 1 and the V
//     fragments of tile t are requested first; the six score MFMAs of tile t + 1 are spread over the exp2 work of tile t, the six PV MFMAs
//     of tile t over its hi / lo packing and the row maximum of tile t + 1 - every MFMA has vector work behind it, every fragment is
//     requested a block before its use.  Online softmax per tile with a LAZY maximum (rescale only when a tile exceeds the reference by
//     2^24: p stays f32 until the split, whose relative precision does not depend on the scale).
//   * the finished row goes through a wave-private LDS tile and leaves as four 16-byte-per-lane stores of WHOLE 128-byte lines.
//   * LDS images are UNPADDED (an LDS-DMA writes 1 KB lane-linear), 16-byte chunks XOR-swizzled through the DMA's per-lane SOURCE
//     address: K by (row >> 1) & 7 (split: 128-B rows; plain 64-B rows: (row >> 2) & 3) - the b128 row reads are conflict free; V (split)
//     swaps the hi / lo halves of rows 2, 3 (mod 4) - the four rows of a transposing read fall into four disjoint 64-B bank windows
//     (the padded image of the kernel above is 2-way conflicted there).  The swizzles depend on the lane only: they fold into 4 + 2
//     per-lane offsets, tile / k-step / slot offsets stay immediates or uniform adds.
//   * An image holds Timg = Tn rounded up to the DMA piece (8 / 16 rows), not to 32: reads of the last key tile run past it into what
//     follows (K: masked to -inf whatever the bytes; V: probability exactly 0 times FINITE bytes - the overrun regions are zeroed once,
//     afterwards they hold zeros or another pair's finite K rows).
template <typename T> struct RingGeo {
    static constexpr bool SP = is_split<T>::value;
    static constexpr int RB = AttnT<T>::RB;    // bytes per image row
    static constexpr int CPR = RB / 16;        // 16-byte chunks per row
    static constexpr int RPP = 64 / CPR;       // rows per LDS-DMA piece (one wave instruction = 1 KB)
    static constexpr int SPB = RB + 16;        // pitch of the output staging tile
    static constexpr int NCW = 7;              // computing waves (wave 7 produces)
    static __host__ __device__ int timg(int Tn) { return (Tn + RPP - 1) / RPP * RPP; }
    static __host__ __device__ int pad_bytes(int Tn) { return (((Tn + 31) & ~31) - timg(Tn)) * RB; }
    static __host__ __device__ int lds_bytes(int Tn) { return 2 * 2 * timg(Tn) * RB + pad_bytes(Tn) + NCW * 32 * SPB; }
    static __device__ __forceinline__ int kswz(int row) { return SP ? (row >> 1) & 7 : (row >> 2) & 3; }
    static __device__ __forceinline__ int vswz(int row) { return SP ? ((row >> 1) & 1) << 2 : 0; }
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_off)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef MFVIT_ATTN_STAMP
// diagnostic build only (tools/attn_stamps.py): cycle stamps of workgroups 0 and 37, [2][16 rounds][8 waves][32 points]
__device__ unsigned long long* g_attn_stamps = nullptr;
#define ATTN_STAMP(i)                                                                                                            \
    do {                                                                                                                         \
        if (g_attn_stamps && (blockIdx.x == 0 || blockIdx.x == 37) && rnd_no < 16) {                                             \
            const unsigned long long t__ = __builtin_amdgcn_s_memtime();                                                         \
            if (lane == 0) g_attn_stamps[(((blockIdx.x ? 1 : 0) * 16 + rnd_no) * 8 + wave) * 32 + (i)] = t__;                   \
        }                                                                                                                        \
    } while (0)
#else
#define ATTN_STAMP(i) do {} while (0)
#endif
#define ATTN_SB() __builtin_amdgcn_sched_barrier(0)
#if defined(MFVIT_ATTN_STAMP) && defined(MFVIT_ATTN_STAMP_FINE)
#define ATTN_FSTAMP(i) do { if (JOB < 0 && t == 4) ATTN_STAMP(i); } while (0)
#else
#define ATTN_FSTAMP(i) do {} while (0)
#endif
// the other half-wave's value (lane ^ 32) by v_permlane32_swap: VALU only - a ds_bpermute would sit in the LDS queue behind the fragment reads
__device__ __forceinline__ float xhalf(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

template <typename T, int NPX>
__global__ __launch_bounds__(512) void attn_fwd_pp_kernel(const typename Vec4<T>::elem* __restrict__ qkv, typename AttnT<T>::OE* __restrict__ out,
                                                          float* __restrict__ lse, int Tn, int H, float scale, int npair) {
    typedef AttnT<T> A;
    typedef RingGeo<T> G;
    typedef typename A::E E;
    typedef typename A::OE OE;
    typedef typename A::frag_t frag_t;
    constexpr int EP = A::EP, RB = G::RB, CPR = G::CPR, RPP = G::RPP, LO = A::SP ? 1 : 0, SPB = G::SPB, NCW = G::NCW, NP = PParts<T, NPX>::value;
    constexpr int NQK = A::SP ? 6 : 2;                                 // MFMAs of one score tile
    constexpr int NPV = A::SP ? 2 * (NP + 1) : 2;                      // ... of one PV tile (P in NP parts)
    constexpr bool PSUM = A::X && NP == 1;                             // row sum taken from the rounded probabilities (pack8x)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nt = (Tn + 31) >> 5;                                     // row tiles = key tiles per pair (<= NCW: checked by the launcher)
    const int Timg = G::timg(Tn);
    const int img = Timg * RB, slotb = 2 * img;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long hs = (long)H * HD * EP, rs = 3 * hs;
    const int npl = (npair - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // pairs of this workgroup: blockIdx.x + k * gridDim.x
    const int NPC = 2 * (Timg / RPP);                                  // LDS-DMA pieces per pair
    char* stage = lds + 2 * slotb + G::pad_bytes(Tn) + wave * (32 * SPB);
    {   // the V overrun of slot 0 (= the head of slot 1) and of slot 1 (= the pad behind it) start out as zeros
        const int pad = G::pad_bytes(Tn);
        for (int i = threadIdx.x * 16; i < pad; i += 512 * 16) {
            *(uint4*)(lds + slotb + i) = make_uint4(0, 0, 0, 0);
            *(uint4*)(lds + 2 * slotb + i) = make_uint4(0, 0, 0, 0);
        }
        // the staging tiles too: the first pair's steps send the (not yet written) tile to the row's own output lines - zeros, not stale LDS
        for (int i = threadIdx.x * 16; i < NCW * 32 * SPB; i += 512 * 16) *(uint4*)(lds + 2 * slotb + pad + i) = make_uint4(0, 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ATTN_SB();
    }
    auto pair_bid = [&](int k) __attribute__((always_inline)) {
        return xcd_remap((int)blockIdx.x + k * (int)gridDim.x, npair);                 // the pairs of a workgroup stay on its XCD
    };
    // LDS-DMA piece pi (0 .. NPC - 1: K image pieces, then V image pieces) of the pair at `base` into slot `slot`
    auto dma_piece = [&](const E* base, int slot, int pi) __attribute__((always_inline)) {
        const int which = pi >= (NPC >> 1) ? 1 : 0;
        const int pj = pi - which * (NPC >> 1);
        const int row = pj * RPP + lane / CPR, cpos = lane % CPR;
        const int cc = cpos ^ (which ? G::vswz(row) : G::kswz(row));                  // the chunk that belongs at this LDS position
        const int rowc = row < Tn ? row : Tn - 1;                                      // image rows past Tn: a finite copy of the last row
        glds16(base + (long)rowc * rs + (1 + which) * hs + 8 * cc, (unsigned)slot * (unsigned)slotb + (unsigned)which * (unsigned)img + (unsigned)pj * 1024u);
    };
    auto load_q = [&](const E* base, int qt, frag_t (&qf)[2], frag_t (&ql)[2]) __attribute__((always_inline)) {
        const int q = qt * 32 + (lane & 31);
        const int qc = q < Tn ? q : Tn - 1;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[s] = *(const frag_t*)(base + (long)qc * rs + 16 * s + 8 * (lane >> 5));
            if constexpr (A::SP) ql[s] = *(const frag_t*)(base + (long)qc * rs + 32 + 16 * s + 8 * (lane >> 5));
            else ql[s] = qf[s];
        }
    };
    // "use" the prefetched Q registers right behind the explicit wait: the compiler then places ITS wait for these loads here, in front of
    // the row's stores - left to the first MFMA of the next round it would be a vmcnt(0) that also waits for those stores
    auto touch_q = [](frag_t (&qf)[2], frag_t (&ql)[2]) __attribute__((always_inline)) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            union { frag_t f; u32x4_t u; } x, y;
            x.f = qf[s];
            y.f = ql[s];
            asm volatile("" : "+v"(x.u), "+v"(y.u));
            qf[s] = x.f;
            ql[s] = y.f;
        }
    };
    const bool producer = wave == NCW;
    const bool act = wave < nt;                                        // this wave owns row tile `wave` of every pair
    frag_t qnf[2], qnl[2];
    {   // prologue: everybody brings in pair 0
        const int bid = pair_bid(0);
        const E* base = qkv + (long)(bid / H) * Tn * rs + (bid % H) * HD * EP;
        if (act) load_q(base, wave, qnf, qnl);
        for (int pi = wave; pi < NPC; pi += 8) dma_piece(base, 0, pi);
        wait_vm<0>();
        touch_q(qnf, qnl);
    }
    const float c = scale * 1.4426950408889634f;
    // per-lane LDS byte offsets (from the start of a slot) of this lane's K row fragments [part][k-step] and V transposing reads [part]
    int koff[2][2], voff[2];
    {
        const int r = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int s = 0; s < 2; ++s) koff[part][s] = r * RB + 16 * (((A::SP ? 4 * part : 0) + 2 * s + hh) ^ G::kswz(r));
        const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3, vr = 4 * hh + q4;
#pragma unroll
        for (int part = 0; part < 2; ++part) voff[part] = img + vr * RB + (((A::SP ? 64 * part : 0) + 32 * g1 + 8 * p4) ^ (16 * G::vswz(vr)));
    }
    constexpr int RPI = 64 / CPR;                                      // rows per output store instruction (whole lines: CPR lanes x 16 B per row)
    constexpr int NST = 32 / RPI;                                      // store instructions per row tile
    // the PREVIOUS pair's row waits in the staging tile: its NST stores (and the log-sum-exp) leave one per key-tile step of the current
    // pair - a burst of stores at the end of a row held the vector-memory path for ~3,000 cycles, and the next loads queued behind it
    // (first pair: nothing is pending yet - the steps then send the unwritten staging tile to the row's OWN output lines, which the same wave
    // overwrites with the real row one pair later, in program order)
    OE* pend_out = nullptr;
    int pend_rows = 1;                                                 // valid rows of the pending tile
    auto flush_one = [&](int i) __attribute__((always_inline)) {
        const int r = i * RPI + lane / CPR, ch = lane % CPR;
        if (r < pend_rows) {
            const uint4 v = *(const uint4*)(stage + r * SPB + 16 * ch);
            *(uint4*)((char*)(pend_out + (long)r * H * HD * EP) + 16 * ch) = v;
        }
    };
    int rnd_no = -1;
    (void)rnd_no;
    for (int kp = 0; kp < npl; ++kp) {
        ++rnd_no;
        ATTN_STAMP(10);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                  // pair kp has landed (producer's wait), slot (kp + 1) & 1 is free
        ATTN_SB();
        ATTN_STAMP(0);
        const int bid = pair_bid(kp);
        const int b = bid / H, h = bid % H;
        const int bidn = pair_bid(kp + 1 < npl ? kp + 1 : kp);
        const E* basen = qkv + (long)(bidn / H) * Tn * rs + (bidn % H) * HD * EP;
        if (producer) {
            if (kp + 1 < npl) {
                for (int pi = 0; pi < NPC; ++pi) {
                    dma_piece(basen, (kp + 1) & 1, pi);
                    __builtin_amdgcn_s_sleep(1);                       // paced: ~50 pieces over the round, never a burst
                }
                wait_vm<0>();
            }
            continue;
        }
        if (!act) continue;
#ifdef MFVIT_FWD_STAGGER
        if (wave >= 4) __builtin_amdgcn_s_sleep(MFVIT_FWD_STAGGER);
#endif
        frag_t qf[2], ql[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) { qf[s] = qnf[s]; ql[s] = qnl[s]; }
        if (kp == 0) {
            pend_out = out + (((long)b * Tn + wave * 32) * H * HD + h * HD) * EP;
            pend_rows = Tn - wave * 32 < 32 ? Tn - wave * 32 : 32;
        }
        // running per-lane LDS addresses of the current key tile (advanced by one tile per step: tile offsets stay immediates)
        const char* slot = lds + (kp & 1) * slotb;
        const char* ka[2][2];
        const char* va[2];
#pragma unroll
        for (int part = 0; part <= LO; ++part) {
#pragma unroll
            for (int s = 0; s < 2; ++s) ka[part][s] = slot + koff[part][s];
            va[part] = slot + voff[part];
        }
        float m2 = -INFINITY, lsum = 0.f;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        // ---- fragment reads: K row fragments of the tile `dt` tiles ahead of the running address, V transposed fragments likewise
        auto read_k = [&](frag_t (&kf)[2][2], int dt) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int part = 0; part <= LO; ++part) kf[part][s] = *(const frag_t*)(ka[part][s] + dt * 32 * RB);
        };
        auto read_v = [&](frag_t (&vf)[2][2], int dt) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int part = 0; part <= LO; ++part) {
                    union { struct { s16x4 a, b; } s2; frag_t v; } u;
                    const char* p0 = va[part] + (dt * 32 + 16 * s) * RB;
                    u.s2.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p0);
                    u.s2.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 8 * RB));
                    vf[part][s] = u.v;
                }
        };
        // MFMA number i of a tile product: split: k-step i / 3, terms lo * hi, hi * lo, hi * hi; plain: k-step i
        auto mma_i = [&](int i, const frag_t (&a)[2][2], const frag_t (&bh)[2], const frag_t (&bl)[2], f32x16& acc) __attribute__((always_inline)) {
            if constexpr (A::SP) {
                const int s = i / 3, term = i % 3;
                acc = MmaTraits_mma(term == 0 ? a[1][s] : a[0][s], term == 1 ? bl[s] : bh[s], acc);
            } else {
                acc = MmaTraits_mma(a[0][i], bh[i], acc);
            }
        };
        // MFMA number i of the PV product (B = P in NP parts): NP == 1: k-step i / 2, terms v_lo * p, v_hi * p
        auto mma_pv = [&](int i, const frag_t (&a)[2][2], const frag_t (&bh)[2], const frag_t (&bl)[2], f32x16& acc) __attribute__((always_inline)) {
            if constexpr (A::SP && NP == 1) {
                const int s = i / 2, term = i % 2;
                acc = MmaTraits_mma(term == 0 ? a[1][s] : a[0][s], bh[s], acc);
            } else {
                mma_i(i, a, bh, bl, acc);
            }
        };
        auto max16 = [&](const f32x16& sc) __attribute__((always_inline)) {   // (v_max3 by hand: fmaxf on MFMA results gets a canonicalising v_max each)
            float cm;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(cm) : "v"(sc[0]), "v"(sc[1]), "v"(sc[2]));
#pragma unroll
            for (int r = 3; r < 15; r += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(cm) : "v"(cm), "v"(sc[r]), "v"(sc[r + 1]));
            asm("v_max_f32 %0, %1, %2" : "=v"(cm) : "v"(cm), "v"(sc[15]));
            return cm;
        };
        // The same for tile 0, whose score MFMAs are the instructions right in front of it: compiler-visible reads.  The hazard recogniser counts the
        // MFMA -> VALU wait states (11 behind an 8-pass MFMA) for its own instructions only - inline asm gets none, and max16 here read accumulator
        // registers the last MFMA had not written yet (round 6: a reference maximum from partial scores - the softmax stays correct for any reference,
        // so results moved by roundings only, ~2 % of the rows, differently from run to run; found by tools/attn_fwd_det_probe.py).  In the steps the
        // tile's last score MFMA is four or more MFMAs (>= 128 cycles) ahead of max16.
        auto max16_first = [&](const f32x16& sc) __attribute__((always_inline)) {
            float cm = fmaxf(sc[0], sc[1]);
#pragma unroll
            for (int r = 2; r < 16; ++r) cm = fmaxf(cm, sc[r]);
            return cm;
        };
        auto mask_tile = [&](f32x16& sc, int kt) __attribute__((always_inline)) {
            const int lim = Tn - kt * 32;                              // (uniform) valid keys of this tile
            if (lim < 32) {                                            // keys past Tn: whatever bytes the image overrun holds
                const int lr = lim - 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = (r & 3) + 8 * (r >> 2) >= lr ? -INFINITY : sc[r];
            }
        };
        // One key tile.  `cur` holds S^T of tile t, `nxt` receives S^T of tile t + 1 (K fragments fin_k), the PV product of tile t uses fin_v;
        // the fragments of the NEXT step (K of tile t + 2, V of tile t + 1) are requested in the middle of this one into fout_k / fout_v.
        // cmax = this lane's maximum over cur (unscaled), produced by the previous step.
        // Side jobs ride on the first steps (JOB = step number 0 .. 3, -1: none): step i sends out store i of the pending tile (i < NST), step 0
        // the pending log-sum-exp, steps 1 and 2 fetch the next pair's Q fragments - all compile-time, the step bodies stay straight-line.
        auto step = [&](f32x16& cur, f32x16& nxt, frag_t (&fin_k)[2][2], frag_t (&fin_v)[2][2], frag_t (&fout_k)[2][2], frag_t (&fout_v)[2][2],
                        int t, float& cmax, auto job) __attribute__((always_inline)) {
            constexpr int JOB = decltype(job)::value;
            {   // running maximum, lazily: rescale only when this tile exceeds the reference by more than 2^24 in the probabilities
                const float cm = fmaxf(cmax, xhalf(cmax)) * c;
                if (__builtin_amdgcn_ballot_w64(cm > m2 + A::LAZY) != 0) {
                    const float mn = fmaxf(m2, cm);
                    const float alpha = __builtin_amdgcn_exp2f(m2 - mn);
                    m2 = mn;
                    lsum *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] *= alpha;
                }
            }
            ATTN_SB();
#pragma unroll
            for (int r = 0; r < 16; ++r) nxt[r] = 0.f;
            auto ex = [&](int r0, int r1) __attribute__((always_inline)) {
#pragma unroll
                for (int r = r0; r < r1; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(cur[r], c, -m2));
                    cur[r] = pr;
                    if constexpr (!PSUM) lsum += pr;
                }
            };
            frag_t ph[2], pl[2];
            auto pk = [&](int s) __attribute__((always_inline)) {
                if constexpr (PSUM) pack8x<1, true>(cur, s, ph[s], pl[s], &lsum);
                else pack8<T, NP>(cur, s, ph[s], pl[s]);
                if constexpr (!A::SP) pl[s] = ph[s];
            };
            // (branch-free: rows past the tile's end repeat its last row - same bytes to the same address; the Q fetch of the last pair
            // repeats the current one)
            const int srow = min(JOB * RPI + lane / CPR, pend_rows - 1);   // row of the pending tile this step sends out
            uint4 sv = make_uint4(0, 0, 0, 0);
            if constexpr (JOB >= 0 && JOB < NST) sv = *(const uint4*)(stage + srow * SPB + 16 * (lane % CPR));
            auto side = [&]() __attribute__((always_inline)) {
                if constexpr (JOB >= 0 && JOB < NST) *(uint4*)((char*)(pend_out + (long)srow * H * HD * EP) + 16 * (lane % CPR)) = sv;
                if constexpr (JOB == 1 || JOB == 2) {
                    constexpr int s = JOB - 1;
                    const int q = wave * 32 + (lane & 31);
                    const int qc = q < Tn ? q : Tn - 1;
                    qnf[s] = *(const frag_t*)(basen + (long)qc * rs + 16 * s + 8 * (lane >> 5));
                    if constexpr (A::SP) qnl[s] = *(const frag_t*)(basen + (long)qc * rs + 32 + 16 * s + 8 * (lane >> 5));
                    else qnl[s] = qnf[s];
                }
            };
            // score MFMAs of tile t + 1 under the exp2 work of tile t
            if constexpr (A::SP) {
                ATTN_FSTAMP(11);
                mma_i(0, fin_k, qf, ql, nxt); ex(0, 3); ATTN_SB();
                ATTN_FSTAMP(12);
                mma_i(1, fin_k, qf, ql, nxt); ex(3, 6); ATTN_SB();
                ATTN_FSTAMP(13);
                mma_i(2, fin_k, qf, ql, nxt); ex(6, 8); ATTN_SB();
                ATTN_FSTAMP(14);
                mma_i(3, fin_k, qf, ql, nxt); pk(0); ATTN_SB();
                ATTN_FSTAMP(15);
                read_k(fout_k, 2);
                read_v(fout_v, 1);
                ATTN_SB();
                ATTN_FSTAMP(16);
                mma_i(4, fin_k, qf, ql, nxt); ex(8, 11); ATTN_SB();
                ATTN_FSTAMP(17);
                mma_i(5, fin_k, qf, ql, nxt); ex(11, 14); ATTN_SB();
                ATTN_FSTAMP(18);
                // PV MFMAs of tile t under the rest of its exp2 / packing and the row maximum of tile t + 1
                mma_pv(0, fin_v, ph, pl, o); ex(14, 16); ATTN_SB();
                ATTN_FSTAMP(19);
                mma_pv(1, fin_v, ph, pl, o); pk(1); ATTN_SB();
                ATTN_FSTAMP(20);
                mma_pv(2, fin_v, ph, pl, o); ATTN_SB();
                side();
                mma_pv(3, fin_v, ph, pl, o); ATTN_SB();
                if constexpr (NPV > 4) {
                    mma_pv(4, fin_v, ph, pl, o); ATTN_SB();
                    mma_pv(5, fin_v, ph, pl, o); ATTN_SB();
                }
                ATTN_FSTAMP(21);
                mask_tile(nxt, t + 1);
                cmax = max16(nxt);
                ATTN_FSTAMP(22);
            } else {
                mma_i(0, fin_k, qf, ql, nxt); ex(0, 8); ATTN_SB();
                read_k(fout_k, 2);
                read_v(fout_v, 1);
                ATTN_SB();
                mma_i(1, fin_k, qf, ql, nxt); pk(0); ex(8, 16); ATTN_SB();
                mma_i(0, fin_v, ph, pl, o); pk(1); ATTN_SB();
                side();
                mma_i(1, fin_v, ph, pl, o); ATTN_SB();
                mask_tile(nxt, t + 1);
                cmax = max16(nxt);
            }
            // advance the running addresses by one key tile
#pragma unroll
            for (int part = 0; part <= LO; ++part) {
#pragma unroll
                for (int s = 0; s < 2; ++s) ka[part][s] += 32 * RB;
                va[part] += 32 * RB;
            }
        };
        // The LAST key tile: no next tile to score, and only the accumulator rows that hold keys below Tn are worked on - row group g
        // (registers 4 g .. 4 g + 3 = keys 8 g .. 8 g + 7 of the tile) is skipped when 8 g >= the tile's valid keys, a whole k-step of the PV
        // product when both its groups are (T = 197: 5 keys left - one group of four, one k-step of two).
        auto last_step = [&](f32x16& cur, frag_t (&fin_v)[2][2], int t, float cmax) __attribute__((always_inline)) {
            {
                const float cm = fmaxf(cmax, xhalf(cmax)) * c;
                if (__builtin_amdgcn_ballot_w64(cm > m2 + A::LAZY) != 0) {
                    const float mn = fmaxf(m2, cm);
                    const float alpha = __builtin_amdgcn_exp2f(m2 - mn);
                    m2 = mn;
                    lsum *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] *= alpha;
                }
            }
            const int ng = (Tn - t * 32 + 7) >> 3;                     // (uniform) row groups with a valid key: 1 .. 4
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (2 * s < ng) {
#pragma unroll
                    for (int g = 2 * s; g < 2 * s + 2; ++g) {
                        if (g < ng) {
#pragma unroll
                            for (int r = 4 * g; r < 4 * g + 4; ++r) {
                                const float pr = __builtin_amdgcn_exp2f(fmaf(cur[r], c, -m2));   // (masked keys: exp2(-inf) = 0)
                                cur[r] = pr;
                                if constexpr (!PSUM) lsum += pr;
                            }
                        } else {
#pragma unroll
                            for (int r = 4 * g; r < 4 * g + 4; ++r) cur[r] = 0.f;
                        }
                    }
                    frag_t ph, pl;
                    if constexpr (PSUM) pack8x<1, true>(cur, s, ph, pl, &lsum);
                    else pack8<T, NP>(cur, s, ph, pl);
                    if constexpr (!A::SP) pl = ph;
                    o = mmap<T, NP>(fin_v[0][s], fin_v[LO][s], ph, pl, o);
                }
            }
        };
        f32x16 sA, sB;
        float cmax;
        frag_t fk0[2][2], fv0[2][2], fk1[2][2], fv1[2][2];
        {   // S^T of tile 0; fragments of step 0 (K of tile 1, V of tile 0)
            frag_t kf[2][2];
            read_k(kf, 0);
            read_k(fk0, 1);
            read_v(fv0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) sA[r] = 0.f;
#pragma unroll
            for (int i = 0; i < NQK; ++i) mma_i(i, kf, qf, ql, sA);
            mask_tile(sA, 0);
            cmax = max16_first(sA);
        }
        ATTN_STAMP(2);
        // (nt >= 4: checked by the launcher)
        step(sA, sB, fk0, fv0, fk1, fv1, 0, cmax, std::integral_constant<int, 0>{});
        step(sB, sA, fk1, fv1, fk0, fv0, 1, cmax, std::integral_constant<int, 1>{});
        step(sA, sB, fk0, fv0, fk1, fv1, 2, cmax, std::integral_constant<int, 2>{});
        step(sB, sA, fk1, fv1, fk0, fv0, 3, cmax, std::integral_constant<int, 3>{});
        // (steps 0 .. 3 were full steps: nt >= 5, checked by the launcher; tiles 4 .. nt - 2 likewise, tile nt - 1 is the last)
        int t = 4;
        for (; t + 2 < nt; t += 2) {
            step(sA, sB, fk0, fv0, fk1, fv1, t, cmax, std::integral_constant<int, -1>{});
            step(sB, sA, fk1, fv1, fk0, fv0, t + 1, cmax, std::integral_constant<int, -1>{});
        }
        if (t + 1 < nt) {
            step(sA, sB, fk0, fv0, fk1, fv1, t, cmax, std::integral_constant<int, -1>{});
            last_step(sB, fv1, t + 1, cmax);
        } else {
            last_step(sA, fv0, t, cmax);
        }
        ATTN_STAMP(3);
        lsum += xhalf(lsum);
        // ---- the row goes into the wave's staging tile: lane (q, h) writes its 8-byte pieces; it leaves during the next pair's steps
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (the pending tile has been read out)
        {
            const float inv = 1.0f / lsum;
            char* srow = stage + (lane & 31) * SPB;
            typedef typename Vec4<typename A::OT>::type V4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                V4 oh, ol;
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    OE h0, h1, l0, l1;
                    cvt_pair<OE, A::SP>(o[4 * g + j] * inv, o[4 * g + j + 1] * inv, h0, h1, l0, l1);
                    oh[j] = h0;
                    oh[j + 1] = h1;
                    if constexpr (A::SP) { ol[j] = l0; ol[j + 1] = l1; }
                }
                *(V4*)(srow + (8 * g + 4 * (lane >> 5)) * 2) = oh;
                if constexpr (A::SP) *(V4*)(srow + 64 + (8 * g + 4 * (lane >> 5)) * 2) = ol;
            }
        }
        pend_out = out + (((long)b * Tn + wave * 32) * H * HD + h * HD) * EP;
        pend_rows = Tn - wave * 32 < 32 ? Tn - wave * 32 : 32;
        if (lane < pend_rows) lse[((long)b * H + h) * Tn + wave * 32 + lane] = (m2 + log2f(lsum)) * 0.6931471805599453f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vm<0>();                                                  // the next pair's Q fragments have landed; the stores in front of them left half a row ago
        touch_q(qnf, qnl);
        ATTN_SB();
        ATTN_STAMP(4);
    }
    if (!producer && act) {                                            // the last row
        for (int i = 0; i < NST; ++i) flush_one(i);
    }
}

// One 8-wave workgroup per (image, head): Q, K, V, dO of the head are read from HBM exactly once into four LDS images
// (padded pitch: conflict-free row reads; unpadded when only that fits the 160 KB of LDS).  -lse/scale and
// -D = -rowsum(dO o O) enter the score MFMAs as accumulator initial values, so S - lse/scale and dP - D come out of the
// matrix core and the VALU work per element is mul, exp2, mul, cvt (+ the hi / lo split of P and dS for split tensors).
// 2^E (and 2^-E) with mx 2^E in [4, 8) for the largest |dO| of a pair: dO 2^E <= 8 leaves dP = dO . v (<= 8 x 32 |v|) and dS = P (dP - D)
// inside fp16's range for |v| up to ~250, and everything above 2^-2 keeps its 22 bits in split fp16.  All-zero dO: E = 100 (anything works).
__device__ __forceinline__ void pow2_scale(float mx, float& s, float& sinv) {
    const int eb = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xffu);
    int e = 129 - eb;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    s = __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
    sinv = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
}

template <typename T, int PITCH, int NPX>
__global__ __launch_bounds__(512) void attn_bwd_mfma_kernel(const typename Vec4<T>::elem* __restrict__ qkv, const typename AttnT<T>::OE* __restrict__ out,
                                                            const typename AttnT<T>::OE* __restrict__ dout, const float* __restrict__ lse,
                                                            typename AttnT<T>::OE* __restrict__ dqkv, int Tn, int H, float scale, int nbh) {
    typedef AttnT<T> A;
    typedef typename A::E E;
    typedef typename A::OE OE;
    typedef typename A::OT OT;
    typedef typename A::frag_t frag_t;
    typedef typename A::ofrag_t ofrag_t;
    constexpr int EP = A::EP, CPR = A::RB / 16, LO = A::SP ? 1 : 0, NP = PParts<T, NPX>::value;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float red[8];                                           // (split fp16: block maximum of |dO|)
    (void)red;
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
        const OE* obase = out + (long)b * Tn * os + h * HD * EP;
        const OE* dobase = dout + (long)b * Tn * os + h * HD * EP;
        // chunk i of this thread: row t0 + RSTEP i, 16-byte piece cidx - ONE per-lane address per tensor, the rows of the other chunks are a
        // uniform stride away (per-chunk addresses were spilled, and every scratch reload waits for ALL loads in flight: the prefetch ran
        // one chunk at a time)
        const E* rp = base + (long)t0 * rs + 8 * cidx;
        const OE* dp_ = dobase + (long)t0 * os + 8 * cidx;
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
            const unsigned voff = ((unsigned)t * (unsigned)os + 8u * (unsigned)cl) * (unsigned)sizeof(OE);
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
            union { uint4 u; ofrag_t f; } cd, co, cx;
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
    OE* dbase = dqkv + (long)b * Tn * rs + h * HD * EP;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's O pieces have landed ...
    __syncthreads();                                                   // ... and everybody else's
    reduce_D();
    float sE = 1.f, sEinv = 1.f;                                       // 2^E, 2^-E of this pair's dO (split fp16 only)
    if constexpr (A::X) {
        // dO (split bf16, gradient scale) -> split fp16 of dO 2^E, E from the pair's largest |dO| (hi parts: lanes cidx < 4): dP, D, dS and with
        // them dQ, dK, dV carry 2^E; the store multipliers take it out
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            union { uint4 u; ofrag_t f; } cd;
            cd.u = pd[i];
            if (cidx < 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf((float)cd.f[j]));
            }
        }
        mx = block_max<512>(mx, red);
        pow2_scale(mx, sE, sEinv);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            union { uint4 u; ofrag_t f; } me, ot;
            union { uint4 u; frag_t f; } res;
            me.u = pd[i];
            ot.u.x = __shfl_xor(me.u.x, 4, 64); ot.u.y = __shfl_xor(me.u.y, 4, 64);     // the other part of the same 8 d's
            ot.u.z = __shfl_xor(me.u.z, 4, 64); ot.u.w = __shfl_xor(me.u.w, 4, 64);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xs = ((float)me.f[j] + (float)ot.f[j]) * sE;
                const f16 hh = (f16)xs;
                res.f[j] = cidx < 4 ? hh : (f16)(xs - (float)hh);
            }
            pd[i] = res.u;
            pD[i] *= sE;
        }
    }
    stage();
    __syncthreads();                                                   // (also: Os may be refilled)
    if (vi + (int)gridDim.x < npair) fetch(vi + gridDim.x);
#ifdef MFVIT_BWD_OLD_STAGGER
    if (wave >= 4) __builtin_amdgcn_s_sleep(MFVIT_BWD_OLD_STAGGER);
#endif
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
                pack8<T, NP>(st, s, sh, sl);
                if constexpr (!A::SP) sl = sh;
                dq = mmap<T, NP>(tr_frag<T>(Ks, PITCH, kt * 32, s, lane, 0), tr_frag<T>(Ks, PITCH, kt * 32, s, lane, LO), sh, sl, dq);
            }
        }
        // (store addresses from a laundered lane id: carried across the persistent loop they were spilled, and the reload - placed right behind
        // the prefetch - waited for every load in flight)
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const int q = qt * 32 + (lane_s & 31);
        if (q < Tn) store_tile_T<OT>(dbase + (long)q * rs, dq, scale * sEinv, lane_s);
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
                pack8<T, NP>(sm, s, ph, pl);
                pack8<T, NP>(dp, s, sh, sl);
                if constexpr (!A::SP) { pl = ph; sl = sh; }
                dv = mmap<T, NP>(tr_frag<T>(dOs, PITCH, qt * 32, s, lane, 0), tr_frag<T>(dOs, PITCH, qt * 32, s, lane, LO), ph, pl, dv);
                dk = mmap<T, NP>(tr_frag<T>(Qs, PITCH, qt * 32, s, lane, 0), tr_frag<T>(Qs, PITCH, qt * 32, s, lane, LO), sh, sl, dk);
            }
        }
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const int k = kt * 32 + (lane_s & 31);
        if (k < Tn) {
            store_tile_T<OT>(dbase + (long)k * rs + hs, dk, scale * sEinv, lane_s);
            store_tile_T<OT>(dbase + (long)k * rs + 2 * hs, dv, sEinv, lane_s);
        }
    }
    __syncthreads();                                               // every wave is done with the images before the next pair overwrites them
    }
}

// X^T[d][col] accumulator tile (col on the lane) -> rows of a [.., HD] tensor of T as 16-byte stores: the half-waves exchange their 8-byte
// pieces (v_permlane32_swap) so that every lane owns 16 contiguous bytes of its row.  (Plain stores on purpose: the system-scope streaming form of
// common.cuh::store16_stream, which saves the tile GEMMs their write-allocate fetches, is acknowledged only when the data is out of the L2 - and the
// persistent kernels WAIT for their stores (vmcnt(0) before the next pair's LDS-DMA): single-pass backward 95 -> 133 us with it, round 5.)
template <typename T> __device__ __forceinline__ void store_tile_T16(typename Vec4<T>::elem* row_ptr, const f32x16& acc, float mul, int lane) {
    typedef typename Vec4<T>::elem E;
    unsigned hi[4][2], lo[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            E h0, h1, l0, l1;
            cvt_pair<E, is_split<T>::value>(acc[4 * g + j] * mul, acc[4 * g + j + 1] * mul, h0, h1, l0, l1);
            union { E e[2]; unsigned u; } a, b;
            a.e[0] = h0; a.e[1] = h1;
            hi[g][j >> 1] = a.u;
            if constexpr (is_split<T>::value) { b.e[0] = l0; b.e[1] = l1; lo[g][j >> 1] = b.u; }
        }
    const int hoff = (lane >> 5) * 16;
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(hi[k][0], hi[k + 1][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(hi[k][1], hi[k + 1][1], false, false);
        *(uint4*)((char*)row_ptr + 16 * k + hoff) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        if constexpr (is_split<T>::value) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(lo[k][0], lo[k + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(lo[k][1], lo[k + 1][1], false, false);
            *(uint4*)((char*)row_ptr + 64 + 16 * k + hoff) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward, SINGLE PASS (round 4): S, P, dP and dS of a (query tile, key tile) pair are computed ONCE.
//
// The two-phase kernel above runs two phases per (image, head): wave = query tile (dQ needs S^T, dS^T: keys in the accumulator ROWS, the contraction
// index of the MFMA) and wave = key tile (dK, dV need S, dS: queries in the rows).  Every score tile is therefore computed twice - 42 MFMAs and
// ~380 VALU instructions per tile pair where 30 and ~200 would do, in a kernel whose time is the SUM of its VALU and MFMA time.  Here:
//   * waves 0 .. 6 own a KEY tile each (K, V fragments of the tile stay in registers for the whole pair) and walk through the query tiles
//     TOGETHER, one step per query tile: S[q][key], dP, P, dS, then dV += dO^T P and dK += Q^T dS as in phase B above;
//   * the hi / lo fragments of dS they have packed for the dK product are ALSO written - the same 16 bytes, split in two 8-byte halves -
//     into a 4 KB tile of LDS laid out [key][query] in the swizzled image format: read back with ds_read_b64_tr_b16 they are dS^T fragments
//     with the keys in the contraction index;
//   * wave 7 (the helper) turns them into dQ: one step behind the others it reads the seven tiles of the step (two buffers, one barrier per
//     step) and accumulates dQ^T[d][q] = sum over key tiles K_j^T dS_j^T in two accumulators (even / odd key tiles: no dependent MFMA pairs) -
//     the K^T fragments of all seven key tiles live in its registers (112 of them) - and stores the finished dQ tile.  No atomics, no second
//     pass.  It runs with issue priority (s_setprio 3): it is the longest wave of every step and shares its SIMD with key wave 3.
//   * one copy of every image (4 x 25.6 KB + 2 x 28 KB of dS^T tiles + row constants = 161.5 KB) and NO V image: a key wave needs only its own
//     key tile of V, as fragments whose 16 bytes per lane are contiguous in a V row - four global loads per lane in front of the last step,
//     into the registers of the fragments that died a step earlier.  The K image is dead once the fragments are in registers (barrier Z),
//     query tile t of Q / dO behind the barrier of step t: the next pair's K pieces, its O rows (where V would be; needed for D only) and the
//     Q / dO rows of tile t - 1 stream in during step t as a WINDOW of 16 LDS-DMA pieces issued by key waves 0 - 2 (the first waves of their
//     SIMDs reach the step barrier ~1,000 cycles ahead of their partners; an LDS-DMA stalls the wave that issues it).  The last tile's pieces
//     are issued by the helper behind the last step barrier and waited for behind barrier Z of the next pair.
//   * images are unpadded (LDS-DMA), 16-byte chunks XOR-swizzled by f(row) = bit 1 of the row -> chunk bit 2, bits 2..3 -> chunk bits 0..1
//     (SpGeo::swz): b128 row reads AND ds_read_b64_tr_b16 transposed reads of the same image are bank-conflict free.
//   * row constants (-lse/scale, -D = -rowsum(dO o O)): wave w computes those of query tile w between the two barriers at the head of a
//     pair from the dO and O images (wave nt - 1: behind step 0, its rows arrive last); -lse/scale rows are written by the helper.
//   * the key waves and the helper run SEPARATE persistent loops (role split at the top): lane-derived offsets are re-derived per pair from
//     a laundered lane id - a spill reload is a compiler-placed vmcnt(0), i.e. a wait for every LDS-DMA in flight.
//   * gradients leave as 16-byte stores (v_permlane32_swap pairs the half-waves' 8-byte pieces).
// Padded keys: the last key tile's K / V fragments (and the helper's K^T elements) are zeroed past Tn in registers; padded queries carry
// -lse/scale = -1e30 (p = 0); rows past Tn of an image are never written (LDS-DMA lanes masked by v_cmpx inside the asm statement) and stay
// zero; tile reads past an image's rows see finite 16-bit data of the next region.
template <typename T> struct SpGeo {
    static constexpr bool SP = is_split<T>::value;
    static constexpr int RB = AttnT<T>::RB, CPR = RB / 16, RPP = 64 / CPR, NCW = 7;
    static constexpr int SCR = 32 * RB;                                // one wave's dS^T tile: 32 key rows
    static __host__ __device__ int timg(int Tn) { return (Tn + RPP - 1) / RPP * RPP; }
    static __host__ __device__ int tpad(int Tn) { return (Tn + 31) & ~31; }
    static __host__ __device__ int img(int Tn) { return timg(Tn) * RB; }
    // [dO][Q][O][K][dS^T tiles x 2][row constants] (off_v: the O rows - V itself never enters LDS): a tile read past an image's rows lands in the
    // next image / in dS^T values (finite 16-bit data)
    static __host__ __device__ int off_do(int) { return 0; }
    static __host__ __device__ int off_q(int Tn) { return img(Tn); }
    static __host__ __device__ int off_v(int Tn) { return 2 * img(Tn); }
    static __host__ __device__ int off_k(int Tn) { return 3 * img(Tn); }
    static __host__ __device__ int off_scr(int Tn, int b) { return 4 * img(Tn) + b * NCW * SCR; }
    static __host__ __device__ int off_ld(int Tn) { return 4 * img(Tn) + 2 * NCW * SCR; }              // -lse/scale rows, then -D rows
    static __host__ __device__ int lds_bytes(int Tn) { return off_ld(Tn) + 2 * tpad(Tn) * 4 + 32; }   // + the tiles' |dO| maxima (split fp16)
    static __device__ __forceinline__ int swz(int row) { return SP ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3); }
};

// NT: row tiles of a pair (= computing waves): 7 (T = 193 .. 224); other lengths stay with the two-phase kernels
template <typename T, int NT, int NPX, bool DM = false>     // DM: the pairs' |dO| maxima come from the producer of dO (domax)
__global__ __launch_bounds__(512) void attn_bwd_sp_kernel(const typename Vec4<T>::elem* __restrict__ qkv, const typename AttnT<T>::OE* __restrict__ out,
                                                          const typename AttnT<T>::OE* __restrict__ dout, const float* __restrict__ lse,
                                                          typename AttnT<T>::OE* __restrict__ dqkv, int Tn, int H, float scale, int npair,
                                                          const unsigned* __restrict__ domax) {
    typedef AttnT<T> A;
    typedef SpGeo<T> G;
    typedef typename A::E E;
    typedef typename A::OE OE;
    typedef typename A::OT OT;
    typedef typename A::frag_t frag_t;
    typedef typename A::ofrag_t ofrag_t;
    constexpr int EP = A::EP, RB = G::RB, CPR = G::CPR, RPP = G::RPP, LO = A::SP ? 1 : 0, NCW = G::NCW, SCR = G::SCR, NP = PParts<T, NPX>::value;
    constexpr int NM = A::SP ? 6 : 2;                                  // MFMAs of one 32 x 32 x 32 product of two tensors
    constexpr int NPT = A::SP ? NP + 1 : 1;                            // MFMAs per k-step of a product with P / dS (NP parts)
    constexpr int PLO = NP - 1;                                        // index of the last P / dS part
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int nt = NT;                                             // (checked by the launcher: (Tn + 31) / 32 == NT <= NCW)
    const int Timg = G::timg(Tn), Tpad = G::tpad(Tn), np = Timg / RPP;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hh = lane >> 5, kr = lane & 31;
    const long hs = (long)H * HD * EP, rs = 3 * hs, os = hs;
    const int npl = (npair - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int i = threadIdx.x * 16; i < G::lds_bytes(Tn); i += 512 * 16) *(uint4*)(lds + i) = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ATTN_SB();
    auto pair_bid = [&](int k) __attribute__((always_inline)) { return xcd_remap((int)blockIdx.x + k * (int)gridDim.x, npair); };
    // piece pj (RPP rows, 1 KB) of an image by LDS-DMA: scalar base of the piece + a per-lane byte offset that only depends on the parity of pj
    // (row = pj RPP + lrow: the swizzle looks at row bits 1 .. 3), lanes whose row is past Tn masked INSIDE the asm statement (no branch in the
    // compiler's view: the steps stay one basic block)
    int lrow = lane / CPR;
    // (plain arithmetic on three values: a two-element array indexed by pj & 1 - and a select between two captured variables just the same -
    // went to scratch memory, and every scratch load waits for ALL LDS-DMA in flight: the pipeline ran one piece at a time)
    unsigned sw_x, vrs_e, vos_e;      // sw_x: what an odd piece flips in the offset (low bits: the strides are multiples of 128 B)
    auto calc_dma = [&](int lane_) __attribute__((always_inline)) {
        lrow = lane_ / CPR;
        const int cpos = lane_ % CPR;
        const unsigned sw_e = 16u * (unsigned)(cpos ^ G::swz(lrow));
        sw_x = sw_e ^ (16u * (unsigned)(cpos ^ G::swz(RPP + lrow)));
        vrs_e = (unsigned)lrow * (unsigned)(rs * 2) + sw_e;
        vos_e = (unsigned)lrow * (unsigned)(os * 2) + sw_e;
    };
    calc_dma(lane);
    auto voff_rs = [&](int pj) __attribute__((always_inline)) { return vrs_e ^ (sw_x & (0u - (unsigned)(pj & 1))); };
    auto voff_os = [&](int pj) __attribute__((always_inline)) { return vos_e ^ (sw_x & (0u - (unsigned)(pj & 1))); };
    auto dma_sv = [&](const char* piece_base_, unsigned voff, int lds_off, int pj) __attribute__((always_inline)) {
        const unsigned long long pbv = (unsigned long long)piece_base_;    // (uniform by construction: pinned to scalar registers for the "s" operand)
        const unsigned pb_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pbv), pb_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pbv >> 32));
        const char* piece_base = (const char*)(((unsigned long long)pb_hi << 32) | pb_lo);   // (unsigned halves: the builtin returns int - OR-ing it in sign-extends)
        unsigned long long keep_exec;
        unsigned keep_m0;
        // (v_cmpx: the compare ANDs into exec itself and writes vcc - no scalar ALU instruction, so SCC is left alone.  The first version used
        // s_and_b64 exec, exec, vcc WITHOUT an "scc" clobber: a compare the compiler held in SCC across the statement was lost - it only showed when
        // an unrelated uniform branch made the compiler do so; WITH the clobber the scalar selects of the window arithmetic turned into vector
        // code, 100 spilled registers and vmcnt(0) waits in the step loop: 108 -> 178 us)
        asm volatile("s_mov_b64 %0, exec\n\tv_cmpx_gt_i32 vcc, %5, %6\n\ts_mov_b32 %1, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep_exec), "=&s"(keep_m0)
                     : "v"(voff), "s"(piece_base), "s"(__builtin_amdgcn_readfirstlane((unsigned)lds_off + (unsigned)pj * 1024u)),
                       "s"(__builtin_amdgcn_readfirstlane(Tn - pj * RPP)), "v"(lrow)
                     : "memory", "vcc");
    };
    // entry i (0 .. 15) of step window w: i < 8: piece 8 w + i of the K-then-O stream (window 0: 16 of them); i >= 8: piece i - 8 of query tile w - 1 of Q (then dO).
    // Everything is a select on scalars - indices past the end repeat the last piece (identical bytes into rows nobody reads any more).
    constexpr int PPT = 32 / RPP;                                      // pieces per tile and image
    auto dma_entry = [&](const E* ksrc, const OE* osrc, const E* qsrc, const OE* dosrc, int w, int i) __attribute__((always_inline)) {
        const bool kv = i < 8 || w == 0;
        int pi = w == 0 ? i : 16 + 8 * (w - 1) + i;                    // piece of the K (first) / O stream
        pi = pi < 2 * np ? pi : 2 * np - 1;
        const int second = pi >= np ? 1 : 0;
        int e = i - 8;                                                 // Q / dO piece of tile w - 1
        e = e < 2 * PPT ? e : 2 * PPT - 1;
        const int isdo = e >= PPT ? 1 : 0;
        int pq = (w - 1) * PPT + (e - isdo * PPT);
        pq = pq < np ? pq : np - 1;
        const int pj = kv ? pi - second * np : pq;
        const bool o_str = (kv && second) || (!kv && isdo);            // O and dO rows are os apart, K and Q rows rs
        const char* src = kv ? (second ? (const char*)osrc : (const char*)ksrc) : (isdo ? (const char*)dosrc : (const char*)qsrc);
        const long stride = o_str ? os : rs;
        const int off = kv ? (second ? G::off_v(Tn) : G::off_k(Tn)) : (isdo ? G::off_do(Tn) : G::off_q(Tn));
        const unsigned vsel = o_str ? 1u : 0u;
        const unsigned vo = (vrs_e + vsel * (vos_e - vrs_e)) ^ (sw_x & (0u - (unsigned)(pj & 1)));
        dma_sv(src + (long)pj * RPP * stride * 2, vo, off, pj);
    };
    const bool helper = wave == NCW;
    char* const dOs = lds + G::off_do(Tn);
    char* const Qs = lds + G::off_q(Tn);
    char* const Ks = lds + G::off_k(Tn);
    float* const Ls = (float*)(lds + G::off_ld(Tn));
    float* const Ds = Ls + Tpad;
    float* const Mx = Ds + Tpad;                                       // [8]: largest |dO| of the query tiles (split fp16; entry 7 stays 0)
    (void)Mx;
    float sE = 1.f, sEinv = 1.f;                                       // 2^E, 2^-E of the pair in hand (split fp16; uniform)

    // V never enters LDS: a key wave needs only ITS key tile of V, as B fragments whose 16 bytes per lane are 16 contiguous bytes of a V row -
    // it loads them straight from global memory into the fragment registers at the start of the last step (their last use of the pair is
    // behind them by then).  The image that would hold V holds the O rows instead (needed once, in the head of a pair, for D = rowsum(dO o O)):
    // they stream in with the K pieces during the steps.  (Round-4 history: O in 16 registers per lane was spilled in the waves that issue
    // LDS-DMA - load, wait, spill, four HBM latencies per pair; O through the idle dS^T buffer could only be requested behind the last step
    // barrier - most of one HBM latency exposed per pair.)
    char* const Os = lds + G::off_v(Tn);
    auto v_fetch = [&](frag_t (&vf)[2][2], const E* vsrc) __attribute__((always_inline)) {   // V rows of key tile `wave` (vsrc: the pair's V base)
        const int key = wave * 32 + kr, kc = key < Tn ? key : Tn - 1;
        const E* vp = vsrc + (long)kc * rs;
#pragma unroll
        for (int part = 0; part <= LO; ++part)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                union { uint4 u; frag_t f; } t;
                t.u = *(const uint4*)(vp + 8 * ((A::SP ? 4 * part : 0) + 2 * s2 + hh));
                vf[part][s2] = t.f;
            }
    };
    // -lse / scale rows of a pair: by the helper (four rows per lane), behind the last step barrier of the pair before (nobody reads the rows any more)
    float lsv[4];
    auto lse_fetch = [&](int bid_) __attribute__((always_inline)) {
        const float* lp = lse + ((long)(bid_ / H) * H + bid_ % H) * Tn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lane + 64 * i;
            lsv[i] = lp[r < Tn ? r : Tn - 1];
        }
    };
    auto lse_write = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lane + 64 * i;
            if (r < Tpad) Ls[r] = r < Tn ? -lsv[i] / scale : -1e30f;   // padded queries: p = exp2(-1e30 c) = 0
        }
    };
    // row constants of query tile `wave` of the pair whose dO image has landed
    auto consts = [&]() __attribute__((always_inline)) {
        const int q = wave * 32 + kr, qc = q < Tn ? q : Tn - 1;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            union { uint4 u; frag_t f; } o0, o1, d0, d1;
            o0.u = *(const uint4*)(Os + qc * RB + 16 * ((2 * hh + i) ^ G::swz(qc)));
            d0.u = *(const uint4*)(dOs + qc * RB + 16 * ((2 * hh + i) ^ G::swz(qc)));
            if constexpr (A::SP) {
                o1.u = *(const uint4*)(Os + qc * RB + 16 * ((4 + 2 * hh + i) ^ G::swz(qc)));
                d1.u = *(const uint4*)(dOs + qc * RB + 16 * ((4 + 2 * hh + i) ^ G::swz(qc)));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float ov = (float)o0.f[j], dv_ = (float)d0.f[j];
                if constexpr (A::SP) { ov += (float)o1.f[j]; dv_ += (float)d1.f[j]; }
                acc = fmaf(ov, dv_, acc);
            }
        }
        acc += xhalf(acc);
        if (hh == 0) Ds[q] = q < Tn ? -acc : 0.f;
    };
    // Split fp16.  The pair's scale 2^E comes from the largest |dO| of the pair, and that must be known before the first row of dO is converted -
    // while the last query tile's rows land after the first step barrier.  So every key wave fetches the hi parts of ITS query tile's dO rows of the
    // NEXT pair straight from global memory (two 16-byte loads per lane, beside the V fragments, in front of the last step), takes their largest
    // magnitude in the head of the next pair and publishes it in Mx[wave] in front of barrier X; behind X everybody (the helper too) reads the seven
    // maxima: no extra barrier, nobody waits for the late rows.  consts_x: D of query tile `wave` as above, times 2^E, and the row goes back into the
    // dO image as split FP16 of dO 2^E (same chunk positions) - every tile by the wave of the same number, the last one behind the first step barrier.
    // Round 6: when the PRODUCER of dO hands the pairs' maxima over (domax[image * H + head], f32 bits of the largest |dO|: the proj data gradient's
    // epilogue, gemm.hip) none of that happens - no prefetch, nothing published - and pair_scale is one scalar load per pair.
    uint4 dpf[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    // (lane-derived values of these helpers come from a laundered copy of the lane id handed in by the caller: derived from `lane` itself they are
    // loop invariants, which the compiler keeps across the pair loop in scratch - and every reload is a drain of the vector-memory queue)
    auto do_prefetch = [&](const OE* dosrc, int lane_) __attribute__((always_inline)) {
#ifdef MFVIT_SP_NOPF
        return;
#endif
        if constexpr (DM) return;
        const int q = wave * 32 + (lane_ & 31), qc = q < Tn ? q : Tn - 1;
        const OE* dp_ = dosrc + (long)qc * os + 16 * (lane_ >> 5);
        dpf[0] = *(const uint4*)dp_;
        dpf[1] = *(const uint4*)(dp_ + 8);
    };
    auto publish_max = [&]() __attribute__((always_inline)) {
        if constexpr (DM) return;
        // 16 bf16 magnitudes as 15-bit integers (monotonic in |x|): packed 16-bit maxima, then the wave's maximum without touching LDS
        const unsigned w[8] = {dpf[0].x, dpf[0].y, dpf[0].z, dpf[0].w, dpf[1].x, dpf[1].y, dpf[1].z, dpf[1].w};
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned a_ = w[i] & 0x7fff7fffu;
            asm("v_pk_max_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(a_));
        }
        unsigned mm = (m & 0xffffu) > (m >> 16) ? (m & 0xffffu) : (m >> 16);
        mm = max(mm, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mm, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
        mm = max(mm, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mm, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
        mm = max(mm, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mm, 0x141, 0xf, 0xf, false));   // row_half_mirror
        mm = max(mm, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mm, 0x140, 0xf, 0xf, false));   // row_mirror: every lane of a row of 16 holds the row's maximum
        const unsigned r0 = __builtin_amdgcn_readlane(mm, 0), r1 = __builtin_amdgcn_readlane(mm, 16), r2 = __builtin_amdgcn_readlane(mm, 32),
                       r3 = __builtin_amdgcn_readlane(mm, 48);
        const unsigned mw = max(max(r0, r1), max(r2, r3));
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) Mx[wave] = __builtin_bit_cast(float, mw << 16);
    };
    auto pair_scale = [&](int bid_) __attribute__((always_inline)) {
        float mx;
        if constexpr (DM) {
            // the producer's maximum is of the f32 values; the prefetch path sees their hi parts (bf16, round to nearest even): round the same way, so
            // that a maximum just below a power of two picks the same exponent on both paths
            const unsigned u = domax[bid_];
            mx = __builtin_bit_cast(float, (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
        } else {
            const float4 m0 = *(const float4*)Mx, m1 = *(const float4*)(Mx + 4);
            mx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
        }
        float s_, si_;
        pow2_scale(mx, s_, si_);
        sE = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_)));
        sEinv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, si_)));
    };
    auto consts_x = [&](int lane_) __attribute__((always_inline)) {
        const int kr = lane_ & 31, hh = lane_ >> 5;
        const int q = wave * 32 + kr, qc = q < Tn ? q : Tn - 1;
        float acc = 0.f;
        uint4 hw[2], lw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            union { uint4 u; ofrag_t f; } o0, o1, d0, d1;
            o0.u = *(const uint4*)(Os + qc * RB + 16 * ((2 * hh + i) ^ G::swz(qc)));
            d0.u = *(const uint4*)(dOs + qc * RB + 16 * ((2 * hh + i) ^ G::swz(qc)));
            o1.u = *(const uint4*)(Os + qc * RB + 16 * ((4 + 2 * hh + i) ^ G::swz(qc)));
            d1.u = *(const uint4*)(dOs + qc * RB + 16 * ((4 + 2 * hh + i) ^ G::swz(qc)));
            union { uint4 u; unsigned w[4]; } h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float oa = (float)o0.f[2 * j] + (float)o1.f[2 * j], ob = (float)o0.f[2 * j + 1] + (float)o1.f[2 * j + 1];
                const float da = (float)d0.f[2 * j] + (float)d1.f[2 * j], db = (float)d0.f[2 * j + 1] + (float)d1.f[2 * j + 1];
                acc = fmaf(oa, da, acc);
                acc = fmaf(ob, db, acc);
                const float x0 = da * sE, x1 = db * sE;
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h.w[j]) : "v"(x0), "v"(x1));
                asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l.w[j]) : "v"(h.w[j]), "v"(x0));
                asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l.w[j]) : "v"(h.w[j]), "v"(x1));
            }
            hw[i] = h.u;
            lw[i] = l.u;
        }
        acc += xhalf(acc);
        if (q < Tn) {
            if (hh == 0) Ds[q] = -acc * sE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *(uint4*)(dOs + q * RB + 16 * ((2 * hh + i) ^ G::swz(q))) = hw[i];
                *(uint4*)(dOs + q * RB + 16 * ((4 + 2 * hh + i) ^ G::swz(q))) = lw[i];
            }
        } else if (hh == 0) {
            Ds[q] = 0.f;
        }
    };

    // per-lane offsets inside an image: row fragments [part][k-step], transposed reads [part][first / second 4-row block], dS^T writes [part][g].
    // The key waves derive them AFRESH behind barrier Z of every pair, from a laundered lane id: carried across the head of the pair (the row
    // constants' arithmetic) they were spilled, the reloads sat in front of the step loop, and their wait - vmcnt(0) at the top of every
    // iteration - also waited for every LDS-DMA of the step before.
    int roff[2][2], toff[2][2], woff[2][4];
    auto calc_offsets = [&](int lane_) __attribute__((always_inline)) {
        const int hh_ = lane_ >> 5, kr_ = lane_ & 31;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int s = 0; s < 2; ++s) roff[part][s] = kr_ * RB + 16 * (((A::SP ? 4 * part : 0) + 2 * s + hh_) ^ G::swz(kr_));
        const int g1 = (lane_ >> 4) & 1, q4 = (lane_ & 15) >> 2, p4 = lane_ & 3;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int row = 4 * hh_ + q4 + 8 * rd;                 // (the swizzle of row tile * 32 + 16 s + row depends on these bits only)
                const int cch = (A::SP ? 4 * part : 0) + 2 * g1 + (p4 >> 1);
                toff[part][rd] = row * RB + 16 * (cch ^ G::swz(row)) + 8 * (p4 & 1);
            }
        // dS^T[key kr][queries 8 g + 4 hh .. + 3]: 8 bytes at byte 16 g + 8 hh of the row's hi (lo) part
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int g = 0; g < 4; ++g) woff[part][g] = kr_ * RB + 16 * (((A::SP ? 4 * part : 0) + g) ^ G::swz(kr_)) + 8 * hh_;
    };
    calc_offsets(lane);
    const float c = scale * 1.4426950408889634f;
    auto rd_row = [&](frag_t (&f)[2][2], const char* image, int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int part = 0; part <= LO; ++part) f[part][s] = *(const frag_t*)(image + tile * 32 * RB + roff[part][s]);
    };
    auto rd_tr = [&](frag_t (&f)[2], const char* image, int tile, int s, int lastpart = (AttnT<T>::SP ? 1 : 0)) __attribute__((always_inline)) {   // [part], k-step s
#pragma unroll
        for (int part = 0; part <= lastpart; ++part) {
            union { struct { s16x4 a, b; } s2; frag_t v; } u;
            const char* p0 = image + (tile * 32 + 16 * s) * RB;
            u.s2.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + toff[part][0]));
            u.s2.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + toff[part][1]));
            f[part] = u.v;
        }
    };
    auto mm = [&](int i, const frag_t (&a)[2][2], const frag_t (&bh)[2], const frag_t (&bl)[2], f32x16& acc, const f32x16& init) __attribute__((always_inline)) {
        if constexpr (A::SP) {
            const int s = i / 3, term = i % 3;
            acc = MmaTraits_mma(term == 0 ? a[1][s] : a[0][s], term == 1 ? bl[s] : bh[s], i == 0 ? init : acc);
        } else {
            acc = MmaTraits_mma(a[0][i], bh[i], i == 0 ? init : acc);
        }
    };
    // MFMA i (0 .. NPT - 1) of one k-step of a product with P / dS: a_lo b_hi, [a_hi b_lo,] a_hi b_hi
    auto mt = [&](int i, const frag_t (&a)[2], const frag_t& bh, const frag_t& bl, f32x16& acc) __attribute__((always_inline)) {
        if constexpr (A::SP) acc = MmaTraits_mma(i == 0 ? a[1] : a[0], (NP == 2 && i == 1) ? bl : bh, acc);
        else acc = MmaTraits_mma(a[0], bh, acc);
    };
    auto zero_frag = [&](frag_t& f) __attribute__((always_inline)) {
        union { uint4 u; frag_t v; } z;
        z.u = make_uint4(0, 0, 0, 0);
        f = z.v;
    };

    {   // prologue: pair 0 by everybody
        const int bid = pair_bid(0);
        const E* base = qkv + (long)(bid / H) * Tn * rs + (bid % H) * HD * EP;
        const OE* dob = dout + (long)(bid / H) * Tn * os + (bid % H) * HD * EP;
        for (int pj = wave; pj < np; pj += 8) {
            dma_sv((const char*)(base + hs) + (long)pj * RPP * rs * 2, voff_rs(pj), G::off_k(Tn), pj);
            dma_sv((const char*)base + (long)pj * RPP * rs * 2, voff_rs(pj), G::off_q(Tn), pj);
            dma_sv((const char*)dob + (long)pj * RPP * os * 2, voff_os(pj), G::off_do(Tn), pj);
            dma_sv((const char*)(out + (long)(bid / H) * Tn * os + (bid % H) * HD * EP) + (long)pj * RPP * os * 2, voff_os(pj), G::off_v(Tn), pj);
        }
        if (helper) lse_fetch(bid);
        wait_vm<0>();
        if (helper) lse_write();
    }

    int rnd_no = -1;
    (void)rnd_no;
    // The three roles run SEPARATE persistent loops (same barrier sequence: X, Z and one per step): values of one role are not live in another's
    // code (in one loop with role branches the helper's 112 K^T registers and the key waves' state were spilled around each other, and a scratch
    // reload waits for every LDS-DMA in flight).
    if (helper) {
        // The helper is the longest wave of every step and shares its SIMD with key wave 3, which reaches the step barriers ~2,000 cycles early:
        // issue priority to the helper (MFVIT_SP_PRIO: A/B macro).
#ifndef MFVIT_SP_NOPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        for (int kp = 0; kp < npl; ++kp) {
            const int bid = pair_bid(kp);
            const int b = bid / H, h = bid % H;
            OE* dbase = dqkv + (long)b * Tn * rs + h * HD * EP;
            const bool more = kp + 1 < npl;
            const int bidn = pair_bid(more ? kp + 1 : kp);
            const E* basen = qkv + (long)(bidn / H) * Tn * rs + (bidn % H) * HD * EP;
            const OE* dobn = dout + (long)(bidn / H) * Tn * os + (bidn % H) * HD * EP;
            const OE* obn = out + (long)(bidn / H) * Tn * os + (bidn % H) * HD * EP;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ++rnd_no;
            ATTN_STAMP(10);
            __builtin_amdgcn_s_barrier();                                  // X: this pair's images are in place, the dS^T tiles are free
            ATTN_SB();
            ATTN_STAMP(0);
        // ------------------------------------------------------------------------------------------- the helper: producer + dQ
            frag_t ktr[NCW][2][2];                                     // K^T fragments [key tile][k-step][part]
#pragma unroll
            for (int j = 0; j < nt; ++j) {
                rd_tr(ktr[j][0], Ks, j, 0);
                rd_tr(ktr[j][1], Ks, j, 1);
            }
            // keys past Tn (last tile): zero (element e of k-step s = key row 16 s + 8 (e >> 2) + 4 hh + (e & 3))
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int part = 0; part <= LO; ++part)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if ((nt - 1) * 32 + 16 * s + 8 * (e >> 2) + 4 * hh + (e & 3) >= Tn) ktr[nt - 1][s][part][e] = (E)0.0f;
            if constexpr (A::X) pair_scale(bid);                       // (the tiles' |dO| maxima were published in front of X)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // Z: K / V are dead
            ATTN_SB();
            ATTN_STAMP(1);
            // The last query tile's Q / dO pieces of THIS pair were requested behind the last step barrier of the pair before; nobody waited for
            // them in front of barrier X (most of an HBM latency per pair): they are waited for HERE, and the first step barrier hands them on -
            // their first readers are behind it (the row constants of the last query tile, the row fragments prefetched in step nt - 3).
            wait_vm<0>();
            // window 0 of the next pair's pieces (16 K pieces): the helper has nothing else to do until the first step's dS^T tiles exist
            for (int i = 0; i < 10; ++i) {                             // (entries 10 .. 15: key waves 0 - 2 in front of their first step)
                dma_entry(basen + hs, obn, basen, dobn, 0, i);
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll 1
            for (int u = 0; u < nt; ++u) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ATTN_STAMP(12 + u);
                __builtin_amdgcn_s_barrier();                          // step u's dS^T tiles are complete; query tile u of Q / dO is dead
                ATTN_SB();
                ATTN_STAMP(2 + u);
                if (u + 1 == nt) {                                     // behind the last step: the last query tile's pieces (nobody else is left to issue them;
                    for (int pj = (nt - 1) * PPT; pj < np; ++pj) {     // the windows of the steps are issued by key waves 0 - 2, which have the slack)
                        dma_sv((const char*)basen + (long)pj * RPP * rs * 2, voff_rs(pj), G::off_q(Tn), pj);
                        dma_sv((const char*)dobn + (long)pj * RPP * os * 2, voff_os(pj), G::off_do(Tn), pj);
                    }
                }
                // -lse / scale rows of the next pair: requested two steps from the end, written one step from the end (the last reader of this
                // pair's rows - the row constants of the last query tile - is in front of barrier nt - 2)
                if (more && u + 3 == nt) lse_fetch(bidn);
                if (more && u + 2 == nt) lse_write();
                const char* sc = lds + G::off_scr(Tn, u & 1);
                f32x16 dq, dq1;                                        // two accumulators (k-step 0 / 1): a single chain of 42 dependent MFMAs issues at ~40 cycles each
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[r] = dq1[r] = 0.f;
                frag_t bfr[2][2];                                      // dS^T fragments [k-step][part] of the key tile in hand
#pragma unroll
                for (int j = 0; j < nt; ++j) {
                    rd_tr(bfr[0], sc + j * SCR, 0, 0, PLO);
                    rd_tr(bfr[1], sc + j * SCR, 0, 1, PLO);
#pragma unroll
                    for (int i = 0; i < NPT; ++i) {
                        mt(i, ktr[j][0], bfr[0][0], bfr[0][PLO], dq);
                        mt(i, ktr[j][1], bfr[1][0], bfr[1][PLO], dq1);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[r] += dq1[r];
                const int q = u * 32 + kr;
                if (q < Tn) store_tile_T16<OT>(dbase + (long)q * rs, dq, scale * sEinv, lane);
            }
        }
        return;
    }
    // Key waves 0 - 2 also issue the next pair's pieces (window w: 16 entries, 6 / 5 / 5 per wave): they are the first waves of SIMDs 0 - 2 and
    // reach the step barrier ~1,000 cycles ahead of their SIMD partners (waves 4 - 6) - an LDS-DMA stalls only the wave that issues it.
    auto key_role = [&](auto dma_tag) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma_tag)::value;
        auto window = [&](const E* ksrc, const OE* osrc, const E* qsrc, const OE* dosrc, int w) __attribute__((always_inline)) {
            if constexpr (DMA) {
#pragma unroll
                for (int e3 = 0; e3 < 6; ++e3) {
                    const int i = wave + 3 * e3;                       // (wave 0: 0 3 6 9 12 15; wave 1: 1 4 .. 13 and 13 again; wave 2: 2 5 .. 14, 14)
                    dma_entry(ksrc, osrc, qsrc, dosrc, w, i < 16 ? i : i - 3);
                }
            }
        };
        // head of a pair: barrier X, this wave's K / V fragments, the row constants D of query tile `wave` (O rows requested a tail ago), barrier Z
        frag_t kfB[2][2], vfB[2][2];
        auto head = [&](int hbid) __attribute__((always_inline)) {
            if constexpr (A::X) publish_max();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ++rnd_no;
            ATTN_STAMP(10);
            __builtin_amdgcn_s_barrier();                                  // X: this pair's images are in place, the dS^T tiles are free
            ATTN_SB();
            ATTN_STAMP(0);
            // ----------------------------------------------------------------------------------------------- key tile `wave`
            rd_row(kfB, Ks, wave);                                         // (vfB: requested from global memory a step / a prologue ago)
            if (wave * 32 + kr >= Tn) {                                    // padded keys: zero K / V (the fragment's column is this lane's key)
    #pragma unroll
                for (int part = 0; part <= LO; ++part)
    #pragma unroll
                    for (int s = 0; s < 2; ++s) { zero_frag(kfB[part][s]); zero_frag(vfB[part][s]); }
            }
            if constexpr (A::X) {
                int lane_h = lane;
                asm volatile("" : "+v"(lane_h));
                pair_scale(hbid);
                if (wave + 1 < nt) consts_x(lane_h);
            } else {
                if (wave + 1 < nt) consts();                               // (the last query tile's: behind the first step barrier, see the helper)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                  // Z: the row constants are in place
            ATTN_SB();
            ATTN_STAMP(1);
        };
        {
            const int bid0 = pair_bid(0);
            v_fetch(vfB, qkv + (long)(bid0 / H) * Tn * rs + (bid0 % H) * HD * EP + 2 * hs);
            if constexpr (A::X) do_prefetch(dout + (long)(bid0 / H) * Tn * os + (bid0 % H) * HD * EP, lane);
        }
        head(pair_bid(0));
        for (int kp = 0; kp < npl; ++kp) {
            const int bid = pair_bid(kp);
            const int b = bid / H, h = bid % H;
            OE* dbase = dqkv + (long)b * Tn * rs + h * HD * EP;
            const bool more = kp + 1 < npl;
            const int bidn = pair_bid(more ? kp + 1 : kp);
            const E* basen = qkv + (long)(bidn / H) * Tn * rs + (bidn % H) * HD * EP;
            const OE* dobn = dout + (long)(bidn / H) * Tn * os + (bidn % H) * HD * EP;
            const OE* obn = out + (long)(bidn / H) * Tn * os + (bidn % H) * HD * EP;
            int lane_l = lane;                                             // (every lane-derived value of the steps comes from this laundered copy)
            asm volatile("" : "+v"(lane_l));
            calc_offsets(lane_l);
            calc_dma(lane_l);
            const int hh_c = lane_l >> 5;
            if (wave >= 4) __builtin_amdgcn_s_sleep(1);                    // (two waves of a SIMD in lockstep behind a barrier cost up to 2 x)
            f32x16 dk, dv;
    #pragma unroll
            for (int r = 0; r < 16; ++r) dk[r] = dv[r] = 0.f;
            frag_t qfr[2][2], dofr[2][2], dotr[2], qtr[2];
            f32x16 smA, dpA, smB, dpB;
            auto ld_consts = [&](f32x16& sm, f32x16& dp, int qt) __attribute__((always_inline)) {
    #pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int q0 = qt * 32 + 8 * g + 4 * hh_c;
                    const float4 L4 = *(const float4*)(Ls + q0);
                    const float4 D4 = *(const float4*)(Ds + q0);
                    sm[4 * g] = L4.x, sm[4 * g + 1] = L4.y, sm[4 * g + 2] = L4.z, sm[4 * g + 3] = L4.w;
                    dp[4 * g] = D4.x, dp[4 * g + 1] = D4.y, dp[4 * g + 2] = D4.z, dp[4 * g + 3] = D4.w;
                }
            };
            rd_row(qfr, Qs, 0);
            rd_row(dofr, dOs, 0);
            ld_consts(smA, dpA, 0);
    #pragma unroll
            for (int i = 0; i < NM; ++i) { mm(i, qfr, kfB[0], kfB[LO], smA, smA); mm(i, dofr, vfB[0], vfB[LO], dpA, dpA); }
            ATTN_SB();
            rd_row(qfr, Qs, 1);
            rd_row(dofr, dOs, 1);
            ATTN_SB();
            char* const scw = lds + G::off_scr(Tn, 0) + wave * SCR;
            // last_tag: the peeled last step has no next tile - no score MFMAs, row constants or row fragments for one (their 64 registers are
            // what lets the next pair's O rows sit in registers across this step without a spill)
            auto step = [&](f32x16& sm, f32x16& dp, f32x16& nsm, f32x16& ndp, int qt, auto last_tag) __attribute__((always_inline)) {
                constexpr bool LAST = decltype(last_tag)::value;
                frag_t ph, pl, sh, sl;                                      // ONE set of packed P / dS fragments, reused by the two k-steps
                char* const sw = scw + (qt & 1) * (NCW * SCR);
                const int qn = qt + 1 < nt ? qt + 1 : nt - 1, qn2 = qt + 2 < nt ? qt + 2 : nt - 1;
                if constexpr (!LAST) ld_consts(nsm, ndp, qn);
                rd_tr(dotr, dOs, qt, 0);
                rd_tr(qtr, Qs, qt, 0);
                ATTN_SB();
                auto ev = [&](int r0, int r1) __attribute__((always_inline)) {
    #pragma unroll
                    for (int r = r0; r < r1; ++r) {
                        sm[r] = __builtin_amdgcn_exp2f(sm[r] * c);
                        dp[r] *= sm[r];
                    }
                };
                auto pkP = [&](int s) __attribute__((always_inline)) {
                    pack8<T, NP>(sm, s, ph, pl);
                    if constexpr (!A::SP) pl = ph;
                };
                auto pkS = [&](int s) __attribute__((always_inline)) {      // dS fragments of k-step s; the same bytes go to the dS^T tile
                    pack8<T, NP>(dp, s, sh, sl);
                    union { frag_t f; uint2 u[2]; } a;
                    a.f = sh;
                    *(uint2*)(sw + woff[0][2 * s]) = a.u[0];
                    *(uint2*)(sw + woff[0][2 * s + 1]) = a.u[1];
                    if constexpr (A::SP && NP == 2) {
                        a.f = sl;
                        *(uint2*)(sw + woff[1][2 * s]) = a.u[0];
                        *(uint2*)(sw + woff[1][2 * s + 1]) = a.u[1];
                    } else {
                        sl = sh;
                    }
                };
                if constexpr (A::SP) {
                    ev(0, 4); ATTN_SB();
                    if constexpr (!LAST) mm(0, qfr, kfB[0], kfB[1], nsm, nsm); ev(4, 8); ATTN_SB();
                    if constexpr (!LAST) mm(0, dofr, vfB[0], vfB[1], ndp, ndp); pkP(0); ATTN_SB();
                    if constexpr (!LAST) mm(1, qfr, kfB[0], kfB[1], nsm, nsm); pkS(0); ATTN_SB();
                    if constexpr (!LAST) mm(1, dofr, vfB[0], vfB[1], ndp, ndp); ev(8, 11); ATTN_SB();
                    if constexpr (!LAST) mm(2, qfr, kfB[0], kfB[1], nsm, nsm); ev(11, 14); ATTN_SB();
                    if constexpr (!LAST) mm(2, dofr, vfB[0], vfB[1], ndp, ndp); ev(14, 16); ATTN_SB();
                    mt(0, dotr, ph, pl, dv); mt(0, qtr, sh, sl, dk); ATTN_SB();
                    mt(1, dotr, ph, pl, dv); mt(1, qtr, sh, sl, dk); ATTN_SB();
                    if constexpr (NPT == 3) { mt(2, dotr, ph, pl, dv); mt(2, qtr, sh, sl, dk); ATTN_SB(); }
                    rd_tr(dotr, dOs, qt, 1);
                    rd_tr(qtr, Qs, qt, 1);
                    ATTN_SB();
                    if constexpr (!LAST) mm(3, qfr, kfB[0], kfB[1], nsm, nsm); if constexpr (!LAST) mm(3, dofr, vfB[0], vfB[1], ndp, ndp); pkP(1); ATTN_SB();
                    if constexpr (!LAST) mm(4, qfr, kfB[0], kfB[1], nsm, nsm); if constexpr (!LAST) mm(4, dofr, vfB[0], vfB[1], ndp, ndp); pkS(1); ATTN_SB();
                    if constexpr (!LAST) mm(5, qfr, kfB[0], kfB[1], nsm, nsm); if constexpr (!LAST) mm(5, dofr, vfB[0], vfB[1], ndp, ndp); ATTN_SB();
                    if constexpr (!LAST) {
                        rd_row(qfr, Qs, qn2);
                        rd_row(dofr, dOs, qn2);
                    }
                    ATTN_SB();
                    mt(0, dotr, ph, pl, dv); mt(0, qtr, sh, sl, dk); ATTN_SB();
                    mt(1, dotr, ph, pl, dv); mt(1, qtr, sh, sl, dk); ATTN_SB();
                    if constexpr (NPT == 3) { mt(2, dotr, ph, pl, dv); mt(2, qtr, sh, sl, dk); ATTN_SB(); }
                } else {
                    ev(0, 8); ATTN_SB();
                    if constexpr (!LAST) mm(0, qfr, kfB[0], kfB[0], nsm, nsm); pkP(0); pkS(0); ATTN_SB();
                    if constexpr (!LAST) mm(0, dofr, vfB[0], vfB[0], ndp, ndp); ev(8, 16); ATTN_SB();
                    mt(0, dotr, ph, pl, dv); mt(0, qtr, sh, sl, dk); ATTN_SB();
                    rd_tr(dotr, dOs, qt, 1);
                    rd_tr(qtr, Qs, qt, 1);
                    ATTN_SB();
                    if constexpr (!LAST) mm(1, qfr, kfB[0], kfB[0], nsm, nsm); if constexpr (!LAST) mm(1, dofr, vfB[0], vfB[0], ndp, ndp); pkP(1); pkS(1); ATTN_SB();
                    if constexpr (!LAST) {
                        rd_row(qfr, Qs, qn2);
                        rd_row(dofr, dOs, qn2);
                    }
                    ATTN_SB();
                    mt(0, dotr, ph, pl, dv); mt(0, qtr, sh, sl, dk); ATTN_SB();
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the dS^T tile is written, every read of query tile qt has returned
                ATTN_STAMP(12 + qt);
                __builtin_amdgcn_s_barrier();
                ATTN_SB();
                ATTN_STAMP(2 + qt);
                if (qt + 1 < nt) window(basen + hs, obn, basen, dobn, qt + 1);    // the window this barrier opened (the last one is the helper's)
            };
            static_assert(NT % 2 == 1 && NT >= 5, "steps 0, NT - 2 and NT - 1 are peeled off the two-step loop");
            // Nothing of the compiler's own may be in flight when the loop starts: a spill reload placed in front of the loop gets its wait - vmcnt(0),
            // which also covers every LDS-DMA - at the value's first use INSIDE the loop, i.e. in every iteration.  So: a compiler-visible wait
            // here (only scratch reloads can be pending), and window 0 of the next pair's pieces goes out behind it.
            __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0)
            if constexpr (DMA) {                                           // the last six entries of window 0 (the helper issues ten: more would make it late for the first step barrier)
                dma_entry(basen + hs, obn, basen, dobn, 0, 10 + wave);
                dma_entry(basen + hs, obn, basen, dobn, 0, 13 + wave);
            }
                step(smA, dpA, smB, dpB, 0, std::false_type{});
            if (wave + 1 == nt) {                                          // the last query tile's Q / dO rows are guaranteed from here on
                if constexpr (A::X) consts_x(lane_l);
                else consts();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (read four step barriers from here)
            }
#pragma unroll 1
            for (int qt = 1; qt + 3 < nt; qt += 2) {
                step(smB, dpB, smA, dpA, qt, std::false_type{});
                step(smA, dpA, smB, dpB, qt + 1, std::false_type{});
            }
            step(smB, dpB, smA, dpA, nt - 2, std::false_type{});
            v_fetch(vfB, basen + 2 * hs);                                  // the next pair's V fragments (this pair's last dP MFMAs are behind us)
            if constexpr (A::X) do_prefetch(dobn, lane_l);
            step(smA, dpA, smB, dpB, nt - 1, std::true_type{});
            const int k = wave * 32 + kr;
            if (k < Tn) {
                store_tile_T16<OT>(dbase + (long)k * rs + hs, dk, scale * sEinv, lane);
                store_tile_T16<OT>(dbase + (long)k * rs + 2 * hs, dv, sEinv, lane);
            }
            ATTN_STAMP(11);
            wait_vm<0>();                                                  // this wave's pieces have landed (the stores above have left)
            if (more) head(bidn);                                      // (the pair loop is rotated: the O rows requested above die inside this iteration)
        }
    };
    if (wave < 3) key_role(std::true_type{});
    else key_role(std::false_type{});
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

int attn_cus() {   // CUs of the CURRENT device, a multiple of 8 (a persistent workgroup's pairs stay on its XCD)
    const int n = device_cus() & ~7;
    return n >= 8 ? n : 8;
}

// parts of P / dS in the split-fp16 kernels (see the top of the file): forward 2 (hi + lo: f32-grade outputs - one part measured 2.2e-4 on the output
// and 8e-5 ... 9e-5 on the logits of a 12-block encoder for 46 instead of 48 us, and was removed), backward 1 (11 bits, like the saved activation
// derivative of the MLP: gradients at 2^-11 relative; worst parameter gradient of the encoder unchanged at 3.3e-4); MFVIT_ATTN_PB=2: two parts
// in the backward too (dqkv at 5e-6 against float64, +14 % time)
static int parts_bwd() { static int sw = INT_MIN; return env_switch("MFVIT_ATTN_PB", 1, sw) == 2 ? 2 : 1; }

template <typename T, int NPX> int launch_fwd_t(const void* qkv, void* out, float* lse, int B, int Tn, int H, hipStream_t st) {
    typedef typename Vec4<T>::elem E;
    typedef typename AttnT<T>::OE OE;
    if constexpr (is_split<T>::value) {
        // persistent pair-synchronous kernel: enough pairs to fill every CU twice, 5 - 7 row tiles per pair (one per computing wave).  The split types
        // only (53 vs 57 - 60 us at the bench shape in split bf16); the plain 16-bit types are faster on the per-pair kernel (27.5 vs 30 us: their steps
        // are too short for the per-step overheads of the pipeline) and are no longer instantiated for it.
        const int cus = attn_cus();
        const int nt = (Tn + 31) >> 5;
        const int bytes = RingGeo<T>::lds_bytes(Tn);
        if (B * H >= 2 * cus && nt >= 5 && nt <= RingGeo<T>::NCW && bytes <= 160 * 1024) {
            static PerDeviceOnce attr_pp;
            if (attr_pp.first()) (void)hipFuncSetAttribute((const void*)attn_fwd_pp_kernel<T, NPX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            ProfScope ps(PROF_ATTN_FWD, 4.0 * B * H * (double)Tn * Tn * HD, 0, st);
            MFVIT_LAUNCH((attn_fwd_pp_kernel<T, NPX>), dim3(cus), dim3(512), bytes, st, (const E*)qkv, (OE*)out, lse, Tn, H, 1.0f / sqrtf((float)HD), B * H);
            MFVIT_CHECK_LAUNCH();
            return MFVIT_OK;
        }
    }
    const int Tpad = (Tn + 31) & ~31;
    const int bytes = Tpad * (AttnT<T>::RSB + AttnT<T>::RB);
    static PerDeviceOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<T, NPX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    ProfScope ps(PROF_ATTN_FWD, 4.0 * B * H * (double)Tn * Tn * HD, 0, st);
    MFVIT_LAUNCH((attn_fwd_mfma_kernel<T, NPX>), dim3(B * H), dim3(256), bytes, st, (const E*)qkv, (OE*)out, lse, Tn, H, 1.0f / sqrtf((float)HD));
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
template <typename T, int NPX> int launch_bwd_t(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn,
                                                int H, hipStream_t st, const unsigned* domax = nullptr) {
    typedef typename Vec4<T>::elem E;
    typedef typename AttnT<T>::OE OE;
    typedef typename AttnT<T>::OT OT;
    constexpr int RSB = AttnT<T>::RSB, RB = AttnT<T>::RB;
    const int Tpad = (Tn + 31) & ~31;
    const bool wide = 4 * Tpad * RSB + 2 * Tpad * 4 + Tpad * RB + 64 <= 160 * 1024;
    const int bytes = 4 * Tpad * (wide ? RSB : RB) + 2 * Tpad * 4 + Tpad * RB;      // Q, K, V, dO images, lse / D rows, the next pair's O rows
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<T, RSB, NPX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<T, RB, NPX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    }
    auto colsum = [&]() -> int {
        if (!dbias) return MFVIT_OK;
        const int M = B * Tn, N = 3 * H * HD;
        MFVIT_LAUNCH((colsum_t_kernel<OT>), dim3((M + 63) / 64), dim3(256), 0, st, (const OE*)dqkv, (long)N * AttnT<T>::EP, dbias, M, N);
        MFVIT_CHECK_LAUNCH();
        return MFVIT_OK;
    };
    {   // single-pass kernel: seven row tiles (T = 193 .. 224), enough pairs to give every CU two.  Default for all 16-bit types
        // (bench shape: split bf16 115 vs 155 us, bf16 / fp16 62.5 vs 71 us: profiles/r04_kernel_experiments.txt); MFVIT_ATTN_BWD_SP=0: off
        static int sws = INT_MIN;
        const int cus3 = attn_cus();
        const int nt = (Tn + 31) >> 5;
        const int b3 = SpGeo<T>::lds_bytes(Tn);
        if (env_switch("MFVIT_ATTN_BWD_SP", 1, sws) != 0 && B * H >= 2 * cus3 && nt == 7 && b3 <= 160 * 1024) {
            static PerDeviceOnce attr_sp;
            if (attr_sp.first()) {
                (void)hipFuncSetAttribute((const void*)attn_bwd_sp_kernel<T, 7, NPX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if constexpr (AttnT<T>::X)
                    (void)hipFuncSetAttribute((const void*)attn_bwd_sp_kernel<T, 7, NPX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            {
                ProfScope ps(PROF_ATTN_BWD, 8.0 * B * H * (double)Tn * Tn * HD, 0, st);
                bool done = false;
                if constexpr (AttnT<T>::X) {
                    if (domax) {
                        MFVIT_LAUNCH((attn_bwd_sp_kernel<T, 7, NPX, true>), dim3(cus3), dim3(512), b3, st, (const E*)qkv, (const OE*)out, (const OE*)dout, lse,
                                     (OE*)dqkv, Tn, H, 1.0f / sqrtf((float)HD), B * H, domax);
                        done = true;
                    }
                }
                if (!done)
                    MFVIT_LAUNCH((attn_bwd_sp_kernel<T, 7, NPX>), dim3(cus3), dim3(512), b3, st, (const E*)qkv, (const OE*)out, (const OE*)dout, lse, (OE*)dqkv,
                                 Tn, H, 1.0f / sqrtf((float)HD), B * H, (const unsigned*)nullptr);
                MFVIT_CHECK_LAUNCH();
            }
            return colsum();
        }
    }
    {
        ProfScope ps(PROF_ATTN_BWD, 8.0 * B * H * (double)Tn * Tn * HD, 0, st);
        // one persistent workgroup per CU (the LDS images allow no second one): a multiple of 8 so that a workgroup's pairs stay on its XCD
        const int cus = attn_cus();
        const int grid = B * H < cus ? B * H : cus;
        if (wide)
            MFVIT_LAUNCH((attn_bwd_mfma_kernel<T, RSB, NPX>), dim3(grid), dim3(512), bytes, st, (const E*)qkv, (const OE*)out, (const OE*)dout, lse,
                         (OE*)dqkv, Tn, H, 1.0f / sqrtf((float)HD), B * H);
        else
            MFVIT_LAUNCH((attn_bwd_mfma_kernel<T, RB, NPX>), dim3(grid), dim3(512), bytes, st, (const E*)qkv, (const OE*)out, (const OE*)dout, lse,
                         (OE*)dqkv, Tn, H, 1.0f / sqrtf((float)HD), B * H);
        MFVIT_CHECK_LAUNCH();
    }
    return colsum();
}

}  // namespace

bool attn_mfma_supported(int dtype, int Tn, int HDim, bool backward) {
    if ((dtype != MFVIT_BF16 && dtype != MFVIT_BF16X3 && dtype != MFVIT_F16 && dtype != MFVIT_X3F16) || HDim != HD || Tn < 1) return false;
    const int Tpad = (Tn + 31) & ~31;
    const int rb = (dtype == MFVIT_BF16X3 || dtype == MFVIT_X3F16) ? 128 : 64;
    const int bytes = backward ? 4 * Tpad * rb + 2 * Tpad * 4 + Tpad * rb + 64 : Tpad * (rb + 16) + Tpad * rb;
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
    if (dtype == MFVIT_BF16) return launch_fwd_t<bf16, 2>(qkv, out, lse, B, Tn, H, st);
    if (dtype == MFVIT_BF16X3) return launch_fwd_t<sbf16, 2>(qkv, out, lse, B, Tn, H, st);
    if (dtype == MFVIT_F16) return launch_fwd_t<f16, 2>(qkv, out, lse, B, Tn, H, st);
    if (dtype == MFVIT_X3F16) return launch_fwd_t<sf16, 2>(qkv, out, lse, B, Tn, H, st);
    return MFVIT_EINVAL;
}
int attn_bwd_mfma(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn, int H,
                  hipStream_t st, const unsigned* domax) {
    if (dtype == MFVIT_BF16) return launch_bwd_t<bf16, 2>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st);
    if (dtype == MFVIT_BF16X3) return launch_bwd_t<sbf16, 2>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st);
    if (dtype == MFVIT_F16) return launch_bwd_t<f16, 2>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st);
    if (dtype == MFVIT_X3F16)
        return parts_bwd() == 1 ? launch_bwd_t<sf16, 1>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st, domax)
                                : launch_bwd_t<sf16, 2>(qkv, out, dout, lse, dqkv, dbias, B, Tn, H, st, domax);
    return MFVIT_EINVAL;
}

}  // namespace mfvit

#ifdef MFVIT_ATTN_STAMP
extern "C" int mfvit_debug_attn_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(mfvit::g_attn_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
