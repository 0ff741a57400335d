// Tiled ("streaming") MFMA multi-head self-attention for gfx950: any sequence length, head_dim 32 / 64 / 96, element types
// bf16, f16 and sbf16 (split bf16: three MFMAs per product, P / dS split in registers).
//
// attention_mfma.hip keeps the whole K/V (forward) or Q/K/V/dO (backward) of one (image, head) in LDS: fastest for ViT-S at
// T = 197, but bounded by the 160 KB of LDS (split bf16: forward T <= 576, backward T <= 288; head_dim 32 only).  The kernels here
// stream the other operand through LDS in 64-row chunks instead, flash-attention style, with the running row maximum / sum (forward)
// or the saved log-sum-exp (backward) on the lane.  They serve
//   * the TransFuser-GPT fusion (fuseattention.py:21-58: 4 heads x 96 over 394 joint tokens),
//   * split-bf16 attention at 384 x 384 inputs (577 tokens, BASELINE configs[4] shape in the f32-grade mode),
//   * head_dim 64.
// Work split: one workgroup (4 waves) per (image, head, block of 128 queries | keys); a wave owns one 32-row tile.
//   forward : O^T[d][q] accumulators (head_dim / 32 tiles), S^T[key][q] scores of a 64-key chunk, online softmax per chunk
//   dQ      : per wave 32 queries; chunks of K, V stream through LDS          dQ^T[d][q] += K^T x dS^T
//   dK, dV  : per wave 32 keys; chunks of Q, dO (+ lse, D) stream through LDS  dV^T[d][key] += dO^T x P ; dK^T[d][key] += Q^T x dS
// Same operand orientation, accumulator-order transposed reads (ds_read_b64_tr_b16) and -lse/scale / -D accumulator initial values
// as attention_mfma.hip.  No atomics, deterministic.
#include "common.cuh"
#include "prof.h"

namespace mfvit {

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

__device__ __forceinline__ f32x16 tl_mma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 tl_mma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int tl_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// NB = head_dim / 32.  A head's row piece holds NB groups of 32 logical columns; split tensors store a group as [hi x 32 | lo x 32].
template <typename T, int NB> struct TG {
    static constexpr bool SP = is_split<T>::value;
    static constexpr int EP = SP ? 2 : 1;
    static constexpr int HD = 32 * NB;
    static constexpr int RB = HD * 2 * EP;      // bytes of one head row piece
    static constexpr int PITCH = RB + 16;       // LDS row pitch: an odd number of 16-B slots -> conflict-free b128 row reads
    static constexpr int CPR = RB / 16;         // 16-B chunks per row piece
    static constexpr int KSTEPS = HD / 16;      // 16-wide k steps over the head dimension
    typedef typename Vec8<T>::type frag_t;
    typedef typename Vec4<T>::elem E;
    // byte offset inside a row piece of logical column d (part 0 = hi / plain, 1 = lo)
    static __device__ __forceinline__ int boff(int d, int part) { return SP ? ((d >> 5) * 128 + (d & 31) * 2 + 64 * part) : d * 2; }
};

// natural-order fragment of an LDS image: row (rowbase + lane & 31), columns 16 s + 8 (lane >> 5) .. + 8
template <typename T, int NB> __device__ __forceinline__ typename Vec8<T>::type tl_row_frag(const char* img, int rowbase, int s, int lane, int part) {
    typedef TG<T, NB> G;
    return *(const typename Vec8<T>::type*)(img + (rowbase + (lane & 31)) * G::PITCH + G::boff(16 * s + 8 * (lane >> 5), part));
}
// the same fragment straight from a global row pointer (elements of T's storage type)
template <typename T, int NB> __device__ __forceinline__ typename Vec8<T>::type tl_glb_frag(const typename Vec4<T>::elem* row, int s, int lane, int part) {
    typedef TG<T, NB> G;
    return *(const typename Vec8<T>::type*)((const char*)row + G::boff(16 * s + 8 * (lane >> 5), part));
}
// transposed fragment in ACCUMULATOR k order: lane holds column d = 32 n + (lane & 31); element j = row (rowbase + 16 s + 8 (j>>2) + 4 h + (j&3))
template <typename T, int NB> __device__ __forceinline__ typename Vec8<T>::type tl_tr_frag(const char* img, int rowbase, int s, int n, int lane, int part) {
    typedef TG<T, NB> G;
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const char* a = img + (rowbase + 16 * s + 4 * h + q) * G::PITCH + G::boff(32 * n + 16 * g1 + 4 * p, part);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 8 * G::PITCH));
    union { struct { s16x4 a, b; } s; typename Vec8<T>::type v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}
// registers 8 s .. 8 s + 7 of an accumulator as a B fragment (k step s); split: hi and lo parts
template <typename T> __device__ __forceinline__ void tl_pack8(const f32x16& v, int s, typename Vec8<T>::type& hi, typename Vec8<T>::type& lo) {
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
        } else {
            lo[j] = h0;
            lo[j + 1] = h1;
        }
    }
}
// acc += A (x) B: one MFMA, or three for split tensors (a_lo b_hi + a_hi b_lo + a_hi b_hi)
template <typename T> __device__ __forceinline__ f32x16 tl_mma3(typename Vec8<T>::type ah, typename Vec8<T>::type al, typename Vec8<T>::type bh,
                                                                typename Vec8<T>::type bl, f32x16 c) {
    if constexpr (is_split<T>::value) {
        c = tl_mma(al, bh, c);
        c = tl_mma(ah, bl, c);
    }
    return tl_mma(ah, bh, c);
}
// accumulator tile X^T[d][col] (col on the lane) -> 32 logical columns of a row of T (row_ptr at the group's first storage element)
template <typename T> __device__ __forceinline__ void tl_store_tile_T(typename Vec4<T>::elem* grp, const f32x16& acc, float mul, int lane) {
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
        *(V4*)(grp + 8 * g + 4 * (lane >> 5)) = o;
        if constexpr (is_split<T>::value) *(V4*)(grp + 32 + 8 * g + 4 * (lane >> 5)) = l;
    }
}
// cooperative copy of `rows` head row pieces (global row stride rs elements, rows >= limit zero-filled) into an LDS image
template <typename T, int NB> __device__ __forceinline__ void tl_stage(const typename Vec4<T>::elem* base, long rs, int row0, int limit, int rows,
                                                                       char* img) {
    typedef TG<T, NB> G;
    for (int q = threadIdx.x; q < rows * G::CPR; q += blockDim.x) {
        const int t = q / G::CPR, c = q % G::CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + t < limit) v = *(const uint4*)(base + (long)(row0 + t) * rs + 8 * c);
        *(uint4*)(img + t * G::PITCH + 16 * c) = v;
    }
}

constexpr int TL_CH = 64;      // rows of the streamed operand per LDS chunk (two 32-row tiles)
constexpr int TL_BLK = 128;    // rows of the resident operand per workgroup (4 waves x 32)

// decode blockIdx -> (b, h, row block); neighbours in the remapped id share an XCD (heads / blocks of one image share L2 lines)
__device__ __forceinline__ void tl_ids(int H, int nblk, int& b, int& h, int& blk) {
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    blk = bid % nblk;
    const int bh = bid / nblk;
    b = bh / H;
    h = bh % H;
}

// DROP: attention dropout (fuseattention.py:52 `att = self.attn_drop(att)`): the mask multiplies the PV operand only - the row sum that
// normalises the output stays the sum of the unmasked probabilities
template <typename T, int NB, bool DROP = false>
__global__ __launch_bounds__(256) void attn_tiled_fwd_kernel(const typename Vec4<T>::elem* __restrict__ qkv, typename Vec4<T>::elem* __restrict__ out,
                                                             float* __restrict__ lse, int Tn, int H, float scale, DropP drop) {
    typedef TG<T, NB> G;
    typedef typename G::E E;
    typedef typename G::frag_t frag_t;
    constexpr int LO = G::SP ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Ks = lds;
    char* Vs = lds + TL_CH * G::PITCH;
    const int nblk = (Tn + TL_BLK - 1) / TL_BLK;
    int b, h, blk;
    tl_ids(H, nblk, b, h, blk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long hs = (long)H * G::HD * G::EP, rs = 3 * hs;         // storage elements per token of one of q / k / v, and of all three
    const E* base = qkv + (long)b * Tn * rs + (long)h * G::HD * G::EP;
    const int q = blk * TL_BLK + wave * 32 + (lane & 31);
    const int qc = q < Tn ? q : Tn - 1;
    frag_t qf[G::KSTEPS], ql[G::KSTEPS];
#pragma unroll
    for (int s = 0; s < G::KSTEPS; ++s) {
        qf[s] = tl_glb_frag<T, NB>(base + (long)qc * rs, s, lane, 0);
        ql[s] = tl_glb_frag<T, NB>(base + (long)qc * rs, s, lane, LO);
    }
    const float c = scale * 1.4426950408889634f;
    float m2 = -INFINITY, lsum = 0.f;
    f32x16 o[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
    for (int k0 = 0; k0 < Tn; k0 += TL_CH) {
        __syncthreads();                                          // the previous chunk's readers are done
        tl_stage<T, NB>(base + hs, rs, k0, Tn, TL_CH, Ks);
        tl_stage<T, NB>(base + 2 * hs, rs, k0, Tn, TL_CH, Vs);
        __syncthreads();
        f32x16 sc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
#pragma unroll
            for (int s = 0; s < G::KSTEPS; ++s)
                sc[t] = tl_mma3<T>(tl_row_frag<T, NB>(Ks, t * 32, s, lane, 0), tl_row_frag<T, NB>(Ks, t * 32, s, lane, LO), qf[s], ql[s], sc[t]);
            if (k0 + (t + 1) * 32 > Tn) {                          // mask the zero-filled keys past the end
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + t * 32 + tl_acc_row(r, lane) >= Tn) sc[t][r] = -INFINITY;
            }
        }
        float cm = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) cm = fmaxf(cm, sc[t][r]);
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64)) * c;               // every chunk holds at least one real key: finite
        const float mn = fmaxf(m2, cm);
        const float alpha = exp2f(m2 - mn);
        m2 = mn;
        lsum *= alpha;
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sc[t][r], c, -m2));
                lsum += p;
                if constexpr (DROP)
                    sc[t][r] = p * drop_mul(drop, (unsigned)((((long)b * H + h) * Tn + qc) * Tn + k0 + t * 32 + tl_acc_row(r, lane)));
                else
                    sc[t][r] = p;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                frag_t ph, pl;
                tl_pack8<T>(sc[t], s, ph, pl);
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    o[n] = tl_mma3<T>(tl_tr_frag<T, NB>(Vs, t * 32, s, n, lane, 0), tl_tr_frag<T, NB>(Vs, t * 32, s, n, lane, LO), ph, pl, o[n]);
            }
        }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (q < Tn) {
        E* orow = out + (((long)b * Tn + q) * H + h) * G::HD * G::EP;
#pragma unroll
        for (int n = 0; n < NB; ++n) tl_store_tile_T<T>((E*)((char*)orow + G::boff(32 * n, 0)), o[n], 1.0f / lsum, lane);
        if (lane < 32) lse[((long)b * H + h) * Tn + q] = (m2 + log2f(lsum)) * 0.6931471805599453f;
    }
}

// dQ: wave = 32 queries; K and V stream through LDS.
template <typename T, int NB, bool DROP = false>
__global__ __launch_bounds__(256) void attn_tiled_bwd_dq_kernel(const typename Vec4<T>::elem* __restrict__ qkv, const typename Vec4<T>::elem* __restrict__ out,
                                                                const typename Vec4<T>::elem* __restrict__ dout, const float* __restrict__ lse,
                                                                typename Vec4<T>::elem* __restrict__ dqkv, int Tn, int H, float scale, DropP drop) {
    typedef TG<T, NB> G;
    typedef typename G::E E;
    typedef typename G::frag_t frag_t;
    constexpr int LO = G::SP ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Ks = lds;
    char* Vs = lds + TL_CH * G::PITCH;
    const int nblk = (Tn + TL_BLK - 1) / TL_BLK;
    int b, h, blk;
    tl_ids(H, nblk, b, h, blk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long hs = (long)H * G::HD * G::EP, rs = 3 * hs, os = hs;
    const E* base = qkv + (long)b * Tn * rs + (long)h * G::HD * G::EP;
    E* dbase = dqkv + (long)b * Tn * rs + (long)h * G::HD * G::EP;
    const E* obase = out + (long)b * Tn * os + (long)h * G::HD * G::EP;
    const E* dobase = dout + (long)b * Tn * os + (long)h * G::HD * G::EP;
    const int q = blk * TL_BLK + wave * 32 + (lane & 31);
    const int qc = q < Tn ? q : Tn - 1;
    frag_t qf[G::KSTEPS], ql[G::KSTEPS], dof[G::KSTEPS], dol[G::KSTEPS];
    float D = 0.f;                                                // D_q = sum_d dO[q][d] O[q][d]: this lane's half of every k step
#pragma unroll
    for (int s = 0; s < G::KSTEPS; ++s) {
        qf[s] = tl_glb_frag<T, NB>(base + (long)qc * rs, s, lane, 0);
        ql[s] = tl_glb_frag<T, NB>(base + (long)qc * rs, s, lane, LO);
        dof[s] = tl_glb_frag<T, NB>(dobase + (long)qc * os, s, lane, 0);
        dol[s] = tl_glb_frag<T, NB>(dobase + (long)qc * os, s, lane, LO);
        const frag_t oh = tl_glb_frag<T, NB>(obase + (long)qc * os, s, lane, 0);
        const frag_t ol = tl_glb_frag<T, NB>(obase + (long)qc * os, s, lane, LO);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (G::SP) D = fmaf((float)dof[s][j] + (float)dol[s][j], (float)oh[j] + (float)ol[j], D);
            else D = fmaf((float)dof[s][j], (float)oh[j], D);
        }
    }
    D += __shfl_xor(D, 32, 64);
    const float L = q < Tn ? -lse[((long)b * H + h) * Tn + q] / scale : -1e30f, Dq = -D;
    const float c = scale * 1.4426950408889634f;
    f32x16 dq[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[n][r] = 0.f;
    for (int k0 = 0; k0 < Tn; k0 += TL_CH) {
        __syncthreads();
        tl_stage<T, NB>(base + hs, rs, k0, Tn, TL_CH, Ks);
        tl_stage<T, NB>(base + 2 * hs, rs, k0, Tn, TL_CH, Vs);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = L; dp[r] = Dq; }
#pragma unroll
            for (int s = 0; s < G::KSTEPS; ++s) {
                st = tl_mma3<T>(tl_row_frag<T, NB>(Ks, t * 32, s, lane, 0), tl_row_frag<T, NB>(Ks, t * 32, s, lane, LO), qf[s], ql[s], st);
                dp = tl_mma3<T>(tl_row_frag<T, NB>(Vs, t * 32, s, lane, 0), tl_row_frag<T, NB>(Vs, t * 32, s, lane, LO), dof[s], dol[s], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {  // dS^T / scale (keys past the end: K rows are zero, so their dQ contribution vanishes)
                if constexpr (DROP) {       // dP = (dO . v) * mask / (1 - p); dp holds (dO . v) - D
                    const float mk = drop_mul(drop, (unsigned)((((long)b * H + h) * Tn + qc) * Tn + k0 + t * 32 + tl_acc_row(r, lane)));
                    st[r] = __builtin_amdgcn_exp2f(st[r] * c) * fmaf(dp[r] - Dq, mk, Dq);
                } else {
                    st[r] = __builtin_amdgcn_exp2f(st[r] * c) * dp[r];
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                frag_t sh, sl;
                tl_pack8<T>(st, s, sh, sl);
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    dq[n] = tl_mma3<T>(tl_tr_frag<T, NB>(Ks, t * 32, s, n, lane, 0), tl_tr_frag<T, NB>(Ks, t * 32, s, n, lane, LO), sh, sl, dq[n]);
            }
        }
    }
    if (q < Tn) {
#pragma unroll
        for (int n = 0; n < NB; ++n) tl_store_tile_T<T>((E*)((char*)(dbase + (long)q * rs) + G::boff(32 * n, 0)), dq[n], scale, lane);
    }
}

// dK, dV: wave = 32 keys; Q, dO and the per-query -lse/scale, -D stream through LDS.
template <typename T, int NB, bool DROP = false>
__global__ __launch_bounds__(256) void attn_tiled_bwd_dkv_kernel(const typename Vec4<T>::elem* __restrict__ qkv, const typename Vec4<T>::elem* __restrict__ out,
                                                                 const typename Vec4<T>::elem* __restrict__ dout, const float* __restrict__ lse,
                                                                 typename Vec4<T>::elem* __restrict__ dqkv, int Tn, int H, float scale, DropP drop) {
    typedef TG<T, NB> G;
    typedef typename G::E E;
    typedef typename G::frag_t frag_t;
    constexpr int LO = G::SP ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Qs = lds;
    char* dOs = lds + TL_CH * G::PITCH;
    float* Ls = (float*)(lds + 2 * TL_CH * G::PITCH);
    float* Ds = Ls + TL_CH;
    const int nblk = (Tn + TL_BLK - 1) / TL_BLK;
    int b, h, blk;
    tl_ids(H, nblk, b, h, blk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long hs = (long)H * G::HD * G::EP, rs = 3 * hs, os = hs;
    const E* base = qkv + (long)b * Tn * rs + (long)h * G::HD * G::EP;
    E* dbase = dqkv + (long)b * Tn * rs + (long)h * G::HD * G::EP;
    const E* obase = out + (long)b * Tn * os + (long)h * G::HD * G::EP;
    const E* dobase = dout + (long)b * Tn * os + (long)h * G::HD * G::EP;
    const int k = blk * TL_BLK + wave * 32 + (lane & 31);
    const int kc = k < Tn ? k : Tn - 1;
    frag_t kf[G::KSTEPS], kl[G::KSTEPS], vf[G::KSTEPS], vl[G::KSTEPS];
#pragma unroll
    for (int s = 0; s < G::KSTEPS; ++s) {
        kf[s] = tl_glb_frag<T, NB>(base + (long)kc * rs + hs, s, lane, 0);
        kl[s] = tl_glb_frag<T, NB>(base + (long)kc * rs + hs, s, lane, LO);
        vf[s] = tl_glb_frag<T, NB>(base + (long)kc * rs + 2 * hs, s, lane, 0);
        vl[s] = tl_glb_frag<T, NB>(base + (long)kc * rs + 2 * hs, s, lane, LO);
    }
    const float c = scale * 1.4426950408889634f;
    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[n][r] = dv[n][r] = 0.f;
    for (int q0 = 0; q0 < Tn; q0 += TL_CH) {
        __syncthreads();
        tl_stage<T, NB>(base, rs, q0, Tn, TL_CH, Qs);
        tl_stage<T, NB>(dobase, os, q0, Tn, TL_CH, dOs);
        if (threadIdx.x < TL_CH) {                                // -lse/scale and -D of the chunk's queries (rows past the end: p = 0)
            const int qq = q0 + threadIdx.x;
            float Lv = -1e30f, Dv = 0.f;
            if (qq < Tn) {
                Lv = -lse[((long)b * H + h) * Tn + qq] / scale;
                const E* orow = obase + (long)qq * os;
                const E* drow = dobase + (long)qq * os;
                for (int d0 = 0; d0 < G::HD; d0 += 8) {
                    const frag_t oh = *(const frag_t*)((const char*)orow + G::boff(d0, 0)), dh = *(const frag_t*)((const char*)drow + G::boff(d0, 0));
                    if constexpr (G::SP) {
                        const frag_t ol = *(const frag_t*)((const char*)orow + G::boff(d0, 1)), dl = *(const frag_t*)((const char*)drow + G::boff(d0, 1));
#pragma unroll
                        for (int j = 0; j < 8; ++j) Dv = fmaf((float)dh[j] + (float)dl[j], (float)oh[j] + (float)ol[j], Dv);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) Dv = fmaf((float)dh[j], (float)oh[j], Dv);
                    }
                }
            }
            Ls[threadIdx.x] = Lv;
            Ds[threadIdx.x] = -Dv;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 sm, dp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qi = t * 32 + 8 * g + 4 * (lane >> 5);
                const float4 L4 = *(const float4*)(Ls + qi);
                const float4 D4 = *(const float4*)(Ds + qi);
                sm[4 * g] = L4.x, sm[4 * g + 1] = L4.y, sm[4 * g + 2] = L4.z, sm[4 * g + 3] = L4.w;
                dp[4 * g] = D4.x, dp[4 * g + 1] = D4.y, dp[4 * g + 2] = D4.z, dp[4 * g + 3] = D4.w;
            }
            f32x16 dneg;                   // DROP: the -D each dp register started from
            if constexpr (DROP) dneg = dp;
#pragma unroll
            for (int s = 0; s < G::KSTEPS; ++s) {
                sm = tl_mma3<T>(tl_row_frag<T, NB>(Qs, t * 32, s, lane, 0), tl_row_frag<T, NB>(Qs, t * 32, s, lane, LO), kf[s], kl[s], sm);    // S[q][key] - lse[q]/scale
                dp = tl_mma3<T>(tl_row_frag<T, NB>(dOs, t * 32, s, lane, 0), tl_row_frag<T, NB>(dOs, t * 32, s, lane, LO), vf[s], vl[s], dp);  // dP[q][key] - D[q]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __builtin_amdgcn_exp2f(sm[r] * c);
                if constexpr (DROP) {       // dV takes P * mask / (1 - p), dS = P * ((dO . v) * mask / (1 - p) - D)
                    const int qq = q0 + t * 32 + tl_acc_row(r, lane);
                    const float mk = drop_mul(drop, (unsigned)((((long)b * H + h) * Tn + (qq < Tn ? qq : Tn - 1)) * Tn + kc));
                    sm[r] = pr * mk;
                    dp[r] = pr * fmaf(dp[r] - dneg[r], mk, dneg[r]);
                } else {
                    sm[r] = pr;
                    dp[r] *= pr;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                frag_t ph, pl, sh, sl;
                tl_pack8<T>(sm, s, ph, pl);
                tl_pack8<T>(dp, s, sh, sl);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    dv[n] = tl_mma3<T>(tl_tr_frag<T, NB>(dOs, t * 32, s, n, lane, 0), tl_tr_frag<T, NB>(dOs, t * 32, s, n, lane, LO), ph, pl, dv[n]);
                    dk[n] = tl_mma3<T>(tl_tr_frag<T, NB>(Qs, t * 32, s, n, lane, 0), tl_tr_frag<T, NB>(Qs, t * 32, s, n, lane, LO), sh, sl, dk[n]);
                }
            }
        }
    }
    if (k < Tn) {
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            tl_store_tile_T<T>((E*)((char*)(dbase + (long)k * rs + hs) + G::boff(32 * n, 0)), dk[n], scale, lane);
            tl_store_tile_T<T>((E*)((char*)(dbase + (long)k * rs + 2 * hs) + G::boff(32 * n, 0)), dv[n], 1.0f, lane);
        }
    }
}

template <typename T, int NB> int tl_launch_fwd(const void* qkv, void* out, float* lse, int B, int Tn, int H, hipStream_t st, DropP drop) {
    typedef TG<T, NB> G;
    typedef typename G::E E;
    const int bytes = 2 * TL_CH * G::PITCH;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)attn_tiled_fwd_kernel<T, NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        (void)hipFuncSetAttribute((const void*)attn_tiled_fwd_kernel<T, NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
    const int nblk = (Tn + TL_BLK - 1) / TL_BLK;
    ProfScope ps(PROF_ATTN_FWD, 4.0 * B * H * (double)Tn * Tn * G::HD, 0, st);
    if (drop.thr)
        MFVIT_LAUNCH((attn_tiled_fwd_kernel<T, NB, true>), dim3(B * H * nblk), dim3(256), bytes, st, (const E*)qkv, (E*)out, lse, Tn, H,
                     1.0f / sqrtf((float)G::HD), drop);
    else
        MFVIT_LAUNCH((attn_tiled_fwd_kernel<T, NB, false>), dim3(B * H * nblk), dim3(256), bytes, st, (const E*)qkv, (E*)out, lse, Tn, H,
                     1.0f / sqrtf((float)G::HD), drop);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
template <typename T, int NB> int tl_launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int Tn, int H,
                                                hipStream_t st, DropP drop) {
    typedef TG<T, NB> G;
    typedef typename G::E E;
    const int bytes_q = 2 * TL_CH * G::PITCH, bytes_kv = 2 * TL_CH * G::PITCH + 2 * TL_CH * 4;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)attn_tiled_bwd_dq_kernel<T, NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes_q);
        (void)hipFuncSetAttribute((const void*)attn_tiled_bwd_dkv_kernel<T, NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes_kv);
        (void)hipFuncSetAttribute((const void*)attn_tiled_bwd_dq_kernel<T, NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes_q);
        (void)hipFuncSetAttribute((const void*)attn_tiled_bwd_dkv_kernel<T, NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes_kv);
    }
    const int nblk = (Tn + TL_BLK - 1) / TL_BLK;
    const float scale = 1.0f / sqrtf((float)G::HD);
    ProfScope ps(PROF_ATTN_BWD, 8.0 * B * H * (double)Tn * Tn * G::HD, 0, st);
    if (drop.thr) {
        MFVIT_LAUNCH((attn_tiled_bwd_dq_kernel<T, NB, true>), dim3(B * H * nblk), dim3(256), bytes_q, st, (const E*)qkv, (const E*)out, (const E*)dout,
                     lse, (E*)dqkv, Tn, H, scale, drop);
        MFVIT_CHECK_LAUNCH();
        MFVIT_LAUNCH((attn_tiled_bwd_dkv_kernel<T, NB, true>), dim3(B * H * nblk), dim3(256), bytes_kv, st, (const E*)qkv, (const E*)out,
                     (const E*)dout, lse, (E*)dqkv, Tn, H, scale, drop);
    } else {
        MFVIT_LAUNCH((attn_tiled_bwd_dq_kernel<T, NB, false>), dim3(B * H * nblk), dim3(256), bytes_q, st, (const E*)qkv, (const E*)out, (const E*)dout,
                     lse, (E*)dqkv, Tn, H, scale, drop);
        MFVIT_CHECK_LAUNCH();
        MFVIT_LAUNCH((attn_tiled_bwd_dkv_kernel<T, NB, false>), dim3(B * H * nblk), dim3(256), bytes_kv, st, (const E*)qkv, (const E*)out,
                     (const E*)dout, lse, (E*)dqkv, Tn, H, scale, drop);
    }
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

}  // namespace

bool attn_tiled_supported(int dtype, int Tn, int HDim) {
    return (dtype == MFVIT_BF16 || dtype == MFVIT_BF16X3 || dtype == MFVIT_F16) && (HDim == 32 || HDim == 64 || HDim == 96) && Tn >= 1;
}

template <typename T> static int tl_fwd_by_hd(const void* qkv, void* out, float* lse, int B, int Tn, int H, int HDim, hipStream_t st, DropP drop) {
    if (HDim == 32) return tl_launch_fwd<T, 1>(qkv, out, lse, B, Tn, H, st, drop);
    if (HDim == 64) return tl_launch_fwd<T, 2>(qkv, out, lse, B, Tn, H, st, drop);
    if (HDim == 96) return tl_launch_fwd<T, 3>(qkv, out, lse, B, Tn, H, st, drop);
    return MFVIT_EINVAL;
}
template <typename T> static int tl_bwd_by_hd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int Tn, int H,
                                              int HDim, hipStream_t st, DropP drop) {
    if (HDim == 32) return tl_launch_bwd<T, 1>(qkv, out, dout, lse, dqkv, B, Tn, H, st, drop);
    if (HDim == 64) return tl_launch_bwd<T, 2>(qkv, out, dout, lse, dqkv, B, Tn, H, st, drop);
    if (HDim == 96) return tl_launch_bwd<T, 3>(qkv, out, dout, lse, dqkv, B, Tn, H, st, drop);
    return MFVIT_EINVAL;
}
// drop: attention dropout of the TransFuser GPT (make_drop; thr == 0: none).  Needs B * H * Tn * Tn < 2^32 mask indices.
int attn_fwd_tiled_drop(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, int HDim, DropP drop, hipStream_t st) {
    if (drop.thr && (double)B * H * Tn * Tn >= 4294967296.0) return MFVIT_EINVAL;
    if (dtype == MFVIT_BF16) return tl_fwd_by_hd<bf16>(qkv, out, lse, B, Tn, H, HDim, st, drop);
    if (dtype == MFVIT_BF16X3) return tl_fwd_by_hd<sbf16>(qkv, out, lse, B, Tn, H, HDim, st, drop);
    if (dtype == MFVIT_F16) return tl_fwd_by_hd<f16>(qkv, out, lse, B, Tn, H, HDim, st, drop);
    return MFVIT_EINVAL;
}
int attn_bwd_tiled_drop(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int Tn, int H, int HDim,
                        DropP drop, hipStream_t st) {
    if (drop.thr && (double)B * H * Tn * Tn >= 4294967296.0) return MFVIT_EINVAL;
    if (dtype == MFVIT_BF16) return tl_bwd_by_hd<bf16>(qkv, out, dout, lse, dqkv, B, Tn, H, HDim, st, drop);
    if (dtype == MFVIT_BF16X3) return tl_bwd_by_hd<sbf16>(qkv, out, dout, lse, dqkv, B, Tn, H, HDim, st, drop);
    if (dtype == MFVIT_F16) return tl_bwd_by_hd<f16>(qkv, out, dout, lse, dqkv, B, Tn, H, HDim, st, drop);
    return MFVIT_EINVAL;
}
int attn_fwd_tiled(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, int HDim, hipStream_t st) {
    return attn_fwd_tiled_drop(dtype, qkv, out, lse, B, Tn, H, HDim, make_drop(0.f, 0, 0), st);
}
int attn_bwd_tiled(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int Tn, int H, int HDim,
                   hipStream_t st) {
    return attn_bwd_tiled_drop(dtype, qkv, out, dout, lse, dqkv, B, Tn, H, HDim, make_drop(0.f, 0, 0), st);
}

}  // namespace mfvit
