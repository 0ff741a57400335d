// Fused multi-head self-attention forward for gfx950: QKV projection -> softmax(q k^T / sqrt(d)) v inside ONE workgroup per
// (image, head) - the "fused-attention kernel" of BASELINE.json's north star.  head_dim 32, element types bf16 / f16 / sbf16 (split).
//
//   phase 1  [q; k; v]^T (96 x T) = W_h (96 x D) . x^T (D x T) + bias : the head's 96 rows of the packed qkv weight against the image's
//            LayerNorm-ed tokens x = y1[b] (T x D), K = D streamed through LDS in 128-byte k tiles (register-staged, double buffered).
//            The product is taken TRANSPOSED (weight rows on the MFMA A side, tokens on the B side) so that a wave's accumulators are
//            q^T / k^T / v^T [d][token] with the token on the lane: the layout the attention core wants -
//              * q^T in accumulator form IS the B operand of S^T = K q^T (its k order is the accumulator's row order, so the K rows are
//                read in that order: two ds_read_b64 per fragment instead of one b128) - Q never touches LDS,
//              * k^T / v^T leave the registers as [token][32] rows (8-byte pieces) into the LDS images the core reads, and - when the
//                backward will need them - into the global qkv tensor.
//   phase 2  the attention core of attention_mfma.hip (scores of a 32-query tile per wave against all keys, exact softmax over register
//            resident scores, P^T straight from the accumulator into the PV MFMA).
// The x tile is re-read by the 12 heads of an image (same XCD: xcd_remap keeps an image's heads together), qkv is never READ back from
// HBM in the forward: 77 MB (bf16) / 155 MB (split) of reads per launch at the bench shape disappear, and the write disappears too for
// no-grad forwards (momentum encoder, frozen backbones).
// LDS: max(2 stages x (96 + Tpad) x 128 B, K and V images) = 80 KB at T = 197: two workgroups per CU.
#include "gemm.cuh"
#include "prof.h"

namespace mfvit {

namespace {

constexpr int FH = 32;          // head_dim
constexpr int FNT = 512;        // 8 waves: wave w owns tokens 32 w .. 32 w + 31
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

template <typename T> struct FT {
    static constexpr bool SP = is_split<T>::value;
    static constexpr int EP = SP ? 2 : 1;
    static constexpr int RB = 64 * EP;           // bytes of one head row piece
    // K / V image pitch.  Every access to these images is 8 bytes wide (accumulator-order ds_read_b64 row reads, ds_read_b64_tr_b16,
    // 8-byte row-piece stores), so the rows only need 8-byte alignment: with RB + 8 (18 / 34 dwords) the 32 rows of a ds_read_b64
    // group fall into 32 distinct bank pairs (conflict free); RB + 16 made rows r and r + 16 collide (6.3 M conflict cycles per launch).
    static constexpr int PITCH = RB + 8;
    // V image: read only through ds_read_b64_tr_b16 (4 rows x 32 B per 16 lanes): plain 64-byte rows WITHOUT padding put the 4 rows of a
    // block into 4 disjoint 16-dword windows (conflict free); split rows keep the K pitch (rows q, q + 2 overlap partly either way)
    static constexpr int VPITCH = SP ? RB + 8 : RB;
    typedef typename Vec8<T>::type frag_t;
    typedef typename Vec4<T>::type vec4_t;
    typedef typename Vec4<T>::elem E;
};

__device__ __forceinline__ f32x16 f_mma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 f_mma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <typename T> __device__ __forceinline__ f32x16 f_mma3(typename Vec8<T>::type ah, typename Vec8<T>::type al, typename Vec8<T>::type bh,
                                                               typename Vec8<T>::type bl, f32x16 c) {
    if constexpr (is_split<T>::value) {
        c = f_mma(al, bh, c);
        c = f_mma(ah, bl, c);
    }
    return f_mma(ah, bh, c);
}
__device__ __forceinline__ int f_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// registers 8 s .. 8 s + 7 of an accumulator as a B fragment (k step s, ACCUMULATOR k order); split: hi and lo parts
template <typename T> __device__ __forceinline__ void f_pack8(const f32x16& v, int s, typename Vec8<T>::type& hi, typename Vec8<T>::type& lo) {
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
// K-image fragment in ACCUMULATOR k order: row (rowbase + lane & 31), element j = column 16 s + 8 (j >> 2) + 4 (lane >> 5) + (j & 3)
template <typename T> __device__ __forceinline__ typename Vec8<T>::type f_row_frag_acc(const char* img, int rowbase, int s, int lane, int part) {
    const char* a = img + (rowbase + (lane & 31)) * FT<T>::PITCH + 64 * part + (16 * s + 4 * (lane >> 5)) * 2;
    union { struct { uint2 a, b; } s; typename Vec8<T>::type v; } u;
    u.s.a = *(const uint2*)a;
    u.s.b = *(const uint2*)(a + 16);
    return u.v;
}
// transposed fragment of the V image in ACCUMULATOR k order (as attention_mfma.hip::tr_frag)
template <typename T> __device__ __forceinline__ typename Vec8<T>::type f_tr_frag(const char* img, int rowbase, int s, int lane, int part) {
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const char* a = img + (rowbase + 16 * s + 4 * h + q) * FT<T>::VPITCH + 64 * part + (16 * g1 + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 8 * FT<T>::VPITCH));
    union { struct { s16x4 a, b; } s; typename Vec8<T>::type v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}
// accumulator tile X^T[d][token] (token on the lane) -> the token's 32-wide row piece (4 x 8-byte stores; split: hi and lo pieces)
template <typename T, typename PTR> __device__ __forceinline__ void f_store_tile_T(PTR row_ptr, const f32x16& acc, float mul, int lane) {
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

constexpr int F_NKC = 4;   // key tiles per register-resident score chunk

template <typename T>
__global__ __launch_bounds__(FNT, 1) void mhsa_fused_fwd_kernel(const typename Vec4<T>::elem* __restrict__ x, long ldx,
                                                                const typename Vec4<T>::elem* __restrict__ wqkv, long ldw,
                                                                const float* __restrict__ bias, typename Vec4<T>::elem* __restrict__ qkv_out,
                                                                typename Vec4<T>::elem* __restrict__ out, float* __restrict__ lse, int Tn, int H,
                                                                int D, float scale) {
    typedef FT<T> F;
    typedef typename F::E E;
    typedef typename F::frag_t frag_t;
    constexpr int EP = F::EP, LO = F::SP ? 1 : 0;
    constexpr int BKB = 128, BK = BKB / 2;                       // storage elements per k tile (bf16: 64 k; split: one 32-k group)
    typedef KTile<T, 96, BKB> TA;                                // weight tile: the head's q, k, v rows
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Tpad = (Tn + 31) & ~31;
    const int stage_bytes = (96 + Tpad) * BKB;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);            // the heads of one image share an XCD (the x tile is re-read from its L2)
    const int b = bid / H, h = bid % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const E* xb = x + (long)b * Tn * ldx;
    const int ntile = Tpad >> 5;                                 // token tiles (= active waves in phase 1), <= 8
    const bool active = wave < ntile;

    // ---------------------------------------------------------------- phase 1: [q; k; v]^T = W_h x^T
    // chunk q of the stage: 0 .. 96*8-1 = weight tile (row a -> global row (a / 32) D + h 32 + a % 32), then Tpad*8 chunks of x rows
    const int nchunks = (96 + Tpad) * 8;
    constexpr int MAXCH = (96 + 256) * 8 / FNT + 1;              // chunks per thread (Tpad <= 256)
    uint4 reg[MAXCH];
    auto load_stage = [&](int kt) {
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            int q = tid + i * FNT;
            q = q < nchunks ? q : nchunks - 1;                   // clamped: every load is unconditional (no branch around a load)
            const int row = q >> 3, c = q & 7;
            const E* src;
            if (row < 96) {
                src = wqkv + ((long)(row >> 5) * D + h * FH + (row & 31)) * ldw + kt * BK + c * 8;
            } else {
                int t = row - 96;
                t = t < Tn ? t : Tn - 1;
                src = xb + (long)t * ldx + kt * BK + c * 8;
            }
            reg[i] = *(const uint4*)src;
        }
    };
    auto store_stage = [&](char* st) {
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int q = tid + i * FNT;
            if (q < nchunks) {
                const int row = q >> 3, c = q & 7;
                if (row < 96) TA::put(st, row, c, reg[i]);
                else KTile<T, 256, BKB>::put(st + TA::BYTES, row - 96, c, reg[i]);     // same swizzle function, image of Tpad rows
            }
        }
    };
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nk = D * EP / BK;
    load_stage(0);
    store_stage(lds);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* ta = lds + cur * stage_bytes;
        const char* tb = ta + TA::BYTES;
        load_stage(kt + 1 < nk ? kt + 1 : kt);
        if (active) {
            if constexpr (F::SP) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const frag_t bh = KTile<T, 256, BKB>::frag(tb, wave * 32, s, lane), bl = KTile<T, 256, BKB>::frag(tb, wave * 32, s + 2, lane);
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[j] = f_mma3<T>(TA::frag(ta, j * 32, s, lane), TA::frag(ta, j * 32, s + 2, lane), bh, bl, acc[j]);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const frag_t bf = KTile<T, 256, BKB>::frag(tb, wave * 32, s, lane);
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[j] = f_mma(TA::frag(ta, j * 32, s, lane), bf, acc[j]);
                }
            }
        }
        store_stage(lds + (cur ^ 1) * stage_bytes);
        __syncthreads();
        cur ^= 1;
    }
    // bias: accumulator row = output channel d of q / k / v
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] += bias ? bias[(long)j * D + h * FH + f_acc_row(r, lane)] : 0.f;
    // the staging buffers are dead (last barrier passed): K and V images go on top of them
    char* Ks = lds;
    char* Vs = lds + Tpad * F::PITCH;
    const int tok = wave * 32 + (lane & 31);
    if (active) {
        const bool live = tok < Tn;
        // keys / values past the end: zero rows (their scores are masked below, their V rows must not hold NaNs)
        f_store_tile_T<T>((E*)(Ks + tok * F::PITCH), acc[1], live ? 1.0f : 0.0f, lane);
        f_store_tile_T<T>((E*)(Vs + tok * F::VPITCH), acc[2], live ? 1.0f : 0.0f, lane);
        if (qkv_out && live) {
            E* dst = qkv_out + ((long)b * Tn + tok) * 3 * H * FH * EP + (long)h * FH * EP;
            f_store_tile_T<T>(dst, acc[0], 1.0f, lane);
            f_store_tile_T<T>(dst + (long)H * FH * EP, acc[1], 1.0f, lane);
            f_store_tile_T<T>(dst + 2L * H * FH * EP, acc[2], 1.0f, lane);
        }
    }
    __syncthreads();
    if (!active) return;

    // ---------------------------------------------------------------- phase 2: attention core, this wave's 32 queries
    frag_t qf[2], ql[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) f_pack8<T>(acc[0], s, qf[s], ql[s]);       // q^T accumulator = B operand (accumulator k order)
    const float c = scale * 1.4426950408889634f;
    float m2 = -INFINITY, lsum = 0.f;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    for (int k0 = 0; k0 < ntile; k0 += F_NKC) {
        const int n = ntile - k0 < F_NKC ? ntile - k0 : F_NKC;
        f32x16 sc[F_NKC];
#pragma unroll
        for (int t = 0; t < F_NKC; ++t) {
            if (t < n) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    sc[t] = f_mma3<T>(f_row_frag_acc<T>(Ks, (k0 + t) * 32, s, lane, 0), f_row_frag_acc<T>(Ks, (k0 + t) * 32, s, lane, LO), qf[s],
                                      ql[s], sc[t]);
                if ((k0 + t + 1) * 32 > Tn) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((k0 + t) * 32 + f_acc_row(r, lane) >= Tn) sc[t][r] = -INFINITY;
                }
            }
        }
        float cm = -INFINITY;
#pragma unroll
        for (int t = 0; t < F_NKC; ++t)
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
        for (int t = 0; t < F_NKC; ++t)
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
                    f_pack8<T>(sc[t], s, ph, pl);
                    o = f_mma3<T>(f_tr_frag<T>(Vs, (k0 + t) * 32, s, lane, 0), f_tr_frag<T>(Vs, (k0 + t) * 32, s, lane, LO), ph, pl, o);
                }
            }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (tok < Tn) {
        f_store_tile_T<T>(out + (((long)b * Tn + tok) * H + h) * FH * EP, o, 1.0f / lsum, lane);
        if (lane < 32) lse[((long)b * H + h) * Tn + tok] = (m2 + log2f(lsum)) * 0.6931471805599453f;
    }
}

template <typename T> int launch_fused(const void* x, long ldx, const void* wqkv, long ldw, const float* bias, void* qkv_out, void* out, float* lse,
                                       int B, int Tn, int H, int D, hipStream_t st) {
    typedef typename Vec4<T>::elem E;
    const int Tpad = (Tn + 31) & ~31;
    const int stage = (96 + Tpad) * 128, images = Tpad * (FT<T>::PITCH + FT<T>::VPITCH);
    const int bytes = 2 * stage > images ? 2 * stage : images;
    static PerDeviceOnce attr;
    if (attr.first()) { (void)hipFuncSetAttribute((const void*)mhsa_fused_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    // algorithmic flops: the projection of this head + the attention core
    ProfScope ps(PROF_ATTN_FWD, 2.0 * B * (double)Tn * 3 * H * FH * D + 4.0 * B * H * (double)Tn * Tn * FH, 0, st);
    MFVIT_LAUNCH((mhsa_fused_fwd_kernel<T>), dim3(B * H), dim3(FNT), bytes, st, (const E*)x, ldx, (const E*)wqkv, ldw, bias, (E*)qkv_out, (E*)out, lse,
                 Tn, H, D, 1.0f / sqrtf((float)FH));
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

}  // namespace

bool mhsa_fused_supported(int dtype, int Tn, int HDim, int D) {
    if (dtype != MFVIT_BF16 && dtype != MFVIT_BF16X3 && dtype != MFVIT_F16) return false;
    const int ep = dtype == MFVIT_BF16X3 ? 2 : 1;
    if (HDim != FH || Tn < 1 || Tn > 256 || D % 64 || D * ep % 64) return false;
    return true;
}

int mhsa_fused_fwd(int dtype, const void* x, long ldx, const void* wqkv, long ldw, const float* bias, void* qkv_out, void* out, float* lse, int B,
                   int Tn, int H, int D, hipStream_t st) {
    if (dtype == MFVIT_BF16) return launch_fused<bf16>(x, ldx, wqkv, ldw, bias, qkv_out, out, lse, B, Tn, H, D, st);
    if (dtype == MFVIT_BF16X3) return launch_fused<sbf16>(x, ldx, wqkv, ldw, bias, qkv_out, out, lse, B, Tn, H, D, st);
    if (dtype == MFVIT_F16) return launch_fused<f16>(x, ldx, wqkv, ldw, bias, qkv_out, out, lse, B, Tn, H, D, st);
    return MFVIT_EINVAL;
}

}  // namespace mfvit
