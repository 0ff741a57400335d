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
#include "common.cuh"
#include "prof.h"

namespace mfvit {

namespace {

constexpr int HD = 32;
constexpr int RSB = 80;  // LDS row pitch in bytes (64 B of data + 16): b128 row reads of 16 consecutive rows are conflict-free

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// natural-order fragment: row (rowbase + lane&31), elements d = 16 s + 8 (lane>>5) .. +8
template <int PITCH = RSB> __device__ __forceinline__ bf16x8 row_frag(const char* img, int rowbase, int s, int lane) {
    return *(const bf16x8*)(img + (rowbase + (lane & 31)) * PITCH + 32 * s + 16 * (lane >> 5));
}
// transposed fragment in ACCUMULATOR k order: lane holds column d = lane&31; element j = row (rowbase + 16 s + 8 (j>>2) + 4 h + (j&3))
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int pitch, int rowbase, int s, int lane) {
    const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const char* a = img + (rowbase + 16 * s + 4 * h + q) * pitch + (16 * g1 + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 8 * pitch));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)v[8 * s + j];
    return o;
}
__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// store an accumulator tile X^T[d][col] (col on the lane) as rows of a [.., HD] bf16 tensor: 4 x 8-byte stores per lane
__device__ __forceinline__ void store_tile_T(bf16* row_ptr, const f32x16& acc, float mul, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[4 * g + j] * mul);
        *(bf16x4*)(row_ptr + 8 * g + 4 * (lane >> 5)) = o;
    }
}

constexpr int NKC = 4;  // key tiles per register-resident chunk (4 x 16 = 64 score registers -> 2 waves per SIMD)

__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, float* __restrict__ lse,
                                                            int Tn, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Tpad = (Tn + 31) & ~31;
    char* Ks = lds;
    char* Vs = lds + Tpad * RSB;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);  // the heads of one image share an XCD: their 64 B row pieces share L2 lines
    const int b = bid / H, h = bid % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rs = 3L * H * HD;
    const bf16* base = qkv + (long)b * Tn * rs + h * HD;
    for (int q = threadIdx.x; q < Tpad * 4; q += blockDim.x) {   // K and V pieces of a row loaded together (one HBM round trip, not two)
        const int t = q >> 2, cidx = q & 3;
        uint4 vk = make_uint4(0, 0, 0, 0), vv = vk;
        if (t < Tn) {
            const bf16* rp = base + (long)t * rs + (long)H * HD + 8 * cidx;
            vk = *(const uint4*)rp;
            vv = *(const uint4*)(rp + (long)H * HD);
        }
        *(uint4*)(Ks + t * RSB + 16 * cidx) = vk;
        *(uint4*)(Vs + t * 64 + 16 * cidx) = vv;
    }
    __syncthreads();
    const float c = scale * 1.4426950408889634f;
    const int nt = Tpad >> 5;
    for (int qt = wave; qt < nt; qt += 4) {
        const int q = qt * 32 + (lane & 31);
        const int qc = q < Tn ? q : Tn - 1;
        bf16x8 qf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[s] = *(const bf16x8*)(base + (long)qc * rs + 16 * s + 8 * (lane >> 5));
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
                    for (int s = 0; s < 2; ++s) sc[t] = mma(row_frag(Ks, (k0 + t) * 32, s, lane), qf[s], sc[t]);
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
                    for (int s = 0; s < 2; ++s) o = mma(tr_frag(Vs, 64, (k0 + t) * 32, s, lane), pack8(sc[t], s), o);
                }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        if (q < Tn) {
            store_tile_T(out + ((long)b * Tn + q) * H * HD + h * HD, o, 1.0f / lsum, lane);
            if (lane < 32) lse[((long)b * H + h) * Tn + q] = (m2 + log2f(lsum)) * 0.6931471805599453f;
        }
    }
}

// One 8-wave workgroup per (image, head): Q, K, V, dO of the head are read from HBM exactly once into four LDS images
// (PITCH = 80: conflict-free row reads, T <= 480; PITCH = 64 reaches T = 608 inside the 160 KB of LDS).  -lse/scale and
// -D = -rowsum(dO o O) enter the score MFMAs as accumulator initial values, so S - lse/scale and dP - D come out of the
// matrix core and the VALU work per element is mul, exp2, mul, cvt.
template <int PITCH>
__global__ __launch_bounds__(512) void attn_bwd_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                            const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16* __restrict__ dqkv, int Tn, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int Tpad = (Tn + 31) & ~31;
    char* Qs = lds;
    char* Ks = Qs + Tpad * PITCH;
    char* Vs = Ks + Tpad * PITCH;
    char* dOs = Vs + Tpad * PITCH;
    float* Ls = (float*)(dOs + Tpad * PITCH);
    float* Ds = Ls + Tpad;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);  // the heads of one image share an XCD: their 64 B row pieces share L2 lines
    const int b = bid / H, h = bid % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rs = 3L * H * HD, os = (long)H * HD;
    const bf16* base = qkv + (long)b * Tn * rs + h * HD;
    bf16* dbase = dqkv + (long)b * Tn * rs + h * HD;
    const bf16* obase = out + (long)b * Tn * os + h * HD;
    const bf16* dobase = dout + (long)b * Tn * os + h * HD;
    // one fused staging pass: the five loads of a row piece (Q, K, V, dO, O) are issued together - staged image by image, each image
    // paid its own HBM round trip before the next one's loads went out (operands are cold inside the training step)
    for (int q = threadIdx.x; q < Tpad * 4; q += blockDim.x) {  // four lanes per row, 16 B each
        const int t = q >> 2, cidx = q & 3;
        uint4 vq = make_uint4(0, 0, 0, 0), vk = vq, vv = vq, vd = vq;
        float D = 0.f, L = -1e30f;                                 // padded queries: p = exp2(-1e30 c) = 0
        if (t < Tn) {
            const bf16* rp = base + (long)t * rs + 8 * cidx;
            vq = *(const uint4*)rp;
            vk = *(const uint4*)(rp + (long)H * HD);
            vv = *(const uint4*)(rp + 2L * H * HD);
            vd = *(const uint4*)(dobase + (long)t * os + 8 * cidx);
            const bf16x8 o = *(const bf16x8*)(obase + (long)t * os + 8 * cidx);
            if (cidx == 0) L = -lse[((long)b * H + h) * Tn + t] / scale;
            union { uint4 u; bf16x8 b8; } cv;
            cv.u = vd;
#pragma unroll
            for (int j = 0; j < 8; ++j) D = fmaf((float)cv.b8[j], (float)o[j], D);
        }
        *(uint4*)(Qs + t * PITCH + 16 * cidx) = vq;
        *(uint4*)(Ks + t * PITCH + 16 * cidx) = vk;
        *(uint4*)(Vs + t * PITCH + 16 * cidx) = vv;
        *(uint4*)(dOs + t * PITCH + 16 * cidx) = vd;
        D += __shfl_xor(D, 1, 64);
        D += __shfl_xor(D, 2, 64);
        if (cidx == 0) {
            Ds[t] = -D;
            Ls[t] = L;
        }
    }
    __syncthreads();
    const float c = scale * 1.4426950408889634f;
    const int nt = Tpad >> 5;
    // ---------------- phase A: dQ, wave = query tile
    for (int qt = wave; qt < nt; qt += 8) {
        bf16x8 qf[2], dof[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[s] = row_frag<PITCH>(Qs, qt * 32, s, lane);
            dof[s] = row_frag<PITCH>(dOs, qt * 32, s, lane);
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
                st = mma(row_frag<PITCH>(Ks, kt * 32, s, lane), qf[s], st);
                dp = mma(row_frag<PITCH>(Vs, kt * 32, s, lane), dof[s], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)  // dS^T / scale (padded keys: K rows are zero, so their dQ contribution vanishes)
                st[r] = __builtin_amdgcn_exp2f(st[r] * c) * dp[r];
#pragma unroll
            for (int s = 0; s < 2; ++s) dq = mma(tr_frag(Ks, PITCH, kt * 32, s, lane), pack8(st, s), dq);
        }
        const int q = qt * 32 + (lane & 31);
        if (q < Tn) store_tile_T(dbase + (long)q * rs, dq, scale, lane);
    }
    // ---------------- phase B: dK, dV, wave = key tile
    for (int kt = wave; kt < nt; kt += 8) {
        bf16x8 kf[2], vf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[s] = row_frag<PITCH>(Ks, kt * 32, s, lane);
            vf[s] = row_frag<PITCH>(Vs, kt * 32, s, lane);
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
                sm = mma(row_frag<PITCH>(Qs, qt * 32, s, lane), kf[s], sm);   // S[q][key] - lse[q]/scale
                dp = mma(row_frag<PITCH>(dOs, qt * 32, s, lane), vf[s], dp);  // dP[q][key] - D[q]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sm[r] = __builtin_amdgcn_exp2f(sm[r] * c);
                dp[r] *= sm[r];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dv = mma(tr_frag(dOs, PITCH, qt * 32, s, lane), pack8(sm, s), dv);
                dk = mma(tr_frag(Qs, PITCH, qt * 32, s, lane), pack8(dp, s), dk);
            }
        }
        const int k = kt * 32 + (lane & 31);
        if (k < Tn) {
            store_tile_T(dbase + (long)k * rs + (long)H * HD, dk, scale, lane);
            store_tile_T(dbase + (long)k * rs + 2L * H * HD, dv, 1.0f, lane);
        }
    }
}

// column sums of a bf16 [M][N] matrix into f32 out[N] (atomicAdd); N % 8 == 0.  Used for d qkv.bias.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16* __restrict__ x, long ld, float* __restrict__ out, int M, int N) {
    const int r0 = blockIdx.x * 64, r1 = min(M, r0 + 64);
    for (int cch = threadIdx.x; cch < N / 8; cch += 256) {
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = 0.f;
        for (int r = r0; r < r1; ++r) {
            const bf16x8 v = *(const bf16x8*)(x + (long)r * ld + 8 * cch);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += (float)v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(out + 8 * cch + j, a[j]);
    }
}

}  // namespace

bool attn_mfma_supported(int dtype, int Tn, int HDim, bool backward) {
    if (dtype != MFVIT_BF16 || HDim != HD || Tn < 1) return false;
    const int Tpad = (Tn + 31) & ~31;
    const int bytes = backward ? 4 * Tpad * 64 + 2 * Tpad * 4 : Tpad * RSB + Tpad * 64;
    return bytes <= 160 * 1024;
}

int attn_fwd_mfma(const void* qkv, void* out, float* lse, int B, int Tn, int H, hipStream_t st) {
    const int Tpad = (Tn + 31) & ~31;
    const int bytes = Tpad * RSB + Tpad * 64;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    ProfScope ps(PROF_ATTN_FWD, 4.0 * B * H * (double)Tn * Tn * HD, 0, st);
    MFVIT_LAUNCH(attn_fwd_mfma_kernel, dim3(B * H), dim3(256), bytes, st, (const bf16*)qkv, (bf16*)out, lse, Tn, H,
                       1.0f / sqrtf((float)HD));
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}
int attn_bwd_mfma(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn, int H,
                  hipStream_t st) {
    const int Tpad = (Tn + 31) & ~31;
    const bool wide = 4 * Tpad * RSB + 2 * Tpad * 4 <= 160 * 1024;
    const int bytes = 4 * Tpad * (wide ? RSB : 64) + 2 * Tpad * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<RSB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    {
        ProfScope ps(PROF_ATTN_BWD, 8.0 * B * H * (double)Tn * Tn * HD, 0, st);
        if (wide)
            MFVIT_LAUNCH(attn_bwd_mfma_kernel<RSB>, dim3(B * H), dim3(512), bytes, st, (const bf16*)qkv, (const bf16*)out, (const bf16*)dout,
                         lse, (bf16*)dqkv, Tn, H, 1.0f / sqrtf((float)HD));
        else
            MFVIT_LAUNCH(attn_bwd_mfma_kernel<64>, dim3(B * H), dim3(512), bytes, st, (const bf16*)qkv, (const bf16*)out, (const bf16*)dout,
                         lse, (bf16*)dqkv, Tn, H, 1.0f / sqrtf((float)HD));
        MFVIT_CHECK_LAUNCH();
    }
    if (dbias) {
        const int M = B * Tn, N = 3 * H * HD;
        MFVIT_LAUNCH(colsum_bf16_kernel, dim3((M + 63) / 64), dim3(256), 0, st, (const bf16*)dqkv, (long)N, dbias, M, N);
        MFVIT_CHECK_LAUNCH();
    }
    return MFVIT_OK;
}

}  // namespace mfvit
