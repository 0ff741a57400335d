// Cross-attention token exchange + two-stream late fusion (Fus_CrossViT) on gfx950, forward and backward, all f32.
//
// Reference (paths relative to /root/reference/moco_pretraining/moco):
//   model/module.py:15-21      PreNorm (LayerNorm eps 1e-5)          model/module.py:108-137  CrossAttention (3 heads x 128)
//   model/crossvit_2vits_..._std002_sum.py:35-65   MultiScaleTransformerEncoder.forward (bidirectional cls<->patch exchange)
//   model/crossvit_2vits_..._std002_sum.py:126-157 Fus_CrossViT.forward (residual, cls pool, two heads, sum)
//
// Algebra: the query is ONE row (the cls token), so per head h
//     scores_h[t] = scale * z_t . (Wk_h^T (Wq z_0)_h)            =: scale * z_t . kq_h
//     attn_h . V_h = Wv_h (sum_t a_h[t] z_t)                     =: Wv_h u_h
// i.e. the two (B*T) x 384 x 384 K/V projections (99% of the reference's 117 MFLOP per direction) fold into three
// 384-vectors per sample.  What remains is HBM-bound: one streaming pass pair over the (B, T, 384) tokens of the other
// stream (roofline: HBM bytes = tokens read twice, second time from L2), plus batch-sized f32 MFMA GEMMs.
// Only row 0 of the post-exchange LayerNorm / residual is consumed (pool = 'cls', FUS:144-145), so only that row is computed.
//
// Parameter arena (f32, state-dict order of Fus_CrossViT; 8 D^2 + 10 D + 2 (C D + C) floats):
//   [0].norm.{w,b} [0].fn.{wq,wk,wv,proj.w,proj.b}  [1].{w,b}  [2].norm.{w,b} [2].fn.{...}  [3].{w,b}
//   mlp_head_cxr.0.{w,b}  mlp_head_enh.0.{w,b}
// Direction 0: CXR cls attends ENH patches through [0], post-norm [3], head cxr   (FUS:57-63)
// Direction 1: ENH cls attends CXR patches through [2], post-norm [1], head enh   (FUS:48-55)
#include "kernels.h"
#include "prof.h"

#include <string.h>

using namespace mfvit;

namespace {

constexpr int D = 384, NPL = 6, NH = 3, DH = 128;

struct FusLayout {  // parameter offsets (floats)
    long ca[2], post[2], head[2];
    long n_w, n_b, wq, wk, wv, wp, bp;  // inside a CA block
    long ca_stride;
    long total;
    int no_norm;   // bare CrossAttention (MOD:123-137 without the PreNorm around it): z = x, the norm entries of a block are never touched
};
FusLayout fus_layout(int C) {
    FusLayout L;
    const long blk = 4L * D * D + 3L * D;
    L.n_w = 0; L.n_b = D; L.wq = 2L * D; L.wk = L.wq + (long)D * D; L.wv = L.wk + (long)D * D; L.wp = L.wv + (long)D * D;
    L.bp = L.wp + (long)D * D;
    const long o0 = 0, o1 = blk, o2 = o1 + 2 * D, o3 = o2 + blk, oh = o3 + 2 * D;
    L.ca[0] = o0; L.ca[1] = o2;
    L.post[0] = o3; L.post[1] = o1;
    L.head[0] = oh; L.head[1] = oh + (long)C * D + C;
    L.ca_stride = o2 - o0;
    L.total = oh + 2 * ((long)C * D + C);
    L.no_norm = 0;
    return L;
}

struct FusWs {  // workspace offsets (floats)
    long z0, st0, qv, kq, u, a, st, o, outp, fus, stc;          // forward (kept for backward)
    long wT;                                                     // [2][4][D*D] transposed wq, wk, wv, wp
    long dout, dqp, dob, du, dkq, dz0p, dqv, dz0q;               // backward scratch
    // per-sample partials of the small parameter gradients (round 6: plain stores + one fixed-order reduce instead of float atomics - the same bits on every run):
    //   pa [2][B][3 D]    d post-LN weight | bias | d proj.bias          (x_finish_bwd)
    //   pb [2][B][2 C D]  d head weight | d backbone-head weight          (x_finish_bwd)
    //   pc [2][B][2 C]    d head bias | d backbone-head bias              (x_finish_bwd)
    //   pd [2][2 B][2 D]  d pre-norm weight | bias: rows 0 .. B - 1 from x_stream_bwd (token rows 1 ..), rows B .. 2 B - 1 from x_row0_bwd (the cls row)
    long pa, pb, pc, pd;
    long total;
};
FusWs fus_ws(int B, int T, int C) {
    FusWs W;
    long o = 0;
    auto take = [&](long n) { long r = o; o += (n + 63) & ~63L; return r; };
    W.z0 = take(2L * B * D); W.st0 = take(2L * B * 2); W.qv = take(2L * B * D); W.kq = take(2L * B * NH * D);
    W.u = take(2L * B * NH * D); W.a = take(2L * B * NH * T); W.st = take(2L * B * 2 * T); W.o = take(2L * B * D);
    W.outp = take(2L * B * D); W.fus = take(2L * B * D); W.stc = take(2L * B * 2);
    W.wT = take(2L * 4 * D * D);
    W.dout = take(2L * B * D); W.dqp = take(2L * B * D); W.dob = take(2L * B * D); W.du = take(2L * B * NH * D);
    W.dkq = take(2L * B * NH * D); W.dz0p = take(2L * B * D); W.dqv = take(2L * B * D); W.dz0q = take(2L * B * D);
    W.pa = take(2L * B * 3 * D); W.pb = take(2L * B * 2 * C * D); W.pc = take(2L * B * 2 * C); W.pd = take(2L * 2 * B * 2 * D);
    W.total = o;
    return W;
}

// ---------------------------------------------------------------------------------------------- kernels
// z0[dir][b] = LN_pre(own cls row); one wave per (b, dir)
__global__ __launch_bounds__(64) void x_cls_ln_kernel(const float* __restrict__ fc, const float* __restrict__ fe,
                                                       const float* __restrict__ params, FusLayout L, float eps, int B, int T,
                                                       float* __restrict__ z0, float* __restrict__ st0) {
    const int b = blockIdx.x, dir = blockIdx.y, lane = threadIdx.x;
    const float* row = (dir == 0 ? fc : fe) + (long)b * T * D;
    const float* g = params + L.ca[dir] + L.n_w;
    const float* be = params + L.ca[dir] + L.n_b;
    float v[NPL], s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { v[i] = row[lane + 64 * i]; s += v[i]; }
    if (L.no_norm) {
#pragma unroll
        for (int i = 0; i < NPL; ++i) z0[((long)dir * B + b) * D + lane + 64 * i] = v[i];
        if (lane == 0) { st0[((long)dir * B + b) * 2] = 0.f; st0[((long)dir * B + b) * 2 + 1] = 1.f; }
        return;
    }
    const float mu = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) * (1.f / D) + eps);
#pragma unroll
    for (int i = 0; i < NPL; ++i) z0[((long)dir * B + b) * D + lane + 64 * i] = (v[i] - mu) * rs * g[lane + 64 * i] + be[lane + 64 * i];
    if (lane == 0) { st0[((long)dir * B + b) * 2] = mu; st0[((long)dir * B + b) * 2 + 1] = rs; }
}

// Streaming pass over the T rows [own cls ; other stream's patches] of one (sample, direction):
//   stats, scores against kq_h, softmax over t, u_h = sum_t a_h[t] z_t.   LDS: sc[3][T] | st[2][T] | red[XW][18*64]
// XW waves per workgroup, a wave per row: ONE workgroup per CU (256 (sample, direction) pairs at the bench shape), so the rows in flight per
// CU are the waves of that workgroup - with four (round 1 - 3) the pass ran at 1.7 TB/s.
template <int XW>
__global__ __launch_bounds__(XW * 64) void x_stream_fwd_kernel(const float* __restrict__ fc, const float* __restrict__ fe,
                                                           const float* __restrict__ params, FusLayout L, float eps, float scale, int B,
                                                           int T, const float* __restrict__ kq, float* __restrict__ u,
                                                           float* __restrict__ a_out, float* __restrict__ st_out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    float* sc = xs;
    float* st = sc + NH * T;
    float* red = st + 2 * T;
    const int b = blockIdx.x, dir = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* own = (dir == 0 ? fc : fe) + (long)b * T * D;
    const float* oth = (dir == 0 ? fe : fc) + (long)b * T * D;
    float g[NPL], be[NPL], kqv[NH][NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        g[i] = L.no_norm ? 1.f : params[L.ca[dir] + L.n_w + lane + 64 * i];
        be[i] = L.no_norm ? 0.f : params[L.ca[dir] + L.n_b + lane + 64 * i];
#pragma unroll
        for (int h = 0; h < NH; ++h) kqv[h][i] = kq[(((long)dir * B + b) * NH + h) * D + lane + 64 * i];
    }
    for (int t = w; t < T; t += XW) {
        const float* row = t == 0 ? own : oth + (long)t * D;
        float v[NPL], s = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) { v[i] = row[lane + 64 * i]; s += v[i]; }
        const float mu = L.no_norm ? 0.f : wave_sum(s) * (1.f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) { v[i] -= mu; q += v[i] * v[i]; }
        const float rs = L.no_norm ? 1.f : rsqrtf(wave_sum(q) * (1.f / D) + eps);
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const float z = v[i] * rs * g[i] + be[i];
            d0 = fmaf(z, kqv[0][i], d0); d1 = fmaf(z, kqv[1][i], d1); d2 = fmaf(z, kqv[2][i], d2);
        }
        d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
        if (lane == 0) {
            sc[t] = d0 * scale; sc[T + t] = d1 * scale; sc[2 * T + t] = d2 * scale;
            st[t] = mu; st[T + t] = rs;
        }
    }
    __syncthreads();
    if (w < NH) {  // softmax over t for head w
        float m = -INFINITY;
        for (int t = lane; t < T; t += 64) m = fmaxf(m, sc[w * T + t]);
        m = wave_max(m);
        float s = 0.f;
        for (int t = lane; t < T; t += 64) { const float e = __expf(sc[w * T + t] - m); sc[w * T + t] = e; s += e; }
        s = 1.f / wave_sum(s);
        for (int t = lane; t < T; t += 64) {
            const float p = sc[w * T + t] * s;
            sc[w * T + t] = p;
            a_out[(((long)dir * B + b) * NH + w) * T + t] = p;
        }
    } else if (w == NH) {
        for (int t = lane; t < T; t += 64) {
            st_out[(((long)dir * B + b) * 2) * T + t] = st[t];
            st_out[(((long)dir * B + b) * 2 + 1) * T + t] = st[T + t];
        }
    }
    __syncthreads();
    float acc[NH][NPL];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < NPL; ++i) acc[h][i] = 0.f;
    for (int t = w; t < T; t += XW) {
        const float* row = t == 0 ? own : oth + (long)t * D;
        const float mu = st[t], rs = st[T + t];
        const float a0 = sc[t], a1 = sc[T + t], a2 = sc[2 * T + t];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const float z = (row[lane + 64 * i] - mu) * rs * g[i] + be[i];
            acc[0][i] = fmaf(a0, z, acc[0][i]); acc[1][i] = fmaf(a1, z, acc[1][i]); acc[2][i] = fmaf(a2, z, acc[2][i]);
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < NPL; ++i) red[(w * NH + h) * D + lane + 64 * i] = acc[h][i];
    __syncthreads();
    for (int q = threadIdx.x; q < NH * D; q += XW * 64) {
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < XW; ++w2) v += red[w2 * NH * D + q];
        u[((long)dir * B + b) * NH * D + q] = v;
    }
}

// cal = cls + outp ; c = LN_post(cal) ; fus_cls = cls + c ; ds = head(fus_cls) ; fused = ds_cxr + ds_enh ;
// x_S = backbone head_S(cls_S) (optional).  One block per sample, wave = direction.
__global__ __launch_bounds__(128) void x_finish_fwd_kernel(const float* __restrict__ fc, const float* __restrict__ fe,
                                                           const float* __restrict__ params, FusLayout L, float eps, int B, int T, int C,
                                                           const float* __restrict__ outp, float* __restrict__ fus,
                                                           float* __restrict__ stc, const float* __restrict__ hw_c,
                                                           const float* __restrict__ hb_c, const float* __restrict__ hw_e,
                                                           const float* __restrict__ hb_e, float* __restrict__ fused,
                                                           float* __restrict__ x_c, float* __restrict__ x_e) {
    __shared__ float dsm[2][64];
    const int b = blockIdx.x, dir = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* cls = (dir == 0 ? fc : fe) + (long)b * T * D;
    const float* op = outp + ((long)dir * B + b) * D;
    const float* g = params + L.post[dir];
    const float* be = g + D;
    float c0[NPL], v[NPL], s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { c0[i] = cls[lane + 64 * i]; v[i] = c0[i] + op[lane + 64 * i]; s += v[i]; }
    const float mu = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { v[i] -= mu; q += v[i] * v[i]; }
    const float rs = rsqrtf(wave_sum(q) * (1.f / D) + eps);
    float f[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        f[i] = c0[i] + v[i] * rs * g[lane + 64 * i] + be[lane + 64 * i];
        fus[((long)dir * B + b) * D + lane + 64 * i] = f[i];
    }
    if (lane == 0) { stc[((long)dir * B + b) * 2] = mu; stc[((long)dir * B + b) * 2 + 1] = rs; }
    const float* hw = params + L.head[dir];
    const float* hb = hw + (long)C * D;
    for (int c = 0; c < C; ++c) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) d = fmaf(f[i], hw[(long)c * D + lane + 64 * i], d);
        d = wave_sum(d);
        if (lane == 0) dsm[dir][c] = d + hb[c];
    }
    const float* bw = dir == 0 ? hw_c : hw_e;
    const float* bb = dir == 0 ? hb_c : hb_e;
    float* xo = dir == 0 ? x_c : x_e;
    if (bw && xo) {
        for (int c = 0; c < C; ++c) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < NPL; ++i) d = fmaf(c0[i], bw[(long)c * D + lane + 64 * i], d);
            d = wave_sum(d);
            if (lane == 0) xo[(long)b * C + c] = d + (bb ? bb[c] : 0.f);
        }
    }
    __syncthreads();
    if (threadIdx.x < C) fused[(long)b * C + threadIdx.x] = dsm[0][threadIdx.x] + dsm[1][threadIdx.x];
}

// Backward of x_finish_fwd: head, residual, post-LN; emits dout (= d outp) and the partial cls gradient dqp.
__global__ __launch_bounds__(128) void x_finish_bwd_kernel(const float* __restrict__ fc, const float* __restrict__ fe,
                                                           const float* __restrict__ params, FusLayout L, int B, int T, int C,
                                                           const float* __restrict__ outp, const float* __restrict__ fus,
                                                           const float* __restrict__ stc, const float* __restrict__ hw_c,
                                                           const float* __restrict__ hw_e, const float* __restrict__ dfused,
                                                           const float* __restrict__ dx_c, const float* __restrict__ dx_e,
                                                           float* __restrict__ dparams, float* __restrict__ dhw_c,
                                                           float* __restrict__ dhb_c, float* __restrict__ dhw_e,
                                                           float* __restrict__ dhb_e, float* __restrict__ dout, float* __restrict__ dqp,
                                                           float* __restrict__ pa, float* __restrict__ pb, float* __restrict__ pc) {
    const int b = blockIdx.x, dir = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // this sample's terms of the small parameter gradients: plain stores into its rows of the partial buffers (FusWs::pa / pb / pc), summed over the batch in a
    // fixed order by fus_reduce_partials
    float* pa_r = pa + ((long)dir * B + b) * 3 * D;
    float* pb_r = pb + ((long)dir * B + b) * 2 * C * D;
    float* pc_r = pc + ((long)dir * B + b) * 2 * C;
    const float* cls = (dir == 0 ? fc : fe) + (long)b * T * D;
    const float* op = outp + ((long)dir * B + b) * D;
    const float* g = params + L.post[dir];
    const float* hw = params + L.head[dir];
    const float mu = stc[((long)dir * B + b) * 2], rs = stc[((long)dir * B + b) * 2 + 1];
    float e[NPL], c0[NPL], f[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) { e[i] = 0.f; c0[i] = cls[lane + 64 * i]; f[i] = fus[((long)dir * B + b) * D + lane + 64 * i]; }
    for (int c = 0; c < C; ++c) {
        const float gc = dfused ? dfused[(long)b * C + c] : 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            e[i] = fmaf(gc, hw[(long)c * D + lane + 64 * i], e[i]);
            pb_r[(long)c * D + lane + 64 * i] = gc * f[i];
        }
        if (lane == 0) pc_r[c] = gc;
    }
    // post-LN backward: dc = e ; cal = cls + outp
    float xh[NPL], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        xh[i] = (c0[i] + op[lane + 64 * i] - mu) * rs;
        const float t = e[i] * g[lane + 64 * i];
        s1 += t; s2 += t * xh[i];
        pa_r[lane + 64 * i] = e[i] * xh[i];
        pa_r[D + lane + 64 * i] = e[i];
    }
    const float c1 = wave_sum(s1) * (1.f / D), c2 = wave_sum(s2) * (1.f / D);
    float dq[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const float dcal = rs * (e[i] * g[lane + 64 * i] - c1 - xh[i] * c2);
        dout[((long)dir * B + b) * D + lane + 64 * i] = dcal;
        pa_r[2 * D + lane + 64 * i] = dcal;
        dq[i] = e[i] + dcal;
    }
    // backbone classifier head on the cls row (x_S = head_S(cls_S)): FUS:131,135
    const float* bw = dir == 0 ? hw_c : hw_e;
    const float* dxs = dir == 0 ? dx_c : dx_e;
    for (int c = 0; c < C; ++c) {
        const float gc = bw && dxs ? dxs[(long)b * C + c] : 0.f;      // (no backbone head: zero rows - the reduce has no destination for them)
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            if (bw && dxs) dq[i] = fmaf(gc, bw[(long)c * D + lane + 64 * i], dq[i]);
            pb_r[(long)(C + c) * D + lane + 64 * i] = gc * c0[i];
        }
        if (lane == 0) pc_r[C + c] = gc;
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) dqp[((long)dir * B + b) * D + lane + 64 * i] = dq[i];
}

// Backward of the streaming pass.  LDS: sa[3][T] (a) | sd[3][T] (da -> ds) | st[2][T] | red[XW][30*64]
template <int XW>
__global__ __launch_bounds__(XW * 64) void x_stream_bwd_kernel(const float* __restrict__ fc, const float* __restrict__ fe,
                                                           const float* __restrict__ params, FusLayout L, float scale, int B, int T,
                                                           const float* __restrict__ kq, const float* __restrict__ a_in,
                                                           const float* __restrict__ st_in, const float* __restrict__ du,
                                                           float* __restrict__ dkq, float* __restrict__ dz0p, float* __restrict__ pd,
                                                           float* __restrict__ dfc, float* __restrict__ dfe) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    float* sa = xs;
    float* sd = sa + NH * T;
    float* st = sd + NH * T;
    float* red = st + 2 * T;
    const int b = blockIdx.x, dir = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* own = (dir == 0 ? fc : fe) + (long)b * T * D;
    const float* oth = (dir == 0 ? fe : fc) + (long)b * T * D;
    float* doth = dir == 0 ? dfe : dfc;
    float g[NPL], be[NPL], kqv[NH][NPL], duv[NH][NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        g[i] = L.no_norm ? 1.f : params[L.ca[dir] + L.n_w + lane + 64 * i];
        be[i] = L.no_norm ? 0.f : params[L.ca[dir] + L.n_b + lane + 64 * i];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            kqv[h][i] = kq[(((long)dir * B + b) * NH + h) * D + lane + 64 * i];
            duv[h][i] = du[(((long)dir * B + b) * NH + h) * D + lane + 64 * i];
        }
    }
    for (int q = threadIdx.x; q < NH * T; q += XW * 64) sa[q] = a_in[((long)dir * B + b) * NH * T + q];
    for (int q = threadIdx.x; q < 2 * T; q += XW * 64) st[q] = st_in[((long)dir * B + b) * 2 * T + q];
    __syncthreads();
    for (int t = w; t < T; t += XW) {
        const float* row = t == 0 ? own : oth + (long)t * D;
        const float mu = st[t], rs = st[T + t];
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const float z = (row[lane + 64 * i] - mu) * rs * g[i] + be[i];
            d0 = fmaf(z, duv[0][i], d0); d1 = fmaf(z, duv[1][i], d1); d2 = fmaf(z, duv[2][i], d2);
        }
        d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
        if (lane == 0) { sd[t] = d0; sd[T + t] = d1; sd[2 * T + t] = d2; }
    }
    __syncthreads();
    if (w < NH) {  // ds_h[t] = a_h[t] (da_h[t] - sum_t' a_h[t'] da_h[t'])
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s = fmaf(sa[w * T + t], sd[w * T + t], s);
        s = wave_sum(s);
        for (int t = lane; t < T; t += 64) sd[w * T + t] = sa[w * T + t] * (sd[w * T + t] - s);
    }
    __syncthreads();
    float akq[NH][NPL], ag[NPL], ab[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        ag[i] = ab[i] = 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h) akq[h][i] = 0.f;
    }
    for (int t = w; t < T; t += XW) {
        const float* row = t == 0 ? own : oth + (long)t * D;
        const float mu = st[t], rs = st[T + t];
        const float a0 = sa[t], a1 = sa[T + t], a2 = sa[2 * T + t];
        const float s0 = sd[t] * scale, s1 = sd[T + t] * scale, s2 = sd[2 * T + t] * scale;
        float xh[NPL], dz[NPL], p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            xh[i] = (row[lane + 64 * i] - mu) * rs;
            const float z = xh[i] * g[i] + be[i];
            akq[0][i] = fmaf(s0, z, akq[0][i]); akq[1][i] = fmaf(s1, z, akq[1][i]); akq[2][i] = fmaf(s2, z, akq[2][i]);
            dz[i] = a0 * duv[0][i] + a1 * duv[1][i] + a2 * duv[2][i] + s0 * kqv[0][i] + s1 * kqv[1][i] + s2 * kqv[2][i];
            const float tg = dz[i] * g[i];
            p1 += tg; p2 += tg * xh[i];
        }
        if (t == 0) {  // the cls row also receives Wq^T dqv later: LN backward for it runs in x_row0_bwd_kernel
#pragma unroll
            for (int i = 0; i < NPL; ++i) dz0p[((long)dir * B + b) * D + lane + 64 * i] = dz[i];
        } else {
#pragma unroll
            for (int i = 0; i < NPL; ++i) { ag[i] = fmaf(dz[i], xh[i], ag[i]); ab[i] += dz[i]; }
            if (doth) {
                const float c1 = L.no_norm ? 0.f : wave_sum(p1) * (1.f / D), c2 = L.no_norm ? 0.f : wave_sum(p2) * (1.f / D);
#pragma unroll
                for (int i = 0; i < NPL; ++i) doth[((long)b * T + t) * D + lane + 64 * i] = rs * (dz[i] * g[i] - c1 - xh[i] * c2);
            }
        }
    }
    // cross-wave reduce: 18 (dkq) + 6 (dgamma) + 6 (dbeta) values per lane
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
#pragma unroll
        for (int h = 0; h < NH; ++h) red[(w * 5 + h) * D + lane + 64 * i] = akq[h][i];
        red[(w * 5 + 3) * D + lane + 64 * i] = ag[i];
        red[(w * 5 + 4) * D + lane + 64 * i] = ab[i];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 5 * D; q += XW * 64) {
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < XW; ++w2) v += red[w2 * 5 * D + q];
        if (q < NH * D) dkq[((long)dir * B + b) * NH * D + q] = v;
        else if (L.no_norm) continue;
        else pd[((long)dir * 2 * B + b) * 2 * D + (q - 3 * D)] = v;        // this sample's d pre-norm weight | bias over its token rows 1 ..: row b of FusWs::pd
    }
}

// cls row of the pre-norm: dz0 = dz0p + dz0q (through Wq); LN backward; total cls gradient of the own stream.
__global__ __launch_bounds__(128) void x_row0_bwd_kernel(const float* __restrict__ fc, const float* __restrict__ fe,
                                                         const float* __restrict__ params, FusLayout L, int B, int T,
                                                         const float* __restrict__ st0, const float* __restrict__ dz0p,
                                                         const float* __restrict__ dz0q, const float* __restrict__ dqp,
                                                         float* __restrict__ pd, float* __restrict__ dfc, float* __restrict__ dfe) {
    const int b = blockIdx.x, dir = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* cls = (dir == 0 ? fc : fe) + (long)b * T * D;
    float* down = dir == 0 ? dfc : dfe;
    const float* g = params + L.ca[dir] + L.n_w;
    const float mu = st0[((long)dir * B + b) * 2], rs = st0[((long)dir * B + b) * 2 + 1];
    float dz[NPL], xh[NPL], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const long o = ((long)dir * B + b) * D + lane + 64 * i;
        dz[i] = dz0p[o] + dz0q[o];
        xh[i] = (cls[lane + 64 * i] - mu) * rs;
        if (L.no_norm) continue;
        const float t = dz[i] * g[lane + 64 * i];
        s1 += t; s2 += t * xh[i];
        pd[((long)dir * 2 * B + B + b) * 2 * D + lane + 64 * i] = dz[i] * xh[i];        // the cls row's terms: row B + b of FusWs::pd
        pd[((long)dir * 2 * B + B + b) * 2 * D + D + lane + 64 * i] = dz[i];
    }
    if (!down) return;
    if (L.no_norm) {   // (st0 = (0, 1): xh = x; no normalisation to differentiate)
#pragma unroll
        for (int i = 0; i < NPL; ++i) down[(long)b * T * D + lane + 64 * i] = dqp[((long)dir * B + b) * D + lane + 64 * i] + dz[i];
        return;
    }
    const float c1 = wave_sum(s1) * (1.f / D), c2 = wave_sum(s2) * (1.f / D);
#pragma unroll
    for (int i = 0; i < NPL; ++i)
        down[(long)b * T * D + lane + 64 * i] = dqp[((long)dir * B + b) * D + lane + 64 * i] + rs * (dz[i] * g[lane + 64 * i] - c1 - xh[i] * c2);
}

GemmP zg() { GemmP p; memset(&p, 0, sizeof(p)); return p; }

constexpr int XW_FWD = 16, XW_BWD = 8;      // waves per workgroup of the streaming passes (the backward holds ~100 registers per lane)
#define FUS_TRY(expr) do { int rc__ = (expr); if (rc__ != MFVIT_OK) return rc__; } while (0)

bool fus_ok(const mfvit_fusion_cfg* c) {
    return c && c->batch > 0 && c->tokens > 1 && c->dim == D && c->heads == NH && c->num_classes > 0 && c->num_classes <= 64;
}

}  // namespace

static int xattn_core_backward(int ndir, const FusLayout& L, const FusWs& W, const float* params, const float* f_cxr, const float* f_enh,
                               float* ws, int B, int T, float* dparams, float* df_cxr, float* df_enh, hipStream_t st);
static int fus_prenorm_job(int dir, const FusLayout& L, const FusWs& W, float* ws, int B, float* dparams, hipStream_t st);

extern "C" {

size_t mfvit_fusion_param_count(const mfvit_fusion_cfg* cfg) { return fus_ok(cfg) ? (size_t)fus_layout(cfg->num_classes).total : 0; }
size_t mfvit_fusion_workspace_bytes(const mfvit_fusion_cfg* cfg) { return fus_ok(cfg) ? (size_t)fus_ws(cfg->batch, cfg->tokens, cfg->num_classes).total * 4 : 0; }

// PreNorm -> CrossAttention of `ndir` directions up to the projection output outp[dir][b][D] (MOD:20,123-137)
static int xattn_core_forward(int ndir, const FusLayout& L, const FusWs& W, const float* params, const float* f_cxr, const float* f_enh,
                              float* ws, int B, int T, float eps_pre, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)DH);
    // transposed copies of wq, wk, wv, wp (used by kq here and by the backward)
    for (int dir = 0; dir < ndir; ++dir) {
        const long offs[4] = {L.wq, L.wk, L.wv, L.wp};
        for (int k = 0; k < 4; ++k)
            FUS_TRY(cast_transpose(MFVIT_F32, params + L.ca[dir] + offs[k], nullptr, ws + W.wT + ((long)dir * 4 + k) * D * D, D, D, st));
    }
    MFVIT_LAUNCH(x_cls_ln_kernel, dim3(B, ndir), dim3(64), 0, st, f_cxr, f_enh, params, L, eps_pre, B, T, ws + W.z0, ws + W.st0);
    MFVIT_CHECK_LAUNCH();
    {   // qv = z0 Wq^T
        GemmP p = zg();
        p.A = ws + W.z0; p.lda = D; p.W = params + L.ca[0] + L.wq; p.ldw = D; p.M = B; p.N = D; p.K = D;
        p.out0 = ws + W.qv; p.ldo0 = D;
        p.nb = ndir; p.nbi = 1; p.sAo = (long)B * D; p.sWo = L.ca_stride; p.sOo = (long)B * D;
        FUS_TRY(gemm_nt_tile(MFVIT_F32, EPI_NONE, p, st));
    }
    {   // kq_h = qv_h Wk_h  (W operand = WkT[:, h*128 ..])
        GemmP p = zg();
        p.A = ws + W.qv; p.lda = D; p.W = ws + W.wT + 1L * D * D; p.ldw = D; p.M = B; p.N = D; p.K = DH;
        p.out0 = ws + W.kq; p.ldo0 = NH * D;
        p.nb = ndir * NH; p.nbi = NH; p.sAo = (long)B * D; p.sAi = DH; p.sWo = 4L * D * D; p.sWi = DH; p.sOo = (long)B * NH * D; p.sOi = D;
        FUS_TRY(gemm_nt_tile(MFVIT_F32, EPI_NONE, p, st));
    }
    {
        const size_t lds = (size_t)(5 * T + XW_FWD * NH * D) * 4;
        static PerDeviceOnce attr;
        if (attr.first()) { (void)hipFuncSetAttribute((const void*)x_stream_fwd_kernel<XW_FWD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        ProfScope ps(PROF_XATTN_FWD, 0, 2.0 * B * T * D * 4 * 2, st);
        MFVIT_LAUNCH(x_stream_fwd_kernel<XW_FWD>, dim3(B, ndir), dim3(XW_FWD * 64), lds, st, f_cxr, f_enh, params, L, eps_pre, scale, B, T,
                           ws + W.kq, ws + W.u, ws + W.a, ws + W.st);
        MFVIT_CHECK_LAUNCH();
    }
    {   // o_h = u_h Wv_h^T
        GemmP p = zg();
        p.A = ws + W.u; p.lda = NH * D; p.W = params + L.ca[0] + L.wv; p.ldw = D; p.M = B; p.N = DH; p.K = D;
        p.out0 = ws + W.o; p.ldo0 = D;
        p.nb = ndir * NH; p.nbi = NH; p.sAo = (long)B * NH * D; p.sAi = D; p.sWo = L.ca_stride; p.sWi = (long)DH * D; p.sOo = (long)B * D; p.sOi = DH;
        FUS_TRY(gemm_nt_tile(MFVIT_F32, EPI_NONE, p, st));
    }
    {   // outp = o Wp^T + bp
        GemmP p = zg();
        p.A = ws + W.o; p.lda = D; p.W = params + L.ca[0] + L.wp; p.ldw = D; p.M = B; p.N = D; p.K = D;
        p.bias = params + L.ca[0] + L.bp;
        p.out0 = ws + W.outp; p.ldo0 = D;
        p.nb = ndir; p.nbi = 1; p.sAo = (long)B * D; p.sWo = L.ca_stride; p.sOo = (long)B * D; p.sBo = L.ca_stride;
        FUS_TRY(gemm_nt_tile(MFVIT_F32, EPI_BIAS, p, st));
    }
    return MFVIT_OK;
}

int mfvit_fusion_forward(const mfvit_fusion_cfg* cfg, const float* params, const float* f_cxr, const float* f_enh, const float* hw_cxr,
                         const float* hb_cxr, const float* hw_enh, const float* hb_enh, void* workspace, float* fused, float* x_cxr,
                         float* x_enh, mfvit_stream_t stream) {
    if (!fus_ok(cfg) || !params || !f_cxr || !f_enh || !workspace || !fused) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int B = cfg->batch, T = cfg->tokens, C = cfg->num_classes;
    const FusLayout L = fus_layout(C);
    const FusWs W = fus_ws(B, T, cfg->num_classes);
    float* ws = (float*)workspace;
    FUS_TRY(xattn_core_forward(2, L, W, params, f_cxr, f_enh, ws, B, T, cfg->eps_pre, st));
    MFVIT_LAUNCH(x_finish_fwd_kernel, dim3(B), dim3(128), 0, st, f_cxr, f_enh, params, L, cfg->eps_post, B, T, C, ws + W.outp,
                       ws + W.fus, ws + W.stc, hw_cxr, hb_cxr, hw_enh, hb_enh, fused, x_cxr, x_enh);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

int mfvit_fusion_backward(const mfvit_fusion_cfg* cfg, const float* params, const float* f_cxr, const float* f_enh, const float* hw_cxr,
                          const float* hw_enh, void* workspace, const float* dfused, const float* dx_cxr, const float* dx_enh,
                          float* dparams, float* df_cxr, float* df_enh, float* dhw_cxr, float* dhb_cxr, float* dhw_enh, float* dhb_enh,
                          mfvit_stream_t stream) {
    if (!fus_ok(cfg) || !params || !f_cxr || !f_enh || !workspace || !dparams) return MFVIT_EINVAL;
    if ((df_cxr == nullptr) != (df_enh == nullptr)) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int B = cfg->batch, T = cfg->tokens, C = cfg->num_classes;
    const FusLayout L = fus_layout(C);
    const FusWs W = fus_ws(B, T, cfg->num_classes);
    float* ws = (float*)workspace;
    MFVIT_LAUNCH(x_finish_bwd_kernel, dim3(B), dim3(128), 0, st, f_cxr, f_enh, params, L, B, T, C, ws + W.outp, ws + W.fus,
                       ws + W.stc, hw_cxr, hw_enh, dfused, dx_cxr, dx_enh, dparams, dhw_cxr, dhb_cxr, dhw_enh, dhb_enh, ws + W.dout,
                       ws + W.dqp, ws + W.pa, ws + W.pb, ws + W.pc);
    MFVIT_CHECK_LAUNCH();
    FUS_TRY(xattn_core_backward(2, L, W, params, f_cxr, f_enh, ws, B, T, dparams, df_cxr, df_enh, st));
    // the batch sums of the small gradients, every one in a fixed order (one launch)
    ColpartBatch batch;
    ColpartBatch* prev = colpart_batch_begin(&batch);
    int rc = MFVIT_OK;
    for (int dir = 0; dir < 2 && rc == MFVIT_OK; ++dir) {
        float* dhw = dparams + L.head[dir];
        float* dg = dparams + L.post[dir];
        rc = colpart_reduce(ws + W.pa + (long)dir * B * 3 * D, B, D, 3, dg, dg + D, dparams + L.ca[dir] + L.bp, st);
        if (rc == MFVIT_OK) rc = colpart_reduce(ws + W.pb + (long)dir * B * 2 * C * D, B, C * D, 2, dhw, dir == 0 ? dhw_cxr : dhw_enh, nullptr, st);
        if (rc == MFVIT_OK) rc = colpart_reduce(ws + W.pc + (long)dir * B * 2 * C, B, C, 2, dhw + (long)C * D, dir == 0 ? dhb_cxr : dhb_enh, nullptr, st);
        if (rc == MFVIT_OK) rc = fus_prenorm_job(dir, L, W, ws, B, dparams, st);
    }
    if (rc == MFVIT_OK) rc = colpart_batch_flush(st);
    colpart_batch_begin(prev);
    return rc;
}

}  // extern "C"

// d pre-norm weight | bias of direction `dir` += the sum over its 2 B partial rows (x_stream_bwd: token rows 1 .., x_row0_bwd: the cls row), fixed order
static int fus_prenorm_job(int dir, const FusLayout& L, const FusWs& W, float* ws, int B, float* dparams, hipStream_t st) {
    if (L.no_norm) return MFVIT_OK;
    return colpart_reduce(ws + W.pd + (long)dir * 2 * B * 2 * D, 2 * B, D, 2, dparams + L.ca[dir] + L.n_w, dparams + L.ca[dir] + L.n_b, nullptr, st);
}

// backward of xattn_core_forward: consumes ws.dout (= d outp) and ws.dqp (cls gradient that bypasses the attention)
static int xattn_core_backward(int ndir, const FusLayout& L, const FusWs& W, const float* params, const float* f_cxr, const float* f_enh,
                               float* ws, int B, int T, float* dparams, float* df_cxr, float* df_enh, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)DH);
    const float* wT = ws + W.wT;
    {   // dWp += dout^T o
        GemmP p = zg();
        p.A = ws + W.dout; p.lda = D; p.W = ws + W.o; p.ldw = D; p.M = B; p.N = D; p.K = D;
        p.out0 = dparams + L.ca[0] + L.wp; p.ldo0 = D;
        p.nb = ndir; p.nbi = 1; p.sAo = (long)B * D; p.sWo = (long)B * D; p.sOo = L.ca_stride;
        FUS_TRY(gemm_tn(MFVIT_F32, p, st));
    }
    {   // do = dout Wp   (W operand = WpT)
        GemmP p = zg();
        p.A = ws + W.dout; p.lda = D; p.W = wT + 3L * D * D; p.ldw = D; p.M = B; p.N = D; p.K = D;
        p.out0 = ws + W.dob; p.ldo0 = D;
        p.nb = ndir; p.nbi = 1; p.sAo = (long)B * D; p.sWo = 4L * D * D; p.sOo = (long)B * D;
        FUS_TRY(gemm_nt_tile(MFVIT_F32, EPI_NONE, p, st));
    }
    {   // dWv_h += do_h^T u_h
        GemmP p = zg();
        p.A = ws + W.dob; p.lda = D; p.W = ws + W.u; p.ldw = NH * D; p.M = B; p.N = DH; p.K = D;
        p.out0 = dparams + L.ca[0] + L.wv; p.ldo0 = D;
        p.nb = ndir * NH; p.nbi = NH; p.sAo = (long)B * D; p.sAi = DH; p.sWo = (long)B * NH * D; p.sWi = D; p.sOo = L.ca_stride; p.sOi = (long)DH * D;
        FUS_TRY(gemm_tn(MFVIT_F32, p, st));
    }
    {   // du_h = do_h Wv_h   (W operand = WvT[:, h*128 ..])
        GemmP p = zg();
        p.A = ws + W.dob; p.lda = D; p.W = wT + 2L * D * D; p.ldw = D; p.M = B; p.N = D; p.K = DH;
        p.out0 = ws + W.du; p.ldo0 = NH * D;
        p.nb = ndir * NH; p.nbi = NH; p.sAo = (long)B * D; p.sAi = DH; p.sWo = 4L * D * D; p.sWi = DH; p.sOo = (long)B * NH * D; p.sOi = D;
        FUS_TRY(gemm_nt_tile(MFVIT_F32, EPI_NONE, p, st));
    }
    {
        const size_t lds = (size_t)(8 * T + XW_BWD * 5 * D) * 4;
        static PerDeviceOnce attr;
        if (attr.first()) { (void)hipFuncSetAttribute((const void*)x_stream_bwd_kernel<XW_BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        ProfScope ps(PROF_XATTN_BWD, 0, 2.0 * B * T * D * 4 * 3, st);
        MFVIT_LAUNCH(x_stream_bwd_kernel<XW_BWD>, dim3(B, ndir), dim3(XW_BWD * 64), lds, st, f_cxr, f_enh, params, L, scale, B, T, ws + W.kq, ws + W.a,
                           ws + W.st, ws + W.du, ws + W.dkq, ws + W.dz0p, ws + W.pd, df_cxr, df_enh);
        MFVIT_CHECK_LAUNCH();
    }
    {   // dWk_h += qv_h^T dkq_h
        GemmP p = zg();
        p.A = ws + W.qv; p.lda = D; p.W = ws + W.dkq; p.ldw = NH * D; p.M = B; p.N = DH; p.K = D;
        p.out0 = dparams + L.ca[0] + L.wk; p.ldo0 = D;
        p.nb = ndir * NH; p.nbi = NH; p.sAo = (long)B * D; p.sAi = DH; p.sWo = (long)B * NH * D; p.sWi = D; p.sOo = L.ca_stride; p.sOi = (long)DH * D;
        FUS_TRY(gemm_tn(MFVIT_F32, p, st));
    }
    {   // dqv_h = dkq_h Wk_h^T
        GemmP p = zg();
        p.A = ws + W.dkq; p.lda = NH * D; p.W = params + L.ca[0] + L.wk; p.ldw = D; p.M = B; p.N = DH; p.K = D;
        p.out0 = ws + W.dqv; p.ldo0 = D;
        p.nb = ndir * NH; p.nbi = NH; p.sAo = (long)B * NH * D; p.sAi = D; p.sWo = L.ca_stride; p.sWi = (long)DH * D; p.sOo = (long)B * D; p.sOi = DH;
        FUS_TRY(gemm_nt_tile(MFVIT_F32, EPI_NONE, p, st));
    }
    {   // dWq += dqv^T z0
        GemmP p = zg();
        p.A = ws + W.dqv; p.lda = D; p.W = ws + W.z0; p.ldw = D; p.M = B; p.N = D; p.K = D;
        p.out0 = dparams + L.ca[0] + L.wq; p.ldo0 = D;
        p.nb = ndir; p.nbi = 1; p.sAo = (long)B * D; p.sWo = (long)B * D; p.sOo = L.ca_stride;
        FUS_TRY(gemm_tn(MFVIT_F32, p, st));
    }
    {   // dz0q = dqv Wq   (W operand = WqT)
        GemmP p = zg();
        p.A = ws + W.dqv; p.lda = D; p.W = wT; p.ldw = D; p.M = B; p.N = D; p.K = D;
        p.out0 = ws + W.dz0q; p.ldo0 = D;
        p.nb = ndir; p.nbi = 1; p.sAo = (long)B * D; p.sWo = 4L * D * D; p.sOo = (long)B * D;
        FUS_TRY(gemm_nt_tile(MFVIT_F32, EPI_NONE, p, st));
    }
    MFVIT_LAUNCH(x_row0_bwd_kernel, dim3(B), dim3(64 * ndir), 0, st, f_cxr, f_enh, params, L, B, T, ws + W.st0, ws + W.dz0p, ws + W.dz0q,
                       ws + W.dqp, ws + W.pd, df_cxr, df_enh);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

// ------------------------------------------------------------------------------------------------ stand-alone PreNorm(CrossAttention)
__global__ void x_copy_out_kernel(const float* __restrict__ src, float* __restrict__ dst, long n) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

extern "C" {

// layout of ONE cross-attention block as the parameter arena: norm.{w,b}, wq, wk, wv, proj.{w,b}; bare: the block starts at wq
// (L.ca[0] = -2 D puts wq at offset 0; the norm offsets are never dereferenced under no_norm)
static FusLayout one_block_layout(const mfvit_fusion_cfg* cfg, bool bare) {
    FusLayout L = fus_layout(cfg->num_classes);
    L.ca[0] = bare ? -2L * D : 0;
    L.no_norm = bare ? 1 : 0;
    return L;
}
static int xattn_one_forward(const mfvit_fusion_cfg* cfg, bool bare, const float* params, const float* x_own, const float* x_oth,
                             void* workspace, float* out, mfvit_stream_t stream) {
    if (!fus_ok(cfg) || !params || !x_own || !x_oth || !workspace || !out) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int B = cfg->batch, T = cfg->tokens;
    const FusLayout L = one_block_layout(cfg, bare);
    const FusWs W = fus_ws(B, T, cfg->num_classes);
    float* ws = (float*)workspace;
    FUS_TRY(xattn_core_forward(1, L, W, params, x_own, x_oth, ws, B, T, cfg->eps_pre, st));
    MFVIT_LAUNCH(x_copy_out_kernel, dim3((unsigned)(((long)B * D + 255) / 256)), dim3(256), 0, st, ws + W.outp, out, (long)B * D);
    MFVIT_CHECK_LAUNCH();
    return MFVIT_OK;
}

static int xattn_one_backward(const mfvit_fusion_cfg* cfg, bool bare, const float* params, const float* x_own, const float* x_oth,
                              void* workspace, const float* dout, float* dparams, float* dx_own, float* dx_oth, mfvit_stream_t stream) {
    if (!fus_ok(cfg) || !params || !x_own || !x_oth || !workspace || !dout || !dparams) return MFVIT_EINVAL;
    if ((dx_own == nullptr) != (dx_oth == nullptr)) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int B = cfg->batch, T = cfg->tokens;
    const FusLayout L = one_block_layout(cfg, bare);
    const FusWs W = fus_ws(B, T, cfg->num_classes);
    float* ws = (float*)workspace;
    MFVIT_LAUNCH(x_copy_out_kernel, dim3((unsigned)(((long)B * D + 255) / 256)), dim3(256), 0, st, dout, ws + W.dout, (long)B * D);
    MFVIT_CHECK_LAUNCH();
    if (hipMemsetAsync(ws + W.dqp, 0, sizeof(float) * B * D, st) != hipSuccess) return MFVIT_ELAUNCH;
    // d proj.bias = column sums of dout (x_finish_bwd does this in the fused model)
    FUS_TRY(colsum_rows(dout, D, dparams + L.ca[0] + L.bp, B, 1, 0, D, st));
    FUS_TRY(xattn_core_backward(1, L, W, params, x_own, x_oth, ws, B, T, dparams, dx_own, dx_oth, st));
    return fus_prenorm_job(0, L, W, ws, B, dparams, st);
}

int mfvit_prenorm_xattn_forward(const mfvit_fusion_cfg* cfg, const float* params, const float* x_own, const float* x_oth,
                                void* workspace, float* out, mfvit_stream_t stream) {
    return xattn_one_forward(cfg, false, params, x_own, x_oth, workspace, out, stream);
}
int mfvit_prenorm_xattn_backward(const mfvit_fusion_cfg* cfg, const float* params, const float* x_own, const float* x_oth,
                                 void* workspace, const float* dout, float* dparams, float* dx_own, float* dx_oth,
                                 mfvit_stream_t stream) {
    return xattn_one_backward(cfg, false, params, x_own, x_oth, workspace, dout, dparams, dx_own, dx_oth, stream);
}
int mfvit_xattn_forward(const mfvit_fusion_cfg* cfg, const float* params, const float* x, void* workspace, float* out, mfvit_stream_t stream) {
    return xattn_one_forward(cfg, true, params, x, x, workspace, out, stream);
}
int mfvit_xattn_backward(const mfvit_fusion_cfg* cfg, const float* params, const float* x, void* workspace, const float* dout,
                         float* dparams, float* dx, mfvit_stream_t stream) {
    return xattn_one_backward(cfg, true, params, x, x, workspace, dout, dparams, dx, dx, stream);
}

}  // extern "C"
