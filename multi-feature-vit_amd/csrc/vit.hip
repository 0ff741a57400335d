// Composite ViT-S/16 encoder forward / backward: host-side launch sequences over the gfx950 kernels, plus the
// C ABI entry points of include/mfvit.h for the encoder and the single ops.  No allocation, no synchronisation:
// every launch goes to the caller's stream, all memory (parameters, shadows, workspace) is caller-owned.
#include "../../include/mfvit.h"
#include "kernels.h"
#include "prof.h"

#include <atomic>
#include <stdlib.h>
#include <string.h>

using namespace mfvit;

namespace mfvit {
static std::atomic<int> g_stream_share{1};
// the hint of the encoder call this thread is inside of (mfvit_vit_cfg::stream_share), or 0: the process-wide default
static thread_local int t_stream_share = 0;
int stream_share() { return t_stream_share > 0 ? t_stream_share : g_stream_share.load(std::memory_order_relaxed); }
namespace {
struct ShareScope {
    int prev;
    explicit ShareScope(const mfvit_vit_cfg* cfg) : prev(t_stream_share) { if (cfg && cfg->stream_share > 0) t_stream_share = cfg->stream_share > 8 ? 8 : cfg->stream_share; }
    ~ShareScope() { t_stream_share = prev; }
};
}  // namespace
}  // namespace mfvit

namespace {

// Round 4: the fc1 bias gradient comes from the accumulators of the fc2-dgrad tile epilogue (float atomics on 1536 addresses)
// instead of the ones-fragment MFMAs of the fc1 weight gradient - those cost that launch 8 % (98.9 vs 91.5 us for the same flops without them: a third
// more MFMAs on half the waves of a third of its workgroups).  Same box, 0 / 1 / 0 / 1: weight-gradient class 97.8 / 94.3 / 97.4 / 94.8 us per launch, tile
// class unchanged (93.2 / 93.3 / 93.4 / 93.6), step 27.23 / 27.16 / 27.26 / 27.19 ms.
constexpr bool fc1b_in_tile() { return true; }

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Dims {
    int dtype, B, T, np, D, depth, H, HD, F, M, Mp, es, ep;   // es: bytes per LOGICAL element of a dtype tensor; ep: storage elements per logical one
    bool save, tok, use_pos;
    int act;
    float p_embd, p_attn, p_resid;      // token-input mode: dropout sites (0 = off)
    unsigned long long seed;
    bool drop;                          // any site active
    bool unfused;                       // residual + LayerNorm (and its backward) as row passes behind plain tile GEMMs instead of the row-complete GEMM kernels
};

bool get_dims(const mfvit_vit_cfg* c, Dims& d) {
    if (!c) return false;
    if (c->dtype != MFVIT_F32 && c->dtype != MFVIT_BF16 && c->dtype != MFVIT_BF16X3 && c->dtype != MFVIT_F16) return false;
    d.tok = c->token_input != 0;
    if (c->batch <= 0) return false;
    if (d.tok ? c->tokens <= 0 : (c->img_h <= 0 || c->img_w <= 0 || c->img_h % 16 || c->img_w % 16)) return false;
    if (c->act != 0 && c->act != 1) return false;
    // embed dim: 384 (vit_small: the row-complete GEMM kernels with their fused LayerNorm epilogues) or 768 (vit_base: plain tile GEMMs + row passes)
    if ((c->dim != 384 && c->dim != 768) || c->depth <= 0 || c->heads <= 0 || c->dim % c->heads) return false;
    if (c->mlp_dim % 128 || c->mlp_dim <= 0) return false;
    d.dtype = c->dtype;
    d.B = c->batch;
    d.np = d.tok ? c->tokens : (c->img_h / 16) * (c->img_w / 16);
    d.T = d.tok ? c->tokens : d.np + 1;
    d.use_pos = c->use_pos != 0;
    d.act = c->act;
    d.D = c->dim;
    d.depth = c->depth;
    d.H = c->heads;
    d.HD = c->dim / c->heads;
    d.F = c->mlp_dim;
    d.M = d.B * d.T;
    d.Mp = d.B * d.np;
    d.es = (c->dtype == MFVIT_BF16 || c->dtype == MFVIT_F16) ? 2 : 4;   // split bf16: hi + lo = 4 bytes
    d.ep = c->dtype == MFVIT_BF16X3 ? 2 : 1;
    d.save = c->save_for_backward != 0;
    d.p_embd = d.tok ? c->p_embd : 0.f;
    d.p_attn = d.tok ? c->p_attn : 0.f;
    d.p_resid = d.tok ? c->p_resid : 0.f;
    for (float pp : {d.p_embd, d.p_attn, d.p_resid})
        if (!(pp >= 0.f && pp < 1.f)) return false;
    d.drop = d.p_embd > 0.f || d.p_attn > 0.f || d.p_resid > 0.f;
    if (d.drop && c->dtype == MFVIT_F32) return false;               // the dropout stages exist for the 16-bit operand types
    d.seed = ((unsigned long long)c->seed_hi << 32) | c->seed_lo;
    if (d.HD != 32 && d.HD != 64 && d.HD != 96) return false;
    if (d.HD == 96 && c->dtype == MFVIT_F32) return false;      // head_dim 96 runs on the streaming MFMA kernels (16-bit types, split bf16)
    if (c->dtype == MFVIT_BF16X3 && d.HD % 32) return false;    // a head's row piece is whole [hi x 32 | lo x 32] groups
    // MFVIT_UNFUSED_ROWS=1 forces the unfused path at dim 384 too (tests: the same encoder through both paths; tools: A/B at small M)
    // (read on every call, not cached: the workspace layout depends on it, and a test flips it between two encoders of one process)
    const char* ue = getenv("MFVIT_UNFUSED_ROWS");
    d.unfused = d.D != 384 || (ue && ue[0] == '1');
    return true;
}

// ------------------------------------------------------------------ parameter arena layout (floats)
struct ParamLayout {
    long cls, pos, pe_w, pe_b, blk0, blk_stride, norm_w, norm_b, total;
    // offsets inside a block
    long ln1_w, ln1_b, qkv_w, qkv_b, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b;
};
ParamLayout param_layout(const Dims& d) {
    ParamLayout L;
    const long D = d.D, F = d.F;
    long o = 0;
    L.cls = o; o += d.tok ? 0 : D;                   // token-input mode: no cls token, no patch embedding
    L.pos = o; o += (long)d.T * D;
    L.pe_w = o; o += d.tok ? 0 : D * 768;
    L.pe_b = o; o += d.tok ? 0 : D;
    L.blk0 = o;
    long b = 0;
    L.ln1_w = b; b += D;
    L.ln1_b = b; b += D;
    L.qkv_w = b; b += 3 * D * D;
    L.qkv_b = b; b += 3 * D;
    L.proj_w = b; b += D * D;
    L.proj_b = b; b += D;
    L.ln2_w = b; b += D;
    L.ln2_b = b; b += D;
    L.fc1_w = b; b += F * D;
    L.fc1_b = b; b += F;
    L.fc2_w = b; b += D * F;
    L.fc2_b = b; b += D;
    L.blk_stride = b;
    o += b * d.depth;
    L.norm_w = o; o += D;
    L.norm_b = o; o += D;
    L.total = o;
    return L;
}

// ------------------------------------------------------------------ weight shadow layout (bytes)
struct ShadowLayout {
    size_t pe_w;                       // [D][768] dtype (bf16 only; f32 reads the arena)
    size_t blk0, blk_stride;           // per block:
    size_t qkv_w, qkv_t, proj_w, proj_t, fc1_w, fc1_t, fc2_w, fc2_t;
    size_t total;
};
ShadowLayout shadow_layout(const Dims& d) {
    ShadowLayout S;
    const size_t D = d.D, F = d.F, es = d.es;
    const bool hw = d.dtype != MFVIT_F32;  // keep straight copies only when a cast is needed
    size_t o = 0;
    S.pe_w = o; o += (hw && !d.tok) ? align256(D * 768 * es) : 0;
    S.blk0 = o;
    size_t b = 0;
    S.qkv_w = b; b += hw ? align256(3 * D * D * es) : 0;
    S.qkv_t = b; b += align256(3 * D * D * es);
    S.proj_w = b; b += hw ? align256(D * D * es) : 0;
    S.proj_t = b; b += align256(D * D * es);
    S.fc1_w = b; b += hw ? align256(F * D * es) : 0;
    S.fc1_t = b; b += align256(F * D * es);
    S.fc2_w = b; b += hw ? align256(D * F * es) : 0;
    S.fc2_t = b; b += align256(D * F * es);
    S.blk_stride = b;
    o += b * d.depth;
    S.total = o;
    return S;
}

// ------------------------------------------------------------------ workspace layout (bytes)
constexpr int TN_SLOTS = 5;
struct WsLayout {
    size_t patches;
    size_t x0, x_stride;        // f32 [M][D], depth+1 (save) or 1 copies
    size_t st0, st_stride;      // LN1 / final-norm stats: mean [M] then rstd [M], depth+1 or 1 copies
    size_t blk0, blk_stride;    // per block (depth or 1 copies):
    size_t y1, qkv, attn, lse, xmid, st2, y2, hpre, hact;
    // backward scratch
    size_t gx, gmid, gxT, gmidT, dhpre, dattn, dqkv, colscratch, colpart, colpart_stride = 0, tnpart, tnpart_stride = 0;
    size_t dtmp;                // token-input mode with residual dropout: [M][D] of the operand type (branch output before the dropout; masked dY)
    size_t kpart;               // gemm_rowp with K splits (small M): partial accumulator tiles, 264 workgroups x 7 x 12 x 512 floats (gemm_rowp.hip)
    size_t utmp;                // unfused path: [M][D] of the operand type (output of the plain tile GEMM in front of a LayerNorm / LayerNorm-backward row pass)
    size_t domax;               // backward: the largest |d attn| of every (image, head) of every block, f32 bits [depth][B][H] (proj data gradient -> attention backward)
    size_t pp_stride;           // distance between the two ping-pong copies of gxT / gmidT / dhpre / dqkv (0 = none)
    size_t total;
};
WsLayout ws_layout(const Dims& d) {
    WsLayout W;
    const size_t M = d.M, D = d.D, F = d.F, es = d.es;
    const size_t nl = d.save ? d.depth : 1;
    size_t o = 0;
    W.patches = o; o += d.tok ? 0 : align256((size_t)d.Mp * 768 * es);
    W.x0 = o; W.x_stride = align256(M * D * 4); o += W.x_stride * (d.save ? d.depth + 1 : 1);
    W.st0 = o; W.st_stride = align256(2 * M * 4); o += W.st_stride * (d.save ? d.depth + 1 : 1);
    W.blk0 = o;
    size_t b = 0;
    W.y1 = b; b += align256(M * D * es);
    W.qkv = b; b += align256(M * 3 * D * es);
    W.attn = b; b += align256(M * D * es);
    W.lse = b; b += align256((size_t)d.B * d.H * d.T * 4);
    W.xmid = b; b += align256(M * D * 4);
    W.st2 = b; b += align256(2 * M * 4);
    W.y2 = b; b += align256(M * D * es);
    W.hpre = b; b += d.save ? align256(M * F * (d.dtype == MFVIT_BF16X3 ? 2 : es)) : 0;   // act'(pre): plain fp16 for split tensors; not kept by no-grad forwards
    W.hact = b; b += align256(M * F * es);
    W.blk_stride = b;
    o += b * nl;
    if (d.save) {
        W.gx = o; o += align256(M * D * 4);
        W.gmid = o; o += align256(M * D * 4);
        // dY buffers read by the weight-gradient kernels on the side stream: two copies, used by layer parity
        W.gxT = o; o += align256(M * D * es);
        W.gmidT = o; o += align256(M * D * es);
        W.dhpre = o; o += align256(M * F * es);
        W.dqkv = o; o += align256(M * 3 * D * es);
        W.pp_stride = o - W.gxT;
        o += W.pp_stride;
        W.dattn = o; o += align256(M * D * es);
        W.colscratch = o; o += align256(2 * D * 4);
        {   // per-workgroup column-sum partials: max over the kernels that use them
            // row kernels: one partial row set [3][D] per workgroup - gemm_rowp launches at most one tile per CU and round (<= max(#CUs, M / 16) tiles),
            // gemm_nt_row full 64-row tiles (the former M / 64 + 256 bound belonged to the removed balanced-rows option)
            const size_t tiles_row = (M + 63) / 64 > 256 ? (M + 63) / 64 : 256;
            size_t a = tiles_row * 3 * D, b2 = 4 * ((M + 191) / 192) * F, c2 = 256 * 3 * D;      // (b2: the x act' tile epilogue - one row of F sums per wave-row block of a tile; tiles are 256 or 192 rows)
            size_t n = a > b2 ? a : b2;
            n = n > c2 ? n : c2;
            W.colpart = o; o += (3 * nl + 2) * align256(n * 4);          // one partial buffer per launch of a backward call that leaves column sums (batched reduce)
            W.colpart_stride = align256(n * 4);
        }
        // split partials of the weight-gradient GEMMs: one slot per launch of a BLOCK (at most 4) + 1 for the patch embedding, each tiles * splits <= 384
        // blocks of 128 x 128 floats; reduced by one batched launch per block, in a fixed order (deterministic sums, round 5), then reused
        W.tnpart_stride = align256((size_t)384 * 128 * 128 * 4);
        W.tnpart = o; o += (size_t)TN_SLOTS * W.tnpart_stride;
    } else {
        W.gx = W.gmid = W.gxT = W.gmidT = W.dhpre = W.dattn = W.dqkv = W.colscratch = W.colpart = W.tnpart = o;
        W.pp_stride = 0;
    }
    W.dtmp = o; o += (d.tok && d.p_resid > 0.f) ? align256(M * D * es) : 0;
    W.kpart = o; o += d.D == 384 ? align256((size_t)264 * 7 * 12 * 512 * 4) : 0;
    W.utmp = o; o += align256(M * D * es);     // always there (1 / 200 of the workspace): the layout does not depend on which path a call takes
    W.domax = o; o += d.save ? align256((size_t)d.depth * d.B * d.H * 4) : 0;
    W.total = o;
    return W;
}

GemmP zero_gemm() {
    GemmP p;
    memset(&p, 0, sizeof(p));
    return p;
}

// Weight-gradient GEMMs are leaves of the backward graph: they CAN run on a second (library-owned, lazily created) HIP stream beside the
// dgrad chain, ordered by events (MFVIT_WGRAD_STREAM=1).  Default since round 3: the caller's stream.  The side stream was worth 3 - 4 % while the
// weight-gradient and row kernels left half of every CU's LDS and issue slots free (round 2); now they are one-workgroup-per-CU kernels that own
// the chip while they run, nothing co-resides with them, and the second queue only adds event waits and a worse launch order: measured on one box,
// three alternating repeats, 31.25 / 31.33 / 31.36 ms per step with the side stream, 30.81 / 30.80 / 30.86 without (profiles/r03_streams_ab.txt).
struct SideStream {
    hipStream_t owner = nullptr;    // caller stream this side stream is paired with
    hipStream_t s = nullptr;
    hipEvent_t in = nullptr, end = nullptr;
    hipEvent_t done[64] = {};
    bool ok = false;
};
std::atomic<int> g_wgrad_stream{1};   // runtime switch (mfvit_set_wgrad_stream): attribution passes serialise the backward
SideStream& side_stream(hipStream_t caller) {
    // one side stream per (thread, caller stream): two encoders running on two caller streams do not share a side queue
    static thread_local SideStream ss[8];
    static thread_local int used = 0;
    SideStream* px = nullptr;
    for (int i = 0; i < used; ++i)
        if (ss[i].owner == caller) px = &ss[i];
    if (!px) {
        px = &ss[used < 8 ? used++ : 7];
        if (px->owner != caller && px->s) return *px;   // table full: share the last one
        px->owner = caller;
    }
    SideStream& x = *px;
    if (!x.s) {
        static const bool enabled = [] { const char* e = getenv("MFVIT_WGRAD_STREAM"); return !(e && e[0] == '0'); }();   // (0: never; created on first use)
        if (enabled && hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&x.in, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&x.end, hipEventDisableTiming) == hipSuccess) {
            x.ok = true;
            for (auto& e : x.done) x.ok = x.ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        }
        if (!x.ok && !x.s) x.s = (hipStream_t)-1;   // do not retry
    }
    return x;
}

#define MFVIT_TRY(expr)            \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != MFVIT_OK) return rc__; \
    } while (0)

}  // namespace

extern "C" {

int mfvit_abi_version(void) { return 5; }
int mfvit_set_stream_share(int n) {
    mfvit::g_stream_share.store(n < 1 ? 1 : (n > 8 ? 8 : n), std::memory_order_relaxed);
    return MFVIT_OK;
}
int mfvit_set_wgrad_stream(int enabled) {
    g_wgrad_stream.store(enabled ? 1 : 0, std::memory_order_relaxed);
    return MFVIT_OK;
}
const char* mfvit_build_info(void) { return "libmfvit_hip gfx950 (MFMA bf16 | split-bf16 x3 | f16 32x32x16, f32 32x32x2), wave64, abi 4"; }

size_t mfvit_vit_param_count(const mfvit_vit_cfg* cfg) {
    Dims d;
    if (!get_dims(cfg, d)) return 0;
    return (size_t)param_layout(d).total;
}
int mfvit_vit_param_layout(const mfvit_vit_cfg* cfg, int64_t out[9]) {
    Dims d;
    if (!get_dims(cfg, d) || !out) return MFVIT_EINVAL;
    const ParamLayout L = param_layout(d);
    out[0] = L.cls; out[1] = L.pos; out[2] = L.pe_w; out[3] = L.pe_b; out[4] = L.blk0; out[5] = L.blk_stride;
    out[6] = L.norm_w; out[7] = L.norm_b; out[8] = L.total;
    return MFVIT_OK;
}
size_t mfvit_vit_shadow_bytes(const mfvit_vit_cfg* cfg) {
    Dims d;
    if (!get_dims(cfg, d)) return 0;
    return shadow_layout(d).total;
}
size_t mfvit_vit_workspace_bytes(const mfvit_vit_cfg* cfg) {
    Dims d;
    if (!get_dims(cfg, d)) return 0;
    return ws_layout(d).total;
}

int mfvit_vit_prepare_shadow(const mfvit_vit_cfg* cfg, const float* params, void* shadow, mfvit_stream_t stream) {
    Dims d;
    if (!get_dims(cfg, d) || !params || !shadow) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const ParamLayout L = param_layout(d);
    const ShadowLayout S = shadow_layout(d);
    char* sh = (char*)shadow;
    const bool hw = d.dtype != MFVIT_F32;
    if (hw && !d.tok) MFVIT_TRY(cast_transpose(d.dtype, params + L.pe_w, sh + S.pe_w, nullptr, d.D, 768, st));
    // one launch per weight type covers all `depth` blocks (identical shapes at fixed arena / shadow strides)
    const float* pb = params + L.blk0;
    char* sb = sh + S.blk0;
    const long ss = L.blk_stride, sd = (long)S.blk_stride;
    MFVIT_TRY(cast_transpose_batched(d.dtype, pb + L.qkv_w, hw ? sb + S.qkv_w : nullptr, sb + S.qkv_t, 3 * d.D, d.D, d.depth, ss, sd, sd, st));
    MFVIT_TRY(cast_transpose_batched(d.dtype, pb + L.proj_w, hw ? sb + S.proj_w : nullptr, sb + S.proj_t, d.D, d.D, d.depth, ss, sd, sd, st));
    MFVIT_TRY(cast_transpose_batched(d.dtype, pb + L.fc1_w, hw ? sb + S.fc1_w : nullptr, sb + S.fc1_t, d.F, d.D, d.depth, ss, sd, sd, st));
    MFVIT_TRY(cast_transpose_batched(d.dtype, pb + L.fc2_w, hw ? sb + S.fc2_w : nullptr, sb + S.fc2_t, d.D, d.F, d.depth, ss, sd, sd, st));
    return MFVIT_OK;
}

}  // extern "C"

// Shared forward of the two front ends: ViT-S/16 (img = (B,3,H,W) image, patch embedding + cls token) and the token-input GPT
// (img = (B,T,dim) tokens, + pos_emb).  Everything behind x_0 / LN1_0 is the same pre-LN block stack.
static int encoder_forward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, const float* img, void* workspace,
                           float* features, mfvit_stream_t stream, bool want_tokens) {
    Dims d;
    if (!get_dims(cfg, d) || !params || !shadow || !img || !workspace || !features) return MFVIT_EINVAL;
    if (d.tok != want_tokens) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const ParamLayout L = param_layout(d);
    const ShadowLayout S = shadow_layout(d);
    const WsLayout W = ws_layout(d);
    char* ws = (char*)workspace;
    const char* sh = (const char*)shadow;
    const bool hw = d.dtype != MFVIT_F32;
    const long D = d.D, F = d.F;
    const long e = d.ep;   // leading dimensions of dtype tensors are in storage elements
    const float eps = cfg->ln_eps;
    auto xbuf = [&](int l) { return (float*)(ws + W.x0 + (d.save ? (size_t)l : 0) * W.x_stride); };
    auto stat = [&](int l) { return (float*)(ws + W.st0 + (d.save ? (size_t)l : 0) * W.st_stride); };
    auto blk = [&](int l) { return ws + W.blk0 + (d.save ? (size_t)l : 0) * W.blk_stride; };
    auto pblk = [&](int l) { return params + L.blk0 + (long)l * L.blk_stride; };
    auto sblk = [&](int l) { return sh + S.blk0 + (size_t)l * S.blk_stride; };

    // dropout sites of the token-input (GPT) mode, see mfvit_vit_cfg
    auto site = [](int l, int which) { return 16u * (unsigned)l + (unsigned)which; };
    if (d.tok && d.p_embd > 0.f) {
        // x_0 = drop(tokens + pos_emb) (fuseattention.py:187 `self.drop(self.pos_emb + token_embeddings)`); LN1_0 -> y1_0
        MFVIT_TRY(drop_add_ln_rows(d.dtype, d.D, nullptr, 0, img, D, d.use_pos ? params + L.pos : nullptr, D, d.T, nullptr, 0,
                                   make_drop(d.p_embd, d.seed, 1), xbuf(0), D, blk(0) + W.y1, D * e, 0, pblk(0) + L.ln1_w, pblk(0) + L.ln1_b,
                                   eps, stat(0), stat(0) + d.M, d.M, st));
    } else if (d.tok) {
        // token input (fuseattention.py:186-189): x_0 = tokens (+ pos_emb, shared by the batch); LN1_0 -> y1_0.  One row kernel pass.
        MFVIT_TRY(ln_rows(d.dtype, d.D, img, D, d.use_pos ? params + L.pos : nullptr, D, d.T, xbuf(0), D, blk(0) + W.y1, D * e, 0,
                          pblk(0) + L.ln1_w, pblk(0) + L.ln1_b, eps, stat(0), stat(0) + d.M, d.M, 1, 0, 0, st));
    } else {
    // patch embedding: im2col -> row-complete GEMM (+bias +pos_embed) -> x_0, LN1_0 -> y1_0
    MFVIT_TRY(im2col16(d.dtype, img, ws + W.patches, d.B, cfg->img_h, cfg->img_w, st));
    if (d.unfused) {
        // plain GEMM + bias -> scratch, then one row pass: + pos_embed, the patch rows of image b behind its cls row, LN1_0
        GemmP q = zero_gemm();
        q.A = ws + W.patches; q.lda = 768 * e;
        q.W = hw ? (const void*)(sh + S.pe_w) : (const void*)(params + L.pe_w); q.ldw = 768 * e;
        q.M = d.Mp; q.N = d.D; q.K = 768;
        q.bias = params + L.pe_b;
        q.out0 = ws + W.utmp; q.ldo0 = D * e;
        MFVIT_TRY(gemm_nt_tile(d.dtype, EPI_BIAS, q, st));
        MFVIT_TRY(add_ln_rows(d.dtype, d.D, ws + W.utmp, D * e, params + L.pos, D, d.T, nullptr, 0, d.np, d.T, 1, xbuf(0), D, blk(0) + W.y1, D * e, 0,
                              pblk(0) + L.ln1_w, pblk(0) + L.ln1_b, eps, stat(0), stat(0) + d.M, d.Mp, st));
    } else {
        GemmP p = zero_gemm();
        p.A = ws + W.patches; p.lda = 768 * e;
        p.W = hw ? (const void*)(sh + S.pe_w) : (const void*)(params + L.pe_w); p.ldw = 768 * e;
        p.M = d.Mp; p.N = d.D; p.K = 768;
        p.bias = params + L.pe_b;
        p.res = params + L.pos; p.ldres = D; p.res_mod = d.np; p.res_off = 1;
        p.orow_in = d.np; p.orow_out = d.T; p.orow_off = 1;
        p.out0 = xbuf(0); p.ldo0 = D;
        p.out1 = blk(0) + W.y1; p.ldo1 = D * e;
        p.gamma = pblk(0) + L.ln1_w; p.beta = pblk(0) + L.ln1_b; p.eps = eps;
        p.mean = stat(0); p.rstd = stat(0) + d.M;
        MFVIT_TRY(gemm_nt_row(d.dtype, REPI_RES_LN, p, st));
    }
    // cls rows: x_0[b,0] = cls_token + pos_embed[0]; LN1_0
    MFVIT_TRY(ln_rows(d.dtype, d.D, params + L.cls, 0, params + L.pos, D, 1, xbuf(0), D, blk(0) + W.y1, D * e, 0, pblk(0) + L.ln1_w,
                      pblk(0) + L.ln1_b, eps, stat(0), stat(0) + d.M, d.B, d.T, 0, 1, st));
    }

    for (int l = 0; l < d.depth; ++l) {
        char* b = blk(l);
        const float* pb = pblk(l);
        const char* sb = sblk(l);
        // qkv tensor format: split FP16 for the whole-head attention kernels in bf16x3 mode (attention_mfma.hip), the activation dtype otherwise
        const int qdt = d.p_attn > 0.f ? d.dtype : attn_qkv_dtype(d.dtype, d.T, d.HD);
        {   // qkv = y1 Wqkv^T + b
            GemmP p = zero_gemm();
            p.A = b + W.y1; p.lda = D * e;
            p.W = hw ? (const void*)(sb + S.qkv_w) : (const void*)(pb + L.qkv_w); p.ldw = D * e;
            p.M = d.M; p.N = 3 * d.D; p.K = d.D;
            p.bias = pb + L.qkv_b;
            p.out0 = b + W.qkv; p.ldo0 = 3 * D * e;
            ProfTag tag(PROF_TAG_MHSA_QKV);
            MFVIT_TRY(gemm_nt_tile(d.dtype, qdt == MFVIT_X3F16 ? EPI_BIAS_X3F16 : EPI_BIAS, p, st));
        }
        if (d.p_attn > 0.f) {
            if (!attn_tiled_supported(d.dtype, d.T, d.HD)) return MFVIT_ENOSYS;
            MFVIT_TRY(attn_fwd_tiled_drop(d.dtype, b + W.qkv, b + W.attn, (float*)(b + W.lse), d.B, d.T, d.H, d.HD,
                                          make_drop(d.p_attn, d.seed, site(l, 2)), st));
        } else {
            MFVIT_TRY(attn_fwd(qdt, b + W.qkv, b + W.attn, (float*)(b + W.lse), d.B, d.T, d.H, d.HD, st));
        }
        {   // xmid = x + attn Wproj^T + b ; y2 = LN2(xmid)
            GemmP p = zero_gemm();
            p.A = b + W.attn; p.lda = D * e;
            p.W = hw ? (const void*)(sb + S.proj_w) : (const void*)(pb + L.proj_w); p.ldw = D * e;
            p.M = d.M; p.N = d.D; p.K = d.D;
            p.bias = pb + L.proj_b;
            p.res = xbuf(l); p.ldres = D;
            p.out0 = b + W.xmid; p.ldo0 = D;
            p.out1 = b + W.y2; p.ldo1 = D * e;
            p.gamma = pb + L.ln2_w; p.beta = pb + L.ln2_b; p.eps = eps;
            p.mean = (float*)(b + W.st2); p.rstd = (float*)(b + W.st2) + d.M;
            if (d.p_resid > 0.f) {
                // xmid = x + resid_drop(proj(attn)) (fuseattention.py:57): plain GEMM + bias, then dropout + residual + LN2 in one row pass
                GemmP q = zero_gemm();
                q.A = p.A; q.lda = p.lda; q.W = p.W; q.ldw = p.ldw; q.M = p.M; q.N = p.N; q.K = p.K; q.bias = p.bias;
                q.out0 = ws + W.dtmp; q.ldo0 = D * e;
                MFVIT_TRY(gemm_nt_tile(d.dtype, EPI_BIAS, q, st));
                MFVIT_TRY(drop_add_ln_rows(d.dtype, d.D, ws + W.dtmp, D * e, nullptr, 0, nullptr, 0, 0, xbuf(l), D,
                                           make_drop(d.p_resid, d.seed, site(l, 3)), (float*)(b + W.xmid), D, b + W.y2, D * e, 0, p.gamma, p.beta,
                                           eps, p.mean, p.rstd, d.M, st));
            } else if (d.unfused) {
                // plain GEMM + bias -> scratch, then residual + LN2 as one row pass
                GemmP q = zero_gemm();
                q.A = p.A; q.lda = p.lda; q.W = p.W; q.ldw = p.ldw; q.M = p.M; q.N = p.N; q.K = p.K; q.bias = p.bias;
                q.out0 = ws + W.utmp; q.ldo0 = D * e;
                MFVIT_TRY(gemm_nt_tile(d.dtype, EPI_BIAS, q, st));
                MFVIT_TRY(add_ln_rows(d.dtype, d.D, ws + W.utmp, D * e, nullptr, 0, 0, xbuf(l), D, 0, 0, 0, (float*)(b + W.xmid), D, b + W.y2, D * e, 0,
                                      p.gamma, p.beta, eps, p.mean, p.rstd, d.M, st));
            } else {
                p.kpart = (float*)(ws + W.kpart);
                ProfTag tag(PROF_TAG_MHSA_PROJ);
                MFVIT_TRY(gemm_nt_row(d.dtype, REPI_RES_LN, p, st));
            }
        }
        {   // hpre = y2 W1^T + b1 ; hact = gelu(hpre)
            GemmP p = zero_gemm();
            p.A = b + W.y2; p.lda = D * e;
            p.W = hw ? (const void*)(sb + S.fc1_w) : (const void*)(pb + L.fc1_w); p.ldw = D * e;
            p.M = d.M; p.N = d.F; p.K = d.D;
            p.bias = pb + L.fc1_b;
            p.out0 = d.save ? b + W.hpre : nullptr; p.ldo0 = F;      // act'(pre) in act_grad_type<T> (gemm.hip): F elements per row
            p.out1 = b + W.hact; p.ldo1 = F * e;
            MFVIT_TRY(gemm_nt_tile(d.dtype, d.act == 1 ? EPI_BIAS_RELU : EPI_BIAS_GELU, p, st));
        }
        {   // x_{l+1} = xmid + hact W2^T + b2 ; y = LN(next norm1 | final norm)
            const bool last = l + 1 == d.depth;
            GemmP p = zero_gemm();
            p.A = b + W.hact; p.lda = F * e;
            p.W = hw ? (const void*)(sb + S.fc2_w) : (const void*)(pb + L.fc2_w); p.ldw = F * e;
            p.M = d.M; p.N = d.D; p.K = d.F;
            p.bias = pb + L.fc2_b;
            p.res = (const float*)(b + W.xmid); p.ldres = D;
            p.out0 = xbuf(l + 1); p.ldo0 = D;
            if (last) {
                p.out1 = features; p.ldo1 = D; p.y_f32 = 1;
                p.gamma = params + L.norm_w; p.beta = params + L.norm_b;
            } else {
                p.out1 = blk(l + 1) + W.y1; p.ldo1 = D * e;
                p.gamma = pblk(l + 1) + L.ln1_w; p.beta = pblk(l + 1) + L.ln1_b;
            }
            p.eps = eps;
            p.mean = stat(l + 1); p.rstd = stat(l + 1) + d.M;
            if (d.p_resid > 0.f) {
                // x_{l+1} = xmid + Dropout(fc2(relu(fc1(y2)))) (fuseattention.py:67-72,80)
                GemmP q = zero_gemm();
                q.A = p.A; q.lda = p.lda; q.W = p.W; q.ldw = p.ldw; q.M = p.M; q.N = p.N; q.K = p.K; q.bias = p.bias;
                q.out0 = ws + W.dtmp; q.ldo0 = D * e;
                MFVIT_TRY(gemm_nt_tile(d.dtype, EPI_BIAS, q, st));
                MFVIT_TRY(drop_add_ln_rows(d.dtype, d.D, ws + W.dtmp, D * e, nullptr, 0, nullptr, 0, 0, (const float*)(b + W.xmid), D,
                                           make_drop(d.p_resid, d.seed, site(l, 4)), xbuf(l + 1), D, p.out1, p.ldo1, p.y_f32, p.gamma, p.beta, eps,
                                           p.mean, p.rstd, d.M, st));
            } else if (d.unfused) {
                GemmP q = zero_gemm();
                q.A = p.A; q.lda = p.lda; q.W = p.W; q.ldw = p.ldw; q.M = p.M; q.N = p.N; q.K = p.K; q.bias = p.bias;
                q.out0 = ws + W.utmp; q.ldo0 = D * e;
                MFVIT_TRY(gemm_nt_tile(d.dtype, EPI_BIAS, q, st));
                MFVIT_TRY(add_ln_rows(d.dtype, d.D, ws + W.utmp, D * e, nullptr, 0, 0, (const float*)(b + W.xmid), D, 0, 0, 0, xbuf(l + 1), D, p.out1,
                                      p.ldo1, p.y_f32, p.gamma, p.beta, eps, p.mean, p.rstd, d.M, st));
            } else {
                p.kpart = (float*)(ws + W.kpart);
                MFVIT_TRY(gemm_nt_row(d.dtype, REPI_RES_LN, p, st));
            }
        }
    }
    return MFVIT_OK;
}

// Shared backward; dinput (token-input mode only): d loss / d tokens, written by the embedding stage.
static int encoder_backward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, void* workspace, const float* dfeatures,
                            float* dparams, float* dinput, int stage_hi, int stage_lo, mfvit_stream_t stream, bool want_tokens) {
    Dims d;
    if (!get_dims(cfg, d) || !params || !shadow || !workspace || !dparams) return MFVIT_EINVAL;
    if (!d.save || d.tok != want_tokens) return MFVIT_EINVAL;
    if (stage_hi > d.depth || stage_lo < -1 || stage_lo > stage_hi) return MFVIT_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const ParamLayout L = param_layout(d);
    const ShadowLayout S = shadow_layout(d);
    const WsLayout W = ws_layout(d);
    char* ws = (char*)workspace;
    const char* sh = (const char*)shadow;
    const long D = d.D, F = d.F;
    const long e = d.ep;   // leading dimensions of dtype tensors are in storage elements
    auto xbuf = [&](int l) { return (float*)(ws + W.x0 + (size_t)l * W.x_stride); };
    auto stat = [&](int l) { return (float*)(ws + W.st0 + (size_t)l * W.st_stride); };
    auto blk = [&](int l) { return ws + W.blk0 + (size_t)l * W.blk_stride; };
    auto pblk = [&](int l) { return params + L.blk0 + (long)l * L.blk_stride; };
    auto gblk = [&](int l) { return dparams + L.blk0 + (long)l * L.blk_stride; };
    auto sblk = [&](int l) { return sh + S.blk0 + (size_t)l * S.blk_stride; };
    float* gx = (float*)(ws + W.gx);
    float* gmid = (float*)(ws + W.gmid);
    float* colscr = (float*)(ws + W.colscratch);
    // column-sum partials: every row-kernel launch of this call writes its own buffer, ONE batched launch reduces them (before the embedding
    // stage, which reads a sum, and at the end of the call); with residual dropout (intermediate sums are consumed at once) nothing is deferred
    int colpart_used = 0;
    auto next_colpart = [&]() { return (float*)(ws + W.colpart + (size_t)(colpart_used++ % (3 * d.depth + 2)) * W.colpart_stride); };
    ColpartBatch cbatch;
    struct BatchScope {
        ColpartBatch* prev; bool on;
        BatchScope(ColpartBatch* b, bool on_) : on(on_) { if (on) prev = colpart_batch_begin(b); }
        ~BatchScope() { if (on) colpart_batch_begin(prev); }
    } batch_scope(&cbatch, !(d.p_resid > 0.f));
    // split partials of the weight-gradient GEMMs through scratch - plain stores, every launch into its OWN slot - and ONE batched reduce launch at
    // the end of every BLOCK that adds the splits in a fixed order: dW is the same bits on every run (the float atomics it replaces add in whatever
    // order the workgroups finish).  Round 3 measured the same idea with a reduce launch behind EVERY gradient as a loss (77.2 -> 80.8 us per
    // launch); batched it costs 0.4 - 0.6 % of the step (round 5, profiles/r05_wgrad_partials_ab.txt: weight-gradient class 95.6 -> 90.8 us per launch,
    // the reduce launches take most of it back; per block - partials still in the Infinity Cache - beats per call by 0.07 ms).  MFVIT_TN_PART=0: float
    // atomics.  (With the weight-gradient side stream the partial path stays ON: every weight gradient and every reduce of a call then runs in the side stream's order.)
    static const bool tn_part_env = [] { const char* e = getenv("MFVIT_TN_PART"); return !(e && e[0] == '0'); }();
    int tn_used = 0;
    bool tn_part = tn_part_env;                // (decided below, once use_side is known)
    auto next_tnpart = [&]() -> float* {       // (more launches than slots between two flushes: float atomics for the rest - never a slot still in use)
        return tn_part && tn_used < TN_SLOTS ? (float*)(ws + W.tnpart + (size_t)(tn_used++) * W.tnpart_stride) : nullptr;
    };
    TnPartBatch tbatch;
    struct TnScope {
        TnPartBatch* prev; bool on;
        TnScope(TnPartBatch* b, bool on_) : on(on_) { if (on) prev = tnpart_batch_begin(b); }
        ~TnScope() { if (on) tnpart_batch_begin(prev); }
    } tn_scope(&tbatch, tn_part_env);
    // dY buffers by layer parity (the embed stage counts as layer -1 -> parity 1)
    auto pp = [&](size_t off, int l) { return (void*)(ws + off + (size_t)(l & 1) * W.pp_stride); };
    // (side_wanted, below, decides whether the side stream of this caller stream is looked up - and created - at all)
    static const int side_env = [] { const char* e = getenv("MFVIT_WGRAD_STREAM"); return e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1)) : -1; }();
    // (never inside a caller's stream capture: the library-owned side stream is not part of the caller's capture, and its last weight gradients are joined by the
    // NEXT call's fork, not before this one returns - a capture ending in between would hold unjoined work.  The package itself no longer captures steps:
    // the whole-step HIP graph of round 5 gained nothing, 8.22 vs 8.12 ms, and was removed in round 6)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    const bool side_wanted = (side_env == 1 || (side_env < 0 && d.M >= 1024 && d.M <= 4096)) && g_wgrad_stream.load(std::memory_order_relaxed) != 0 && !(d.p_resid > 0.f) &&
                             !capturing;
    static SideStream no_side;
    SideStream& ss = side_wanted ? side_stream(st) : no_side;
    // (residual dropout: the masked dY copies live in ONE scratch buffer - everything stays on the caller's stream)
    // Default since round 5: the side stream at SMALL M only (1,024 ... 4,096 token rows: 6 - 20 images) - there no kernel fills the chip, the launches of a
    // block are a dependent chain of ~10 us kernels with ~5 us of dispatch gap each, and the weight gradients (leaves) beside the data-gradient chain
    // shorten it: 20 pairs per step 7.31 -> 6.66 ms, 16 pairs 6.31 -> 6.10 (then host-bound), 8 pairs 5.18 -> 4.96; 4 pairs LOSE (host-bound either way, 5.1 ->
    // 6.9 ms) and 32 pairs gain 1 % (profiles/r05_small_batch.txt).  MFVIT_WGRAD_STREAM=1: at every M (round 2 behaviour), 0: never.
    const bool use_side = side_wanted && ss.ok && W.pp_stride != 0;
    hipStream_t wst = use_side ? ss.s : st;                       // stream of the weight-gradient GEMMs
    // (with the side stream every weight gradient AND every reduce of their partials runs on it - the patch embedding's too, below - so the scratch slots
    // are still written and read in one stream's order)
    tn_part = tn_part_env;
    // the side stream may only start after everything already queued on the caller's stream (activations, zeroed gradients)
    auto fork = [&]() -> int {                                    // main -> side dependency at this point of the main stream
        if (!use_side) return MFVIT_OK;
        if (hipEventRecord(ss.in, st) != hipSuccess || hipStreamWaitEvent(ss.s, ss.in, 0) != hipSuccess) return MFVIT_ELAUNCH;
        return MFVIT_OK;
    };
    auto wait_layer = [&](int l) -> int {                         // main waits until the wgrads of layer l have finished
        if (!use_side || l < 0 || l >= d.depth || l > stage_hi) return MFVIT_OK;
        return hipStreamWaitEvent(st, ss.done[l & 63], 0) == hipSuccess ? MFVIT_OK : MFVIT_ELAUNCH;
    };

    // only split bf16 (hi + lo = the f32 value to 2^-17) drops the f32 residual-gradient copies; plain bf16 / fp16 keep them: re-rounding the residual
    // gradient to 8 / 11 mantissa bits at each of the 2 x depth LayerNorm-backward stages departs from the reference's autocast (fp32 residual grads)
    const bool lean_grad = d.dtype == MFVIT_BF16X3 && !d.unfused;   // (the unfused row passes read the f32 residual gradient)
    auto site = [](int l, int which) { return 16u * (unsigned)l + (unsigned)which; };
    const bool rdrop = d.p_resid > 0.f;             // the bias gradients of proj / fc2 then come from the MASKED dY (wgrad column sums)
    // The attention backward's split-fp16 core scales dO (= the proj data gradient) per (image, head) by a power of two from the pair's largest |dO|.  The
    // GEMM that produces dO leaves those maxima behind (GemmP::omax: atomicMax of f32 bits, one slice per block, zeroed here for the blocks of this call), and
    // the attention kernel reads ONE number per pair instead of prefetching the hi parts of all of dO's rows a pair ahead.  MFVIT_DO_MAX=0: the prefetch.
    static int s_domax = INT_MIN;
    const int qdt_bwd = attn_qkv_dtype(d.dtype, d.T, d.HD);
    const bool do_max = env_switch("MFVIT_DO_MAX", 1, s_domax) != 0 && qdt_bwd == MFVIT_X3F16 && (d.HD == 32 || d.HD == 64) && d.T >= 64 && d.D % 128 == 0 && !(d.p_attn > 0.f);
    auto domax = [&](int l) { return (unsigned*)(ws + W.domax) + (size_t)l * d.B * d.H; };
    if (do_max) {
        const int l0 = stage_lo > 0 ? stage_lo : 0, l1 = stage_hi < d.depth - 1 ? stage_hi : d.depth - 1;
        if (l1 >= l0 && hipMemsetAsync(domax(l0), 0, (size_t)(l1 - l0 + 1) * d.B * d.H * 4, st) != hipSuccess) return MFVIT_ELAUNCH;
    }
    for (int s = stage_hi; s >= stage_lo; --s) {
        if (s == d.depth) {
            // final LayerNorm backward: dfeatures -> gx (grad of x_depth); dcol = d fc2_b of the last block
            if (!dfeatures) return MFVIT_EINVAL;
            MFVIT_TRY(ln_bwd_rows(d.dtype, d.D, dfeatures, D, xbuf(d.depth), D, stat(d.depth), stat(d.depth) + d.M, params + L.norm_w,
                                  nullptr, 0, gx, D, pp(W.gxT, d.depth - 1), D * e, dparams + L.norm_w, dparams + L.norm_b,
                                  rdrop ? colscr + D : gblk(d.depth - 1) + L.fc2_b, next_colpart(), d.M, 1, 0, st));
        } else if (s >= 0) {
            const int l = s;
            char* b = blk(l);
            const float* pb = pblk(l);
            float* gb = gblk(l);
            const char* sb = sblk(l);
            void* gxT = pp(W.gxT, l);
            void* gmidT = pp(W.gmidT, l);
            void* dhpre = pp(W.dhpre, l);
            void* dqkv = pp(W.dqkv, l);
            MFVIT_TRY(wait_layer(l + 2));                         // layer l+2's wgrads read this parity's dhpre / gmidT / dqkv
            // Side stream: the four weight gradients of a block are queued TOGETHER, behind the attention backward (their last input), with ONE
            // main -> side dependency per block: an event record + wait costs the host ~9 us on this ROCm, and a fork in front of every gradient made
            // the host the bottleneck of a small-batch step (B = 8: 5.1 -> 6.1 ms).  Their inputs are this layer parity's copies, untouched until then.
            GemmP def_tn[2];
            int ndef = 0;
            auto tn_now_or_later = [&](const GemmP& g) -> int {
                if (use_side) { def_tn[ndef++] = g; return MFVIT_OK; }
                return gemm_tn(d.dtype, g, wst);
            };
            const void* gy2 = gxT;                                // dY of fc2: the residual gradient, masked where the branch was dropped
            if (rdrop) {
                MFVIT_TRY(mask_scale_rows(d.dtype, false, gxT, D * e, ws + W.dtmp, D * e, make_drop(d.p_resid, d.seed, site(l, 4)), d.M, d.D, st));
                gy2 = ws + W.dtmp;
            }
            {   // dW2 += gx^T hact
                GemmP p = zero_gemm();
                p.A = gy2; p.lda = D * e; p.W = b + W.hact; p.ldw = F * e;
                p.M = d.M; p.N = d.D; p.K = d.F;
                if (rdrop) p.cs0 = gb + L.fc2_b;
                p.out0 = gb + L.fc2_w; p.ldo0 = F;
                p.cpart = next_tnpart();                          // split partials: plain stores, reduced in a fixed order at the end of the call
                MFVIT_TRY(tn_now_or_later(p));
            }
            {   // dhpre = (gx W2) * gelu'(hpre)
                GemmP p = zero_gemm();
                p.A = gy2; p.lda = D * e; p.W = sb + S.fc2_t; p.ldw = D * e;
                p.M = d.M; p.N = d.F; p.K = d.D;
                p.aux = b + W.hpre; p.ldaux = F;
                p.out0 = dhpre; p.ldo0 = F * e;
                if (fc1b_in_tile()) { p.cs0 = gb + L.fc1_b; p.cpart = next_colpart(); }   // d fc1_b += column sums of dhpre from the accumulators of this epilogue (partials, fixed-order reduce)
                MFVIT_TRY(gemm_nt_tile(d.dtype, EPI_GELU_BWD, p, st));
            }
            {   // dW1 += dhpre^T y2
                GemmP p = zero_gemm();
                p.A = dhpre; p.lda = F * e; p.W = b + W.y2; p.ldw = D * e;
                p.M = d.M; p.N = d.F; p.K = d.D;
                if (!fc1b_in_tile()) p.cs0 = gb + L.fc1_b;        // d fc1_b += column sums of dhpre (ones-fragment MFMA in the wgrad kernel)
                p.out0 = gb + L.fc1_w; p.ldo0 = D;
                p.cpart = next_tnpart();                          // split partials: plain stores, reduced in a fixed order at the end of the call
                MFVIT_TRY(tn_now_or_later(p));
            }
            {   // gmid = LN2bwd(dhpre W1) + gx ; d ln2_w, d ln2_b, d proj_b
                GemmP p = zero_gemm();
                p.A = dhpre; p.lda = F * e; p.W = sb + S.fc1_t; p.ldw = F * e;
                p.M = d.M; p.N = d.D; p.K = d.F;
                p.aux = b + W.xmid; p.ldaux = D;
                p.mean = (float*)(b + W.st2); p.rstd = (float*)(b + W.st2) + d.M;
                p.gamma = pb + L.ln2_w;
                // 16-bit modes: the residual gradient travels in the operand-type copy alone (gxT / gmidT: hi + lo = the f32 value to 2^-17);
                // the f32 copies gx / gmid were a second 38.7 MB store per launch (14 % of it) read by nobody else but the embedding stage
                if (lean_grad) { p.res_t = gxT; p.ldres_t = D * e; p.out0 = nullptr; }
                else { p.res = gx; p.ldres = D; p.out0 = gmid; }
                p.ldo0 = D; p.out1 = gmidT; p.ldo1 = D * e;
                p.cs0 = gb + L.ln2_w; p.cs1 = gb + L.ln2_b; p.cs2 = rdrop ? colscr + D : gb + L.proj_b; p.cpart = next_colpart();
                if (d.unfused) {
                    // plain data-gradient GEMM -> scratch, then the LayerNorm backward + residual-gradient add + column sums as one row pass
                    GemmP q = zero_gemm();
                    q.A = p.A; q.lda = p.lda; q.W = p.W; q.ldw = p.ldw; q.M = p.M; q.N = p.N; q.K = p.K;
                    q.out0 = ws + W.utmp; q.ldo0 = D * e;
                    MFVIT_TRY(gemm_nt_tile(d.dtype, EPI_NONE, q, st));
                    MFVIT_TRY(ln_bwd_rows(d.dtype, d.D, nullptr, 0, (const float*)(b + W.xmid), D, p.mean, p.rstd, p.gamma, gx, D, gmid, D, gmidT, D * e,
                                          p.cs0, p.cs1, p.cs2, p.cpart, d.M, 1, 0, st, ws + W.utmp, D * e));
                } else {
                    p.kpart = (float*)(ws + W.kpart);
                    MFVIT_TRY(gemm_nt_row(d.dtype, REPI_LNBWD_RES, p, st));
                }
            }
            const void* gyp = gmidT;                              // dY of proj
            if (rdrop) {
                MFVIT_TRY(mask_scale_rows(d.dtype, false, gmidT, D * e, ws + W.dtmp, D * e, make_drop(d.p_resid, d.seed, site(l, 3)), d.M, d.D, st));
                gyp = ws + W.dtmp;
            }
            GemmP pend_proj = zero_gemm();                        // dWproj: launched here, or held back to ride along with dWqkv (one launch for both)
            bool have_pend = false;
            {   // dWproj += gmid^T attn
                GemmP p = zero_gemm();
                p.A = gyp; p.lda = D * e; p.W = b + W.attn; p.ldw = D * e;
                p.M = d.M; p.N = d.D; p.K = d.D;
                if (rdrop) p.cs0 = gb + L.proj_b;
                p.out0 = gb + L.proj_w; p.ldo0 = D;
                p.cpart = next_tnpart();                          // split partials: plain stores, reduced in a fixed order at the end of the call
                // held back by default when the weight gradients run on the caller's stream (nothing to overlap: one launch less is a pure gain,
                // 30.81 -> 30.68 ms per step)
                if (!rdrop) {
                    pend_proj = p;
                    have_pend = true;
                } else {
                    MFVIT_TRY(gemm_tn(d.dtype, p, wst));
                }
            }
            {   // dattn = gmid Wproj
                GemmP p = zero_gemm();
                p.A = gyp; p.lda = D * e; p.W = sb + S.proj_t; p.ldw = D * e;
                p.M = d.M; p.N = d.D; p.K = d.D;
                p.out0 = ws + W.dattn; p.ldo0 = D * e;
                if (do_max) { p.omax = domax(l); p.omax_rows = d.T; p.omax_hd = d.HD; }
                MFVIT_TRY(gemm_nt_tile(d.dtype, EPI_NONE, p, st));
            }
            if (d.p_attn > 0.f)
                MFVIT_TRY(attn_bwd_tiled_drop(d.dtype, b + W.qkv, b + W.attn, ws + W.dattn, (const float*)(b + W.lse), dqkv, d.B, d.T, d.H, d.HD,
                                              make_drop(d.p_attn, d.seed, site(l, 2)), st));
            else
            MFVIT_TRY(attn_bwd(qdt_bwd, b + W.qkv, b + W.attn, ws + W.dattn, (const float*)(b + W.lse), dqkv, nullptr,
                               d.B, d.T, d.H, d.HD, st, do_max ? domax(l) : nullptr));
            MFVIT_TRY(fork());                                    // dqkv - and with it every input of this block's weight gradients - is ready
            for (int i = 0; i < ndef; ++i) MFVIT_TRY(gemm_tn(d.dtype, def_tn[i], wst));
            {   // dWqkv += dqkv^T y1 ; d qkv_b += column sums of dqkv (ones-fragment MFMA inside the wgrad kernel)
                GemmP p = zero_gemm();
                p.A = dqkv; p.lda = 3 * D * e; p.W = b + W.y1; p.ldw = D * e;
                p.M = d.M; p.N = 3 * d.D; p.K = d.D;
                p.cs0 = gb + L.qkv_b;
                p.out0 = gb + L.qkv_w; p.ldo0 = D;
                p.cpart = next_tnpart();                          // split partials: plain stores, reduced in a fixed order at the end of the call
                // dWproj rides along when the LDS-DMA kernel takes both (same rows, same K = 384: 27 + 9 tiles x 7 splits = 252 workgroups): the
                // float atomics of one launch (16.5 MB, 12.5 us) and one prologue less per block
                GemmP pend_nc = pend_proj;
                pend_nc.cpart = nullptr;                          // (in the paired launch both gradients' partials go to dWqkv's slot)
                if (have_pend && gemm_tn_pair_supported(d.dtype, p, pend_nc)) {
                    MFVIT_TRY(gemm_tn_glds_pair(d.dtype, p, pend_nc, wst));
                } else {
                    if (have_pend) MFVIT_TRY(gemm_tn(d.dtype, pend_proj, wst));
                    MFVIT_TRY(gemm_tn(d.dtype, p, wst));
                }
            }
            if (tn_part) {                                        // dW += this block's split partials, splits in a fixed order; the slots are free again
                MFVIT_TRY(tnpart_batch_flush(wst));
                tn_used = 0;
            }
            if (use_side && hipEventRecord(ss.done[l & 63], ss.s) != hipSuccess) return MFVIT_ELAUNCH;
            MFVIT_TRY(wait_layer(l + 1));                         // fc2-wgrad of layer l+1 reads the gxT copy written next
            {   // gx = LN1bwd(dqkv Wqkv) + gmid ; d ln1_w, d ln1_b, d fc2_b of block l-1 (or scratch for the embed stage)
                if (l == 0 && hipMemsetAsync(colscr, 0, 2 * D * sizeof(float), st) != hipSuccess) return MFVIT_ELAUNCH;
                GemmP p = zero_gemm();
                p.A = dqkv; p.lda = 3 * D * e; p.W = sb + S.qkv_t; p.ldw = 3 * D * e;
                p.M = d.M; p.N = d.D; p.K = 3 * d.D;
                p.aux = xbuf(l); p.ldaux = D;
                p.mean = stat(l); p.rstd = stat(l) + d.M;
                p.gamma = pb + L.ln1_w;
                if (lean_grad) { p.res_t = gmidT; p.ldres_t = D * e; p.out0 = l == 0 ? gx : nullptr; }     // (the embedding stage reads gx)
                else { p.res = gmid; p.ldres = D; p.out0 = gx; }
                p.ldo0 = D; p.out1 = pp(W.gxT, l - 1); p.ldo1 = D * e;
                p.cs0 = gb + L.ln1_w; p.cs1 = gb + L.ln1_b; p.cs2 = (l > 0 && !rdrop) ? gblk(l - 1) + L.fc2_b : colscr; p.cpart = next_colpart();
                if (d.unfused) {
                    GemmP q = zero_gemm();
                    q.A = p.A; q.lda = p.lda; q.W = p.W; q.ldw = p.ldw; q.M = p.M; q.N = p.N; q.K = p.K;
                    q.out0 = ws + W.utmp; q.ldo0 = D * e;
                    MFVIT_TRY(gemm_nt_tile(d.dtype, EPI_NONE, q, st));
                    MFVIT_TRY(ln_bwd_rows(d.dtype, d.D, nullptr, 0, xbuf(l), D, p.mean, p.rstd, p.gamma, gmid, D, gx, D, pp(W.gxT, l - 1), D * e,
                                          p.cs0, p.cs1, p.cs2, p.cpart, d.M, 1, 0, st, ws + W.utmp, D * e));
                } else {
                    p.kpart = (float*)(ws + W.kpart);
                    MFVIT_TRY(gemm_nt_row(d.dtype, REPI_LNBWD_RES, p, st));
                }
            }
        } else if (d.tok) {
            MFVIT_TRY(colpart_batch_flush(st));
            // token-input embedding stage: gx = d x_0 = d tokens; d pos_emb = sum over the batch (fuseattention.py:187)
            if (d.p_embd > 0.f)      // d (tokens + pos_emb) = d x_0 * mask / (1 - p)
                MFVIT_TRY(mask_scale_rows(d.dtype, true, gx, D, gx, D, make_drop(d.p_embd, d.seed, 1), d.M, d.D, st));
            if (dinput && hipMemcpyAsync(dinput, gx, (size_t)d.M * D * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
                return MFVIT_ELAUNCH;
            if (d.use_pos) MFVIT_TRY(batch_sum(gx, dparams + L.pos, d.B, (long)d.T * D, st));
        } else {
            MFVIT_TRY(colpart_batch_flush(st));                   // (colscr, read below, is one of the deferred sums)
            // embed stage: gx = d x_0.  d cls_token = sum_b gx[b,0]; d pe_b = sum over patch rows; d pe_w = gx_patch^T patches.
            // pos_embed is a fixed table (requires_grad = False upstream): no gradient.
            MFVIT_TRY(colsum_rows(gx, D, dparams + L.cls, d.B, d.T, 0, d.D, st));
            if (!cfg->stop_grad_conv1) {
                // colscr[0:D] = sum of gx over ALL rows (cs2 of the LN1_0 backward); colscr[D:2D] = sum over the cls rows
                MFVIT_TRY(colsum_rows(gx, D, colscr + D, d.B, d.T, 0, d.D, st));
                MFVIT_TRY(axpy(dparams + L.pe_b, colscr, 1.0f, d.D, st));
                MFVIT_TRY(axpy(dparams + L.pe_b, colscr + D, -1.0f, d.D, st));
                GemmP p = zero_gemm();   // d pe_w += gx[patch rows]^T patches
                p.A = pp(W.gxT, -1); p.lda = D * e; p.W = ws + W.patches; p.ldw = 768 * e;
                p.M = d.Mp; p.N = d.D; p.K = 768;
                p.orow_in = d.np; p.orow_out = d.T; p.orow_off = 1;
                p.out0 = dparams + L.pe_w; p.ldo0 = 768;
                p.cpart = next_tnpart();
                MFVIT_TRY(fork());                                // (the residual gradient of block 0 is ready on the main stream)
                MFVIT_TRY(gemm_tn(d.dtype, p, wst));
            }
        }
    }
    MFVIT_TRY(colpart_batch_flush(st));
    MFVIT_TRY(tnpart_batch_flush(wst));     // dW += the split partials still pending (the patch embedding's), splits in a fixed order
    if (use_side) {   // join: everything the side stream did is ordered before whatever the caller queues next
        if (hipEventRecord(ss.end, ss.s) != hipSuccess || hipStreamWaitEvent(st, ss.end, 0) != hipSuccess) return MFVIT_ELAUNCH;
    }
    return MFVIT_OK;
}

extern "C" {

int mfvit_vit_forward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, const float* img, void* workspace,
                      float* features, mfvit_stream_t stream) {
    ShareScope share(cfg);
    return encoder_forward(cfg, params, shadow, img, workspace, features, stream, false);
}
int mfvit_vit_backward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, void* workspace, const float* dfeatures,
                       float* dparams, int stage_hi, int stage_lo, mfvit_stream_t stream) {
    ShareScope share(cfg);
    return encoder_backward(cfg, params, shadow, workspace, dfeatures, dparams, nullptr, stage_hi, stage_lo, stream, false);
}
int mfvit_gpt_forward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, const float* tokens, void* workspace, float* out,
                      mfvit_stream_t stream) {
    ShareScope share(cfg);
    return encoder_forward(cfg, params, shadow, tokens, workspace, out, stream, true);
}
int mfvit_gpt_backward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, void* workspace, const float* dout, float* dparams,
                       float* dtokens, mfvit_stream_t stream) {
    if (!cfg || !dtokens || !dout) return MFVIT_EINVAL;
    ShareScope share(cfg);
    return encoder_backward(cfg, params, shadow, workspace, dout, dparams, dtokens, cfg->depth, -1, stream, true);
}

// ------------------------------------------------------------------------------------------------ single ops
int mfvit_input_transform(const uint8_t* src, const int64_t* desc, const int32_t* tables, int n, int S, int crop, const float* mean3,
                          const float* std3, float* out, mfvit_stream_t stream) {
    if (!src || !desc || !tables || !mean3 || !std3 || !out) return MFVIT_EINVAL;   // mean3 / std3 are HOST pointers (3 floats each)
    return input_transform(src, (const long long*)desc, tables, n, S, crop, mean3, std3, out, (hipStream_t)stream);
}
int mfvit_eval_counts(const float* scores, int64_t ld, const int64_t* labels, int n, int C, uint64_t* confusion, int64_t* preds,
                      uint64_t* u2, uint64_t* npos, mfvit_stream_t stream) {
    if (!scores || !labels || (!confusion && !(u2 && npos))) return MFVIT_EINVAL;
    return eval_counts(scores, ld, labels, n, C, (unsigned long long*)confusion, (unsigned long long*)u2, (unsigned long long*)npos, preds,
                       (hipStream_t)stream);
}
int mfvit_linear_fwd(int dtype, int epilogue, const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, void* y,
                     int64_t ldy, void* y2, int64_t ldy2, int M, int N, int K, mfvit_stream_t stream) {
    if (!x || !w || (!y && epilogue != EPI_BIAS_GELU)) return MFVIT_EINVAL;      // GELU: y = NULL skips the saved derivative
    if (epilogue != EPI_BIAS && epilogue != EPI_BIAS_GELU && epilogue != EPI_NONE && epilogue != EPI_BIAS_X3F16) return MFVIT_EINVAL;
    if (epilogue == EPI_BIAS_GELU && !y2) return MFVIT_EINVAL;
    GemmP p = zero_gemm();
    p.A = x; p.lda = ldx; p.W = w; p.ldw = ldw; p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.out0 = y; p.ldo0 = ldy; p.out1 = y2; p.ldo1 = ldy2;
    return gemm_nt_tile(dtype, epilogue, p, (hipStream_t)stream);
}
int mfvit_linear_dgrad_act(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const void* act_grad, int64_t ldg,
                           void* dx, int64_t lddx, int M, int N, int K, mfvit_stream_t stream) {
    if (!dy || !wt || !act_grad || !dx) return MFVIT_EINVAL;
    GemmP p = zero_gemm();
    p.A = dy; p.lda = lddy; p.W = wt; p.ldw = ldwt; p.M = M; p.N = N; p.K = K;
    p.aux = act_grad; p.ldaux = ldg; p.out0 = dx; p.ldo0 = lddx;
    return gemm_nt_tile(dtype, EPI_GELU_BWD, p, (hipStream_t)stream);
}
int mfvit_linear_wgrad(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, int64_t lddw, int M, int N, int K,
                       mfvit_stream_t stream) {
    if (!dy || !x || !dw) return MFVIT_EINVAL;
    GemmP p = zero_gemm();
    p.A = dy; p.lda = lddy; p.W = x; p.ldw = ldx; p.M = M; p.N = N; p.K = K;
    p.out0 = dw; p.ldo0 = lddw;
    return gemm_tn(dtype, p, (hipStream_t)stream);
}
int mfvit_linear_wgrad_ws(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, int64_t lddw, int M, int N,
                          int K, float* scratch, mfvit_stream_t stream) {
    if (!dy || !x || !dw) return MFVIT_EINVAL;
    GemmP p = zero_gemm();
    p.A = dy; p.lda = lddy; p.W = x; p.ldw = ldx; p.M = M; p.N = N; p.K = K;
    p.out0 = dw; p.ldo0 = lddw;
    p.cpart = scratch;
    return gemm_tn(dtype, p, (hipStream_t)stream);
}
int mfvit_linear_wgrad_pair(int dtype, const void* dy_a, int64_t lddy_a, const void* x_a, int64_t ldx_a, float* dw_a, int64_t lddw_a, float* dbias_a,
                            int Na, const void* dy_b, int64_t lddy_b, const void* x_b, int64_t ldx_b, float* dw_b, int64_t lddw_b, int Nb, int M, int K,
                            mfvit_stream_t stream) {
    if (!dy_a || !x_a || !dw_a || !dy_b || !x_b || !dw_b) return MFVIT_EINVAL;
    GemmP a = zero_gemm(), b = zero_gemm();
    a.A = dy_a; a.lda = lddy_a; a.W = x_a; a.ldw = ldx_a; a.M = M; a.N = Na; a.K = K; a.out0 = dw_a; a.ldo0 = lddw_a; a.cs0 = dbias_a;
    b.A = dy_b; b.lda = lddy_b; b.W = x_b; b.ldw = ldx_b; b.M = M; b.N = Nb; b.K = K; b.out0 = dw_b; b.ldo0 = lddw_b;
    if (!gemm_tn_pair_supported(dtype, a, b)) return MFVIT_ENOSYS;
    return gemm_tn_glds_pair(dtype, a, b, (hipStream_t)stream);
}
int mfvit_linear_res_ln_fwd(int dtype, const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* res,
                            int64_t ldres, float* x_out, void* y, int y_f32, const float* gamma, const float* beta, float eps,
                            float* mean, float* rstd, int M, int K, mfvit_stream_t stream) {
    return mfvit_linear_res_ln_fwd_ws(dtype, a, lda, w, ldw, bias, res, ldres, x_out, y, y_f32, gamma, beta, eps, mean, rstd, M, K, nullptr, stream);
}
int mfvit_linear_res_ln_fwd_ws(int dtype, const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* res,
                               int64_t ldres, float* x_out, void* y, int y_f32, const float* gamma, const float* beta, float eps,
                               float* mean, float* rstd, int M, int K, float* scratch, mfvit_stream_t stream) {
    if (!a || !w || !y || !gamma || !beta || ((size_t)scratch & 15)) return MFVIT_EINVAL;
    GemmP p = zero_gemm();
    p.kpart = scratch;
    p.A = a; p.lda = lda; p.W = w; p.ldw = ldw; p.M = M; p.N = 384; p.K = K;
    p.bias = bias; p.res = res; p.ldres = ldres;
    p.out0 = x_out; p.ldo0 = 384; p.out1 = y; p.ldo1 = (dtype == MFVIT_BF16X3 && !y_f32) ? 768 : 384; p.y_f32 = y_f32;
    p.gamma = gamma; p.beta = beta; p.eps = eps; p.mean = mean; p.rstd = rstd;
    return gemm_nt_row(dtype, REPI_RES_LN, p, (hipStream_t)stream);
}
int mfvit_linear_dgrad_ln_bwd(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const float* x, const float* mean,
                              const float* rstd, const float* gamma, const float* dres, float* dx, void* dx_t, float* dgamma,
                              float* dbeta, float* dcol, int M, int K, mfvit_stream_t stream) {
    return mfvit_linear_dgrad_ln_bwd_ws(dtype, dy, lddy, wt, ldwt, x, mean, rstd, gamma, dres, dx, dx_t, dgamma, dbeta, dcol, M, K, nullptr, stream);
}
int mfvit_linear_dgrad_ln_bwd_ws(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const float* x, const float* mean,
                                 const float* rstd, const float* gamma, const float* dres, float* dx, void* dx_t, float* dgamma,
                                 float* dbeta, float* dcol, int M, int K, float* scratch, mfvit_stream_t stream) {
    if (!dy || !wt || !x || !mean || !rstd || !gamma || !dx || ((size_t)scratch & 15)) return MFVIT_EINVAL;
    GemmP p = zero_gemm();
    p.kpart = scratch;
    p.A = dy; p.lda = lddy; p.W = wt; p.ldw = ldwt; p.M = M; p.N = 384; p.K = K;
    p.aux = x; p.ldaux = 384; p.mean = (float*)mean; p.rstd = (float*)rstd; p.gamma = gamma;
    p.res = dres; p.ldres = 384;
    p.out0 = dx; p.ldo0 = 384; p.out1 = dx_t; p.ldo1 = dtype == MFVIT_BF16X3 ? 768 : 384;
    p.cs0 = dgamma; p.cs1 = dbeta; p.cs2 = dcol;
    return gemm_nt_row(dtype, REPI_LNBWD_RES, p, (hipStream_t)stream);
}
int mfvit_attention_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int T, int H, int head_dim, mfvit_stream_t stream) {
    if (!qkv || !out || !lse || B <= 0 || T <= 0 || H <= 0) return MFVIT_EINVAL;
    return attn_fwd(dtype, qkv, out, lse, B, T, H, head_dim, (hipStream_t)stream);
}
int mfvit_attention_qkv_dtype(int dtype, int T, int head_dim) { return attn_qkv_dtype(dtype, T, head_dim); }
int mfvit_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias_qkv,
                        int B, int T, int H, int head_dim, mfvit_stream_t stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || B <= 0 || T <= 0 || H <= 0) return MFVIT_EINVAL;
    return attn_bwd(dtype, qkv, out, dout, lse, dqkv, dbias_qkv, B, T, H, head_dim, (hipStream_t)stream);
}
int mfvit_attention_drop_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int T, int H, int head_dim, float p, uint64_t seed,
                             uint32_t site, mfvit_stream_t stream) {
    if (!qkv || !out || !lse || B <= 0 || T <= 0 || H <= 0 || !(p >= 0.f && p < 1.f)) return MFVIT_EINVAL;
    if (!attn_tiled_supported(dtype, T, head_dim)) return MFVIT_ENOSYS;
    return attn_fwd_tiled_drop(dtype, qkv, out, lse, B, T, H, head_dim, make_drop(p, seed, site), (hipStream_t)stream);
}
int mfvit_attention_drop_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int T, int H,
                             int head_dim, float p, uint64_t seed, uint32_t site, mfvit_stream_t stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || B <= 0 || T <= 0 || H <= 0 || !(p >= 0.f && p < 1.f)) return MFVIT_EINVAL;
    if (!attn_tiled_supported(dtype, T, head_dim)) return MFVIT_ENOSYS;
    return attn_bwd_tiled_drop(dtype, qkv, out, dout, lse, dqkv, B, T, H, head_dim, make_drop(p, seed, site), (hipStream_t)stream);
}
int mfvit_dropout_mask(float p, uint64_t seed, uint32_t site, int64_t n, uint8_t* keep, mfvit_stream_t stream) {
    if (!keep || !(p >= 0.f && p < 1.f)) return MFVIT_EINVAL;
    return dropout_mask(make_drop(p, seed, site), n, keep, (hipStream_t)stream);
}
int mfvit_layernorm_fwd(int dtype, const float* x, void* y, int y_f32, const float* gamma, const float* beta, float eps, float* mean,
                        float* rstd, int rows, int N, mfvit_stream_t stream) {
    if (!x || !y || !gamma || !beta) return MFVIT_EINVAL;
    return ln_rows(dtype, N, x, N, nullptr, 0, 0, nullptr, 0, y, (dtype == MFVIT_BF16X3 && !y_f32) ? 2 * N : N, y_f32, gamma, beta, eps, mean, rstd, rows, 1, 0,
                   0, (hipStream_t)stream);
}
int mfvit_layernorm_bwd(int dtype, const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, float* dx, void* dx_t, float* dgamma, float* dbeta, float* dcol, int rows, int N,
                        mfvit_stream_t stream) {
    if (!dy || !x || !mean || !rstd || !gamma) return MFVIT_EINVAL;
    return ln_bwd_rows(dtype, N, dy, N, x, N, mean, rstd, gamma, dres, N, dx, N, dx_t, dtype == MFVIT_BF16X3 ? 2 * N : N, dgamma, dbeta, dcol, nullptr, rows, 1, 0,
                       (hipStream_t)stream);
}
int mfvit_cast_transpose(int dtype, const float* src, void* dst, void* dst_t, int R, int C, mfvit_stream_t stream) {
    if (!src || R <= 0 || C <= 0) return MFVIT_EINVAL;
    return cast_transpose(dtype, src, dst, dst_t, R, C, (hipStream_t)stream);
}
int mfvit_head_fwd(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy, int M, int N, int K, int accumulate,
                   mfvit_stream_t stream) {
    if (!x || !w || !y) return MFVIT_EINVAL;
    return linear_small_fwd(x, ldx, w, b, y, ldy, M, N, K, accumulate, (hipStream_t)stream);
}
int mfvit_head_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* w, float* dx, int64_t lddx, int dx_accumulate,
                   float* dw, float* db, int M, int N, int K, mfvit_stream_t stream) {
    if (!dy || !x || !w) return MFVIT_EINVAL;
    return linear_small_bwd(dy, lddy, x, ldx, w, dx, lddx, dx_accumulate, dw, db, M, N, K, (hipStream_t)stream);
}
int mfvit_cross_entropy(const float* logits, const int64_t* target, float* loss_mean, float* dlogits, int64_t* preds, int B, int C,
                        mfvit_stream_t stream) {
    if (!logits || !target || !loss_mean) return MFVIT_EINVAL;
    return ce_small(logits, (const long*)target, loss_mean, dlogits, (long*)preds, B, C, (hipStream_t)stream);
}

}  // extern "C"
