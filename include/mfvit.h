/* libmfvit_hip.so - C ABI of the MI355X-native (gfx950) Multi-Feature-ViT hot path.
 *
 * The reference (endiqq/Multi-Feature-ViT) is pure Python on stock PyTorch: it has no FFI.  Its drop-in boundary
 * is the Python constructor / train-step API.  This C ABI is the layer underneath that API: every
 * entry point replaces a group of ATen ops dispatched by a cited piece of reference Python, takes plain device
 * pointers and sizes, never allocates, never synchronises, launches on the given HIP stream, and returns
 * 0 or a negative errno-style code (MFVIT_E*).  The caller owns all memory.  Thread-safe and stream-ordered.
 *
 * Paths below are relative to the reference root; MOD = moco_pretraining/moco/model/module.py,
 * FUS = moco_pretraining/moco/model/crossvit_2vits_2additionaloutputs_..._std002_sum.py,
 * BLD = moco_pretraining/moco/moco/builder_vit_mocov3structure_mocov2loss.py, OPT = moco_pretraining/moco/moco/optimizer.py,
 * MAIN_CA / MAIN_SS / MAIN_MOCO = the three main_*.py drivers (SURVEY.md alias table).
 */
#ifndef MFVIT_H
#define MFVIT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mfvit_stream_t; /* hipStream_t */

#define MFVIT_OK 0
#define MFVIT_EINVAL (-22)
#define MFVIT_ENOSYS (-38)
#define MFVIT_ELAUNCH (-5)

#define MFVIT_F32 0  /* exact-f32 MFMA path (parity mode: logits within 1e-3 of the f32 CPU oracle) */
#define MFVIT_BF16 1 /* bf16 operands, f32 accumulate / residual / statistics (throughput mode) */
/* SPLIT bf16 ("bf16x3"): every MFMA operand is kept as hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits) and every product runs as
 * three bf16 MFMAs (hi*hi + lo*hi + hi*lo) with f32 accumulation: f32-grade results (logits within 1e-3 of the f32 CPU path, the
 * reference's CA finetune is fp32: MAIN_CA:862-882 has no autocast) at a third of the bf16 MFMA rate instead of the f32 MFMA rate
 * (1/16).  Tensors of this dtype use the "I32" storage layout: a logical row-major [M][N] matrix (N % 32 == 0) is a bf16 [M][2N]
 * array, every group of 32 logical columns stored as [hi x 32 | lo x 32]; leading dimensions are in STORAGE elements (2 x logical). */
#define MFVIT_BF16X3 2
/* fp16 operands (v_mfma_f32_32x32x16_f16), f32 accumulate / residual / statistics: the arithmetic of the reference's autocast
 * pretraining (MAIN_MOCO:349,533); pair with mfvit_amp_unscale for the GradScaler semantics (MAIN_MOCO:546-548). */
#define MFVIT_F16 3
/* SPLIT fp16, ATTENTION OPERANDS ONLY (round 5): the qkv tensor of mfvit_attention_fwd / _bwd in the I32 layout with hi = f16(x),
 * lo = f16(x - hi) (|x| <= 65504; 22 mantissa bits above 2^-2, an absolute floor of 2^-25 below); out, dout and dqkv of those two calls
 * stay MFVIT_BF16X3.  This is what the encoder runs in bf16x3 mode wherever the whole-head kernels apply (head_dim 32, T <= 224):
 * mfvit_linear_fwd epilogue 5 writes the qkv projection in this form, and every MFMA of the attention core is the f16 one - the
 * probabilities and score gradients are split (or, in the backward, rounded) to fp16 with a third of the vector instructions the
 * bf16 split needs.  No other entry point accepts this tag. */
#define MFVIT_X3F16 4

/* ABI history: 3 = rounds 3 - 4.  4 (round 5, BREAKING): mfvit_linear_fwd_persistent / mfvit_linear_fwd_ws (dropped in round 4 without a
 * version bump) and mfvit_mhsa_fused_fwd are gone; MFVIT_X3F16, linear epilogue 5, mfvit_attention_qkv_dtype and mfvit_adam_step_dev are new.
 * 5 (round 6, BREAKING): mfvit_adam_step_dev is gone (the whole-step HIP graph it served was removed); mfvit_vit_cfg grew `stream_share` (the per-call form of
 * mfvit_set_stream_share); mfvit_prof_collect_tags is new. */
int mfvit_abi_version(void);
const char* mfvit_build_info(void);

/* ------------------------------------------------------------------------------------------------------------
 * ViT-S/16 encoder (the backbone the reference imports as `vits` / `vits_returnftrs`, ABSENT from its tree;
 * call sites MAIN_SS:276,711  MAIN_CA:289-290  FUS:80,83,128-135  BLD:29-30,164,174; spec SURVEY.md Appendix A).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct mfvit_vit_cfg {
    int dtype;           /* MFVIT_F32 | MFVIT_BF16 | MFVIT_BF16X3 | MFVIT_F16 : storage / MFMA operand type of activations and weight shadows */
    int batch;           /* images */
    int img_h, img_w;    /* multiples of 16 */
    int dim;             /* 384 (vit_small) */
    int depth;           /* 12 */
    int heads;           /* 12 -> head_dim 32 */
    int mlp_dim;         /* 1536 */
    int save_for_backward; /* 1: keep per-layer activations in the workspace */
    int stop_grad_conv1; /* 1: no gradient for patch_embed.proj.{weight,bias} (MAIN_MOCO:127-128,274) */
    float ln_eps;        /* 1e-6 */
    /* ---- token-input ("GPT") mode: the TransFuser fusion transformer of fuseattention.py:84-212 is the same pre-LN block stack
     * (separate query / key / value Linears = one packed [3 dim][dim] weight, ReLU MLP, LayerNorm eps 1e-5) fed with tokens
     * instead of image patches; mfvit_gpt_forward / mfvit_gpt_backward run it through the encoder's kernels. */
    int token_input;     /* 0: ViT-S/16 (patch embedding + cls token); 1: the input is a (batch, tokens, dim) f32 tensor */
    int tokens;          /* token_input: sequence length (394 = 2 x 197 joint CXR + ENH tokens) */
    int use_pos;         /* token_input: add the learnable pos_emb (args.pos_embed, fuseattention.py:186-189) */
    int act;             /* MLP activation: 0 erf-GELU (timm Block), 1 ReLU (fuseattention.py:67-72) */
    /* ---- token_input, training: the three dropout sites of the GPT (fuseattention.py:33-34,71,112; config.py:40-42 sets 0.1 each).
     * 0 = off (evaluation, and the whole ViT path).  Masks are counter-based hashes of (seed, site, element index), regenerated by the
     * backward from the same cfg (mfvit_attention_drop_fwd explains the scheme; sites: 1 = embedding, 16 l + 2 = attention of block l,
     * 16 l + 3 = after proj, 16 l + 4 = after the MLP).  The caller draws a fresh seed per forward. */
    float p_embd, p_attn, p_resid;
    uint32_t seed_lo, seed_hi;
    /* ---- launch-geometry hint of THIS call (round 6; ABI 5): how many independent kernel streams run side by side on the GPU while the call's kernels run
     * (the two-stream CA model: 2).  0 = the process-wide default of mfvit_set_stream_share.  Per call and thread-local inside the library: two models in one
     * process no longer share a setting.  Results do not depend on it beyond the summation order of the weight-gradient splits. */
    int stream_share;
} mfvit_vit_cfg;

/* Parameter arena: one contiguous f32 buffer, tensors in timm registration order
 *   cls_token, pos_embed, patch_embed.proj.weight, patch_embed.proj.bias,
 *   blocks.i.{norm1.weight, norm1.bias, attn.qkv.weight, attn.qkv.bias, attn.proj.weight, attn.proj.bias,
 *             norm2.weight, norm2.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias} (i = 0..depth-1),
 *   norm.weight, norm.bias          (the classifier `head` is NOT part of the arena: BLD:218-222 replaces it)
 * The gradient arena has the same layout. */
size_t mfvit_vit_param_count(const mfvit_vit_cfg* cfg);
/* offsets (in floats) of: [0] cls_token [1] pos_embed [2] pe.weight [3] pe.bias [4] blocks.0 [5] block stride
 * [6] norm.weight [7] norm.bias [8] total */
int mfvit_vit_param_layout(const mfvit_vit_cfg* cfg, int64_t out[9]);

/* Weight shadow: dtype-typed copies the MFMA kernels read (bf16: W and W^T of every Linear; f32: W^T only).
 * Must be refreshed (mfvit_vit_prepare_shadow) after every parameter update. */
size_t mfvit_vit_shadow_bytes(const mfvit_vit_cfg* cfg);
int mfvit_vit_prepare_shadow(const mfvit_vit_cfg* cfg, const float* params, void* shadow, mfvit_stream_t stream);

size_t mfvit_vit_workspace_bytes(const mfvit_vit_cfg* cfg);

/* features3D(img): (B,3,H,W) f32 NCHW -> tokens (B, 1+HW/256, dim) f32 after the final LayerNorm
 * (replaces patch_embed -> cat(cls) -> +pos_embed -> 12 x Block -> norm; FUS:128,133; crossvit.py:130-146). */
int mfvit_vit_forward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, const float* img, void* workspace,
                      float* features, mfvit_stream_t stream);

/* Backward of mfvit_vit_forward.  Stages are numbered depth (final norm), depth-1 .. 0 (blocks), -1 (patch embed +
 * cls token); [stage_hi .. stage_lo] are run in descending order so the host can interleave per-block gradient
 * all-reduce with the remaining backward.  dparams is ACCUMULATED into (zero it for a fresh gradient).
 * dfeatures: (B,T,dim) f32, read only when stage_hi == depth. */
int mfvit_vit_backward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, void* workspace, const float* dfeatures,
                       float* dparams, int stage_hi, int stage_lo, mfvit_stream_t stream);

/* Token-input encoder (cfg->token_input = 1): the GPT of the TransFuser fusion (fuseattention.py:84-212), heads x head_dim with
 * head_dim in {32, 64, 96} (config.py: n_embd 384, n_head 4 -> 96), mlp_dim = block_exp * dim.
 * Parameter arena (f32): pos_emb [tokens][dim], then per block
 *   ln1.weight, ln1.bias, attn.{query,key,value}.weight as one [3 dim][dim] matrix, attn.{query,key,value}.bias [3 dim],
 *   attn.proj.weight, attn.proj.bias, ln2.weight, ln2.bias, mlp.0.weight, mlp.0.bias, mlp.2.weight, mlp.2.bias,
 * then ln_f.weight, ln_f.bias.  (mfvit_vit_param_layout: [0] = [1] = 0 pos_emb, [2] = [3] = first block, no patch embedding.)
 * forward : out = ln_f(blocks(tokens + pos_emb))   (B, tokens, dim) f32            (fuseattention.py:186-192; dropouts are identity)
 * backward: dtokens (B, tokens, dim) f32 = d loss / d tokens; dparams accumulated (pos_emb gradient = sum over the batch). */
int mfvit_gpt_forward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, const float* tokens, void* workspace, float* out,
                      mfvit_stream_t stream);
int mfvit_gpt_backward(const mfvit_vit_cfg* cfg, const float* params, const void* shadow, void* workspace, const float* dout, float* dparams,
                       float* dtokens, mfvit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Single ops (exposed for parity tests and for the MoCo projector / predictor path).
 * ------------------------------------------------------------------------------------------------------------ */
/* y[M][N] = x[M][K] W[N][K]^T + bias   (nn.Linear forward; x, W, y of `dtype`; epilogue 0 bias, 1 bias+GELU(erf)
 * writing gelu'(pre-activation) to y (what the backward needs) and the activation to y2, 3 none, 5 bias with y written as
 * MFVIT_X3F16 (MFVIT_BF16X3 only: the qkv projection feeding mfvit_attention_fwd)).  N % 128 == 0, K % 64 == 0.
 * Epilogue 1: y may be NULL (no-grad forward: the derivative is not computed); for MFVIT_BF16X3 y is PLAIN fp16 [M][N] (ldy in
 * fp16 elements) - the derivative only ever multiplies a gradient, the split copy cost 155 MB of stores per fc1 launch. */
int mfvit_linear_fwd(int dtype, int epilogue, const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, void* y,
                     int64_t ldy, void* y2, int64_t ldy2, int M, int N, int K, mfvit_stream_t stream);
/* dx[M][N] = (dy[M][K] Wt[N][K]^T) * act_grad[M][N]   (the fc2 data gradient of timm's Mlp fused with the GELU backward: Wt = W2^T,
 * act_grad = gelu'(pre-activation) as written by mfvit_linear_fwd epilogue 1 - for MFVIT_BF16X3 plain fp16, ldg in fp16 elements).
 * N % 128 == 0, K % 64 == 0 (MFVIT_BF16X3: K % 32 == 0). */
int mfvit_linear_dgrad_act(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const void* act_grad, int64_t ldg,
                           void* dx, int64_t lddx, int M, int N, int K, mfvit_stream_t stream);
/* dW[N][K] (f32, accumulated) += dy[M][N]^T x[M][K]   (nn.Linear weight gradient).  N % 128 == 0, K % 128 == 0. */
int mfvit_linear_wgrad(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, int64_t lddw, int M, int N, int K,
                       mfvit_stream_t stream);
/* Same, with caller-owned scratch for the partial sums of the M-splits (MFVIT_WGRAD_SCRATCH_FLOATS floats, may be NULL): the
 * partials then leave the kernel as plain stores and a second small kernel adds them into dW, instead of float atomics. */
#define MFVIT_WGRAD_SCRATCH_FLOATS (384 * 128 * 128)
int mfvit_linear_wgrad_ws(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, int64_t lddw, int M, int N,
                          int K, float* scratch, mfvit_stream_t stream);
/* TWO weight gradients with the same reduction rows M and the same K in ONE launch (the encoder backward's default for dWqkv + dWproj,
 * the nn.Linear pair of the timm attention block, call sites crossvit_2vits_..._sum.py:128-135):
 *   dw_a[Na][K] += dy_a^T x_a ; dbias_a[Na] += column sums of dy_a (may be NULL) ; dw_b[Nb][K] += dy_b^T x_b.
 * 16-bit dtypes, M >= 2048, Na / Nb / K multiples of 128; anything else returns MFVIT_ENOSYS (the caller then issues two
 * mfvit_linear_wgrad calls). */
int mfvit_linear_wgrad_pair(int dtype, const void* dy_a, int64_t lddy_a, const void* x_a, int64_t ldx_a, float* dw_a, int64_t lddw_a, float* dbias_a,
                            int Na, const void* dy_b, int64_t lddy_b, const void* x_b, int64_t ldx_b, float* dw_b, int64_t lddw_b, int Nb, int M, int K,
                            mfvit_stream_t stream);
/* proj / fc2 (+ residual + following LayerNorm): x_out = a W^T + bias + res ; y = LN(x_out).  N == 384. */
int mfvit_linear_res_ln_fwd(int dtype, const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* res,
                            int64_t ldres, float* x_out, void* y, int y_f32, const float* gamma, const float* beta, float eps,
                            float* mean, float* rstd, int M, int K, mfvit_stream_t stream);
/* dgrad of a Linear feeding a LayerNorm, fused with the LN backward and the residual-gradient add:
 *   dyln = dy W (W given transposed: wt[N=384][K]) ; dx = LNbwd(dyln; x, mean, rstd, gamma) + dres
 *   column sums (accumulated): dgamma, dbeta, dcol = sum_rows dx.  dx_t (dtype copy of dx) optional. */
int mfvit_linear_dgrad_ln_bwd(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const float* x, const float* mean,
                              const float* rstd, const float* gamma, const float* dres, float* dx, void* dx_t, float* dgamma,
                              float* dbeta, float* dcol, int M, int K, mfvit_stream_t stream);
/* The same two with caller-owned scratch (MFVIT_ROWP_SCRATCH_FLOATS floats, 16-byte aligned; NULL = the calls above): at small M the tall-tile row
 * kernel then splits K over up to 4 workgroups per row tile - each streams its share of W[384][K] through its CU - and the workgroup that arrives last
 * adds the partial tiles from the scratch and runs the epilogue (csrc/gemm_rowp.hip; what the encoder does with its own workspace).  Round 5, additive. */
#define MFVIT_ROWP_SCRATCH_FLOATS (264 * 7 * 12 * 512)
int mfvit_linear_res_ln_fwd_ws(int dtype, const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* res,
                               int64_t ldres, float* x_out, void* y, int y_f32, const float* gamma, const float* beta, float eps,
                               float* mean, float* rstd, int M, int K, float* scratch, mfvit_stream_t stream);
int mfvit_linear_dgrad_ln_bwd_ws(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const float* x, const float* mean,
                                 const float* rstd, const float* gamma, const float* dres, float* dx, void* dx_t, float* dgamma,
                                 float* dbeta, float* dcol, int M, int K, float* scratch, mfvit_stream_t stream);
/* softmax(q k^T / sqrt(d)) v per (image, head); qkv [B][T][3][H][d], out [B][T][H*d], lse [B][H][T] (module.py:52-64 is the same math for
 * the cross-attention; the self-attention of the timm block: SURVEY Appendix A).  dtype MFVIT_X3F16: qkv split fp16, out / dout / dqkv
 * MFVIT_BF16X3.  mfvit_attention_qkv_dtype: the tag the encoder uses for the qkv tensor of an activation dtype at (T, head_dim) -
 * MFVIT_X3F16 for MFVIT_BF16X3 where the whole-head kernels apply, the dtype itself otherwise. */
int mfvit_attention_qkv_dtype(int dtype, int T, int head_dim);
int mfvit_attention_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int T, int H, int head_dim, mfvit_stream_t stream);
int mfvit_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias_qkv,
                        int B, int T, int H, int head_dim, mfvit_stream_t stream);
/* The same with dropout on the attention probabilities (TransFuser GPT, fuseattention.py:52 `att = self.attn_drop(att)`), streaming kernels
 * (16-bit dtypes; head_dim 32 / 64 / 96; B * H * T * T < 2^32).  The keep mask is a counter-based hash of (seed, site, element index
 * ((b * H + h) * T + i) * T + j) >= p * 2^32, regenerated by the backward - nothing is stored; kept values are scaled by 1 / (1 - p).
 * p = 0 is plain attention.  mfvit_dropout_mask writes the keep bytes (1 / 0) of the first n elements of a site: what the parity tests
 * feed the reference arithmetic with (the reference's own torch RNG stream cannot be reproduced). */
int mfvit_attention_drop_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int T, int H, int head_dim, float p, uint64_t seed,
                             uint32_t site, mfvit_stream_t stream);
int mfvit_attention_drop_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int T, int H,
                             int head_dim, float p, uint64_t seed, uint32_t site, mfvit_stream_t stream);
int mfvit_dropout_mask(float p, uint64_t seed, uint32_t site, int64_t n, uint8_t* keep, mfvit_stream_t stream);
/* LayerNorm over rows of width N in {384, 768} (f32 in; y of dtype or f32). */
int mfvit_layernorm_fwd(int dtype, const float* x, void* y, int y_f32, const float* gamma, const float* beta, float eps, float* mean,
                        float* rstd, int rows, int N, mfvit_stream_t stream);
int mfvit_layernorm_bwd(int dtype, const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, float* dx, void* dx_t, float* dgamma, float* dbeta, float* dcol, int rows, int N,
                        mfvit_stream_t stream);
/* f32 [R][C] -> dtype [R][C] (dst, optional) and dtype [C][R] (dst_t, optional). */
int mfvit_cast_transpose(int dtype, const float* src, void* dst, void* dst_t, int R, int C, mfvit_stream_t stream);
/* classifier heads with a handful of classes (MAIN_CA:309-310, FUS:105-113): y = x W^T + b over rows x[m*ldx]. */
int mfvit_head_fwd(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy, int M, int N, int K, int accumulate,
                   mfvit_stream_t stream);
int mfvit_head_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* w, float* dx, int64_t lddx, int dx_accumulate,
                   float* dw, float* db, int M, int N, int K, mfvit_stream_t stream);
/* nn.CrossEntropyLoss (mean) for C <= 64 classes (MAIN_CA:873, MAIN_SS:714): loss_mean[1], dlogits = dloss/dlogits,
 * preds = argmax (MAIN_CA:870). */
int mfvit_cross_entropy(const float* logits, const int64_t* target, float* loss_mean, float* dlogits, int64_t* preds, int B, int C,
                        mfvit_stream_t stream);

/* GPU-side input pipeline (replaces the torchvision / Pillow chain of aihc_utils/image_transform.py:50-84 run by
 * the DataLoader workers, moco/loader.py:121-137): Resize((S,S), bilinear) -> horizontal flip -> rotation (nearest, fill 0) -> crop
 * -> ToTensor -> Normalize, fused, bit-exact against Pillow's fixed-point arithmetic.
 *   src    : decoded uint8 HWC (3 channels, channel order as decoded) images of n samples, packed back to back, on the device
 *   desc   : device int64 [n][20]: 0 byte offset in src of the top-left pixel of the source window, 1 in_h, 2 in_w, 3 / 4 offsets (in int32 units) of the x / y
 *            resample tables, 5 / 6 their ksize, 7 flip (0/1), 8 rotation mode (0 none, 1 affine, 2/3/4 = transpose 90/180/270),
 *            9..14 the 16.16 fixed-point affine terms a0 a1 a2 a3 a4 a5 of libImaging's affine_fixed, 15 (crop_i << 32) | crop_j,
 *            16 source row pitch in bytes (3 * image width; a RandomResizedCrop box is a window of the image), 17..19 zero
 *   tables : device int32: per axis S rows of [first source index, tap count, taps[ksize]] (22-bit fixed point)
 *   out    : float32 [n][3][crop][crop]
 * mfvit.input_pipeline builds desc / tables from image sizes and the random draws exactly as Pillow's Python / C code does. */
int mfvit_input_transform(const uint8_t* src, const int64_t* desc, const int32_t* tables, int n, int S, int crop, const float* mean3,
                          const float* std3, float* out, mfvit_stream_t stream);

/* Epoch metrics on the device (replaces the per-batch .cpu() copies MAIN_CA:886-899 and the scikit-learn calls
 * MAIN_CA:901-911).  scores f32 [n][C] (row stride ld), labels int64 [n].  ACCUMULATES into caller-zeroed uint64 arrays:
 * confusion[t][p] (+ optional preds[n] = first-maximum argmax, as torch.max MAIN_CA:870), and per class c the pair counts of the
 * one-vs-rest ROC AUC: u2[c] = 2 #{(pos, neg): s_pos > s_neg} + #{s_pos == s_neg}, npos[c] = #{label == c}; AUC_c = u2 /
 * (2 npos (n - npos)) = the trapezoid area of sklearn's roc_curve.  The pair counts are over THIS call's n samples (call it
 * once on the whole epoch's scores); the confusion matrix may be accumulated batch by batch.  Either half may be NULL. */
int mfvit_eval_counts(const float* scores, int64_t ld, const int64_t* labels, int n, int C, uint64_t* confusion, int64_t* preds,
                      uint64_t* u2, uint64_t* npos, mfvit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Two-stream fusion: bidirectional cls<->patch cross-attention exchange + heads (f32).
 * Replaces PreNorm / CrossAttention (MOD:15-21,108-137), MultiScaleTransformerEncoder.forward (FUS:35-65) and
 * Fus_CrossViT.forward (FUS:126-157) for pool='cls', cross_attn_depth = multi_scale_enc_depth = 1.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct mfvit_fusion_cfg {
    int batch;
    int tokens;       /* 1 + patches (197 at 224^2) */
    int dim;          /* 384 */
    int heads;        /* 3 (FUS:75) -> head_dim 128 */
    int num_classes;  /* 3 */
    float eps_pre;    /* 1e-5: PreNorm's nn.LayerNorm default (MOD:18) */
    float eps_post;   /* 1e-6: FUS:26,31 */
} mfvit_fusion_cfg;

/* Parameter arena (f32), state-dict order of Fus_CrossViT (22 tensors, 8 D^2 + 10 D + 2 (C D + C) floats):
 *   cross_attn_layers.0.{0.norm.{weight,bias}, 0.fn.{wq,wk,wv}.weight, 0.fn.proj.{weight,bias}, 1.{weight,bias},
 *                        2.norm.{weight,bias}, 2.fn.{wq,wk,wv}.weight, 2.fn.proj.{weight,bias}, 3.{weight,bias}},
 *   mlp_head_cxr.0.{weight,bias}, mlp_head_enh.0.{weight,bias}.   The gradient arena has the same layout. */
size_t mfvit_fusion_param_count(const mfvit_fusion_cfg* cfg);
size_t mfvit_fusion_workspace_bytes(const mfvit_fusion_cfg* cfg);
/* f_cxr / f_enh: (B,T,dim) f32 tokens (features3D of each stream).  hw_* / hb_*: the backbones' own classifier heads
 * (C x dim, C) or NULL; when given, x_cxr / x_enh = head(f[:,0]) (FUS:131,135).  fused: (B,C) (FUS:147-155).
 * The workspace keeps what the backward needs. */
int mfvit_fusion_forward(const mfvit_fusion_cfg* cfg, const float* params, const float* f_cxr, const float* f_enh, const float* hw_cxr,
                         const float* hb_cxr, const float* hw_enh, const float* hb_enh, void* workspace, float* fused, float* x_cxr,
                         float* x_enh, mfvit_stream_t stream);
/* dparams, dhw_*, dhb_* are ACCUMULATED into.  df_cxr / df_enh: (B,T,dim) fully written, or both NULL when the
 * backbones are frozen (README default, MAIN_CA:298-305). */
int mfvit_fusion_backward(const mfvit_fusion_cfg* cfg, const float* params, const float* f_cxr, const float* f_enh, const float* hw_cxr,
                          const float* hw_enh, void* workspace, const float* dfused, const float* dx_cxr, const float* dx_enh,
                          float* dparams, float* df_cxr, float* df_enh, float* dhw_cxr, float* dhb_cxr, float* dhw_enh, float* dhb_enh,
                          mfvit_stream_t stream);

/* Stand-alone PreNorm(CrossAttention) (MOD:15-21,108-137): out[b] = proj(attn(LN_1e-5([x_own[b,0] ; x_oth[b,1:]]))) -> (B, dim).
 * params = one block [norm.weight, norm.bias, fn.wq.weight, fn.wk.weight, fn.wv.weight, fn.proj.weight, fn.proj.bias]
 * (4 D^2 + 3 D floats).  x_own == x_oth reproduces PreNorm(dim, CrossAttention(...))(x) exactly.  cfg: batch, tokens, dim 384,
 * heads 3, eps_pre; workspace of mfvit_fusion_workspace_bytes(cfg).  Backward: dparams accumulated; dx_own receives row 0 only,
 * dx_oth rows 1.. only (the caller zero-fills both), or both NULL. */
int mfvit_prenorm_xattn_forward(const mfvit_fusion_cfg* cfg, const float* params, const float* x_own, const float* x_oth,
                                void* workspace, float* out, mfvit_stream_t stream);
int mfvit_prenorm_xattn_backward(const mfvit_fusion_cfg* cfg, const float* params, const float* x_own, const float* x_oth,
                                 void* workspace, const float* dout, float* dparams, float* dx_own, float* dx_oth,
                                 mfvit_stream_t stream);

/* Bare CrossAttention (MOD:123-137 without the PreNorm; the reference's live path never calls it that way, FUS:25,30):
 * out[b] = proj(attn(q = wq x[b,0], k / v = wk / wv x[b,:])) -> (B, dim).  params = [wq.weight, wk.weight, wv.weight, proj.weight,
 * proj.bias] (4 D^2 + D floats); cfg / workspace as above (eps_pre unused).  Backward: dparams accumulated, dx (B,T,dim) fully
 * written (zero-filled by the caller, or NULL). */
int mfvit_xattn_forward(const mfvit_fusion_cfg* cfg, const float* params, const float* x, void* workspace, float* out,
                        mfvit_stream_t stream);
int mfvit_xattn_backward(const mfvit_fusion_cfg* cfg, const float* params, const float* x, void* workspace, const float* dout,
                         float* dparams, float* dx, mfvit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Opt-in kernel timing (HIP events on the launch stream), used by bench.py for the roofline of the dominant kernel.
 * Classes: 0 gemm_nt_tile 1 gemm_nt_row_res_ln 2 gemm_nt_row_lnbwd 3 gemm_tn_wgrad 4 attention_fwd 5 attention_bwd
 *          6 xattn_stream_fwd 7 xattn_stream_bwd 8 infonce 9 other.
 * mfvit_prof_collect waits for the recorded events and fills out[cls*4 + {0 launches, 1 ms, 2 algorithmic flops,
 * 3 algorithmic bytes}] (ncls <= 10), then clears the records. */
int mfvit_prof_enable(int class_mask); /* bit c set = time class c; 0 = off */
/* Weight-gradient GEMMs of mfvit_vit_backward run on a library-owned side stream beside the dgrad chain (default on; also
 * MFVIT_WGRAD_STREAM=0).  0 serialises them on the caller's stream - used by bench.py's attribution pass so that a kernel's
 * event-timed duration is its own, not a share of a co-scheduled GPU.  Results are identical either way. */
int mfvit_set_wgrad_stream(int enabled);
/* How many independent kernel streams the caller runs side by side on this GPU (default 1: a kernel may size its grid for the whole chip).  The
 * two-stream CA model sets 2 (both encoders at once, crossvit_2vits_..._sum.py:128-135 are independent): for SMALL problems the
 * one-workgroup-per-CU kernels (row-complete GEMMs, weight gradients) then launch about half as many, longer workgroups, and the two encoders'
 * kernels run beside each other.  A hint only: results do not depend on it beyond the summation order of the weight-gradient splits. */
int mfvit_set_stream_share(int n);
int mfvit_prof_collect(double* out, int ncls);
/* Sub-attribution of the records the last mfvit_prof_collect call consumed: out[tag*3 + {0 launches, 1 ms, 2 algorithmic flops}] for tag 1 = the qkv
 * projections and tag 2 = the output projections (+ residual + LayerNorm) of the encoder FORWARD - with class attention_fwd the three parts of the
 * fused multi-head self-attention figure BASELINE.json's metric names (Attention.forward, moco_pretraining/moco/model/module.py:52-64). */
int mfvit_prof_collect_tags(double* out, int ntags);
const char* mfvit_prof_class_name(int cls);

/* ------------------------------------------------------------------------------------------------------------
 * MoCo step pieces (BLD = moco/builder_vit_mocov3structure_mocov2loss.py).  The Linear layers of the projector /
 * predictor run on mfvit_linear_fwd / mfvit_linear_wgrad.
 * ------------------------------------------------------------------------------------------------------------ */
/* BatchNorm1d over the batch axis of x[n][C] (BLD:62-78; SyncBatchNorm semantics, MAIN_MOCO:297):
 *   mfvit_bn_stats   : rank-local per-column mean and M2 = sum (x - mean)^2
 *   mfvit_bn_combine : merge W (mean, M2, count) triples (all_gathered by the host) -> global mean, invstd; updates the
 *                      running statistics (momentum, unbiased variance) when given
 *   mfvit_bn_apply   : y = (x - mean) * invstd [* gamma + beta] [-> ReLU]   (gamma == NULL: affine=False) */
int mfvit_bn_stats(int dtype, const void* x, int n, int C, float* mean, float* m2, mfvit_stream_t stream);
int mfvit_bn_combine(const float* means, const float* m2s, const float* counts, int W, int C, float eps, float momentum, float* mean,
                     float* invstd, float* running_mean, float* running_var, mfvit_stream_t stream);
int mfvit_bn_apply(int dtype, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                   void* y, int n, int C, mfvit_stream_t stream);
/* backward: s1 = sum dy', s2 = sum dy' * xhat per column (dy' = dy masked by the fused ReLU via y); after the host has
 * summed s1, s2 over ranks: dx = gamma * invstd * (dy' - S1 * inv_count - xhat * S2 * inv_count).  dgamma = local s2, dbeta = local s1. */
int mfvit_bn_bwd_sums(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* invstd, int relu, int n,
                      int C, float* s1, float* s2, mfvit_stream_t stream);
int mfvit_bn_bwd_apply(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* invstd, const float* gamma,
                       int relu, const float* S1, const float* S2, float inv_count, void* dx, int n, int C, mfvit_stream_t stream);
/* F.normalize(x, dim=1) (BLD:165,175) and its backward (f32). */
int mfvit_l2norm_fwd(const float* x, float* y, float* inv_norm, int n, int C, float eps, mfvit_stream_t stream);
int mfvit_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int n, int C, mfvit_stream_t stream);
/* out[r * ldo] = scale * (a[r] . b[r])   (l_pos, BLD:183). */
int mfvit_rowdot(const float* a, const float* b, float* out, int64_t ldo, float scale, int n, int C, mfvit_stream_t stream);
/* nn.CrossEntropyLoss (mean) over wide rows, e.g. the (n, 1 + 65536) InfoNCE logits (MAIN_MOCO:330,535):
 * loss_mean[1]; optional per-row lse (given: the mean is taken over the rows in a fixed order - the same bits on every run; NULL: one float atomic per
 * row); optional dlogits = (softmax - onehot) / n. */
int mfvit_cross_entropy_rows(const float* logits, int64_t ld, const int64_t* target, float* loss_mean, float* lse, float* dlogits,
                             int64_t ldd, int n, int C, mfvit_stream_t stream);
/* momentum update over a flat arena: dst = dst * m + src * (1 - m)   (BLD:83-89, one launch instead of ~157 x 3). */
int mfvit_ema_update(float* dst, const float* src, float m, int64_t n, mfvit_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-tensor optimizer steps.  `table` is a DEVICE array of int64[nchunks][7]:
 *   [tensor_id, param_ptr, grad_ptr, state0_ptr, state1_ptr, count, flag]   (pointers to the chunk's first float)
 * LARS (OPT:10-43): flag = 1 for tensors with ndim > 1 (weight decay + trust ratio), 0 otherwise; state0 = mu;
 *   `norms` = device scratch float[2 * ntensors].
 * Adam / AdamW (torch.optim semantics; MAIN_CA:455-459, MAIN_MOCO:338-345): state0 = exp_avg, state1 = exp_avg_sq,
 *   flag bit 0 = decoupled weight decay (AdamW); `step` counts from 1.
 * SGD (MAIN_CA:445-448): state0 = momentum buffer; first_step = 1 initialises the buffer with the gradient. */
int mfvit_lars_step(const int64_t* table, int nchunks, int ntensors, float* norms, float lr, float weight_decay, float momentum,
                    float trust_coefficient, mfvit_stream_t stream);
int mfvit_adam_step(const int64_t* table, int nchunks, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    mfvit_stream_t stream);
int mfvit_sgd_step(const int64_t* table, int nchunks, float lr, float momentum, float weight_decay, int first_step,
                   mfvit_stream_t stream);
/* torch.cuda.amp.GradScaler.unscale_ over the same chunk table (MAIN_MOCO:349,546-548: scaler.scale(loss).backward();
 * scaler.step(optimizer); scaler.update()): every gradient element is multiplied by inv_scale and *found_inf (device float,
 * cleared by the caller) is set to 1 when any element was inf / nan before the multiplication. */
int mfvit_amp_unscale(const int64_t* table, int nchunks, float inv_scale, float* found_inf, mfvit_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MFVIT_H */
