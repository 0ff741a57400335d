// Internal launcher declarations (host side) shared by the composite encoder and the C ABI.
#pragma once
#include "gemm.cuh"

namespace mfvit {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_GELU_BWD = 2, EPI_NONE = 3, EPI_BIAS_RELU = 4,    // 4: out0 = relu'(pre), out1 = relu(pre)
       EPI_BIAS_X3F16 = 5 };                                                                  // 5: bias, output written as split FP16 (sbf16 inputs only: qkv)
enum { REPI_RES_LN = 0, REPI_LNBWD_RES = 1 };

int gemm_nt_tile(int dtype, int epi, const GemmP& p, hipStream_t st);
bool gemm_nt_pp_supported(int dtype, int epi, const GemmP& p);     // gemm_pp.hip (round 6): persistent 256 x 128 ping-pong tile kernel, 16-bit types, large M
int gemm_nt_pp(int dtype, int epi, const GemmP& p, hipStream_t st);
int gemm_nt_row(int dtype, int repi, const GemmP& p, hipStream_t st);
bool gemm_nt_rowp_supported(int dtype, int repi, const GemmP& p);   // gemm_rowp.hip: one tall row-complete tile per CU (split bf16)
int gemm_nt_rowp(int dtype, int repi, const GemmP& p, hipStream_t st);
int input_transform(const unsigned char* src, const long long* desc, const int* tables, int n, int S, int crop, const float* mean,
                    const float* stdv, float* out, hipStream_t st);   // input.hip
int eval_counts(const float* scores, long ld, const int64_t* labels, int n, int C, unsigned long long* conf, unsigned long long* u2,
                unsigned long long* npos, int64_t* preds, hipStream_t st);   // metrics.hip
int gemm_tn(int dtype, const GemmP& p, hipStream_t st);
bool gemm_nt_small_supported(int dtype, int epi, const GemmP& p);   // gemm_small.hip: f32, a handful of rows (the fusion module's per-sample rows)
int gemm_nt_small(int epi, GemmP p, hipStream_t st);
bool gemm_tn_small_supported(int dtype, const GemmP& p);
int gemm_tn_small(GemmP p, hipStream_t st);
bool gemm_tn_glds_supported(int dtype, const GemmP& p);   // gemm_tn2.hip
int gemm_tn_glds(int dtype, GemmP p, hipStream_t st);
int colpart_reduce(const float* part, int G, int ncols, int nq, float* d0, float* d1, float* d2, hipStream_t st);
// deferred / batched form: between colpart_batch_begin(&batch) and colpart_batch_begin(previous) every colpart_reduce call of this thread
// is queued (its partial buffer must stay untouched until the flush) and colpart_batch_flush reduces all queued jobs in ONE launch
struct ColpartJob { const float* part; float* d[3]; int G, ncols, nq; };
struct ColpartBatch { static constexpr int MAXJ = 48; int n; ColpartJob job[MAXJ]; };
ColpartBatch* colpart_batch_begin(ColpartBatch* b);
int colpart_batch_flush(hipStream_t st);
bool gemm_tn_pair_supported(int dtype, const GemmP& a, const GemmP& b);   // gemm_tn2.hip: two weight gradients in one launch
int gemm_tn_glds_pair(int dtype, GemmP a, const GemmP& b, hipStream_t st);
// out[n][k] += sum over the splits of part[s * stride + n * K + k]  (fixed order: deterministic).  Between tnpart_batch_begin(&batch) and
// tnpart_batch_begin(previous) every call of this thread is queued (its partial buffer must stay untouched until the flush) and
// tnpart_batch_flush reduces all queued jobs in ONE launch - the weight gradients of a whole encoder-backward call (round 5).
int tn_partial_reduce(const float* part, int splits, long stride, int N, int K, float* out, long ldo, hipStream_t st);
struct TnPartJob { const float* part; float* out; long stride, ldo; int splits, N, K; long first4; };   // first4: prefix of float4 counts
struct TnPartBatch { static constexpr int MAXJ = 56; int n; long total4; TnPartJob job[MAXJ]; };
TnPartBatch* tnpart_batch_begin(TnPartBatch* b);
int tnpart_batch_flush(hipStream_t st);

int attn_fwd_exact(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, int HD, hipStream_t st);
int attn_bwd_exact(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn,
                   int H, int HD, hipStream_t st);

// dtype of the qkv tensor the attention kernels want for activations of `dtype` (MFVIT_X3F16 for split bf16 where the whole-head kernels apply)
int attn_qkv_dtype(int dtype, int Tn, int HD);
int attn_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, int HD, hipStream_t st);
// domax (optional, MFVIT_X3F16 only): the largest |dout| of every (image, head) as f32 bits, [B][H], when the producer of dout has it (GemmP::omax)
int attn_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias, int B, int Tn,
             int H, int HD, hipStream_t st, const unsigned* domax = nullptr);

int im2col16(int dtype, const float* img, void* P, int B, int H, int W, hipStream_t st);
int ln_rows(int dtype, int N, const float* in0, long ld0, const float* in1, long ld1, int mod1, float* xout, long ldx, void* y, long ldy,
            int y_f32, const float* gamma, const float* beta, float eps, float* mean, float* rstd, int rows, int row_stride, int row_off,
            int in0_bcast, hipStream_t st);
int ln_bwd_rows(int dtype, int N, const float* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd,
                const float* gamma, const float* dres, long ldres, float* dx, long lddx, void* dxT, long lddxT, float* dgamma, float* dbeta,
                float* dcol, float* cpart, int rows, int row_stride, int row_off, hipStream_t st, const void* dyT = nullptr, long lddyT = 0);
// v = tin[g] (operand type) (+ pos[go % pmod]) (+ res[go]) -> x[go], y[go] = LN(v), statistics; go = (g / rin) * rout + roff + g % rin (rin = 0: g)
int add_ln_rows(int dtype, int N, const void* tin, long ldt, const float* pos, long ldp, int pmod, const float* res, long ldres, int rin, int rout,
                int roff, float* xout, long ldx, void* y, long ldy, int y_f32, const float* gamma, const float* beta, float eps, float* mean,
                float* rstd, int rows, hipStream_t st);
int cast_transpose(int dtype, const float* src, void* dst, void* dstT, int R, int C, hipStream_t st);
int cast_transpose_batched(int dtype, const float* src, void* dst, void* dstT, int R, int C, int nb, long s_src, long s_dst, long s_dstT,
                           hipStream_t st);
int linear_small_fwd(const float* x, long ldx, const float* W, const float* b, float* y, long ldy, int M, int N, int K, int accumulate,
                     hipStream_t st);
int linear_small_bwd(const float* dy, long lddy, const float* x, long ldx, const float* W, float* dx, long lddx, int dx_accumulate, float* dW,
                     float* db, int M, int N, int K, hipStream_t st);
int ce_small(const float* logits, const long* target, float* loss_mean, float* dlogits, long* preds, int B, int C, hipStream_t st);
int add_rows(float* dst, long ldd, const float* src, long lds_, int rows, int N, hipStream_t st);
int axpy(float* y, const float* x, float a, long n, hipStream_t st);
int colsum_rows(const float* x, long ld, float* out, int rows, int row_stride, int row_off, int N, hipStream_t st);
int batch_sum(const float* x, float* out, int B, long n, hipStream_t st);   // out[i] += sum_b x[b * n + i]

// attention with dropout on the probabilities (TransFuser GPT); streaming kernels only (attention_tiled.hip)
int attn_fwd_tiled_drop(int dtype, const void* qkv, void* out, float* lse, int B, int Tn, int H, int HDim, DropP drop, hipStream_t st);
int attn_bwd_tiled_drop(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int Tn, int H, int HDim,
                        DropP drop, hipStream_t st);
bool attn_tiled_supported(int dtype, int Tn, int HDim);
// elementwise.hip: the keep mask (1 / 0 bytes) of n elements of one dropout site - what the kernels regenerate from (seed, site, index)
int dropout_mask(DropP drop, long n, unsigned char* out, hipStream_t st);
// x = res + drop(t) -> LayerNorm (t: a GEMM output in the operand type, or a0 (+ a1[row % mod1]) in f32); see elementwise.hip
int drop_add_ln_rows(int dtype, int N, const void* tin, long ldt, const float* a0, long lda0, const float* a1, long lda1, int mod1,
                     const float* res, long ldres, DropP drop, float* xout, long ldx, void* y, long ldy, int y_f32, const float* gamma,
                     const float* beta, float eps, float* mean, float* rstd, int rows, hipStream_t st);
int mask_scale_rows(int dtype, bool f32, const void* src, long lds_, void* dst, long ldd, DropP drop, int rows, int N, hipStream_t st);
}  // namespace mfvit
