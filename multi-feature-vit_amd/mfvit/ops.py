"""Tensor-level wrappers over the single-op entry points of libmfvit_hip.so.

Used by the parity tests (tests/test_ops_gpu.py) and by the MoCo projector / predictor path.  Every function
launches on torch's current HIP stream and returns freshly allocated tensors owned by the caller.
"""
import torch

from . import _lib
from ._lib import BF16, BF16X3, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_X3F16, EPI_NONE, F16, F32, X3F16, check, lib, ptr, require_cuda, stream

_TORCH_OF = {BF16: torch.bfloat16, BF16X3: torch.bfloat16, F16: torch.float16, F32: torch.float32}


def _tdtype(code):
    return _TORCH_OF[code]


def _code_of(t, split=False):
    """dtype code of a tensor.  split=True: a bf16 tensor holding the I32 split layout (include/mfvit.h, MFVIT_BF16X3)."""
    if t.dtype == torch.bfloat16:
        return BF16X3 if split else BF16
    if t.dtype == torch.float16 and not split:
        return F16
    if t.dtype == torch.float32 and not split:
        return F32
    raise _lib.MfvitError(f"unsupported dtype {t.dtype}{' for a split tensor' if split else ''}")


def split_pack(x):
    """f32 [..., N] (N % 32 == 0) -> split-bf16 storage [..., 2N]: every group of 32 columns as [hi x 32 | lo x 32] with
    hi = bf16(x), lo = bf16(x - hi)   (torch restatement of the layout the kernels write; used by tests and tools)."""
    x = x.float()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    *lead, N = x.shape
    g = torch.stack([hi.reshape(*lead, N // 32, 32), lo.reshape(*lead, N // 32, 32)], dim=-2)
    return g.reshape(*lead, 2 * N).contiguous()


def split_unpack(s):
    """split-bf16 storage [..., 2N] -> f32 [..., N] (hi + lo)."""
    *lead, N2 = s.shape
    g = s.reshape(*lead, N2 // 64, 2, 32).float()
    return (g[..., 0, :] + g[..., 1, :]).reshape(*lead, N2 // 2)


def split_pack_f16(x):
    """f32 [..., N] -> split-FP16 storage [..., 2N] (dtype tag X3F16: the attention core's qkv operand in bf16x3 mode): the I32 layout
    of split_pack with hi = f16(x), lo = f16(x - hi)."""
    x = x.float()
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    *lead, N = x.shape
    g = torch.stack([hi.reshape(*lead, N // 32, 32), lo.reshape(*lead, N // 32, 32)], dim=-2)
    return g.reshape(*lead, 2 * N).contiguous()


def attention_qkv_dtype(code, T, head_dim):
    """dtype tag of the qkv tensor the encoder hands the attention core for activations of dtype `code` (X3F16 for BF16X3 where the
    whole-head kernels apply)."""
    return lib().mfvit_attention_qkv_dtype(code, T, head_dim)


def linear_fwd(x, w, bias=None, gelu=False, split=False, want_grad=True, qkv_f16=False):
    """y = x @ w.T + bias (x [M,K], w [N,K] of the same dtype).  gelu=True returns (gelu'(y), gelu(y)); want_grad=False skips the
    derivative (returns (None, gelu(y))).  split=True: x, w and the results are split-bf16 storage ([M,2K], [N,2K] -> [M,2N]) - except
    gelu'(y), which the library keeps as plain fp16 [M,N] for split tensors (it only ever multiplies a gradient)."""
    require_cuda(x, w, bias)
    code = _code_of(x, split)
    e = 2 if split else 1
    M, K = x.shape[0], x.shape[1] // e
    N = w.shape[0]
    epi = EPI_BIAS_GELU if gelu else (EPI_BIAS if bias is not None else EPI_NONE)
    if qkv_f16:   # split inputs only: y comes back as split FP16 (same shape; a float16 view of the storage)
        epi = EPI_BIAS_X3F16
    if gelu:
        y2 = torch.empty(M, N * e, device=x.device, dtype=x.dtype)
        y = torch.empty(M, N, device=x.device, dtype=torch.float16 if split else x.dtype) if want_grad else None
        ldy = N
    else:
        y, y2, ldy = torch.empty(M, N * e, device=x.device, dtype=x.dtype), None, N * e
    check(lib().mfvit_linear_fwd(code, epi, ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(y), ldy, ptr(y2), N * e, M, N, K,
                                 stream()), "mfvit_linear_fwd")
    if qkv_f16:
        return y.view(torch.float16)
    return (y, y2) if gelu else y


def linear_dgrad_act(dy, wt, act_grad, split=False):
    """dx = (dy @ wt.T) * act_grad   (dy [M,K], wt [N,K]; act_grad [M,N] as linear_fwd(gelu=True) returns it: plain fp16 for split
    tensors).  The fc2 data gradient fused with the GELU backward."""
    require_cuda(dy, wt, act_grad)
    code = _code_of(dy, split)
    e = 2 if split else 1
    M, K = dy.shape[0], dy.shape[1] // e
    N = wt.shape[0]
    dx = torch.empty(M, N * e, device=dy.device, dtype=dy.dtype)
    check(lib().mfvit_linear_dgrad_act(code, ptr(dy), dy.stride(0), ptr(wt), wt.stride(0), ptr(act_grad), act_grad.stride(0), ptr(dx),
                                       N * e, M, N, K, stream()), "mfvit_linear_dgrad_act")
    return dx


WGRAD_SCRATCH_FLOATS = 384 * 128 * 128   # include/mfvit.h: MFVIT_WGRAD_SCRATCH_FLOATS


_wgrad_scratch = {}


def wgrad_scratch(device):
    """A cached WGRAD_SCRATCH_FLOATS buffer per (device, current stream) for callers that run ONE weight gradient at a time on that stream (the MoCo MLP heads, the
    InfoNCE dq): with it the split partials are plain stores reduced in a fixed order - the same bits on every run - instead of float atomics."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _wgrad_scratch.get(key)
    if t is None:
        t = _wgrad_scratch[key] = torch.empty(WGRAD_SCRATCH_FLOATS, device=device, dtype=torch.float32)
    return t


def linear_wgrad(dy, x, out=None, scratch=None, split=False):
    """dW [N,K] (f32) += dy[M,N].T @ x[M,K].  scratch: optional f32 tensor of WGRAD_SCRATCH_FLOATS elements (split partials
    as plain stores + a reduce pass instead of float atomics).  split=True: dy, x are split-bf16 storage."""
    require_cuda(dy, x)
    code = _code_of(dy, split)
    e = 2 if split else 1
    M, N = dy.shape[0], dy.shape[1] // e
    K = x.shape[1] // e
    if out is None:
        out = torch.zeros(N, K, device=dy.device, dtype=torch.float32)
    if scratch is not None:
        if scratch.dtype != torch.float32 or scratch.numel() < WGRAD_SCRATCH_FLOATS:
            raise _lib.MfvitError("wgrad scratch must be float32 with at least WGRAD_SCRATCH_FLOATS elements")
        check(lib().mfvit_linear_wgrad_ws(code, ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(out), out.stride(0), M, N, K,
                                          ptr(scratch), stream()), "mfvit_linear_wgrad_ws")
    else:
        check(lib().mfvit_linear_wgrad(code, ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(out), out.stride(0), M, N, K, stream()),
              "mfvit_linear_wgrad")
    return out


def linear_wgrad_pair(dy_a, x_a, dy_b, x_b, want_dbias=True, split=False):
    """Two weight gradients over the same M rows and the same K in ONE launch (the encoder backward's dWqkv + dWproj):
    returns (dW_a [Na,K], dbias_a [Na] or None, dW_b [Nb,K]), all f32.  Raises MfvitError (ENOSYS) for shapes the paired kernel
    does not take (M < 4096, N / K not multiples of 128, f32)."""
    require_cuda(dy_a, x_a, dy_b, x_b)
    code = _code_of(dy_a, split)
    e = 2 if split else 1
    M, Na, Nb = dy_a.shape[0], dy_a.shape[1] // e, dy_b.shape[1] // e
    K = x_a.shape[1] // e
    if dy_b.shape[0] != M or x_a.shape[0] != M or x_b.shape[0] != M or x_b.shape[1] // e != K:
        raise _lib.MfvitError("linear_wgrad_pair: the two GEMMs must share M and K")
    dw_a = torch.zeros(Na, K, device=dy_a.device, dtype=torch.float32)
    dw_b = torch.zeros(Nb, K, device=dy_a.device, dtype=torch.float32)
    db_a = torch.zeros(Na, device=dy_a.device, dtype=torch.float32) if want_dbias else None
    check(lib().mfvit_linear_wgrad_pair(code, ptr(dy_a), dy_a.stride(0), ptr(x_a), x_a.stride(0), ptr(dw_a), dw_a.stride(0), ptr(db_a), Na,
                                        ptr(dy_b), dy_b.stride(0), ptr(x_b), x_b.stride(0), ptr(dw_b), dw_b.stride(0), Nb, M, K, stream()),
          "mfvit_linear_wgrad_pair")
    return dw_a, db_a, dw_b


ROWP_SCRATCH_FLOATS = 264 * 7 * 12 * 512      # include/mfvit.h: MFVIT_ROWP_SCRATCH_FLOATS


def linear_res_ln_fwd(a, w, bias, res, gamma, beta, eps, y_f32=False, split=False, scratch=None):
    """x_out = a @ w.T + bias + res ; y = LayerNorm(x_out).  Returns (x_out f32, y, mean, rstd).  scratch (f32[ROWP_SCRATCH_FLOATS]): lets the
    row kernel split K over several workgroups per row tile at small M (mfvit_linear_res_ln_fwd_ws)."""
    require_cuda(a, w, res)
    code = _code_of(a, split)
    e = 2 if split else 1
    M, K = a.shape[0], a.shape[1] // e
    x_out = torch.empty(M, 384, device=a.device, dtype=torch.float32)
    y = torch.empty(M, 384 if y_f32 else 384 * e, device=a.device, dtype=torch.float32 if y_f32 else a.dtype)
    mean = torch.empty(M, device=a.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    check(lib().mfvit_linear_res_ln_fwd_ws(code, ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(res),
                                           res.stride(0) if res is not None else 0, ptr(x_out), ptr(y), int(y_f32), ptr(gamma),
                                           ptr(beta), eps, ptr(mean), ptr(rstd), M, K, ptr(scratch), stream()), "mfvit_linear_res_ln_fwd_ws")
    return x_out, y, mean, rstd


def linear_dgrad_ln_bwd(dy, wt, x, mean, rstd, gamma, dres, want_copy=True, split=False, scratch=None):
    """dx = LNbwd(dy @ wt.T ; x, mean, rstd, gamma) + dres.  wt is the TRANSPOSED weight [384, K].
    Returns (dx f32, dx copy in dy.dtype, dgamma, dbeta, dcol)."""
    require_cuda(dy, wt, x)
    code = _code_of(dy, split)
    e = 2 if split else 1
    M, K = dy.shape[0], dy.shape[1] // e
    dx = torch.empty(M, 384, device=dy.device, dtype=torch.float32)
    dx_t = torch.empty(M, 384 * e, device=dy.device, dtype=dy.dtype) if want_copy else None
    dgamma = torch.zeros(384, device=dy.device, dtype=torch.float32)
    dbeta = torch.zeros_like(dgamma)
    dcol = torch.zeros_like(dgamma)
    check(lib().mfvit_linear_dgrad_ln_bwd_ws(code, ptr(dy), dy.stride(0), ptr(wt), wt.stride(0), ptr(x), ptr(mean), ptr(rstd),
                                             ptr(gamma), ptr(dres), ptr(dx), ptr(dx_t), ptr(dgamma), ptr(dbeta), ptr(dcol), M, K,
                                             ptr(scratch), stream()), "mfvit_linear_dgrad_ln_bwd_ws")
    return dx, dx_t, dgamma, dbeta, dcol


def _attn_code(qkv, split):
    # a float16 tensor in the split layout is the X3F16 qkv operand (out / dout / dqkv: split bf16)
    if split and qkv.dtype == torch.float16:
        return X3F16, torch.bfloat16
    return _code_of(qkv, split), qkv.dtype


def attention_fwd(qkv, heads, split=False):
    """qkv [B,T,3*D] (layout [B][T][3][H][d]) -> (out [B,T,D], lse [B,H,T]).  split=True: split-bf16 storage ([B,T,6*D] -> [B,T,2*D]);
    split=True with a float16 qkv: split-FP16 qkv (split_pack_f16), split-bf16 out."""
    require_cuda(qkv)
    code, odt = _attn_code(qkv, split)
    e = 2 if split else 1
    B, T, D3 = qkv.shape
    D = D3 // (3 * e)
    out = torch.empty(B, T, D * e, device=qkv.device, dtype=odt)
    lse = torch.empty(B, heads, T, device=qkv.device, dtype=torch.float32)
    check(lib().mfvit_attention_fwd(code, ptr(qkv), ptr(out), ptr(lse), B, T, heads, D // heads, stream()), "mfvit_attention_fwd")
    return out, lse


def attention_bwd(qkv, out, dout, lse, heads, want_dbias=True, split=False):
    require_cuda(qkv, out, dout, lse)
    code, odt = _attn_code(qkv, split)
    e = 2 if split else 1
    B, T, D3 = qkv.shape
    D = D3 // (3 * e)
    dqkv = torch.empty(qkv.shape, device=qkv.device, dtype=odt)
    dbias = torch.zeros(3 * D, device=qkv.device, dtype=torch.float32) if want_dbias else None
    check(lib().mfvit_attention_bwd(code, ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(dbias), B, T, heads, D // heads,
                                    stream()), "mfvit_attention_bwd")
    return dqkv, dbias


def attention_drop_fwd(qkv, heads, p, seed, site, split=False):
    """attention_fwd with dropout on the probabilities (streaming kernels; TransFuser GPT): the keep mask is a counter-based hash of
    (seed, site, ((b * H + h) * T + i) * T + j), see include/mfvit.h."""
    require_cuda(qkv)
    code = _code_of(qkv, split)
    e = 2 if split else 1
    B, T, D3 = qkv.shape
    D = D3 // (3 * e)
    out = torch.empty(B, T, D * e, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(B, heads, T, device=qkv.device, dtype=torch.float32)
    check(lib().mfvit_attention_drop_fwd(code, ptr(qkv), ptr(out), ptr(lse), B, T, heads, D // heads, float(p), int(seed), int(site), stream()),
          "mfvit_attention_drop_fwd")
    return out, lse


def attention_drop_bwd(qkv, out, dout, lse, heads, p, seed, site, split=False):
    require_cuda(qkv, out, dout, lse)
    code, odt = _attn_code(qkv, split)
    e = 2 if split else 1
    B, T, D3 = qkv.shape
    D = D3 // (3 * e)
    dqkv = torch.empty(qkv.shape, device=qkv.device, dtype=odt)
    check(lib().mfvit_attention_drop_bwd(code, ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), B, T, heads, D // heads, float(p), int(seed),
                                         int(site), stream()), "mfvit_attention_drop_bwd")
    return dqkv


def dropout_mask(p, seed, site, n, device="cuda:0"):
    """The keep mask (bool, n elements) the kernels regenerate for one dropout site."""
    keep = torch.empty(n, device=device, dtype=torch.uint8)
    check(lib().mfvit_dropout_mask(float(p), int(seed), int(site), n, ptr(keep), stream()), "mfvit_dropout_mask")
    return keep.bool()


def _code_of_dtype(dt, split=False):
    if dt == torch.bfloat16:
        return BF16X3 if split else BF16
    return F16 if dt == torch.float16 else F32


def layernorm_fwd(x, gamma, beta, eps, out_dtype=torch.float32, split=False):
    require_cuda(x, gamma, beta)
    rows, N = x.shape
    y = torch.empty(rows, N * (2 if split else 1), device=x.device, dtype=out_dtype)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    code = _code_of_dtype(out_dtype, split)
    check(lib().mfvit_layernorm_fwd(code, ptr(x), ptr(y), int(out_dtype == torch.float32), ptr(gamma), ptr(beta), eps, ptr(mean),
                                    ptr(rstd), rows, N, stream()), "mfvit_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, copy_dtype=None, split=False):
    require_cuda(dy, x)
    rows, N = x.shape
    dx = torch.empty_like(x)
    dx_t = torch.empty(rows, N * (2 if split else 1), device=x.device, dtype=copy_dtype) if copy_dtype is not None else None
    dgamma = torch.zeros(N, device=x.device, dtype=torch.float32)
    dbeta = torch.zeros_like(dgamma)
    dcol = torch.zeros_like(dgamma)
    code = _code_of_dtype(copy_dtype, split)
    check(lib().mfvit_layernorm_bwd(code, ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx), ptr(dx_t), ptr(dgamma),
                                    ptr(dbeta), ptr(dcol), rows, N, stream()), "mfvit_layernorm_bwd")
    return dx, dx_t, dgamma, dbeta, dcol


def cast_transpose(src, dtype, want_straight=True, want_transposed=True, split=False):
    require_cuda(src)
    R, C = src.shape
    code = _code_of_dtype(dtype, split)
    e = 2 if split else 1
    dst = torch.empty(R, C * e, device=src.device, dtype=dtype) if want_straight else None
    dst_t = torch.empty(C, R * e, device=src.device, dtype=dtype) if want_transposed else None
    check(lib().mfvit_cast_transpose(code, ptr(src), ptr(dst), ptr(dst_t), R, C, stream()), "mfvit_cast_transpose")
    return dst, dst_t


def head_fwd(x, w, b, ldx=None, out=None, accumulate=False):
    """y[m] = x[m*ldx : +K] @ w.T + b   (f32, small N)."""
    require_cuda(x, w, b)
    N, K = w.shape
    if ldx is None:
        ldx = x.stride(0)
    M = x.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    check(lib().mfvit_head_fwd(ptr(x), ldx, ptr(w), ptr(b), ptr(out), out.stride(0), M, N, K, int(accumulate), stream()),
          "mfvit_head_fwd")
    return out


def head_bwd(dy, x, w, ldx=None, dx=None, lddx=None, dx_accumulate=False, dw=None, db=None):
    require_cuda(dy, x, w)
    N, K = w.shape
    M = dy.shape[0]
    if ldx is None:
        ldx = x.stride(0)
    if dx is not None and lddx is None:
        lddx = dx.stride(0)
    check(lib().mfvit_head_bwd(ptr(dy), dy.stride(0), ptr(x), ldx, ptr(w), ptr(dx), lddx or 0, int(dx_accumulate), ptr(dw), ptr(db),
                               M, N, K, stream()), "mfvit_head_bwd")
    return dx, dw, db


def cross_entropy(logits, target, want_grad=True):
    """Returns (loss_mean [1], dlogits or None, preds int64)."""
    require_cuda(logits, target)
    B, C = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    dlogits = torch.empty_like(logits) if want_grad else None
    preds = torch.empty(B, device=logits.device, dtype=torch.int64)
    check(lib().mfvit_cross_entropy(ptr(logits), ptr(target), ptr(loss), ptr(dlogits), ptr(preds), B, C, stream()),
          "mfvit_cross_entropy")
    return loss, dlogits, preds
