"""Stand-alone entry points of the exchange modules on the HIP path:
  PreNorm(dim, CrossAttention(dim, num_heads=3))(x)                 (MOD:15-21,108-137)
  CrossAttention(dim, num_heads=3)(x)                               (MOD:123-137; never called bare by the reference)
  MultiScaleTransformerEncoder(...)(xs, xl) -> (xs_out, xl_out)     (FUS:35-65)
Same kernels as the fused Fus_CrossViT pipeline (csrc/fusion.hip), run for one direction at a time; the post-exchange
LayerNorm over all token rows (dead inside Fus_CrossViT) is the row LayerNorm kernel."""
import torch

from . import _lib, ops
from .arena import ParamArena
from .fusion import fusion_cfg
from ._lib import check, lib, ptr, stream


class _PreNormXAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, arena, x_own, x_oth, *params):
        _lib.require_cuda(x_own, x_oth)
        x_own = x_own.contiguous().float()
        x_oth = x_own if x_oth is None else x_oth.contiguous().float()
        B, T, D = x_own.shape
        if D != 384 or tuple(x_oth.shape) != (B, T, D):
            raise _lib.MfvitError("PreNorm(CrossAttention) is built for (B, T, 384) inputs, 3 heads")
        flat = arena.ensure()
        cfg = fusion_cfg(B, T, 3)
        ws = torch.empty(lib().mfvit_fusion_workspace_bytes(cfg), device=x_own.device, dtype=torch.uint8)
        out = torch.empty(B, 1, D, device=x_own.device, dtype=torch.float32)
        check(lib().mfvit_prenorm_xattn_forward(cfg, ptr(flat), ptr(x_own), ptr(x_oth), ptr(ws), ptr(out), stream()),
              "mfvit_prenorm_xattn_forward")
        ctx.arena, ctx.cfg, ctx.ws = arena, cfg, ws
        ctx.save_for_backward(x_own, x_oth)
        return out

    @staticmethod
    def backward(ctx, dout):
        x_own, x_oth = ctx.saved_tensors
        arena = ctx.arena
        flat = arena.ensure()
        gflat = torch.zeros_like(flat)
        need_dx = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx_own = torch.zeros_like(x_own) if need_dx else None
        dx_oth = torch.zeros_like(x_oth) if need_dx else None
        check(lib().mfvit_prenorm_xattn_backward(ctx.cfg, ptr(flat), ptr(x_own), ptr(x_oth), ptr(ctx.ws), ptr(dout.contiguous().float()),
                                                 ptr(gflat), ptr(dx_own), ptr(dx_oth), stream()), "mfvit_prenorm_xattn_backward")
        ctx.ws = None
        return (None, dx_own if ctx.needs_input_grad[1] else None, dx_oth if ctx.needs_input_grad[2] else None) + \
            tuple(arena.grad_views(gflat))


def _arena_of(prenorm):
    a = getattr(prenorm, "_mfvit_arena", None)
    if a is None or not a.intact():
        if prenorm.fn.num_heads != 3 or prenorm.norm.normalized_shape != (384,):
            raise _lib.MfvitError("the fused cross-attention is built for dim 384 / 3 heads (FUS:73-75 defaults)")
        a = ParamArena(list(prenorm.named_parameters()))          # norm.{weight,bias}, fn.{wq,wk,wv}.weight, fn.proj.{weight,bias}
        object.__setattr__(prenorm, "_mfvit_arena", a)
    return a


def prenorm_cross_attention(prenorm, x, x_other=None):
    """PreNorm(CrossAttention)(x): query = row 0 of x, keys / values = all rows of [x[:, :1] ; (x_other or x)[:, 1:]] -> (B, 1, C)."""
    a = _arena_of(prenorm)
    return _PreNormXAttnFn.apply(a, x, x if x_other is None else x_other, *a.params)


class _XAttnFn(torch.autograd.Function):
    """Bare CrossAttention(x) (MOD:123-137): the same folded kernels with the normalisation switched off."""

    @staticmethod
    def forward(ctx, arena, x, *params):
        _lib.require_cuda(x)
        x = x.contiguous().float()
        B, T, D = x.shape
        if D != 384:
            raise _lib.MfvitError("CrossAttention is built for (B, T, 384) inputs, 3 heads")
        flat = arena.ensure()
        cfg = fusion_cfg(B, T, 3)
        ws = torch.empty(lib().mfvit_fusion_workspace_bytes(cfg), device=x.device, dtype=torch.uint8)
        out = torch.empty(B, 1, D, device=x.device, dtype=torch.float32)
        check(lib().mfvit_xattn_forward(cfg, ptr(flat), ptr(x), ptr(ws), ptr(out), stream()), "mfvit_xattn_forward")
        ctx.arena, ctx.cfg, ctx.ws = arena, cfg, ws
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, = ctx.saved_tensors
        arena = ctx.arena
        flat = arena.ensure()
        gflat = torch.zeros_like(flat)
        dx = torch.zeros_like(x) if ctx.needs_input_grad[1] else None
        check(lib().mfvit_xattn_backward(ctx.cfg, ptr(flat), ptr(x), ptr(ctx.ws), ptr(dout.contiguous().float()), ptr(gflat), ptr(dx), stream()),
              "mfvit_xattn_backward")
        ctx.ws = None
        return (None, dx) + tuple(arena.grad_views(gflat))


def cross_attention(ca, x):
    """CrossAttention(dim, num_heads=3)(x) without a PreNorm around it -> (B, 1, C)."""
    a = getattr(ca, "_mfvit_arena", None)
    if a is None or not a.intact():
        if ca.num_heads != 3 or ca.wq.weight.shape != (384, 384):
            raise _lib.MfvitError("the fused cross-attention is built for dim 384 / 3 heads (FUS:73-75 defaults)")
        a = ParamArena(list(ca.named_parameters()))                # wq.weight, wk.weight, wv.weight, proj.weight, proj.bias
        object.__setattr__(ca, "_mfvit_arena", a)
    return _XAttnFn.apply(a, x, *a.params)


class _LNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        shape = x.shape
        x2 = x.contiguous().float().reshape(-1, shape[-1])
        y, mean, rstd = ops.layernorm_fwd(x2, w, b, eps)
        ctx.save_for_backward(x2, mean, rstd, w)
        ctx.shape = shape
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, w = ctx.saved_tensors
        dx, _, dg, db, _ = ops.layernorm_bwd(dy.contiguous().float().reshape(x2.shape), x2, mean, rstd, w)
        return dx.reshape(ctx.shape), dg, db, None


def exchange(encoder, xs, xl):
    """MultiScaleTransformerEncoder.forward (FUS:35-65), cross_attn_depth = 1."""
    (pn_s, n_l, pn_l, n_s), = [tuple(layer) for layer in encoder.cross_attn_layers]
    cal_l = xl[:, 0:1] + prenorm_cross_attention(pn_l, xl, xs)                 # large cls attends small patches   FUS:50-53
    xl_out = _LNFn.apply(torch.cat((cal_l, xl[:, 1:]), dim=1), n_l.weight, n_l.bias, n_l.eps)    # FUS:54-55
    cal_s = xs[:, 0:1] + prenorm_cross_attention(pn_s, xs, xl)                 # small cls attends large patches   FUS:58-61
    xs_out = _LNFn.apply(torch.cat((cal_s, xs[:, 1:]), dim=1), n_s.weight, n_s.bias, n_s.eps)    # FUS:62-63
    return xs_out, xl_out
