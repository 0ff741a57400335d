"""Two-stream fusion engine: one autograd node over mfvit_fusion_forward / mfvit_fusion_backward (f32)."""
import torch

from . import _lib
from ._lib import FusionCfg, check, lib, ptr, stream


def fusion_cfg(B, T, C):
    c = FusionCfg()
    c.batch, c.tokens, c.dim, c.heads, c.num_classes = B, T, 384, 3, C
    c.eps_pre, c.eps_post = 1e-5, 1e-6
    return c


class FusionFn(torch.autograd.Function):
    """(f_cxr, f_enh, backbone heads, 22 fusion params) -> (fused, x_cxr, x_enh)   [FUS:126-157]"""

    @staticmethod
    def forward(ctx, arena, f_cxr, f_enh, hw_c, hb_c, hw_e, hb_e, *params):
        _lib.require_cuda(f_cxr, f_enh)
        B, T, D = f_cxr.shape
        flat = arena.ensure()
        C = (flat.numel() - 8 * D * D - 10 * D) // (2 * (D + 1))
        cfg = fusion_cfg(B, T, C)
        if lib().mfvit_fusion_param_count(cfg) != flat.numel():
            raise _lib.MfvitError("fusion parameter arena does not match the C ABI layout")
        f_cxr = f_cxr.contiguous().float()
        f_enh = f_enh.contiguous().float()
        ws = torch.empty(lib().mfvit_fusion_workspace_bytes(cfg), device=f_cxr.device, dtype=torch.uint8)
        fused = torch.empty(B, C, device=f_cxr.device, dtype=torch.float32)
        have_heads = hw_c is not None and hw_e is not None
        x_c = torch.empty_like(fused) if have_heads else None
        x_e = torch.empty_like(fused) if have_heads else None
        check(lib().mfvit_fusion_forward(cfg, ptr(flat), ptr(f_cxr), ptr(f_enh), ptr(hw_c), ptr(hb_c), ptr(hw_e), ptr(hb_e), ptr(ws),
                                         ptr(fused), ptr(x_c), ptr(x_e), stream()), "mfvit_fusion_forward")
        ctx.arena, ctx.cfg, ctx.ws = arena, cfg, ws
        ctx.have_heads = have_heads
        ctx.save_for_backward(f_cxr, f_enh, hw_c, hw_e)
        if not have_heads:
            x_c = fused.new_zeros(B, C)
            x_e = fused.new_zeros(B, C)
        return fused, x_c, x_e

    @staticmethod
    def backward(ctx, dfused, dx_c, dx_e):
        f_cxr, f_enh, hw_c, hw_e = ctx.saved_tensors
        arena, cfg = ctx.arena, ctx.cfg
        flat = arena.ensure()
        need_df = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        # ONE zeroed buffer for every accumulated gradient of this node (the fusion arena + the four head tensors): one fill launch per step
        # instead of five; every view starts 16-byte aligned
        sizes = [flat.numel()]
        want = [False] * 4
        if ctx.have_heads:
            want = [bool(ctx.needs_input_grad[i]) for i in (3, 4, 5, 6)]
            sizes += [hw_c.numel() if want[0] else 0, hw_c.shape[0] if want[1] else 0, hw_e.numel() if want[2] else 0, hw_e.shape[0] if want[3] else 0]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 3) // 4 * 4
        zbuf = torch.zeros(tot, device=flat.device, dtype=torch.float32)
        gflat = zbuf[:flat.numel()]
        df_c = torch.empty_like(f_cxr) if need_df else None
        df_e = torch.empty_like(f_enh) if need_df else None
        dhw_c = dhb_c = dhw_e = dhb_e = None
        if ctx.have_heads:
            if want[0]:
                dhw_c = zbuf[offs[1]:offs[1] + hw_c.numel()].view_as(hw_c)
            if want[1]:
                dhb_c = zbuf[offs[2]:offs[2] + hw_c.shape[0]]
            if want[2]:
                dhw_e = zbuf[offs[3]:offs[3] + hw_e.numel()].view_as(hw_e)
            if want[3]:
                dhb_e = zbuf[offs[4]:offs[4] + hw_e.shape[0]]
        dfused = dfused.contiguous().float()
        dx_c = dx_c.contiguous().float() if ctx.have_heads else None
        dx_e = dx_e.contiguous().float() if ctx.have_heads else None
        check(lib().mfvit_fusion_backward(cfg, ptr(flat), ptr(f_cxr), ptr(f_enh), ptr(hw_c) if ctx.have_heads else None,
                                          ptr(hw_e) if ctx.have_heads else None, ptr(ctx.ws), ptr(dfused), ptr(dx_c), ptr(dx_e),
                                          ptr(gflat), ptr(df_c), ptr(df_e), ptr(dhw_c), ptr(dhb_c), ptr(dhw_e), ptr(dhb_e), stream()),
              "mfvit_fusion_backward")
        ctx.ws = None
        return (None, df_c, df_e, dhw_c, dhb_c, dhw_e, dhb_e) + tuple(arena.grad_views(gflat))
