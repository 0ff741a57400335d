"""Engine plumbing of the TransFuser fusion transformer (fuseattention.py:84-212) on the encoder's gfx950 kernels.

The GPT of the reference is the same pre-LN block stack as the ViT (LayerNorm -> q/k/v -> softmax(q k^T / sqrt(d)) v -> proj -> residual;
LayerNorm -> Linear -> act -> Linear -> residual; final LayerNorm) with three differences the C ABI's token-input mode covers
(include/mfvit.h, mfvit_gpt_forward): tokens instead of image patches (+ learnable pos_emb), a ReLU MLP (fuseattention.py:69), and
separate query / key / value Linears - which are ONE packed [3 dim][dim] weight once their parameters are laid out back to back in
the flat arena.  4 heads x 96 (config.py:37-40) run on the streaming MFMA attention kernels (csrc/attention_tiled.hip).
"""
import torch

from . import _lib
from ._lib import VitCfg, check, lib, ptr, stream
from .arena import ParamArena


class _GptFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, tokens, need_grad, drop, *params):
        out, ws, cfg = eng.run_forward(tokens, need_grad, drop)
        ctx.eng, ctx.ws, ctx.cfg = eng, ws, cfg
        return out

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.eng
        if ctx.ws is None:
            raise _lib.MfvitError("the GPT's saved activations were already released: run the forward again")
        dtokens, grads = eng.run_backward(ctx.cfg, ctx.ws, dout)
        eng.release_ws(ctx.ws)
        ctx.ws = None
        return (None, dtokens, None, None) + tuple(grads)


class GptEngine:
    """Owns the flat arena (parameter order of include/mfvit.h), the weight shadow and the workspaces of one GPT module."""

    def __init__(self, named_params, dim, depth, heads, mlp_dim, tokens, use_pos, precision, ln_eps=1e-5):
        self.named = list(named_params)
        self.dim, self.depth, self.heads, self.mlp_dim, self.tokens, self.use_pos = dim, depth, heads, mlp_dim, tokens, bool(use_pos)
        self.precision = precision
        self.ln_eps = ln_eps
        if _lib.dtype_code(precision) == _lib.F32 and dim // heads not in (32, 64):
            raise NotImplementedError("head_dim 96 runs on the streaming MFMA attention kernels: use precision 'bf16x3' (f32-grade), "
                                      "'fp16' or 'bf16' for the TransFuser GPT")
        self.arena = ParamArena(self.named)
        self._shadow = None
        self._shadow_key = None
        self._pool = {}

    def cfg(self, batch, save, drop=None):
        """drop: None, or (p_embd, p_attn, p_resid, seed) for a training-mode forward (the backward re-uses the SAME cfg: the masks are
        functions of the seed in it)."""
        c = VitCfg()
        c.dtype = _lib.dtype_code(self.precision)
        c.batch, c.img_h, c.img_w = batch, 0, 0
        c.dim, c.depth, c.heads, c.mlp_dim = self.dim, self.depth, self.heads, self.mlp_dim
        c.save_for_backward = int(bool(save))
        c.stop_grad_conv1 = 0
        c.ln_eps = self.ln_eps
        c.token_input, c.tokens, c.use_pos, c.act = 1, self.tokens, int(self.use_pos), 1
        if drop is not None:
            c.p_embd, c.p_attn, c.p_resid = float(drop[0]), float(drop[1]), float(drop[2])
            c.seed_lo, c.seed_hi = int(drop[3]) & 0xFFFFFFFF, (int(drop[3]) >> 32) & 0xFFFFFFFF
        return c

    def _ensure(self, cfg):
        flat = self.arena.ensure()
        if flat.numel() != lib().mfvit_vit_param_count(cfg):
            raise _lib.MfvitError("GPT parameter arena does not match the C ABI's layout")
        key = (self.arena.version(), cfg.dtype, str(flat.device), flat.data_ptr())
        if self._shadow is None or key != self._shadow_key:
            nbytes = lib().mfvit_vit_shadow_bytes(cfg)
            if self._shadow is None or self._shadow.numel() != nbytes or self._shadow.device != flat.device:
                self._shadow = torch.empty(nbytes, device=flat.device, dtype=torch.uint8)
            check(lib().mfvit_vit_prepare_shadow(cfg, ptr(flat), ptr(self._shadow), stream()), "mfvit_vit_prepare_shadow")
            self._shadow_key = key
        return flat

    def release_ws(self, ws):
        pool = self._pool.setdefault(ws.numel(), [])
        if len(pool) < 2:
            pool.append(ws)

    def run_forward(self, tokens, save, drop=None):
        _lib.require_cuda(tokens)
        tokens = tokens.contiguous().float()
        B, T, D = tokens.shape
        if T != self.tokens or D != self.dim:
            raise _lib.MfvitError(f"expected (B, {self.tokens}, {self.dim}) tokens, got {tuple(tokens.shape)} (pos_emb is length-bound)")
        cfg = self.cfg(B, save, drop)
        flat = self._ensure(cfg)
        nbytes = lib().mfvit_vit_workspace_bytes(cfg)
        if nbytes == 0:
            raise _lib.MfvitError("invalid GPT configuration for the HIP encoder (dim 384, head_dim in {32, 64, 96}, mlp_dim % 128 == 0)")
        pool = self._pool.setdefault(nbytes, [])
        ws = pool.pop() if pool else torch.empty(nbytes, device=tokens.device, dtype=torch.uint8)
        out = torch.empty(B, T, D, device=tokens.device, dtype=torch.float32)
        check(lib().mfvit_gpt_forward(cfg, ptr(flat), ptr(self._shadow), ptr(tokens), ptr(ws), ptr(out), stream()), "mfvit_gpt_forward")
        if not save:
            self.release_ws(ws)
            ws = None
        return out, ws, cfg

    def run_backward(self, cfg, ws, dout):
        dout = dout.contiguous().float()
        flat = self.arena.flat
        gflat = torch.zeros_like(flat)
        dtokens = torch.empty_like(dout)
        check(lib().mfvit_gpt_backward(cfg, ptr(flat), ptr(self._shadow), ptr(ws), ptr(dout), ptr(gflat), ptr(dtokens), stream()),
              "mfvit_gpt_backward")
        return dtokens, self.arena.grad_views(gflat)

    def __call__(self, tokens, drop=None):
        """drop = (p_embd, p_attn, p_resid, seed): the training-mode dropout sites of the GPT (None / all zero: none)."""
        need = torch.is_grad_enabled() and (tokens.requires_grad or any(p.requires_grad for p in self.arena.params))
        if drop is not None and not any(float(p) > 0 for p in drop[:3]):
            drop = None
        return _GptFn.apply(self, tokens, need, drop, *self.arena.params)
