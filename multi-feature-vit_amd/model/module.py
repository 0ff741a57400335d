"""Drop-in for the reference's ``model/module.py`` (moco_pretraining/moco/model/module.py): same class names and
constructor signatures, parameters under the same attribute names (-> same state-dict keys).

On the accelerated path these classes are PARAMETER CONTAINERS: ``Fus_CrossViT`` (the only live user, FUS:22-33) runs
PreNorm -> CrossAttention -> residual -> LayerNorm of both directions as one fused f32 HIP pipeline
(csrc/fusion.hip) and never calls the per-module ``forward``.  A stand-alone ``PreNorm(dim, CrossAttention(..))(x)``
call runs the same kernels on a single direction (mfvit/xattn.py -> mfvit_prenorm_xattn_forward / _backward).  ``Residual`` / ``FeedForward`` / ``Attention`` are dead code in
the reference (defined MOD:8-64, instantiated by no live path); the names exist because FUS:6 imports them.
"""
import torch
import torch.nn as nn


class Residual(nn.Module):  # MOD:8-13
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(x, **kwargs) + x


class PreNorm(nn.Module):
    """MOD:15-21: ``fn(LayerNorm(x))`` with nn.LayerNorm's default eps = 1e-5."""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        if isinstance(self.fn, CrossAttention):
            from mfvit.xattn import prenorm_cross_attention
            return prenorm_cross_attention(self, x)
        raise NotImplementedError("PreNorm is accelerated only around CrossAttention (its sole live use, FUS:25,30)")


class _DeadCode(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError(f"{type(self).__name__} is dead code in the reference (instantiated by no live path, "
                                  "SURVEY.md §2 row 2) and is not built")


class FeedForward(_DeadCode):  # MOD:23-34
    pass


class Attention(_DeadCode):  # MOD:36-64
    pass


class CrossAttention(nn.Module):
    """MOD:108-137: single-query cross attention; wq/wk/wv bias-free by default, proj with bias; dropouts must be 0."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        if attn_drop or proj_drop:
            raise NotImplementedError("dropout > 0 is not used by the reference (FUS:16 default 0.) and is not built")
        if qkv_bias or qk_scale is not None:
            raise NotImplementedError("qkv_bias / qk_scale are not used by the reference's live path (MOD:109 defaults)")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.wq = nn.Linear(dim, dim, bias=False)
        self.wk = nn.Linear(dim, dim, bias=False)
        self.wv = nn.Linear(dim, dim, bias=False)
        self.attn_drop = nn.Dropout(0.)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.)

    def forward(self, x):
        """MOD:123-137: (B, N, C) -> (B, 1, C).  The reference never calls the module without its PreNorm (FUS:25,30); when it is,
        the same folded kernels run with the normalisation switched off."""
        from mfvit.xattn import cross_attention
        return cross_attention(self, x)
