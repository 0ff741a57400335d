"""Drop-in for the ``vits`` module the reference imports (MAIN_MOCO:39, MAIN_SS:44) but does not ship
(facebookresearch/moco-v3 ``vits.py`` on timm ``VisionTransformer``).

``vits.__dict__[arch](**kw)`` (MAIN_SS:276, MAIN_MOCO:274) returns a module whose encoder runs on the hand-written
gfx950 kernels of libmfvit_hip.so.  ``vit_small`` is the hot path: its 384-wide rows run on the row-complete GEMM kernels
with fused LayerNorm epilogues.  ``vit_base`` (MAIN_MOCO:50) runs on the same tile GEMM / attention / weight-gradient kernels with the residual +
LayerNorm stages as separate row passes (csrc/vit.hip, "unfused").  The conv-stem names exist so that ``-a`` parsing works and fail loudly when called.
"""
from functools import partial  # noqa: F401  (the reference wraps constructors in functools.partial)

from mfvit.encoder import VisionTransformerMoCo

__all__ = ["vit_small", "vit_base", "vit_conv_small", "vit_conv_base", "vit_small_ori", "vit_base_ori"]


def vit_small(**kwargs):
    """ViT-S/16: embed 384, depth 12, 12 heads, mlp 4x, qkv bias, LN eps 1e-6 (moco-v3 ``vit_small``).
    kwargs: num_classes (BLD:29-30), stop_grad_conv1 (MAIN_MOCO:274), img_size, precision ('bf16x3' default | 'bf16' | 'fp16' | 'fp32')."""
    cfg = dict(patch_size=16, embed_dim=384, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True)
    cfg.update(kwargs)
    return VisionTransformerMoCo(**cfg)


def vit_small_ori(**kwargs):  # listed in MAIN_CA:56-57, defined nowhere in the reference: alias of vit_small
    return vit_small(**kwargs)


def _out_of_scope(name):
    def ctor(**_kwargs):
        raise NotImplementedError(f"{name} is not on the accelerated hot path (only `-a vit_small` is; SURVEY.md §8)")
    ctor.__name__ = name
    return ctor


def vit_base(**kwargs):
    """ViT-B/16: embed 768, depth 12, 12 heads (head_dim 64), mlp 4x (moco-v3 ``vit_base``); same kwargs as vit_small."""
    cfg = dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True)
    cfg.update(kwargs)
    return VisionTransformerMoCo(**cfg)


def vit_base_ori(**kwargs):  # listed in MAIN_CA:56-57, defined nowhere in the reference: alias of vit_base
    return vit_base(**kwargs)


vit_conv_small = _out_of_scope("vit_conv_small")
vit_conv_base = _out_of_scope("vit_conv_base")
