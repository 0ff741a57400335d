"""Oracle: ViT-S/16 backbone (moco-v3 ``vits.vit_small`` on timm-0.4.9 ``VisionTransformer``).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for this file: the backbone
source is not in the reference tree.  Reference call sites that constrain it:
  main_vit_covid_..._vitsmall.py:276,711  (``vits.__dict__[arch]()``, ``model(images)``)
  main_vit_covid_..._crossvit_..._sum.py:289-290
  moco/model/crossvit_2vits_..._sum.py:80,83,128-135  (``features3D`` -> (B,197,384), ``__call__`` -> (B,C))
  moco/moco/builder_vit_mocov3structure_mocov2loss.py:29-30,164,174,217-222 (``num_classes``, ``.head``)
  moco/model/crossvit.py:130-146 (commented ``vit_features``: patch_embed -> cat cls -> +pos -> blocks -> norm)

Everything is written functionally over a ``dict[str, Tensor]`` keyed by the timm parameter names
(cls_token, pos_embed, patch_embed.proj.*, blocks.i.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*,
norm.*, head.*) so the same dict can be loaded into the product module with ``load_state_dict``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ---- optional OPERAND-ROUNDING mode (fp16 / bf16 arithmetic of the 16-bit kernels restated on the CPU) -----------------------------
# Inside ``with rounded_matmul(torch.float16):`` every matrix product of this file (and of ref_moco's MLPs) rounds BOTH operands to
# that type first and accumulates exactly (the caller's float64); the backward products autograd derives round their operands the same
# way (dY included).  That is the arithmetic of the reference's autocast path (MAIN_MOCO:349,533: Linear / matmul in fp16, everything
# else in fp32) and of the HIP 'fp16' mode (operand tensors stored in fp16, f32 accumulate, f32 residual stream / LayerNorm / softmax).
# Used by tests/test_moco_gpu.py to bound the fp16 gradients against an oracle that makes the SAME roundings, instead of only
# against float64 (where the ill-conditioned step shows ~10 % rounding-induced difference).
_MM_DTYPE = None


class _RoundedMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, dt):
        ra, rb = a.to(dt).to(a.dtype), b.to(dt).to(b.dtype)
        ctx.save_for_backward(ra, rb)
        ctx.dt, ctx.sa, ctx.sb = dt, a.shape, b.shape
        return ra @ rb

    @staticmethod
    def backward(ctx, g):
        ra, rb = ctx.saved_tensors
        rg = g.to(ctx.dt).to(g.dtype)
        da = (rg @ rb.transpose(-1, -2)).sum_to_size(ctx.sa) if ctx.needs_input_grad[0] else None
        db = (ra.transpose(-1, -2) @ rg).sum_to_size(ctx.sb) if ctx.needs_input_grad[1] else None
        return da, db, None


def mm(a, b):
    """a @ b, with both operands rounded to the active ``rounded_matmul`` type (if any)."""
    return a @ b if _MM_DTYPE is None else _RoundedMM.apply(a, b, _MM_DTYPE)


class rounded_matmul:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _MM_DTYPE
        self.prev, _MM_DTYPE = _MM_DTYPE, self.dtype
        return self

    def __exit__(self, *exc):
        global _MM_DTYPE
        _MM_DTYPE = self.prev


EMBED = {"vit_small": 384, "vit_base": 768}
DEPTH = 12
HEADS = 12
PATCH = 16
LN_EPS = 1e-6


def sincos_pos_embed(gh, gw, dim, temperature=10000.0, dtype=torch.float32):
    """Fixed 2-D sin-cos position embedding with a zero cls slot (moco-v3
    ``build_2d_sincos_position_embedding``).  Computed in float64 then cast so
    the product (which does the same) matches bit for bit."""
    assert dim % 4 == 0
    a = torch.arange(gw, dtype=torch.float64)
    b = torch.arange(gh, dtype=torch.float64)
    grid_w, grid_h = torch.meshgrid(a, b, indexing="ij")  # upstream uses the (old default) ij indexing
    pos_dim = dim // 4
    omega = torch.arange(pos_dim, dtype=torch.float64) / pos_dim
    omega = 1.0 / (temperature ** omega)
    out_w = grid_w.flatten()[:, None] * omega[None, :]
    out_h = grid_h.flatten()[:, None] * omega[None, :]
    pe = torch.cat([out_w.sin(), out_w.cos(), out_h.sin(), out_h.cos()], dim=1)[None]
    return torch.cat([torch.zeros(1, 1, dim, dtype=torch.float64), pe], dim=1).to(dtype)


def param_shapes(arch="vit_small", num_classes=1000, img_size=224, depth=DEPTH):
    """Ordered (name, shape) list in timm registration order (cls_token, pos_embed, patch_embed,
    blocks, norm, head) - the order ``zip(base.parameters(), momentum.parameters())`` relies on
    (builder_vit_mocov3structure_mocov2loss.py:52,88)."""
    d = EMBED[arch]
    t = (img_size // PATCH) ** 2 + 1
    out = [("cls_token", (1, 1, d)), ("pos_embed", (1, t, d)),
           ("patch_embed.proj.weight", (d, 3, PATCH, PATCH)), ("patch_embed.proj.bias", (d,))]
    for i in range(depth):
        p = f"blocks.{i}."
        out += [(p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)),
                (p + "attn.qkv.weight", (3 * d, d)), (p + "attn.qkv.bias", (3 * d,)),
                (p + "attn.proj.weight", (d, d)), (p + "attn.proj.bias", (d,)),
                (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,)),
                (p + "mlp.fc1.weight", (4 * d, d)), (p + "mlp.fc1.bias", (4 * d,)),
                (p + "mlp.fc2.weight", (d, 4 * d)), (p + "mlp.fc2.bias", (d,))]
    out += [("norm.weight", (d,)), ("norm.bias", (d,))]
    if num_classes:
        out += [("head.weight", (num_classes, d)), ("head.bias", (num_classes,))]
    return out


def seeded_params(seed, arch="vit_small", num_classes=3, img_size=224, depth=DEPTH, dtype=torch.float32):
    """Deterministic, platform-independent weights for fixtures: numpy PCG64 stream ``seed``;
    matrices ~ N(0, 0.02 * 2.5) (large enough that attention is not uniform), LN weight ~ 1 + N(0,.1),
    biases ~ N(0,.05), cls ~ N(0,.02); pos_embed is the fixed sin-cos table.  Recipe, not trained weights."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d = EMBED[arch]
    g = img_size // PATCH
    out = {}
    for name, shape in param_shapes(arch, num_classes, img_size, depth):
        if name == "pos_embed":
            out[name] = sincos_pos_embed(g, g, d, dtype=dtype)
            continue
        x = rng.standard_normal(size=shape, dtype=np.float64)
        if name == "cls_token":
            x *= 0.02
        elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
            x = 1.0 + 0.1 * x
        elif name.endswith(".bias"):
            x *= 0.05
        else:
            x *= 0.05
        out[name] = torch.from_numpy(x).to(dtype)
    return out


def layer_norm(x, w, b, eps):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def patch_embed(p, img):
    """Conv2d(3, D, k=16, s=16) as an explicit patch GEMM: (B,3,H,W) -> (B, N, D)."""
    B, C, H, W = img.shape
    gh, gw = H // PATCH, W // PATCH
    w = p["patch_embed.proj.weight"]
    D = w.shape[0]
    x = img.reshape(B, C, gh, PATCH, gw, PATCH).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * PATCH * PATCH)
    return mm(x, w.reshape(D, -1).t()) + p["patch_embed.proj.bias"]


def mhsa(p, pre, y, heads):
    """timm-0.4.9 ``Attention.forward`` (same math as moco/model/module.py:52-64):
    qkv Linear(D,3D,bias) -> (3,B,h,T,d); softmax(q k^T d^-1/2) v; proj."""
    B, T, D = y.shape
    d = D // heads
    qkv = mm(y, p[pre + "attn.qkv.weight"].t()) + p[pre + "attn.qkv.bias"]
    qkv = qkv.reshape(B, T, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = mm(q, k.transpose(-2, -1)) * (d ** -0.5)
    a = a.softmax(dim=-1)
    o = mm(a, v).transpose(1, 2).reshape(B, T, D)
    return mm(o, p[pre + "attn.proj.weight"].t()) + p[pre + "attn.proj.bias"]


def block(p, i, x, heads):
    pre = f"blocks.{i}."
    x = x + mhsa(p, pre, layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], LN_EPS), heads)
    y = layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], LN_EPS)
    h = gelu_erf(mm(y, p[pre + "mlp.fc1.weight"].t()) + p[pre + "mlp.fc1.bias"])
    return x + mm(h, p[pre + "mlp.fc2.weight"].t()) + p[pre + "mlp.fc2.bias"]


def depth_of(p):
    i = 0
    while f"blocks.{i}.norm1.weight" in p:
        i += 1
    return i


def features3d(p, img, heads=HEADS):
    """``features3D`` (crossvit_2vits_..._sum.py:128: "b, 197, 384"; spec crossvit.py:130-146):
    patch_embed -> cat(cls) -> + pos_embed -> blocks -> norm; returns ALL tokens."""
    B = img.shape[0]
    x = patch_embed(p, img)
    x = torch.cat([p["cls_token"].expand(B, -1, -1), x], dim=1) + p["pos_embed"]
    for i in range(depth_of(p)):
        x = block(p, i, x, heads)
    return layer_norm(x, p["norm.weight"], p["norm.bias"], LN_EPS)


def head_linear(p, cls):
    return cls @ p["head.weight"].t() + p["head.bias"]


def forward(p, img, heads=HEADS):
    """``model(img)`` = head(features3D(img)[:, 0])  (timm forward_features + head; dropouts 0)."""
    return head_linear(p, features3d(p, img, heads)[:, 0])


def single_stream_loss(p, img, target, heads=HEADS):
    """Single-stream step, BASELINE config #1 (main_vit_covid_..._vitsmall.py:711-714)."""
    logits = forward(p, img, heads)
    return logits, F.cross_entropy(logits, target)
