"""TEST INFRASTRUCTURE ONLY - CPU restatement of the TransFuser fusion (reference moco_pretraining/moco/model/fuseattention.py).

Functional, parameter-dict based (keys = the reference module's state-dict keys), any float dtype:
  gpt_forward       fuseattention.py:84-212 (ViT branch :183-184, 186-192, 207-208) with SelfAttention :40-58 and Block :75-82;
                    dropouts are the identity (eval mode / p = 0) unless `drop` hands in the keep masks of a training-mode forward:
                    nn.Dropout(p)(x) == x * keep / (1 - p) for the mask it drew (embd :187, attn :52, resid :57, mlp :71)
  transfuser_logits fuseattention.py:280-320 (Encoder, ViT branch) + :386-393 (TransFuser.output)
Pinned against the reference's own GPT / TransFuser classes by tests/golden/transfuser.npz (oracle/make_golden.py imports the
reference file with an in-memory `torchvision.models` module object, which only the out-of-scope CNN classes further down touch).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def gpt_param_shapes(n_embd=384, block_exp=3, n_layer=8, n_tokens=394, prefix=""):
    s = {prefix + "pos_emb": (1, n_tokens, n_embd)}
    for i in range(n_layer):
        b = f"{prefix}blocks.{i}."
        for ln in ("ln1", "ln2"):
            s[b + ln + ".weight"] = (n_embd,)
            s[b + ln + ".bias"] = (n_embd,)
        for lin in ("key", "query", "value", "proj"):
            s[b + f"attn.{lin}.weight"] = (n_embd, n_embd)
            s[b + f"attn.{lin}.bias"] = (n_embd,)
        s[b + "mlp.0.weight"] = (block_exp * n_embd, n_embd)
        s[b + "mlp.0.bias"] = (block_exp * n_embd,)
        s[b + "mlp.2.weight"] = (n_embd, block_exp * n_embd)
        s[b + "mlp.2.bias"] = (n_embd,)
    s[prefix + "ln_f.weight"] = (n_embd,)
    s[prefix + "ln_f.bias"] = (n_embd,)
    return s


def seeded_gpt_params(seed, dtype=torch.float32, **kw):
    """Deterministic non-trivial parameters (numpy PCG64, one stream, dict order of gpt_param_shapes)."""
    g = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in gpt_param_shapes(**kw).items():
        x = g.standard_normal(size=shape)
        if name.endswith("pos_emb"):
            x = 0.1 * x
        elif len(shape) == 1:
            x = (1.0 if ("ln" in name and name.endswith("weight")) else 0.0) + 0.1 * x
        else:
            x = x / math.sqrt(shape[1])
        out[name] = torch.from_numpy(x).to(dtype)
    return out


def _drop(x, drop, key, pdrop):
    """nn.Dropout with a GIVEN keep mask (drop: {key: bool tensor}; absent / p = 0: identity)."""
    if drop is None or key not in drop or not pdrop:
        return x
    return x * drop[key].to(x.dtype) / (1.0 - pdrop)


def self_attention(p, pre, x, n_head, drop=None, layer=0, pdrops=(0.0, 0.0, 0.0)):
    B, T, C = x.shape
    hs = C // n_head
    k = F.linear(x, p[pre + "key.weight"], p[pre + "key.bias"]).view(B, T, n_head, hs).transpose(1, 2)       # :44
    q = F.linear(x, p[pre + "query.weight"], p[pre + "query.bias"]).view(B, T, n_head, hs).transpose(1, 2)   # :45
    v = F.linear(x, p[pre + "value.weight"], p[pre + "value.bias"]).view(B, T, n_head, hs).transpose(1, 2)   # :46
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hs))                                                   # :49
    att = F.softmax(att, dim=-1)                                                                              # :50
    att = _drop(att, drop, ("attn", layer), pdrops[1])                                                        # :52 attn_drop
    y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)                                                  # :52-53
    return _drop(F.linear(y, p[pre + "proj.weight"], p[pre + "proj.bias"]), drop, ("proj", layer), pdrops[2])   # :56-57 resid_drop


def gpt_forward(p, cxr, enh, n_head=4, pos_embed=True, prefix="", drop=None, pdrops=(0.0, 0.0, 0.0)):
    """drop: {"embd": (B,T,C), ("attn", l): (B,H,T,T), ("proj", l): (B,T,C), ("mlp", l): (B,T,C)} bool keep masks; pdrops = (embd, attn, resid)."""
    ftrs = cxr.shape[1]
    x = torch.cat([cxr, enh], dim=1)                                                                          # :184
    if pos_embed:
        x = p[prefix + "pos_emb"] + x                                                                         # :187
    x = _drop(x, drop, "embd", pdrops[0])                                                                     # :187 self.drop(...) (both branches of :186-189)
    C = x.shape[-1]
    i = 0
    while f"{prefix}blocks.{i}.ln1.weight" in p:
        b = f"{prefix}blocks.{i}."
        x = x + self_attention(p, b + "attn.", F.layer_norm(x, (C,), p[b + "ln1.weight"], p[b + "ln1.bias"], 1e-5), n_head, drop, i, pdrops)    # :78
        h = F.layer_norm(x, (C,), p[b + "ln2.weight"], p[b + "ln2.bias"], 1e-5)
        m = F.linear(F.relu(F.linear(h, p[b + "mlp.0.weight"], p[b + "mlp.0.bias"])), p[b + "mlp.2.weight"], p[b + "mlp.2.bias"])
        x = x + _drop(m, drop, ("mlp", i), pdrops[2])                                                         # :79 (mlp[3] = nn.Dropout(resid_pdrop), :71)
        i += 1
    x = F.layer_norm(x, (C,), p[prefix + "ln_f.weight"], p[prefix + "ln_f.bias"], 1e-5)                        # :192
    return x[:, :ftrs], x[:, ftrs:]                                                                           # :207-208


def transfuser_logits(p, feat_cxr, feat_enh, n_head=4, pos_embed=True):
    """p: TransFuser state dict ('encoder.transformer4.*', 'output.*'); feat_*: features3D of the two streams (B, 197, 384)."""
    a, b = gpt_forward(p, feat_cxr, feat_enh, n_head, pos_embed, prefix="encoder.transformer4.")
    fused = (feat_cxr + a)[:, 0] + (feat_enh + b)[:, 0]                                                       # :307-320
    return F.linear(fused, p["output.weight"], p["output.bias"])                                              # :393
