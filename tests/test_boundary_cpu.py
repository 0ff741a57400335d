"""CPU: the C-ABI boundary and host logic - the shared library loads without a GPU, exports every symbol
include/mfvit.h declares (and the ctypes table binds exactly those), layouts agree between C and Python, the
drop-in modules expose the reference's names / state-dict keys, and the product never touches the oracle."""
import importlib
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multi-feature-vit_amd")
FUS_MOD = ("model.crossvit_2vits_2additionaloutputs_changenormlayer_location_removeextralclayer_"
           "changemodelinputlocation_std002_sum")


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mfvit.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mfvit_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from mfvit import _lib
    h = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(h, s), f"{s} declared in include/mfvit.h but not exported"
    assert sorted(_lib.SIGNATURES.keys()) == syms, "ctypes table and header disagree"
    assert h.mfvit_abi_version() == 5
    assert b"gfx950" in h.mfvit_build_info()


def test_vit_layouts_agree_between_c_and_python():
    import ctypes
    import vits
    from mfvit import _lib
    m = vits.vit_small(num_classes=3)
    cfg = m._cfg(torch.zeros(2, 3, 224, 224), True)
    h = _lib.lib()
    assert h.mfvit_vit_param_count(cfg) == m.flat_parameters().numel() == 21665664
    out = (ctypes.c_int64 * 9)()
    assert h.mfvit_vit_param_layout(cfg, out) == 0
    assert out[0] == m.arena_slice("cls_token")[0] and out[1] == m.arena_slice("pos_embed")[0]
    assert out[2] == m.arena_slice("patch_embed.proj.weight")[0] and out[3] == m.arena_slice("patch_embed.proj.bias")[0]
    assert out[4] == m.arena_slice("blocks.0.norm1.weight")[0]
    assert out[5] == m.arena_slice("blocks.1.norm1.weight")[0] - out[4] == 1774464
    assert out[6] == m.arena_slice("norm.weight")[0] and out[8] == 21665664
    assert m.block_slice(3) == (out[4] + 3 * out[5], out[5])
    assert h.mfvit_vit_workspace_bytes(cfg) > 0 and h.mfvit_vit_shadow_bytes(cfg) > 0
    bad = m._cfg(torch.zeros(2, 3, 224, 220), True)
    assert h.mfvit_vit_workspace_bytes(bad) == 0                                          # invalid config -> 0, not a crash


def test_vits_drop_in_surface():
    import vits
    import vits_returnftrs
    for name in ("vit_small", "vit_base", "vit_conv_small", "vit_conv_base"):             # MAIN_MOCO:50
        assert callable(vits.__dict__[name]) and callable(vits_returnftrs.__dict__[name])
    m = vits_returnftrs.__dict__["vit_small"](num_classes=4096, stop_grad_conv1=True)     # BLD:29-30, MAIN_MOCO:274
    assert isinstance(m.head, torch.nn.Linear) and m.head.in_features == 384 and m.head.weight.shape[1] == 384
    assert not m.patch_embed.proj.weight.requires_grad and not m.pos_embed.requires_grad
    assert callable(m.features3D)
    keys = list(m.state_dict().keys())
    assert keys[:4] == ["cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"]
    assert "blocks.11.mlp.fc2.bias" in keys and keys[-4:] == ["norm.weight", "norm.bias", "head.weight", "head.bias"]
    m.head = torch.nn.Linear(m.head.in_features, 3)                                       # MAIN_CA:309
    del m.head                                                                            # BLD:218
    m.head = torch.nn.Sequential(torch.nn.Linear(384, 8))
    a = vits.vit_small()
    b = vits.vit_small()
    assert [tuple(p.shape) for p in a.parameters()] == [tuple(p.shape) for p in b.parameters()]   # BLD:52,88 zip order
    vb = vits.vit_base(num_classes=3, depth=1)                                            # MAIN_MOCO:50 `-a vit_base`: 768 wide, head_dim 64
    assert vb.head.in_features == 768 and vb.blocks[0].attn.qkv.weight.shape == (2304, 768) and vb.blocks[0].mlp.fc1.weight.shape == (3072, 768)
    with pytest.raises(NotImplementedError):
        vits.vit_conv_small()
    # MoCo checkpoint prefix handling (MAIN_SS:327-337): strict=False load leaves exactly the head missing
    sd = {k: v for k, v in a.state_dict().items() if not k.startswith("head")}
    msg = vits.vit_small(num_classes=3).load_state_dict(sd, strict=False)
    assert set(msg.missing_keys) == {"head.weight", "head.bias"}


def test_fusion_drop_in_surface():
    import vits_returnftrs as vits
    from mfvit import _lib
    from mfvit.fusion import fusion_cfg
    fus = importlib.import_module(FUS_MOD)
    mod = importlib.import_module("model.module")
    for n in ("PreNorm", "CrossAttention", "Attention", "FeedForward"):                   # FUS:6
        assert hasattr(mod, n)
    a, b = vits.vit_small(num_classes=3), vits.vit_small(num_classes=3)
    model = fus.Fus_CrossViT(a, b)                                                        # MAIN_CA:393
    sd = model.state_dict()
    assert len(sd) == 22 and sum(p.numel() for p in model.parameters()) == 1185798
    assert not any(k.startswith(("vit", "cxr", "enh")) for k in sd)
    L = "multi_scale_transformers.0.cross_attn_layers.0."
    assert list(sd.keys())[:9] == [L + "0.norm.weight", L + "0.norm.bias", L + "0.fn.wq.weight", L + "0.fn.wk.weight",
                                   L + "0.fn.wv.weight", L + "0.fn.proj.weight", L + "0.fn.proj.bias", L + "1.weight", L + "1.bias"]
    assert list(sd.keys())[-4:] == ["mlp_head_cxr.0.weight", "mlp_head_cxr.0.bias", "mlp_head_enh.0.weight", "mlp_head_enh.0.bias"]
    assert _lib.lib().mfvit_fusion_param_count(fusion_cfg(4, 197, 3)) == model.flat_parameters().numel()
    assert model._arena.intact()
    assert model.multi_scale_transformers[0].cross_attn_layers[0][0].norm.eps == 1e-5     # MOD:18
    assert model.multi_scale_transformers[0].cross_attn_layers[0][1].eps == 1e-6          # FUS:26
    with pytest.raises(_lib.MfvitError):
        model(a, b, torch.zeros(1, 3, 224, 224), torch.zeros(1, 3, 224, 224))             # CPU tensors: loud failure


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
                assert "ref_vit" not in text and "ref_fusion" not in text and "ref_moco" not in text, f
