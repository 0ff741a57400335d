#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (CPU) in the build container.

TEST INFRASTRUCTURE ONLY.  Run from the repo root:   python oracle/make_golden.py
Needs /root/reference (read-only; imported with sys.dont_write_bytecode).  The reference cannot travel
to the GPU box, so only the produced vectors (inputs are re-derived from numpy PCG64 seeds, outputs are
stored) are committed.  No reference source text is stored in the fixtures.

What is exercised from the reference (paths relative to /root/reference/moco_pretraining/moco):
  model/module.py                      PreNorm, CrossAttention              (imported as is)
  model/crossvit_2vits_..._sum.py      MultiScaleTransformerEncoder, Fus_CrossViT
        imports ``timm.models.layers`` only for the initialiser ``trunc_normal_`` (its line 9/119); timm is
        not installed, so a 3-symbol in-memory module providing torch's own ``trunc_normal_`` is registered
        (SURVEY.md §8c).  Every weight is overwritten with seeded values afterwards, so the initialiser
        cannot influence any stored number.
  moco/builder_vit_mocov3structure_mocov2loss.py   MoCo._build_mlp, _momentum_update_key_encoder,
        _dequeue_and_enqueue (1-rank gloo group), concat_all_gather
  moco/builder_vit.py                  MoCo.contrastive_loss, MoCo_ViT.forward (symmetric MoCo-v3 loss; 1-rank gloo group,
        torch.Tensor.cuda mapped to the identity while it runs because builder_vit.py:93 calls .cuda() on the labels)
  moco/builder_vit_mocov3structure_mocov2loss.py   MoCo.forward itself (its lines 154-199) on a toy encoder: same 1-rank gloo group and
        Tensor.cuda -> identity shim (its :121, :194 call .cuda())  -> moco_forward.npz
  model/fuseattention.py               GPT, Encoder (ViT branch), TransFuser (eval mode) -> transfuser.npz; the file's module-level
        ``from torchvision import models`` (torchvision absent: ordinary ModuleNotFoundError) is satisfied by a bare in-memory module object
        that none of the exercised classes touch
  moco/optimizer.py                    LARS
The ViT backbone is absent from the reference; where a backbone is needed the oracle's own restatement
(oracle/ref_vit.py, parity unpinned) is wrapped in an object exposing ``features3D`` / ``__call__``.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/moco_pretraining/moco"
sys.path.insert(0, REF)

from oracle import ref_fusion, ref_moco, ref_vit  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FUS_MOD = ("model.crossvit_2vits_2additionaloutputs_changenormlayer_location_removeextralclayer_"
           "changemodelinputlocation_std002_sum")


def rng_tensor(seed, shape, scale=1.0, dtype=torch.float32):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(size=shape, dtype=np.float64) * scale).to(dtype)


def sample(t, n=4096):
    """Strided subsample + moments, to keep fixtures small while still pinning a whole tensor."""
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return dict(idx=idx.numpy(), val=f[idx].numpy(), sum=np.float64(f.sum()), abssum=np.float64(f.abs().sum()))


def put(d, key, t, full=False, n=4096):
    if full:
        d[key] = t.detach().double().numpy()
    else:
        for k, v in sample(t, n).items():
            d[f"{key}.{k}"] = v


def install_timm_stub():
    m_timm = types.ModuleType("timm")
    m_models = types.ModuleType("timm.models")
    m_layers = types.ModuleType("timm.models.layers")
    m_layers.trunc_normal_ = torch.nn.init.trunc_normal_
    m_layers.DropPath = torch.nn.Identity
    m_layers.to_2tuple = lambda x: (x, x)
    sys.modules.setdefault("timm", m_timm)
    sys.modules.setdefault("timm.models", m_models)
    sys.modules.setdefault("timm.models.layers", m_layers)


def golden_cross_attention():
    from model.module import CrossAttention, PreNorm
    torch.manual_seed(0)
    B, T, D = 2, 197, 384
    fp = ref_fusion.seeded_fusion_params(101, dtype=torch.float64)
    mod = PreNorm(D, CrossAttention(D, num_heads=3)).double()
    L = ref_fusion._L
    sd = {"norm.weight": fp[L + "0.norm.weight"], "norm.bias": fp[L + "0.norm.bias"],
          "fn.wq.weight": fp[L + "0.fn.wq.weight"], "fn.wk.weight": fp[L + "0.fn.wk.weight"],
          "fn.wv.weight": fp[L + "0.fn.wv.weight"], "fn.proj.weight": fp[L + "0.fn.proj.weight"],
          "fn.proj.bias": fp[L + "0.fn.proj.bias"]}
    mod.load_state_dict(sd, strict=True)
    x = rng_tensor(102, (B, T, D), dtype=torch.float64).requires_grad_(True)
    r = rng_tensor(103, (B, 1, D), dtype=torch.float64)
    y = mod(x)
    (y * r).sum().backward()
    d = dict(seed_params=101, seed_x=102, seed_r=103)
    put(d, "y", y, full=True)
    put(d, "dx", x.grad)
    for n, p in mod.named_parameters():
        put(d, "d." + n, p.grad)
    np.savez_compressed(os.path.join(OUT, "fusion_cross_attention.npz"), **d)
    print("fusion_cross_attention.npz", y.abs().mean().item())


def golden_cross_attention_bare():
    """The reference's CrossAttention called WITHOUT its PreNorm (MOD:123-137; its live path never does that, FUS:25,30)."""
    from model.module import CrossAttention
    torch.manual_seed(0)
    B, T, D = 3, 197, 384
    fp = ref_fusion.seeded_fusion_params(111, dtype=torch.float64)
    mod = CrossAttention(D, num_heads=3).double()
    L = ref_fusion._L
    mod.load_state_dict({k: fp[L + "2.fn." + k] for k in ("wq.weight", "wk.weight", "wv.weight", "proj.weight", "proj.bias")}, strict=True)
    x = rng_tensor(112, (B, T, D), dtype=torch.float64).requires_grad_(True)
    r = rng_tensor(113, (B, 1, D), dtype=torch.float64)
    y = mod(x)
    (y * r).sum().backward()
    d = dict(seed_params=111, seed_x=112, seed_r=113)
    put(d, "y", y, full=True)
    put(d, "dx", x.grad)
    for n, p in mod.named_parameters():
        put(d, "d." + n, p.grad)
    np.savez_compressed(os.path.join(OUT, "fusion_cross_attention_bare.npz"), **d)
    print("fusion_cross_attention_bare.npz", y.abs().mean().item())


class OracleBackbone:
    """Stands in for the absent vits_returnftrs model (oracle restatement; parity unpinned)."""

    def __init__(self, params):
        self.p = params

    def features3D(self, img):
        return ref_vit.features3d(self.p, img)

    def __call__(self, img):
        return ref_vit.forward(self.p, img)

    def to(self, *_a, **_k):
        return self


def golden_fusion():
    install_timm_stub()
    import importlib
    fus = importlib.import_module(FUS_MOD)
    B, T, D = 2, 197, 384
    fp = ref_fusion.seeded_fusion_params(201, dtype=torch.float64)
    # (a) exchange alone on seeded feature tensors
    enc = fus.MultiScaleTransformerEncoder().double()
    pre = "multi_scale_transformers.0."
    enc.load_state_dict({k[len(pre):]: v for k, v in fp.items() if k.startswith(pre)}, strict=True)
    xs = rng_tensor(202, (B, T, D), dtype=torch.float64)
    xl = rng_tensor(203, (B, T, D), dtype=torch.float64)
    xs_o, xl_o = enc(xs, xl)
    d = dict(seed_params=201, seed_xs=202, seed_xl=203)
    put(d, "xs_out", xs_o)
    put(d, "xl_out", xl_o)
    put(d, "xs_out_cls", xs_o[:, 0], full=True)
    put(d, "xl_out_cls", xl_o[:, 0], full=True)
    np.savez_compressed(os.path.join(OUT, "fusion_exchange.npz"), **d)
    print("fusion_exchange.npz")

    # (b) Fus_CrossViT end to end on seeded feature providers + CA-step loss and fusion grads
    #     depth-2 oracle backbones keep generation fast; the fusion code under test is the reference's.
    vit_c = ref_vit.seeded_params(211, num_classes=3, depth=2, dtype=torch.float64)
    vit_e = ref_vit.seeded_params(212, num_classes=3, depth=2, dtype=torch.float64)
    model = fus.Fus_CrossViT(OracleBackbone(vit_c), OracleBackbone(vit_e)).double()
    model.load_state_dict(fp, strict=True)
    assert sorted(model.state_dict().keys()) == sorted(fp.keys())
    img_c = rng_tensor(213, (B, 3, 224, 224), dtype=torch.float64)
    img_e = rng_tensor(214, (B, 3, 224, 224), dtype=torch.float64)
    target = torch.tensor([2, 0])
    fused, x_cxr, x_enh = model(OracleBackbone(vit_c), OracleBackbone(vit_e), img_c, img_e)
    output = fused + x_cxr + x_enh                         # main_..._crossvit_..._sum.py:868
    loss = torch.nn.CrossEntropyLoss()(output, target)     # :432,873
    loss.backward()
    d = dict(seed_params=201, seed_vit_cxr=211, seed_vit_enh=212, seed_img_cxr=213, seed_img_enh=214,
             vit_depth=2, target=target.numpy())
    put(d, "fused", fused, full=True)
    put(d, "x_cxr", x_cxr, full=True)
    put(d, "x_enh", x_enh, full=True)
    put(d, "output", output, full=True)
    put(d, "loss", loss, full=True)
    d["preds"] = output.argmax(1).numpy()
    for n, p in model.named_parameters():
        put(d, "d." + n, p.grad)
    d["n_params"] = sum(p.numel() for p in model.parameters())
    d["keys"] = np.array(sorted(model.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "fusion_e2e.npz"), **d)
    print("fusion_e2e.npz loss", loss.item(), "n_params", d["n_params"])


def golden_moco():
    import torch.distributed as dist
    import moco.builder_vit_mocov3structure_mocov2loss as bld
    d = {}
    # (a) projector / predictor MLPs (train-mode BN) built by the reference's own _build_mlp
    n, hid, mlp_dim, dim = 8, 384, 512, 256      # mlp_dim shrunk from 4096 to keep the fixture small
    proj = bld.MoCo._build_mlp(None, 3, hid, mlp_dim, dim).double().train()
    pred = bld.MoCo._build_mlp(None, 2, dim, mlp_dim, dim).double().train()
    pp = ref_moco.seeded_mlp_params(301, "", 3, hid, mlp_dim, dim, dtype=torch.float64)
    qp = ref_moco.seeded_mlp_params(302, "", 2, dim, mlp_dim, dim, dtype=torch.float64)
    proj.load_state_dict(pp, strict=False)
    pred.load_state_dict(qp, strict=False)
    x = rng_tensor(303, (n, hid), dtype=torch.float64).requires_grad_(True)
    z = proj(x)
    q = pred(z)
    qn = torch.nn.functional.normalize(q, dim=1)
    r = rng_tensor(304, (n, dim), dtype=torch.float64)
    (qn * r).sum().backward()
    d.update(seed_proj=301, seed_pred=302, seed_x=303, seed_r=304, n=n, hid=hid, mlp_dim=mlp_dim, dim=dim)
    put(d, "proj_out", z, full=True)
    put(d, "pred_out", q, full=True)
    put(d, "q_norm", qn, full=True)
    put(d, "dx", x.grad, full=True)
    for name, p in proj.named_parameters():
        put(d, "dproj." + name, p.grad)
    for name, p in pred.named_parameters():
        put(d, "dpred." + name, p.grad)
    for name, b in list(proj.named_buffers()) + [("pred." + k, v) for k, v in pred.named_buffers()]:
        put(d, "buf." + name, b.double(), full=True)

    # (b) EMA + enqueue through a real MoCo_ViT instance built on a toy encoder
    class Toy(torch.nn.Module):
        def __init__(self, num_classes=1000, **_):
            super().__init__()
            self.body = torch.nn.Linear(12, 16)
            self.head = torch.nn.Linear(16, num_classes)

        def forward(self, x):
            return self.head(self.body(x))

    args = types.SimpleNamespace(arch="vit_small")
    torch.manual_seed(0)
    m = bld.MoCo_ViT(Toy, args, dim=256, mlp_dim=64, T=0.2)
    base = {k: rng_tensor(310 + i, v.shape) for i, (k, v) in enumerate(m.base_encoder.named_parameters())}
    mom = {k: rng_tensor(340 + i, v.shape) for i, (k, v) in enumerate(m.momentum_encoder.named_parameters())}
    with torch.no_grad():
        for k, v in m.base_encoder.named_parameters():
            v.copy_(base[k])
        for k, v in m.momentum_encoder.named_parameters():
            v.copy_(mom[k])
    m._momentum_update_key_encoder(0.99)
    d["ema_m"] = 0.99
    d["ema_names"] = np.array(list(base.keys()))
    d["ema_seed_base0"] = 310
    d["ema_seed_mom0"] = 340
    for k, v in m.momentum_encoder.named_parameters():
        put(d, "ema." + k, v, full=True)
    d["state_keys"] = np.array(sorted(m.state_dict().keys()))

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    K = m.K
    d["K"] = K
    q0 = rng_tensor(360, (256, K))
    m.queue.copy_(torch.nn.functional.normalize(q0, dim=0))
    m.queue_ptr[0] = K - 32                       # next enqueue wraps the pointer to 0
    keys = torch.nn.functional.normalize(rng_tensor(361, (32, 256)), dim=1)
    qv = torch.nn.functional.normalize(rng_tensor(362, (4, 256)), dim=1)
    kv = torch.nn.functional.normalize(rng_tensor(363, (4, 256)), dim=1)
    # logits exactly as builder:183-191 evaluates them (einsum forms), on the reference's own queue buffer
    l_pos = torch.einsum('nc,nc->n', [qv, kv]).unsqueeze(-1)
    l_neg = torch.einsum('nc,ck->nk', [qv, m.queue.clone().detach()])
    logits = torch.cat([l_pos, l_neg], dim=1)
    logits /= m.T
    put(d, "nce_logits", logits)
    put(d, "nce_loss", torch.nn.CrossEntropyLoss()(logits, torch.zeros(4, dtype=torch.long)), full=True)
    m._dequeue_and_enqueue(keys)
    d["enq_ptr_before"] = K - 32
    d["enq_ptr_after"] = int(m.queue_ptr)
    put(d, "enq_cols", m.queue[:, K - 32:], full=True)
    put(d, "enq_untouched", m.queue[:, :64], full=True)
    d.update(seed_queue=360, seed_keys=361, seed_q=362, seed_k=363, T=0.2)
    dist.destroy_process_group()
    np.savez_compressed(os.path.join(OUT, "moco_pieces.npz"), **d)
    print("moco_pieces.npz ptr", d["enq_ptr_after"])


def golden_moco_v3():
    """moco/builder_vit.py (the symmetric MoCo-v3 loss): MoCo.contrastive_loss and MoCo.forward of the reference on a
    toy encoder, 1-rank gloo group.  The reference moves its labels with `.cuda()` (builder_vit.py:93): Tensor.cuda is mapped to
    the identity while the reference code runs (torch-level shim, nothing of the reference is altered)."""
    import torch.distributed as dist
    import moco.builder_vit as bv

    class Toy(torch.nn.Module):
        def __init__(self, num_classes=1000, **_):
            super().__init__()
            self.body = torch.nn.Linear(12, 16)
            self.head = torch.nn.Linear(16, num_classes)

        def forward(self, x):
            return self.head(self.body(x))

    d = {}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=0, world_size=1)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        torch.manual_seed(0)
        m = bv.MoCo_ViT(Toy, types.SimpleNamespace(arch="vit_small"), dim=256, mlp_dim=64, T=0.2).double().train()
        d["state_keys"] = np.array(sorted(m.state_dict().keys()))
        # (a) the loss alone
        q = rng_tensor(501, (8, 256), dtype=torch.float64).requires_grad_(True)
        k = rng_tensor(502, (8, 256), dtype=torch.float64)
        loss = m.contrastive_loss(q, k)
        loss.backward()
        d.update(seed_q=501, seed_k=502, T=0.2, n=8)
        put(d, "ctr_loss", loss, full=True)
        put(d, "ctr_dq", q.grad, full=True)
        # (b) forward on seeded weights: loss, and the momentum encoder after its update
        with torch.no_grad():
            for i, (name, p) in enumerate(m.named_parameters()):
                p.copy_(rng_tensor(520 + i, p.shape, scale=0.3, dtype=torch.float64))
        d["param_names"] = np.array([n_ for n_, _ in m.named_parameters()])
        d["param_shapes"] = np.array([",".join(map(str, p.shape)) for _, p in m.named_parameters()])
        d["seed_param0"] = 520
        x1 = rng_tensor(511, (8, 12), dtype=torch.float64)
        x2 = rng_tensor(512, (8, 12), dtype=torch.float64)
        out = m(x1, x2, 0.99)
        d.update(seed_x1=511, seed_x2=512, m=0.99)
        put(d, "fwd_loss", out, full=True)
        for name, p in m.momentum_encoder.named_parameters():
            put(d, "mom." + name, p, full=True)
    finally:
        torch.Tensor.cuda = real_cuda
        dist.destroy_process_group()
    np.savez_compressed(os.path.join(OUT, "moco_v3.npz"), **d)
    print("moco_v3.npz loss", float(d["fwd_loss"]))


def golden_moco_forward():
    """MoCo.forward ITSELF (builder_vit_mocov3structure_mocov2loss.py:154-199: EMA, shuffle, key encoding with the shared predictor,
    unshuffle, logits against the queue, temperature, labels, enqueue) of the reference, on a toy encoder, under a 1-rank gloo group
    with torch.Tensor.cuda mapped to the identity while the reference code runs (its :121 and :194 call .cuda()).  Nothing
    of the reference is altered.  Toy widths are MFMA-tile friendly (128) so that the HIP builder can run the same case."""
    import torch.distributed as dist
    import moco.builder_vit_mocov3structure_mocov2loss as bld

    class Toy(torch.nn.Module):
        def __init__(self, num_classes=1000, **_):
            super().__init__()
            self.body = torch.nn.Linear(12, 128)
            self.head = torch.nn.Linear(128, num_classes)

        def forward(self, x):
            return self.head(self.body(x))

    d = {}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29535"
    dist.init_process_group("gloo", rank=0, world_size=1)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        torch.manual_seed(0)
        n, T, mval = 8, 0.2, 0.99
        m = bld.MoCo_ViT(Toy, types.SimpleNamespace(arch="vit_small"), dim=256, mlp_dim=128, T=T).double().train()
        d["state_keys"] = np.array(sorted(m.state_dict().keys()))
        with torch.no_grad():
            for i, (name, p) in enumerate(m.named_parameters()):
                if p.ndim == 1:      # BN affine / Linear bias: around 1 for BN weights, small otherwise
                    p.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * rng_tensor(540 + i, p.shape, dtype=torch.float64))
                else:
                    p.copy_(rng_tensor(540 + i, p.shape, dtype=torch.float64) / p.shape[1] ** 0.5)
            m.queue.copy_(torch.nn.functional.normalize(rng_tensor(539, tuple(m.queue.shape), dtype=torch.float64), dim=0))
            m.queue_ptr[0] = m.K - n                                   # the enqueue wraps the pointer to 0
        d["param_names"] = np.array([n_ for n_, _ in m.named_parameters()])
        d["param_shapes"] = np.array([",".join(map(str, p.shape)) for _, p in m.named_parameters()])
        d.update(seed_param0=540, seed_queue=539, seed_q=531, seed_k=532, n=n, T=T, m=mval, K=m.K, ptr_before=m.K - n)
        im_q = rng_tensor(531, (n, 12), dtype=torch.float64)
        im_k = rng_tensor(532, (n, 12), dtype=torch.float64)
        logits, labels = m(im_q, im_k, mval)
        loss = torch.nn.functional.cross_entropy(logits, labels)       # MAIN_MOCO:535
        loss.backward()
        put(d, "logits", logits)
        put(d, "logits_head", logits[:, :16], full=True)
        d["labels"] = labels.numpy()
        put(d, "loss", loss, full=True)
        d["ptr_after"] = int(m.queue_ptr)
        put(d, "queue_tail", m.queue[:, m.K - n:], full=True)          # the n enqueued keys
        put(d, "queue", m.queue)
        for name, p in m.momentum_encoder.named_parameters():
            put(d, "mom." + name, p)
        for name, p in list(m.base_encoder.named_parameters()) + [("predictor." + k_, v) for k_, v in m.predictor.named_parameters()]:
            put(d, "d." + name, p.grad)
        put(d, "pred_bn_running_mean", m.predictor[1].running_mean, full=True)   # updated TWICE per step (Q6): q pass and k pass
        put(d, "pred_bn_running_var", m.predictor[1].running_var, full=True)     # (the mean is ~0: its input is a mean-free BN output)
        d["pred_bn_batches"] = int(m.predictor[1].num_batches_tracked)
    finally:
        torch.Tensor.cuda = real_cuda
        dist.destroy_process_group()
    np.savez_compressed(os.path.join(OUT, "moco_forward.npz"), **d)
    print("moco_forward.npz loss", float(d["loss"]), "ptr", d["ptr_after"])


def golden_transfuser():
    """fuseattention.py: GPT.forward (8 blocks, 4 heads x 96, 394 joint tokens, ReLU MLP) and TransFuser.forward of the REFERENCE, eval
    mode (its dropouts are then the identity), on seeded parameters; the encoders are stand-ins exposing `features3D` / `head`
    (the backbone is absent from the reference).  The reference file imports torchvision (absent here: an ordinary
    ModuleNotFoundError) for its CNN classes only; a bare in-memory module object named `torchvision.models` is registered, which GPT /
    Encoder(ViT branch) / TransFuser never touch."""
    from oracle import ref_gpt
    m_tv = types.ModuleType("torchvision")
    m_tv.models = types.ModuleType("torchvision.models")
    sys.modules.setdefault("torchvision", m_tv)
    sys.modules.setdefault("torchvision.models", m_tv.models)
    from config.config import GlobalConfig
    from model import fuseattention as fa

    class Stream(torch.nn.Module):          # what Encoder / TransFuser touch of a backbone: features3D and head.in_features
        def __init__(self, feats):
            super().__init__()
            self.feats = feats
            self.head = torch.nn.Linear(384, 3)

        def features3D(self, x):
            return self.feats

    cfg = GlobalConfig()
    args = types.SimpleNamespace(arch="vit_small", pos_embed=True)
    B = 2
    fc = rng_tensor(601, (B, 197, 384), dtype=torch.float64).requires_grad_(True)
    fe = rng_tensor(602, (B, 197, 384), dtype=torch.float64).requires_grad_(True)
    torch.manual_seed(0)
    model = fa.TransFuser(Stream(fc), Stream(fe), cfg, args).double().eval()
    gp = ref_gpt.seeded_gpt_params(603, dtype=torch.float64, prefix="encoder.transformer4.")
    sd = dict(gp)
    sd["output.weight"] = rng_tensor(604, (3, 384), scale=0.05, dtype=torch.float64)
    sd["output.bias"] = rng_tensor(605, (3,), scale=0.05, dtype=torch.float64)
    missing = model.load_state_dict(sd, strict=True)
    d = dict(seed_fc=601, seed_fe=602, seed_gpt=603, seed_ow=604, seed_ob=605, B=B, n_head=cfg.n_head, n_layer=cfg.n_layer,
             block_exp=cfg.block_exp, n_tokens=int(model.encoder.transformer4.pos_emb.shape[1]))
    d["state_keys"] = np.array(sorted(model.state_dict().keys()))
    # (a) GPT alone
    a, b = model.encoder.transformer4(fc, fe)
    put(d, "gpt_cxr", a)
    put(d, "gpt_enh", b)
    put(d, "gpt_cxr_cls", a[:, 0], full=True)
    # (b) TransFuser logits + gradients wrt the features and every parameter
    img = torch.zeros(B, 3, 224, 224, dtype=torch.float64)
    logits = model(img, img)
    r = rng_tensor(606, (B, 3), dtype=torch.float64)
    (logits * r).sum().backward()
    d["seed_r"] = 606
    put(d, "logits", logits, full=True)
    put(d, "d.fc", fc.grad)
    put(d, "d.fe", fe.grad)
    for name, p in model.named_parameters():
        if p.grad is not None:
            put(d, "d." + name, p.grad, n=256)
    # (c) without the positional embedding (args.pos_embed False, fuseattention.py:188-189)
    args.pos_embed = False
    a2, _ = model.encoder.transformer4(fc, fe)
    put(d, "gpt_cxr_nopos_cls", a2[:, 0], full=True)
    np.savez_compressed(os.path.join(OUT, "transfuser.npz"), **d)
    print("transfuser.npz logits", logits.detach().numpy().ravel()[:3])


def golden_lars():
    from moco.optimizer import LARS
    shapes = [(6, 5), (5,), (4, 3), (3, 2)]
    ps = [torch.nn.Parameter(rng_tensor(400 + i, s)) for i, s in enumerate(shapes)]
    with torch.no_grad():
        ps[3].zero_()                                       # param_norm == 0 branch (optimizer.py:31-34)
    wd = 2.0 ** -6                                           # exact in binary: g = -wd*p cancels exactly
    opt = LARS(ps, lr=0.3, weight_decay=wd, momentum=0.9)
    d = dict(lr=0.3, weight_decay=wd, momentum=0.9, trust=0.001, n=len(shapes),
             shapes=np.array([str(s) for s in shapes]))
    for step in range(3):
        for i, p in enumerate(ps):
            p.grad = rng_tensor(410 + 10 * step + i, p.shape)
            if step == 1 and i == 2:
                p.grad.zero_()
                with torch.no_grad():
                    p.grad -= wd * p                      # update_norm == 0 branch
        opt.step()
        for i, p in enumerate(ps):
            d[f"p{i}.step{step}"] = p.detach().double().numpy()
    np.savez_compressed(os.path.join(OUT, "lars.npz"), **d)
    print("lars.npz")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                 # python oracle/make_golden.py golden_moco_forward ... : only the named generators
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    golden_cross_attention()
    golden_cross_attention_bare()
    golden_fusion()
    golden_moco()
    golden_moco_v3()
    golden_moco_forward()
    golden_transfuser()
    golden_lars()
