"""usage (GPU box): python tools/determinism_moco_bisect.py [batch] [precision]: two identically seeded MoCo models in one process - the first tensor of the forward that
differs between them (parameters after init, encoder features, projector / predictor outputs, keys, logits), several rounds."""
import os
import sys
import types
from functools import partial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-feature-vit_amd")]
import torch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
import vits  # noqa: E402
import moco.builder_vit_mocov3structure_mocov2loss as bld  # noqa: E402
from mfvit.moco_ops import l2_normalize, neg_logits, pos_logits  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x1 = torch.randn(B, 3, 224, 224, generator=g).to(dev)
x2 = torch.randn(B, 3, 224, 224, generator=g).to(dev)


def build():
    torch.manual_seed(7)
    return bld.MoCo_ViT(partial(vits.vit_small, stop_grad_conv1=True, precision=prec), types.SimpleNamespace(arch="vit_small"), 256, 4096, 0.2).to(dev)


def stages(m):
    out = {}
    with torch.no_grad():
        f = m.base_encoder.features3D(x1) if hasattr(m.base_encoder, "features3D") else None
        if f is not None:
            out["base features3D"] = f.clone()
    h = m.base_encoder(x1)
    out["base encoder + projector"] = h.detach().clone()
    p = m.predictor(h)
    out["predictor"] = p.detach().clone()
    q = l2_normalize(p)
    out["q"] = q.detach().clone()
    with torch.no_grad():
        hk = m.momentum_encoder(x2)
        out["momentum encoder + projector"] = hk.clone()
        k = l2_normalize(m.predictor(hk))
        out["k"] = k.clone()
        out["l_pos"] = pos_logits(q, k).clone()
        out["l_neg"] = neg_logits(q, m._queue_t()).clone()
    torch.cuda.synchronize()
    return out


def full(tag):
    """one seeded model, built, stepped once through the REAL forward (momentum update, enqueue) + backward, with hooks on the sub-modules; then freed"""
    m = build()
    cap = {}
    hooks = []
    for name, mod in (("base encoder + projector", m.base_encoder), ("predictor (calls in order)", m.predictor), ("momentum encoder + projector", m.momentum_encoder)):
        def hook(_m, _i, o, name=name):
            cap.setdefault(name, []).append(o.detach().clone())
        hooks.append(mod.register_forward_hook(hook))
    logits, labels = m(x1, x2, 0.99)
    cap["logits"] = [logits.detach().clone()]
    from mfvit.moco_ops import cross_entropy_rows
    loss = cross_entropy_rows(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    for n, p_ in m.named_parameters():
        if p_.grad is not None and ("blocks.11" in n or "blocks.0." in n or "head" in n or "predictor" in n):
            cap["grad " + n] = [p_.grad.detach().clone()]
    cap["momentum params after the EMA"] = [m.momentum_encoder.flat_parameters().detach().clone()]
    for h in hooks:
        h.remove()
    return cap


for rnd in range(3):
    a = full("a")
    b = full("b")
    diff = [n for n in a if not all(torch.equal(x, y) for x, y in zip(a[n], b[n]))]
    print(f"sequential models, round {rnd}: {len(a) - len(diff)} of {len(a)} captured tensors identical; first differing: {diff[:10]}", flush=True)
m1, m2 = build(), build()
pd = [n for (n, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()) if not torch.equal(a, b)]
print(f"batch {B}, {prec}: parameters / buffers that differ after the seeded init: {pd[:8]}")
for rnd in range(3):
    a, b, a2 = stages(m1), stages(m2), stages(m1)
    print(f"round {rnd}: " + "; ".join(f"{n}: {'same' if torch.equal(a[n], b[n]) else 'DIFFERENT'} / {'same' if torch.equal(a[n], a2[n]) else 'DIFFERENT'}" for n in a) + "   (two models / one model twice)")
