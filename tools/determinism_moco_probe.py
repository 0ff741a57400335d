"""usage (GPU box): python tools/determinism_moco_probe.py [batch] [precision]: the MoCo pretraining step (BASELINE configs[3] slice: two views, momentum encoder, queue,
InfoNCE, AdamW) - the same seeded model trained for three steps twice; which parameters / buffers (queue included) and losses are the same bits at the end."""
import os
import sys
import types
from functools import partial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-feature-vit_amd")]
import torch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
import vits  # noqa: E402
import moco.builder_vit_mocov3structure_mocov2loss as bld  # noqa: E402
from mfvit.amp import GradScaler  # noqa: E402
from mfvit.moco_ops import cross_entropy_rows  # noqa: E402
from mfvit.optim import AdamW  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x1 = torch.randn(B, 3, 224, 224, generator=g).to(dev)
x2 = torch.randn(B, 3, 224, 224, generator=g).to(dev)


def train():
    torch.manual_seed(7)
    model = bld.MoCo_ViT(partial(vits.vit_small, stop_grad_conv1=True, precision=prec), types.SimpleNamespace(arch="vit_small"), 256, 4096, 0.2).to(dev)
    opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=1.5e-4, weight_decay=0.1)
    scaler = GradScaler(enabled=prec == "fp16")
    losses, first = [], {}
    for it in range(3):
        logits, labels = model(x1, x2, 0.99)
        loss = cross_entropy_rows(logits, labels)
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        if it == 0:
            first["logits"] = logits.detach().clone()
            for n, p_ in model.named_parameters():
                if p_.grad is not None:
                    first["grad " + n] = p_.grad.detach().clone()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    state = {n: t.detach().clone() for n, t in list(model.named_parameters()) + list(model.named_buffers())}
    for i, l in enumerate(losses):
        state[f"loss[{i}]"] = l
    return state, first


(a, fa), (b, fb) = train(), train()
d0 = [n for n in fa if not torch.equal(fa[n], fb[n])]
print(f"  step 0: {len(fa) - len(d0)} of {len(fa)} of (logits, gradients) bit-identical; differing: {d0[:14]}{' ...' if len(d0) > 14 else ''}")
diff = [n for n in a if not torch.equal(a[n], b[n])]
same = [n for n in a if n not in diff]
print(f"MoCo step x 3, batch {B}, {prec}: {len(a) - len(diff)} of {len(a)} tensors bit-identical between two trainings; differing: {diff[:12]}{' ...' if len(diff) > 12 else ''}"
      + (f"; identical: {same[:30]}" if diff and len(same) <= 40 else ""))
