"""Which host-side torch calls launch the small fill / copy kernels of one MoCo step (the step of bench.py --workload moco) - torch.profiler with stacks."""
import os, sys, types, collections
from functools import partial
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
import vits
import moco.builder_vit_mocov3structure_mocov2loss as bld
from mfvit.moco_ops import cross_entropy_rows
from mfvit.optim import AdamW
from mfvit.amp import GradScaler
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = 128
model = bld.MoCo_ViT(partial(vits.vit_small, stop_grad_conv1=True, precision=prec, img_size=224), types.SimpleNamespace(arch="vit_small"), 256, 4096, 0.2).to(dev)
x1, x2 = torch.randn(B, 3, 224, 224, device=dev), torch.randn(B, 3, 224, 224, device=dev)
opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=1.5e-4, weight_decay=0.1)
scaler = GradScaler(enabled=prec == "fp16")


def step():
    logits, labels = model(x1, x2, 0.99)
    loss = cross_entropy_rows(logits, labels)
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()


for _ in range(3):
    step()
torch.cuda.synchronize()
# call sites of the small tensor methods (python-level counter: the profiler's stacks come back empty on this build)
import traceback
sites = collections.Counter()


def wrap(cls, name):
    orig = getattr(cls, name)

    def f(self, *a, **k):
        if getattr(self, "is_cuda", False) or any(getattr(x, "is_cuda", False) for x in a):
            fr = [x for x in traceback.extract_stack(limit=8) if "moco_small_ops" not in x.filename][-1]
            sites[(name, f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70] if fr.line else ''}")] += 1
        return orig(self, *a, **k)
    setattr(cls, name, f)


for nm in ("copy_", "to", "float", "clone", "zero_", "fill_", "add_", "contiguous", "half"):
    wrap(torch.Tensor, nm)
step()
for (nm, where), c in sites.most_common(30):
    print(f"{c:5d} / step  {nm:10s} {where}")
sys.exit(0)
