"""Host-side time per phase of the CA step (no synchronisation inside the loop): shows whether the Python / launch path keeps ahead of
the GPU (sum of phases << GPU step time) or is the bottleneck."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
import bench
from mfvit.losses import cross_entropy
from mfvit.optim import Adam
dev = torch.device("cuda:0")
class A: pass
args = A(); args.precision = "bf16"; args.img = 224; args.mode = "T"
model, backs = bench.build_models(args, dev)
g = torch.Generator().manual_seed(1)
B = 128
x = torch.randn(B, 3, 224, 224, generator=g).to(dev); xe = torch.randn(B, 3, 224, 224, generator=g).to(dev)
t = torch.randint(0, 3, (B,), generator=g).to(dev)
params = list(model.parameters()) + [p for m in backs for p in m.parameters() if p.requires_grad]
opt = Adam(params, lr=1e-4, betas=(0.9, 0.999))
acc = [0.0] * 5
def step(rec):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True); t1 = time.perf_counter()
    fused, xc, xen = model(backs[0], backs[1], x, xe); t2 = time.perf_counter()
    loss, _ = cross_entropy(fused + xc + xen, t); t3 = time.perf_counter()
    loss.backward(); t4 = time.perf_counter()
    opt.step(); t5 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[i] += d
for _ in range(5): step(False)
torch.cuda.synchronize()
N = 30
T0 = time.perf_counter()
for _ in range(N): step(True)
Th = time.perf_counter() - T0
torch.cuda.synchronize()
Tg = time.perf_counter() - T0
print(f"host loop {1e3*Th/N:.2f} ms/step, with final sync {1e3*Tg/N:.2f} ms/step")
for n, v in zip(("zero_grad", "forward", "loss", "backward", "opt.step"), acc): print(f"  {n:10s} {1e3*v/N:7.3f} ms/step (host)")
