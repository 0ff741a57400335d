"""Adam step over the bench's parameter set size (two vit_small encoders + fusion: ~45 M f32 parameters in 3 param groups): time per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit.optim import Adam
dev = torch.device("cuda:0")
groups = [[torch.nn.Parameter(torch.randn(n, device=dev)) for n in ns] for ns in ([1_200_000, 300_000, 1152, 384], [21_600_000, 75_648, 384], [21_600_000, 75_648, 384])]
opt = Adam([{"params": g} for g in groups], lr=1e-4)
for g in groups:
    for p in g:
        p.grad = torch.randn_like(p)
for _ in range(3):
    opt.step()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    opt.step()
e.record()
torch.cuda.synchronize()
n = sum(p.numel() for g in groups for p in g)
us = s.elapsed_time(e) * 1e3 / 20
print(f"Adam step, {n / 1e6:.1f} M parameters: {us:.1f} us  ({28.0 * n / us / 1e6:.2f} TB/s of the 28 B/element)")
