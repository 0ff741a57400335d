"""HipBatchNorm1d / HipLinear forward + backward against torch float64 at the MoCo config-4 MLP sizes (n = 128 rows, 256 / 4096 columns)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import mlp
dev = "cuda:0"
def err(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())
def l2(a, b):
    return float((a.double().cpu() - b.double()).norm() / b.double().norm())
for n, C in [(8, 512), (128, 512), (128, 4096), (128, 256), (200, 4096)]:
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, C, generator=g) * 2 + 0.5
    r = torch.randn(n, C, generator=g)
    for relu in (False, True):
        bn = mlp.HipBatchNorm1d(C, relu=relu).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(1 + 0.1 * torch.randn(C, generator=g)); bn.bias.copy_(0.1 * torch.randn(C, generator=g))
        xr = x.double().requires_grad_(True)
        wr = bn.weight.detach().cpu().double().requires_grad_(True)
        br = bn.bias.detach().cpu().double().requires_grad_(True)
        ref = torch.nn.functional.batch_norm(xr, None, None, wr, br, True, 0.1, bn.eps)
        if relu:
            ref = torch.relu(ref)
        (ref * r.double()).sum().backward()
        xg = x.to(dev).requires_grad_(True)
        y = bn(xg)
        (y.float() * r.to(dev)).sum().backward()
        print(f"BN n={n} C={C} relu={relu}: y {err(y, ref):.2e}  dx {err(xg.grad, xr.grad):.2e} (L2 {l2(xg.grad, xr.grad):.2e})  "
              f"dgamma {err(bn.weight.grad, wr.grad):.2e}  dbeta {err(bn.bias.grad, br.grad):.2e}", flush=True)
for n, K, N in [(128, 256, 4096), (128, 4096, 256), (128, 4096, 4096), (128, 384, 4096)]:
    g = torch.Generator().manual_seed(5)
    x, r = torch.randn(n, K, generator=g), torch.randn(n, N, generator=g)
    lin = mlp.HipLinear(K, N, precision="bf16x3").to(dev)
    xr = x.double().requires_grad_(True)
    wr = lin.weight.detach().cpu().double().requires_grad_(True)
    ((xr @ wr.t()) * r.double()).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    y = lin(xg)
    (y * r.to(dev)).sum().backward()
    print(f"Linear n={n} {K}->{N}: y {err(y, x.double() @ wr.detach().t()):.2e}  dx {err(xg.grad, xr.grad):.2e}  dW {err(lin.weight.grad, wr.grad):.2e}", flush=True)
