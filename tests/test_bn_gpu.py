"""GPU parity of the BatchNorm kernels of the MoCo projector / predictor (mfvit_bn_stats / _combine / _apply / _bwd_sums / _bwd_apply,
csrc/moco.hip; reference: nn.BatchNorm1d / SyncBatchNorm of builder_vit_mocov3structure_mocov2loss.py:62-78, MAIN_MOCO:297) in EVERY element
type they run in - fp16 (the autocast pretraining of configs[3]), bf16, and f32 (what precision='bf16x3' hands them: mfvit/mlp.py::_tdtype) -
at the op level, against float64 BatchNorm on the SAME rounded inputs and the SAME ReLU mask (the kernels were op-tested
in fp32 only; promoted from tools/bn_check.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_bn.txt")
DTYPES = {"fp16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
# outputs / dx leave the kernels in the element type: one rounding of the result (2^-11 fp16, 2^-8 bf16) on top of f32 arithmetic
TOL = {"fp16": 2e-3, "bf16": 1.6e-2, "f32": 5e-5}


def log(msg):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(msg + "\n")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("affine", [True, False], ids=["affine", "noaffine"])
@pytest.mark.parametrize("relu", [False, True], ids=["plain", "relu"])
@pytest.mark.parametrize("n,C", [(32, 256), (32, 512), (128, 256), (128, 512), (128, 4096), (32, 4096)])
@pytest.mark.parametrize("prec", list(DTYPES))
def test_batchnorm_forward_backward_in_every_element_type(prec, n, C, relu, affine):
    from mfvit import mlp
    dt = DTYPES[prec]
    g = torch.Generator().manual_seed(1000 + n + C)
    x = (torch.randn(n, C, generator=g) * 2 + 0.5).to(dt)                 # the ROUNDED input is the input
    r = torch.randn(n, C, generator=g).to(dt)                              # upstream gradient as the previous kernel hands it over
    bn = mlp.HipBatchNorm1d(C, affine=affine, relu=relu).to(DEV).train()
    if affine:
        with torch.no_grad():
            bn.weight.copy_(1 + 0.1 * torch.randn(C, generator=g))
            bn.bias.copy_(0.1 * torch.randn(C, generator=g))
    gamma = bn.weight.detach().cpu().double() if affine else torch.ones(C, dtype=torch.float64)
    beta = bn.bias.detach().cpu().double() if affine else torch.zeros(C, dtype=torch.float64)
    xg = x.to(DEV).requires_grad_(True)
    y = bn(xg)
    assert y.dtype == dt
    (y.float() * r.to(DEV).float()).sum().backward()
    # float64 BatchNorm on the same inputs; the ReLU mask is the kernel's own (pre-activations within a rounding of zero may fall either way)
    xd = x.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + bn.eps)
    xhat = (xd - mean) * rstd
    pre = xhat * gamma + beta
    mask = (y.detach().cpu() > 0).double() if relu else torch.ones_like(pre)
    y_ref = pre * mask
    dy = r.double() * mask
    s0, s1 = dy.sum(0), (dy * xhat).sum(0)
    dx_ref = gamma * rstd * (dy - s0 / n - xhat * s1 / n)
    e = {"y": rel(y, y_ref), "dx": rel(xg.grad, dx_ref)}
    if affine:
        e["dgamma"], e["dbeta"] = rel(bn.weight.grad, s1), rel(bn.bias.grad, s0)
    e["run_mean"] = rel(bn.running_mean, 0.1 * mean)
    e["run_var"] = rel(bn.running_var, 0.9 + 0.1 * xd.var(0, unbiased=True))
    flips = float(((pre > 0).double() != mask).double().mean()) if relu else 0.0
    log(f"BatchNorm[{prec}, n={n}, C={C}, relu={relu}, affine={affine}] " + " ".join(f"{k} {v:.2e}" for k, v in e.items()) +
        f"  mask flips vs float64 {flips:.1e}")
    t = TOL[prec]
    assert e["y"] < t and e["dx"] < t, e
    assert all(e[k] < 5e-5 for k in e if k not in ("y", "dx")), e       # sums and statistics are f32 in every type


@pytest.mark.parametrize("prec", ["fp16", "f32"])
def test_batchnorm_eval_mode_uses_the_running_statistics(prec):
    from mfvit import mlp
    dt = DTYPES[prec]
    n, C = 64, 512
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, C, generator=g).to(dt)
    bn = mlp.HipBatchNorm1d(C).to(DEV).eval()
    with torch.no_grad():
        bn.running_mean.copy_(0.2 * torch.randn(C, generator=g))
        bn.running_var.copy_(0.5 + torch.rand(C, generator=g))
        bn.weight.copy_(1 + 0.1 * torch.randn(C, generator=g))
    xg = x.to(DEV).requires_grad_(True)
    y = bn(xg)
    y.float().sum().backward()
    k = bn.weight.detach().cpu().double() / torch.sqrt(bn.running_var.cpu().double() + bn.eps)
    y_ref = (x.double() - bn.running_mean.cpu().double()) * k + bn.bias.detach().cpu().double()
    assert rel(y, y_ref) < TOL[prec] and rel(xg.grad, k.expand(n, C)) < TOL[prec]


@pytest.mark.parametrize("counts", [[16] * 8, [3, 40, 1, 16, 16, 7, 25, 20]], ids=["equal", "ragged"])
def test_chan_combine_of_eight_partial_statistics(counts):
    """SyncBatchNorm at world size 8 (configs[3] / [4]; MAIN_MOCO:297 over BLD:62-78) without 8 GPUs: mfvit_bn_stats on each of 8 row slices (what each
    rank computes), the triples laid out as the all-gather delivers them, mfvit_bn_combine with W = 8 - against float64 BatchNorm statistics of the
    concatenated batch and against the oracle's restatement of the combine (oracle/ref_moco.py::bn_chan_combine, which the 8-rank gloo test runs on the CPU)."""
    from mfvit._lib import check, lib, ptr, stream
    from mfvit import ops
    from oracle import ref_moco
    C, W = 512, len(counts)
    g = torch.Generator().manual_seed(11)
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    x = torch.randn(offs[-1], C, generator=g) * torch.linspace(0.3, 4.0, C) + torch.repeat_interleave(torch.arange(W).float(), torch.tensor(counts))[:, None] * 0.9
    xd = x.to(DEV)
    means, m2s = torch.empty(W, C, device=DEV), torch.empty(W, C, device=DEV)
    for w in range(W):
        part = xd[offs[w]:offs[w + 1]].contiguous()
        check(lib().mfvit_bn_stats(ops._code_of(part), ptr(part), part.shape[0], C, ptr(means[w]), ptr(m2s[w]), stream()), "mfvit_bn_stats")
    cnt = torch.tensor(counts, dtype=torch.float32, device=DEV)
    mean, invstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    check(lib().mfvit_bn_combine(ptr(means), ptr(m2s), ptr(cnt), W, C, 1e-5, 0.1, ptr(mean), ptr(invstd), ptr(rm), ptr(rv), stream()), "mfvit_bn_combine")
    x64 = x.double()
    mu_ref, var_ref = x64.mean(0), x64.var(0, unbiased=False)
    e = {"mean": rel(mean, mu_ref), "invstd": rel(invstd, 1 / torch.sqrt(var_ref + 1e-5)), "run_mean": rel(rm, 0.1 * mu_ref),
         "run_var": rel(rv, 0.9 + 0.1 * x64.var(0, unbiased=True))}
    stats = torch.cat([torch.cat([ref_moco.bn_partial_stats(x[offs[w]:offs[w + 1]])]) for w in range(W)])
    o_mu, o_var, o_invstd, _ = ref_moco.bn_chan_combine(stats, C)
    e["vs_oracle_mean"], e["vs_oracle_invstd"] = rel(mean, o_mu), rel(invstd, o_invstd)
    log(f"Chan combine[W = 8, counts {counts}] " + " ".join(f"{k} {v:.2e}" for k, v in e.items()))
    assert all(v < 2e-5 for v in e.values()), e
