"""GPU parity of the single-op C-ABI entry points (through mfvit.ops -> ctypes -> libmfvit_hip.so) against float64
CPU math on the same (dtype-rounded) inputs.  Tolerances are stated per test: f32 paths are exact-f32 MFMA
(<= 2e-5 relative to the output scale), bf16 paths carry one bf16 rounding of the output (2^-9) on top of f32
accumulation."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_ops.txt")


def dev():
    return torch.device("cuda:0")


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 6e-3


def rel_err(got, ref):
    ref = ref.double().cpu()
    got = got.double().cpu()
    scale = ref.abs().max().clamp_min(1e-30)
    return float((got - ref).abs().max() / scale)


def log(name, err):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(f"{name}: max_abs_err/scale = {err:.3e}\n")


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(300, 384, 384), (128, 1152, 384), (197 * 3, 1536, 384), (64, 384, 1536)])
def test_linear_fwd(dtype, M, N, K):
    from mfvit import ops
    x, w, b = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, 0.05), rnd((N,), torch.float32, 3)
    y = ops.linear_fwd(x.to(dev()), w.to(dev()), b.to(dev()))
    ref = x.double() @ w.double().t() + b.double()
    e = rel_err(y, ref)
    log(f"linear_fwd[{dtype},{M},{N},{K}]", e)
    assert e < tol(dtype)
    dact, act = ops.linear_fwd(x.to(dev()), w.to(dev()), b.to(dev()), gelu=True)
    rg = ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(rg).sum().backward()          # gelu'(pre): what the backward consumes
    e1 = rel_err(dact, rg.grad)
    e2 = rel_err(act, torch.nn.functional.gelu(ref))
    log(f"linear_fwd_gelu[{dtype},{M},{N},{K}]", max(e1, e2))
    assert e1 < tol(dtype) and e2 < tol(dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(777, 256, 384), (197 * 4, 1152, 384), (100, 384, 1536), (31, 128, 128), (197 * 64, 1152, 384),
                                   (197 * 33 + 5, 384, 1536), (4099, 128, 128), (197 * 128, 1536, 384)])   # >= 4096 rows: LDS-DMA kernel (bf16), ragged tails
def test_linear_wgrad(dtype, M, N, K):
    from mfvit import ops
    dy, x = rnd((M, N), dtype, 4), rnd((M, K), dtype, 5)
    dw = ops.linear_wgrad(dy.to(dev()), x.to(dev()))
    ref = dy.double().t() @ x.double()
    e = rel_err(dw, ref)
    log(f"linear_wgrad[{dtype},{M},{N},{K}]", e)
    if e >= 1e-4:   # diagnostics for the transposing-read operand map
        got = dw.double().cpu()
        bad = ((got - ref).abs() > 1e-3 * ref.abs().max()).nonzero()[:16].tolist()
        log(f"  first mismatches {bad}", e)
    assert e < (2e-5 if dtype == torch.float32 else 1e-4)   # f32 accumulation, f32 output in both modes
    # accumulation into an existing gradient
    dw2 = ops.linear_wgrad(dy.to(dev()), x.to(dev()), out=dw.clone())
    assert rel_err(dw2, 2 * ref) < (2e-5 if dtype == torch.float32 else 1e-4)
    # split partials through caller scratch (plain stores + reduce pass) instead of float atomics: same result, accumulating
    scratch = torch.empty(ops.WGRAD_SCRATCH_FLOATS, device=dev())
    dw3 = ops.linear_wgrad(dy.to(dev()), x.to(dev()), out=dw.clone(), scratch=scratch)
    assert rel_err(dw3, 2 * ref) < (2e-5 if dtype == torch.float32 else 1e-4)


@pytest.mark.parametrize("M,N,K", [(1, 384, 384), (37, 96, 64), (130, 128, 128), (128, 384, 384), (512, 160, 96)])
def test_small_f32_gemms(M, N, K):
    """csrc/gemm_small.hip (f32, M <= 512: one 32 x 32 tile per workgroup, the reduction split over its four waves) at ragged M / N / K
    against float64: y = x W^T + b and dW += dy^T x (the fusion module's per-sample rows run on it; M = 1 and partial tiles here)."""
    from mfvit import ops
    x, w, b = rnd((M, K), torch.float32, 41), rnd((N, K), torch.float32, 42, 0.05), rnd((N,), torch.float32, 43)
    y = ops.linear_fwd(x.to(dev()), w.to(dev()), b.to(dev()))
    e = rel_err(y, x.double() @ w.double().t() + b.double())
    dy = rnd((M, N), torch.float32, 44)
    pre = rnd((N, K), torch.float32, 45)
    dw = ops.linear_wgrad(dy.to(dev()), x.to(dev()), out=pre.to(dev()).clone())
    e2 = rel_err(dw, pre.double() + dy.double().t() @ x.double())
    log(f"small_f32_gemms[{M},{N},{K}] fwd / wgrad {e2:.2e}", e)
    assert e < 1e-5 and e2 < 1e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,K", [(200, 384), (197 * 2, 1536), (64, 768)])
def test_linear_res_ln_fwd(dtype, M, K):
    from mfvit import ops
    a, w = rnd((M, K), dtype, 6), rnd((384, K), dtype, 7, 0.05)
    bias, res = rnd((384,), torch.float32, 8), rnd((M, 384), torch.float32, 9)
    gamma, beta = 1 + 0.1 * rnd((384,), torch.float32, 10), rnd((384,), torch.float32, 11, 0.1)
    x_out, y, mean, rstd = ops.linear_res_ln_fwd(*(t.to(dev()) for t in (a, w, bias, res, gamma, beta)), 1e-6)
    xr = a.double() @ w.double().t() + bias.double() + res.double()
    yr = torch.nn.functional.layer_norm(xr, (384,), gamma.double(), beta.double(), 1e-6)
    es = [rel_err(x_out, xr), rel_err(y, yr), rel_err(mean, xr.mean(1)), rel_err(rstd, 1 / torch.sqrt(xr.var(1, unbiased=False) + 1e-6))]
    log(f"linear_res_ln_fwd[{dtype},{M},{K}]", max(es))
    assert es[0] < 2e-5 + (0 if dtype == torch.float32 else 1e-6) and es[1] < tol(dtype) and es[2] < 1e-4 and es[3] < 1e-4


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,K", [(200, 1152), (197 * 2, 1536)])
def test_linear_dgrad_ln_bwd(dtype, M, K):
    from mfvit import ops
    dy, wt = rnd((M, K), dtype, 12), rnd((384, K), dtype, 13, 0.05)
    x, dres = rnd((M, 384), torch.float32, 14), rnd((M, 384), torch.float32, 15)
    gamma = 1 + 0.1 * rnd((384,), torch.float32, 16)
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = torch.zeros(384, dtype=torch.float64, requires_grad=True)
    yln = torch.nn.functional.layer_norm(xd, (384,), gd, bd, 1e-6)
    dyln = dy.double() @ wt.double().t()
    yln.backward(dyln)
    dx_ref = xd.grad + dres.double()
    mean = x.double().mean(1)
    rstd = 1 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-6)
    dx, dx_t, dgamma, dbeta, dcol = ops.linear_dgrad_ln_bwd(dy.to(dev()), wt.to(dev()), x.to(dev()), mean.float().to(dev()),
                                                           rstd.float().to(dev()), gamma.to(dev()), dres.to(dev()))
    es = [rel_err(dx, dx_ref), rel_err(dx_t, dx_ref), rel_err(dgamma, gd.grad), rel_err(dbeta, bd.grad), rel_err(dcol, dx_ref.sum(0))]
    log(f"linear_dgrad_ln_bwd[{dtype},{M},{K}]", max(es))
    assert es[0] < 3e-5 and es[1] < tol(dtype) and es[2] < 1e-4 and es[3] < 1e-4 and es[4] < 1e-4


def _attn_ref(qkv, heads):
    B, T, D3 = qkv.shape
    D = D3 // 3
    d = D // heads
    q, k, v = qkv.reshape(B, T, 3, heads, d).permute(2, 0, 3, 1, 4)
    a = (q @ k.transpose(-2, -1)) * d ** -0.5
    lse = torch.logsumexp(a, dim=-1)
    o = (a.softmax(-1) @ v).transpose(1, 2).reshape(B, T, D)
    return o, lse


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,T", [(2, 197), (1, 577), (3, 50)])
def test_attention_fwd_bwd(dtype, B, T):
    from mfvit import ops
    H, D = 12, 384
    qkv = rnd((B, T, 3 * D), dtype, 17)
    dout = rnd((B, T, D), dtype, 18)
    qd = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qd, H)
    o_ref.backward(dout.double())
    out, lse = ops.attention_fwd(qkv.to(dev()), H)
    e_o, e_l = rel_err(out, o_ref), rel_err(lse, lse_ref)
    dqkv, dbias = ops.attention_bwd(qkv.to(dev()), out, dout.to(dev()), lse, H)
    e_d, e_b = rel_err(dqkv, qd.grad), rel_err(dbias, qd.grad.sum((0, 1)))
    log(f"attention[{dtype},{B},{T}] out/lse/dqkv/dbias", max(e_o, e_l, e_d, e_b))
    t = 3e-5 if dtype == torch.float32 else 1.5e-2   # bf16: out and dout are bf16-rounded operands of the backward
    assert e_o < tol(dtype) and e_l < 1e-5 and e_d < t and e_b < t


def test_layernorm_fwd_bwd():
    from mfvit import ops
    rows, N = 333, 384
    x, dy, dres = rnd((rows, N), torch.float32, 19, 2.0), rnd((rows, N), torch.float32, 20), rnd((rows, N), torch.float32, 21)
    gamma, beta = 1 + 0.1 * rnd((N,), torch.float32, 22), rnd((N,), torch.float32, 23, 0.1)
    xd, gd, bd = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xd, (N,), gd, bd, 1e-5)
    yr.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(x.to(dev()), gamma.to(dev()), beta.to(dev()), 1e-5)
    assert rel_err(y, yr) < 2e-6
    dx, dx_t, dgamma, dbeta, dcol = ops.layernorm_bwd(dy.to(dev()), x.to(dev()), mean, rstd, gamma.to(dev()), dres.to(dev()),
                                                     copy_dtype=torch.bfloat16)
    ref = xd.grad + dres.double()
    assert rel_err(dx, ref) < 1e-5 and rel_err(dx_t, ref) < 6e-3
    assert rel_err(dgamma, gd.grad) < 1e-5 and rel_err(dbeta, bd.grad) < 1e-5 and rel_err(dcol, ref.sum(0)) < 1e-5


def test_cast_transpose_head_ce():
    from mfvit import ops
    src = rnd((384, 1152), torch.float32, 24)
    d, dt = ops.cast_transpose(src.to(dev()), torch.bfloat16)
    assert torch.equal(d.cpu(), src.bfloat16()) and torch.equal(dt.cpu(), src.bfloat16().t().contiguous())
    d, dt = ops.cast_transpose(src.to(dev()), torch.float32, want_straight=False)
    assert d is None and torch.equal(dt.cpu(), src.t().contiguous())
    # ragged shapes: (100, 36) stays on the four-elements-per-thread kernel (partial 64 x 64 tiles), (37, 50) falls back to the element-wise one
    for shape in ((100, 36), (37, 50)):
        s2 = rnd(shape, torch.float32, 29)
        for dt_ in (torch.bfloat16, torch.float16, torch.float32):
            d, dt = ops.cast_transpose(s2.to(dev()), dt_)
            assert torch.equal(d.cpu(), s2.to(dt_)) and torch.equal(dt.cpu(), s2.to(dt_).t().contiguous()), (shape, dt_)
    # classifier head on the cls rows of (B,T,D) tokens, and its backward
    B, T, D, C = 5, 7, 384, 3
    feats, w, b = rnd((B, T, D), torch.float32, 25), rnd((C, D), torch.float32, 26, 0.05), rnd((C,), torch.float32, 27)
    y = ops.head_fwd(feats.to(dev()), w.to(dev()), b.to(dev()), ldx=T * D)
    yr = feats[:, 0].double() @ w.double().t() + b.double()
    assert rel_err(y, yr) < 1e-6
    dy = rnd((B, C), torch.float32, 28)
    dfe = torch.zeros(B, T, D, device=dev())
    dw = torch.zeros(C, D, device=dev())
    db = torch.zeros(C, device=dev())
    ops.head_bwd(dy.to(dev()), feats.to(dev()), w.to(dev()), ldx=T * D, dx=dfe, lddx=T * D, dw=dw, db=db)
    assert rel_err(dfe[:, 0], dy.double() @ w.double()) < 1e-6 and float(dfe[:, 1:].abs().max()) == 0.0
    assert rel_err(dw, dy.double().t() @ feats[:, 0].double()) < 1e-6 and rel_err(db, dy.double().sum(0)) < 1e-6
    # cross entropy (mean), gradient and argmax; ties resolve to the first maximum like torch.max (MAIN_CA:870)
    logits = rnd((9, 3), torch.float32, 29, 3.0)
    logits[4] = torch.tensor([1.0, 1.0, 0.5])
    target = torch.tensor([0, 1, 2, 0, 1, 2, 0, 1, 2])
    ld = logits.double().requires_grad_(True)
    lr = torch.nn.functional.cross_entropy(ld, target)
    lr.backward()
    loss, dlogits, preds = ops.cross_entropy(logits.to(dev()), target.to(dev()))
    assert abs(float(loss) - float(lr)) < 1e-6 * max(1.0, abs(float(lr)))
    assert rel_err(dlogits, ld.grad) < 1e-6
    assert preds.cpu().tolist() == logits.max(1)[1].tolist()
