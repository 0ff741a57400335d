"""GPU: the round-6 tile GEMM (csrc/gemm_pp.hip: persistent 256 x 128 tiles, two wave groups half a stage apart, LDS-DMA ring, per-wave epilogue patches)
through the C ABI (mfvit_linear_fwd / mfvit_linear_dgrad_act) in every element type it is built for - split bf16 (`bf16x3`, the headline), bf16, fp16 - and
every epilogue, against float64 math on the ROUNDED operands, and against the round-5 tile kernel (MFVIT_PP=0) on the same inputs:

  bias (+ split-fp16 output: the qkv projection) | none (proj data gradient) | bias + GELU with the saved derivative, with and without it | x gelu' (fc2 data gradient)

Shapes: the four (N, K) of a ViT-S block at the bench's M = 128 x 197 = 25,216 (BASELINE configs[2]), a ragged M (the last 256-row tile partial, its second
half-tile empty), an M just above the kernel's threshold, fewer tiles than CUs, and the shortest K it takes (two stages: every wait of the prologue, the first
stage behind an epilogue and the peeled last stage at once).  The reference's math: timm Block Linears, call sites
moco_pretraining/moco/model/crossvit_2vits_..._std002_sum.py:128-135."""
import os

import pytest
import torch

from mfvit import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_gemm_pp.txt")
# (name, split, torch dtype, GEMM tolerance vs float64 on the rounded operands (scale-relative), tolerance of a 2-byte output)
MODES = [("bf16x3", True, torch.bfloat16, 3e-5, 3e-5), ("fp16", False, torch.float16, 1e-3, 1e-3), ("bf16", False, torch.bfloat16, 8e-3, 8e-3)]
SHAPES = [(128 * 197, 1536, 384, "fc1"), (128 * 197, 1152, 384, "qkv"), (128 * 197, 384, 384, "proj dgrad"), (128 * 197 - 57, 1536, 384, "ragged M"),
          (2048 + 40, 384, 1536, "threshold M, K = 1536"), (6 * 256 + 130, 256, 64, "12 + 2 tiles, two stages"), (64 * 197, 1152, 384, "B = 64"),
          (256 * 197, 1152, 384, "B = 256: seven tiles per workgroup"), (3000, 128, 384, "one column tile, fewer tiles than CUs"),
          (128 * 197, 384, 1536, "48 stages at the bench M")]


def log(msg):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(msg + "\n")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def gen(seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return g


class Fmt:
    def __init__(self, split, dt):
        self.split, self.dt = split, dt

    def pack(self, x):
        return ops.split_pack(x) if self.split else x.to(self.dt)

    def val(self, t):           # float64 value of a packed tensor
        return (ops.split_unpack(t) if self.split else t).double()


@pytest.mark.parametrize("M,N,K,tag", SHAPES, ids=[s[3] for s in SHAPES])
@pytest.mark.parametrize("name,split,dt,tol,tol16", MODES, ids=[m[0] for m in MODES])
def test_forward_epilogues(monkeypatch, name, split, dt, tol, tol16, M, N, K, tag):
    if not split and K % 64:
        pytest.skip("plain 16-bit stages are 64 k wide")
    f = Fmt(split, dt)
    g = gen(100 + N + K)
    x, w = f.pack(torch.randn(M, K, device=DEV, generator=g)), f.pack(torch.randn(N, K, device=DEV, generator=g) * 0.05)
    b = torch.randn(N, device=DEV, generator=g)
    ref0 = f.val(x) @ f.val(w).T
    ref = ref0 + b.double()
    out = {}
    for pp in ("2", "0"):                                  # 2: the new kernel wherever it CAN run (1, the default, leaves small M / few tiles to the round-5 kernel)
        monkeypatch.setenv("MFVIT_PP", pp)
        y = ops.linear_fwd(x, w, b, split=split)
        y0 = ops.linear_fwd(x, w, None, split=split)
        dact, act = ops.linear_fwd(x, w, b, gelu=True, split=split)
        none, act2 = ops.linear_fwd(x, w, b, gelu=True, split=split, want_grad=False)
        out[pp] = [y, y0, dact, act, act2]
        if split:
            out[pp].append(ops.linear_fwd(x, w, b, split=True, qkv_f16=True))
        assert none is None
    y, y0, dact, act, act2 = out["2"][:5]
    rg = ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(rg).sum().backward()
    e = dict(bias=rel(f.val(y), ref), none=rel(f.val(y0), ref0), gelu=rel(f.val(act), torch.nn.functional.gelu(ref)),
             gelu_nograd=rel(f.val(act2), torch.nn.functional.gelu(ref)), dgelu=rel(dact.double(), rg.grad))
    if split:
        q = out["2"][5]
        assert q.dtype == torch.float16 and q.shape == y.shape
        qv = q.view(M, N // 32, 2, 32).double().sum(2).reshape(M, N)        # I32 layout: [hi x 32 | lo x 32] per 32 columns
        e["qkv_f16"] = rel(qv, ref)
    # against the round-5 kernel: the same products in the same k order - differences are summation-order roundings of the f32 accumulators
    def vals(r):
        return [f.val(r[0]), f.val(r[1]), r[2].double(), f.val(r[3]), f.val(r[4])]
    vs_old = max(rel(a, o) for a, o in zip(vals(out["2"]), vals(out["0"])))
    log(f"gemm_pp forward[{name}, M={M}, N={N}, K={K}: {tag}] " + " ".join(f"{k} {v:.2e}" for k, v in e.items()) + f"  vs round-5 kernel {vs_old:.2e}")
    assert max(e["bias"], e["none"], e["gelu"], e["gelu_nograd"]) < tol, e
    assert e["dgelu"] < (1e-3 if dact.dtype == torch.float16 else tol16), e           # the saved derivative is fp16 (split, fp16) or bf16
    if split:
        assert e["qkv_f16"] < tol, e
    assert vs_old < (1e-3 if split else 2 * tol16), vs_old       # (gelu' of split tensors is plain fp16: 2^-11 flips against the other kernel's rounding)


@pytest.mark.parametrize("M,N,K,tag", [(128 * 197, 1536, 384, "fc2 dgrad"), (128 * 197 - 57, 1536, 384, "ragged M"), (2048 + 40, 384, 1536, "threshold M"),
                                       (6 * 256 + 130, 256, 64, "two stages")])
@pytest.mark.parametrize("name,split,dt,tol,tol16", MODES, ids=[m[0] for m in MODES])
def test_dgrad_times_saved_derivative(monkeypatch, name, split, dt, tol, tol16, M, N, K, tag):
    f = Fmt(split, dt)
    g = gen(200 + N + K)
    dy, wt = f.pack(torch.randn(M, K, device=DEV, generator=g) * 0.1), f.pack(torch.randn(N, K, device=DEV, generator=g) * 0.05)
    ag = (torch.rand(M, N, device=DEV, generator=g) * 1.2 - 0.1).to(torch.float16 if split else dt)          # gelu' lives in [-0.13, 1.13]
    ref = (f.val(dy) @ f.val(wt).T) * ag.double()
    res = {}
    for pp in ("2", "0"):
        monkeypatch.setenv("MFVIT_PP", pp)
        res[pp] = ops.linear_dgrad_act(dy, wt, ag, split=split)
    e, vs_old = rel(f.val(res["2"]), ref), rel(f.val(res["2"]), f.val(res["0"]))
    log(f"gemm_pp dgrad x act'[{name}, M={M}, N={N}, K={K}: {tag}] {e:.2e}  vs round-5 kernel {vs_old:.2e}")
    assert e < tol and vs_old < 2 * tol, (e, vs_old)


def test_results_are_the_same_bits_from_run_to_run():
    """No atomics, no arrival order anywhere in the forward epilogues: two launches on the same inputs give identical bytes (bench shape, fc1 + GELU)."""
    g = gen(7)
    M, N, K = 128 * 197, 1536, 384
    x, w = ops.split_pack(torch.randn(M, K, device=DEV, generator=g)), ops.split_pack(torch.randn(N, K, device=DEV, generator=g) * 0.05)
    b = torch.randn(N, device=DEV, generator=g)
    a1 = ops.linear_fwd(x, w, b, gelu=True, split=True)
    a2 = ops.linear_fwd(x, w, b, gelu=True, split=True)
    assert torch.equal(a1[0], a2[0]) and torch.equal(a1[1].view(torch.int16), a2[1].view(torch.int16))


@pytest.mark.parametrize("M,N,K,tag", [(128 * 197, 1152, 384, "qkv: 56 tall + 57 short row tiles"), (128 * 197 - 57, 1536, 384, "fc1, ragged M"),
                                       (64 * 197, 1536, 384, "B = 64"), (6 * 256 + 130, 256, 64, "two stages, 14 tiles")])
def test_mixed_tile_heights_give_the_same_bits_as_tall_tiles_only(monkeypatch, M, N, K, tag):
    """Second half of round 6: a launch mixes 256-row and 192-row tiles so that every persistent workgroup carries the same load (MFVIT_PP_MIX, default 1).  The k order
    of every output element does not depend on the tile it falls into, so every epilogue's outputs must be BIT-identical with MFVIT_PP_MIX=0 (256-row tiles only) - a row
    mapped to the wrong place of a short tile's LDS image, a skipped fragment or a miscounted wait would show as wrong bits, not as a rounding."""
    g = gen(300 + N + K + M % 97)
    x, w = ops.split_pack(torch.randn(M, K, device=DEV, generator=g)), ops.split_pack(torch.randn(N, K, device=DEV, generator=g) * 0.05)
    b = torch.randn(N, device=DEV, generator=g)
    dy, wt = ops.split_pack(torch.randn(M, K, device=DEV, generator=g) * 0.1), ops.split_pack(torch.randn(N, K, device=DEV, generator=g) * 0.05)
    ag = (torch.rand(M, N, device=DEV, generator=g) * 1.2 - 0.1).to(torch.float16)
    monkeypatch.setenv("MFVIT_PP", "2")
    res = {}
    for mix in ("1", "0"):
        monkeypatch.setenv("MFVIT_PP_MIX", mix)
        dact, act = ops.linear_fwd(x, w, b, gelu=True, split=True)
        res[mix] = [ops.linear_fwd(x, w, b, split=True), ops.linear_fwd(x, w, None, split=True), dact, act, ops.linear_fwd(x, w, b, split=True, qkv_f16=True),
                    ops.linear_dgrad_act(dy, wt, ag, split=True)]
    for i, (a, c) in enumerate(zip(res["1"], res["0"])):
        assert torch.equal(a.view(torch.int16), c.view(torch.int16)), (tag, i)
    log(f"gemm_pp mixed tile heights[M={M}, N={N}, K={K}: {tag}] bias / none / gelu' / gelu / qkv_f16 / dgrad x act' bit-identical with 256-row tiles only")
