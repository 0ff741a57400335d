"""The unfused encoder path (csrc/vit.hip: plain tile GEMM + a LayerNorm / LayerNorm-backward row pass where vit_small runs the row-complete GEMM
kernels with fused epilogues): (1) vit_base (embed 768, head_dim 64; MAIN_MOCO:50 `-a vit_base`) forward + backward against the CPU oracle,
(2) the same vit_small encoder through BOTH paths (MFVIT_UNFUSED_ROWS=1 forces the unfused one at dim 384): features and every gradient agree
at the precision's rounding level - the row passes are checked against the fused epilogues, which the oracle tests pin."""
import os

import pytest
import torch

from conftest import rng_tensor
from oracle import ref_vit

pytestmark = pytest.mark.gpu
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_encoder.txt")


def log(msg):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(msg + "\n")


def scale_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def build(arch, precision, seed, depth, num_classes=3, **kw):
    import vits
    m = vits.__dict__[arch](num_classes=num_classes, depth=depth, precision=precision, **kw)
    p = ref_vit.seeded_params(seed, arch=arch, num_classes=num_classes, depth=depth)
    msg = m.load_state_dict(p, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    return m.to("cuda:0"), p


@pytest.mark.parametrize("precision,tol_f,tol_g", [("fp32", 1e-4, 1e-3), ("bf16x3", 1e-4, 2e-3), ("fp16", 5e-3, 2e-2), ("bf16", 4e-2, 8e-2)])
def test_vit_base_forward_backward_against_the_oracle(precision, tol_f, tol_g):
    """vit_base, 3 blocks, B = 3 (591 token rows: a ragged last tile in every GEMM): features3D / logits against the f32 oracle, all parameter
    gradients of a random linear functional against the float64 oracle.  The gradient bound of bf16x3 is the saved activation derivative's
    (plain fp16), as for vit_small."""
    B, depth = 3, 3
    m, p = build("vit_base", precision, 1701, depth)
    x = rng_tensor(1702, (B, 3, 224, 224))
    r = rng_tensor(1703, (B, 197, 768))
    rl = rng_tensor(1704, (B, 3))
    with torch.no_grad():
        ref_f = ref_vit.features3d(p, x)
        ref_l = ref_vit.head_linear(p, ref_f[:, 0])
        xg = x.to("cuda:0")
        f = m.features3D(xg)
        logits = m(xg)
    e_f, e_l = scale_err(f, ref_f), scale_err(logits, ref_l)
    assert f.shape == (B, 197, 768)
    assert e_f < tol_f and e_l < tol_f, (e_f, e_l)
    if precision in ("fp32", "bf16x3"):
        assert logits.argmax(1).cpu().tolist() == ref_l.argmax(1).tolist()
    pd = {k: v.double().requires_grad_(k != "pos_embed") for k, v in p.items()}
    f_ref = ref_vit.features3d(pd, x.double())
    ((f_ref * r.double()).sum() + (ref_vit.head_linear(pd, f_ref[:, 0]) * rl.double()).sum()).backward()
    f = m.features3D(xg)
    ((f * r.to("cuda:0")).sum() + (m(xg) * rl.to("cuda:0")).sum()).backward()
    worst = ("", 0.0)
    for name, prm in m.named_parameters():
        if name == "pos_embed":
            assert prm.grad is None
            continue
        assert prm.grad is not None, name
        e = scale_err(prm.grad, pd[name].grad)
        worst = max(worst, (name, e), key=lambda t: t[1])
        assert e < tol_g, (name, e)
    log(f"vit_base[{precision}] features {e_f:.3e} logits {e_l:.3e} worst gradient {worst[0]} {worst[1]:.3e}")


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5, ), ("bf16x3", 2e-4), ("fp16", 5e-3)])
@pytest.mark.parametrize("B", [2, 17])
def test_unfused_path_matches_the_fused_kernels_at_dim_384(precision, tol, B):
    """vit_small through the row-complete kernels (default) and through tile GEMM + row passes (MFVIT_UNFUSED_ROWS=1): same features, same gradients
    up to the rounding of the operand type (B = 17: 3,349 rows, the tall-tile row kernel with mixed heights; B = 2: its one-fragment form)."""
    depth = 2
    m, _ = build("vit_small", precision, 1711, depth)
    x = rng_tensor(1712, (B, 3, 224, 224)).to("cuda:0")
    r = rng_tensor(1713, (B, 197, 384)).to("cuda:0")
    out = {}
    for mode in ("0", "1"):
        os.environ["MFVIT_UNFUSED_ROWS"] = mode
        try:
            m.zero_grad(set_to_none=True)
            f = m.features3D(x)
            (f * r).sum().backward()
            torch.cuda.synchronize()
            out[mode] = (f.detach().clone(), {n: q.grad.detach().clone() for n, q in m.named_parameters() if q.grad is not None})
        finally:
            os.environ.pop("MFVIT_UNFUSED_ROWS", None)
    e_f = scale_err(out["1"][0], out["0"][0])
    assert e_f < tol, e_f
    assert set(out["0"][1]) == set(out["1"][1])
    worst = ("", 0.0)
    for n, g in out["0"][1].items():
        e = scale_err(out["1"][1][n], g)
        worst = max(worst, (n, e), key=lambda t: t[1])
        assert e < 20 * tol, (n, e)
    log(f"unfused vs fused[{precision},B={B}] features {e_f:.3e} worst gradient {worst[0]} {worst[1]:.3e}")
