"""The GEMM kernels of the split-bf16 (bf16x3) path that share a job, each selected by the one run-time switch that is left for it (round 5 removed
the variant switches whose A/B had been decided: DESIGN.md 5) and called through the C ABI, against float64 math on the ROUNDED operands (hi + lo),
at the bench's row count (M = 128 x 197 = 25,216), a ragged M and small ones:

  row-complete kernels   csrc/gemm_rowp.hip (default: one tall tile per CU, mixed tile heights, LayerNorm epilogues; LayerNorm-backward epilogue
                         with x staged by LDS-DMA) and csrc/gemm.hip gemm_nt_row (MFVIT_ROWP=0: what the patch embedding and f32 run)
  weight gradient        csrc/gemm_tn2.hip (default from M = 2,048: LDS-DMA ring, 8 waves) and csrc/gemm.hip gemm_tn (MFVIT_TN_GLDS=0)

The switches are read by the library at every launch, so one process covers all of them.  Tolerances: 3e-5 of the largest output element for
the GEMM outputs (f32 accumulation of 384 - 25,216 products of 16-bit-exact hi / lo parts), 2e-4 for the column sums over 25,216 rows."""
import pytest
import torch

from mfvit import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D, F = 384, 1536
SHAPES = [(128 * 197, "full"), (128 * 197 - 57, "ragged"), (4096 + 40, "small"), (16 * 197, "b16")]
# the tall-tile kernel is instantiated per number of 16-row fragments a tile carries (1 .. 7): one M for each, and tiles of a single row
ROW_SHAPES = SHAPES + [(16 * 197, "B16: 1 fragment"), (32 * 197 - 3, "B32: 2"), (48 * 197, "B48: 3"), (64 * 197, "B64: 4"), (96 * 197, "B96: 5"),
                       (112 * 197, "B112: 6"), (197, "one row per tile")]


def sp(x):
    return ops.split_pack(x)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def _gen(seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return g


def rn(g, *shape, sc=1.0):
    return torch.randn(*shape, device=DEV, generator=g) * sc


@pytest.mark.parametrize("M,tag", ROW_SHAPES)
@pytest.mark.parametrize("K", [D, F])
@pytest.mark.parametrize("mode", ["0", "1"])
def test_row_kernel_forward_residual_layernorm(monkeypatch, M, tag, K, mode):
    """x = a @ w.T + bias + res ; y = LayerNorm(x)  (proj / fc2 of timm's Block + the following norm): both row kernels, split and f32 y."""
    monkeypatch.setenv("MFVIT_ROWP", mode)
    g = _gen(11 + K)
    a, w = sp(rn(g, M, K)), sp(rn(g, D, K, sc=.05))
    b, res = rn(g, D), rn(g, M, D)
    gam, bet = torch.rand(D, device=DEV, generator=g) + .5, rn(g, D)
    x64 = ops.split_unpack(a).double() @ ops.split_unpack(w).double().T + b.double() + res.double()
    mu = x64.mean(1, keepdim=True)
    var = ((x64 - mu) ** 2).mean(1, keepdim=True)
    y64 = (x64 - mu) / torch.sqrt(var + 1e-6) * gam.double() + bet.double()
    for y_f32 in (False, True):
        x, y, mean, rstd = ops.linear_res_ln_fwd(a, w, b, res, gam, bet, 1e-6, y_f32=y_f32, split=True)
        yv = y if y_f32 else ops.split_unpack(y)
        errs = dict(x=rel(x, x64), y=rel(yv, y64), mean=rel(mean, mu.squeeze(1)), rstd=rel(rstd, 1 / torch.sqrt(var + 1e-6).squeeze(1)))
        assert max(errs.values()) < 3e-5, (tag, K, mode, y_f32, errs)


@pytest.mark.parametrize("M,tag", ROW_SHAPES)
@pytest.mark.parametrize("K", [3 * D, F])
@pytest.mark.parametrize("mode", ["0", "1"])
def test_row_kernel_dgrad_layernorm_backward(monkeypatch, M, tag, K, mode):
    """dx = LayerNorm-backward(dy @ wt.T; x) + dres, dgamma, dbeta, column sums of dx  (qkv / fc1 data gradients + norm1 / norm2 backward)."""
    monkeypatch.setenv("MFVIT_ROWP", mode)
    g = _gen(23 + K)
    dy, wt = sp(rn(g, M, K, sc=.1)), sp(rn(g, D, K, sc=.05))
    x = rn(g, M, D, sc=1.5) + .3
    mean = x.mean(1)
    rstd = 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
    gam, dres = torch.rand(D, device=DEV, generator=g) + .5, rn(g, M, D, sc=.1)
    d64 = ops.split_unpack(dy).double() @ ops.split_unpack(wt).double().T
    h = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
    gg = d64 * gam.double()
    dx64 = rstd.double()[:, None] * (gg - gg.mean(1, keepdim=True) - h * (gg * h).mean(1, keepdim=True)) + dres.double()
    dx, dxt, dgm, dbt, dcl = ops.linear_dgrad_ln_bwd(dy, wt, x, mean, rstd, gam, dres, split=True)
    errs = dict(dx=rel(dx, dx64), dx_split=rel(ops.split_unpack(dxt), dx64), dgamma=rel(dgm, (d64 * h).sum(0)), dbeta=rel(dbt, d64.sum(0)),
                dcol=rel(dcl, dx64.sum(0)))
    assert max(errs["dx"], errs["dx_split"]) < 3e-5 and max(errs["dgamma"], errs["dbeta"], errs["dcol"]) < 2e-4, (tag, K, mode, errs)


@pytest.mark.parametrize("M", [1024, 3152, 3349, 6304])
@pytest.mark.parametrize("ksplit", ["2", "3", "4", "-1"])
@pytest.mark.parametrize("share", [1, 2])
def test_row_kernels_with_k_splits(monkeypatch, M, ksplit, share):
    """The tall-tile row kernels with K split over 2 - 4 workgroups per row tile (small M, round 5: partial accumulator tiles through scratch, the workgroup
    that arrives last adds them and runs the epilogue): forward (+ residual + LayerNorm) at K = 1536, LayerNorm backward at K = 1152 and 1536, forced split
    counts and the launcher's own choice (-1), alone on the chip and with the half-chip hint of the two-stream model; against float64, and REPEATED on the
    same scratch (the arrival counters must be back at zero after every launch)."""
    from mfvit._lib import lib
    monkeypatch.setenv("MFVIT_ROWP_KSPLIT", ksplit)
    lib().mfvit_set_stream_share(share)
    try:
        scratch = torch.empty(ops.ROWP_SCRATCH_FLOATS, device=DEV, dtype=torch.float32)
        g = _gen(51 + M)
        K = F
        a, w = sp(rn(g, M, K)), sp(rn(g, D, K, sc=.05))
        b, res = rn(g, D), rn(g, M, D)
        gam, bet = torch.rand(D, device=DEV, generator=g) + .5, rn(g, D)
        x64 = ops.split_unpack(a).double() @ ops.split_unpack(w).double().T + b.double() + res.double()
        mu = x64.mean(1, keepdim=True)
        var = ((x64 - mu) ** 2).mean(1, keepdim=True)
        y64 = (x64 - mu) / torch.sqrt(var + 1e-6) * gam.double() + bet.double()
        for rep in range(3):
            x, y, mean, rstd = ops.linear_res_ln_fwd(a, w, b, res, gam, bet, 1e-6, split=True, scratch=scratch)
            errs = dict(x=rel(x, x64), y=rel(ops.split_unpack(y), y64), mean=rel(mean, mu.squeeze(1)), rstd=rel(rstd, 1 / torch.sqrt(var + 1e-6).squeeze(1)))
            assert max(errs.values()) < 3e-5, (M, ksplit, share, rep, errs)
        for K in (3 * D, F):
            dy, wt = sp(rn(g, M, K, sc=.1)), sp(rn(g, D, K, sc=.05))
            xx = rn(g, M, D, sc=1.5) + .3
            mean = xx.mean(1)
            rstd = 1 / torch.sqrt(xx.var(1, unbiased=False) + 1e-6)
            dres = rn(g, M, D, sc=.1)
            d64 = ops.split_unpack(dy).double() @ ops.split_unpack(wt).double().T
            h = (xx.double() - mean.double()[:, None]) * rstd.double()[:, None]
            gg = d64 * gam.double()
            dx64 = rstd.double()[:, None] * (gg - gg.mean(1, keepdim=True) - h * (gg * h).mean(1, keepdim=True)) + dres.double()
            for rep in range(2):
                dx, dxt, dgm, dbt, dcl = ops.linear_dgrad_ln_bwd(dy, wt, xx, mean, rstd, gam, dres, split=True, scratch=scratch)
                errs = dict(dx=rel(dx, dx64), dx_split=rel(ops.split_unpack(dxt), dx64), dgamma=rel(dgm, (d64 * h).sum(0)), dbeta=rel(dbt, d64.sum(0)),
                            dcol=rel(dcl, dx64.sum(0)))
                assert max(errs["dx"], errs["dx_split"]) < 3e-5 and max(errs["dgamma"], errs["dbeta"], errs["dcol"]) < 2e-4, (M, K, ksplit, share, rep, errs)
    finally:
        lib().mfvit_set_stream_share(1)


@pytest.mark.parametrize("M,tag", SHAPES)
@pytest.mark.parametrize("glds", ["1", "0"])
def test_weight_gradient_variants(monkeypatch, M, tag, glds):
    """dW = dy.T @ x for the four linears of a block, split bf16 and plain bf16, on the LDS-DMA kernel of gemm_tn2.hip and on gemm_tn."""
    monkeypatch.setenv("MFVIT_TN_GLDS", glds)
    w8 = il = glds
    g = _gen(37)
    for name, n, k in (("qkv", 3 * D, D), ("fc1", F, D), ("fc2", D, F), ("proj", D, D)):
        a32, b32 = rn(g, M, n, sc=.1), rn(g, M, k)
        for split in (True, False):
            a, b = (sp(a32), sp(b32)) if split else (a32.bfloat16(), b32.bfloat16())
            ref = (ops.split_unpack(a).double().T @ ops.split_unpack(b).double()) if split else (a.double().T @ b.double())
            out = ops.linear_wgrad(a, b, split=split)
            assert rel(out, ref) < 2e-5, (tag, name, split, w8, il, rel(out, ref))
            out2 = ops.linear_wgrad(a, b, out=out.clone(), split=split)          # accumulates into an existing gradient
            assert rel(out2, 2 * ref) < 2e-5, (tag, name, split, w8, il, "accumulate")


def test_two_term_weight_gradient_is_opt_in_and_bf16_grade_in_dy(monkeypatch):
    """MFVIT_WGRAD_TERMS=2 (csrc/gemm_tn2.hip, TWO): the dY_lo x_hi product is dropped - a throughput option (8 % of the weight-gradient class,
    3.6 % of the train step), NOT the default: dW then carries dY at bf16 precision.  Against float64 on the split operands the
    error is that of rounding dY to bf16 (a few 1e-3 of the largest element), against float64 on (bf16(dY), split x) it is f32-grade - and the
    bias column sums, which keep both parts, stay f32-grade."""
    g = _gen(91)
    M = 197 * 128
    dy32, x32 = rn(g, M, 3 * D, sc=.1), rn(g, M, D)
    dy, x = sp(dy32), sp(x32)
    ref = ops.split_unpack(dy).double().T @ ops.split_unpack(x).double()
    ref_hi = dy32.to(torch.bfloat16).double().T @ ops.split_unpack(x).double()
    full = ops.linear_wgrad(dy, x, split=True)
    monkeypatch.setenv("MFVIT_WGRAD_TERMS", "2")
    two = ops.linear_wgrad(dy, x, split=True)
    e_full, e_two, e_two_hi = rel(full, ref), rel(two, ref), rel(two, ref_hi)
    y1, attn = sp(rn(g, M, D)), sp(rn(g, M, D, sc=.1))
    dw_a, db_a, dw_b = ops.linear_wgrad_pair(dy, x, attn, y1, split=True)     # the paired launch of the encoder backward takes the same switch
    e_pair, e_bias = rel(dw_a, ref_hi), rel(db_a, ops.split_unpack(dy).double().sum(0))
    assert e_full < 2e-5 and 2e-4 < e_two < 6e-3 and e_two_hi < 2e-5 and e_pair < 2e-5 and e_bias < 2e-5, (e_full, e_two, e_two_hi, e_pair, e_bias)


@pytest.mark.parametrize("M,tag", SHAPES)
@pytest.mark.parametrize("kind", ["split", "bf16", "fp16"])
def test_paired_weight_gradient_launch(M, tag, kind):
    """`gemm_tn_glds_pair` (csrc/gemm_tn2.hip) through `mfvit_linear_wgrad_pair`: dWqkv (+ d qkv.bias column sums) and dWproj of a timm
    attention block in ONE launch - the default of the timed encoder backward at M >= 2048 - against float64 dY^T X on the rounded
    operands for BOTH outputs and the bias sums, at the bench's M, a ragged M and a small one."""
    g = _gen(71)
    dqkv32, y132 = rn(g, M, 3 * D, sc=.1), rn(g, M, D)
    gmid32, attn32 = rn(g, M, D, sc=.1), rn(g, M, D)
    if kind == "split":
        pk, up = sp, (lambda t: ops.split_unpack(t).double())
    else:
        dt = torch.bfloat16 if kind == "bf16" else torch.float16
        pk, up = (lambda t: t.to(dt)), (lambda t: t.double())
    dqkv, y1, gmid, attn = pk(dqkv32), pk(y132), pk(gmid32), pk(attn32)
    dw_a, db_a, dw_b = ops.linear_wgrad_pair(dqkv, y1, gmid, attn, split=kind == "split")
    ra, rb, rbias = up(dqkv).T @ up(y1), up(gmid).T @ up(attn), up(dqkv).sum(0)
    errs = dict(dWqkv=rel(dw_a, ra), dWproj=rel(dw_b, rb), dbias=rel(db_a, rbias))
    assert errs["dWqkv"] < 2e-5 and errs["dWproj"] < 2e-5 and errs["dbias"] < 2e-4, (tag, kind, errs)
    # and the same numbers as the two single launches it replaces (same tiles, same split chunks; float atomics: rounding-level agreement)
    sa, sb = ops.linear_wgrad(dqkv, y1, split=kind == "split"), ops.linear_wgrad(gmid, attn, split=kind == "split")
    assert rel(dw_a, sa) < 2e-6 and rel(dw_b, sb) < 2e-6, (tag, kind)
    with pytest.raises(Exception):                                       # shapes the paired kernel does not take are refused, not mis-computed
        ops.linear_wgrad_pair(dqkv[:1000], y1[:1000], gmid[:1000], attn[:1000], split=kind == "split")


@pytest.mark.parametrize("M", [128 * 197, 64 * 197, 4096 + 40, 16 * 197, 2048 + 37, 901])
def test_weight_gradient_with_partial_scratch(M):
    """The plain-store split-partial path (scratch given: `mfvit_linear_wgrad_ws`, MFVIT_TN_PART=1 inside the encoder) at row counts where the
    KR-rounded split chunks overshoot M: every split must write its partial tile (an empty split used to leave stale scratch in the sum)."""
    g = _gen(53)
    scratch = torch.full((ops.WGRAD_SCRATCH_FLOATS,), float("nan"), device=DEV)       # stale scratch must never reach the result
    for name, n, k in (("proj", D, D), ("fc1", F, D)):
        a32, b32 = rn(g, M, n, sc=.1), rn(g, M, k)
        for kind in ("split", "bf16", "fp32"):
            if kind == "split":
                a, b = sp(a32), sp(b32)
                ref = ops.split_unpack(a).double().T @ ops.split_unpack(b).double()
            elif kind == "bf16":
                a, b = a32.bfloat16(), b32.bfloat16()
                ref = a.double().T @ b.double()
            else:
                a, b = a32, b32
                ref = a.double().T @ b.double()
            out = ops.linear_wgrad(a, b, scratch=scratch, split=kind == "split")
            assert bool(torch.isfinite(out).all()) and rel(out, ref) < 2e-5, (M, name, kind, rel(out, ref))


@pytest.mark.parametrize("M,tag", [(128 * 197, "full"), (128 * 197 - 57, "ragged"), (16 * 197, "B16: 1 fragment"), (48 * 197, "B48: 3"), (197, "one row per tile")])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("mode", ["0", "1"])
def test_row_kernels_plain_16_bit_types(monkeypatch, M, tag, dt, mode):
    """The tall-tile row kernels in the plain 16-bit types (round 4: gemm_rowp instantiated for bf16 / f16 - a 128-byte row is two k steps of one MFMA
    each instead of one split k group of three) against float64 on the rounded operands, forward (K = 384, 1536) and LayerNorm backward (K = 1152, 1536);
    mode 0 = gemm_nt_row, the kernel they replace.  Tolerances: the output's own rounding (2^-9 bf16, 2^-12 fp16) for the operand-type tensors, 3e-5 / 2e-4
    for the f32 outputs and the column sums."""
    monkeypatch.setenv("MFVIT_ROWP", mode)
    tol_t = 6e-3 if dt == torch.bfloat16 else 8e-4
    g = _gen(71)
    for K in (D, F):
        a, w = rn(g, M, K).to(dt), rn(g, D, K, sc=.05).to(dt)
        b, res = rn(g, D), rn(g, M, D)
        gam, bet = torch.rand(D, device=DEV, generator=g) + .5, rn(g, D)
        x64 = a.double() @ w.double().T + b.double() + res.double()
        mu = x64.mean(1, keepdim=True)
        var = ((x64 - mu) ** 2).mean(1, keepdim=True)
        y64 = (x64 - mu) / torch.sqrt(var + 1e-6) * gam.double() + bet.double()
        for y_f32 in (False, True):
            x, y, mean, rstd = ops.linear_res_ln_fwd(a, w, b, res, gam, bet, 1e-6, y_f32=y_f32)
            errs = dict(x=rel(x, x64), mean=rel(mean, mu.squeeze(1)), rstd=rel(rstd, 1 / torch.sqrt(var + 1e-6).squeeze(1)))
            assert max(errs.values()) < 3e-5 and rel(y, y64) < (3e-5 if y_f32 else tol_t), (tag, K, mode, y_f32, errs, rel(y, y64))
    for K in (3 * D, F):
        dy, wt = rn(g, M, K, sc=.1).to(dt), rn(g, D, K, sc=.05).to(dt)
        x = rn(g, M, D, sc=1.5) + .3
        mean = x.mean(1)
        rstd = 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
        gam, dres = torch.rand(D, device=DEV, generator=g) + .5, rn(g, M, D, sc=.1)
        d64 = dy.double() @ wt.double().T
        h = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
        gg = d64 * gam.double()
        dx64 = rstd.double()[:, None] * (gg - gg.mean(1, keepdim=True) - h * (gg * h).mean(1, keepdim=True)) + dres.double()
        dx, dxt, dgm, dbt, dcl = ops.linear_dgrad_ln_bwd(dy, wt, x, mean, rstd, gam, dres)
        errs = dict(dx=rel(dx, dx64), dgamma=rel(dgm, (d64 * h).sum(0)), dbeta=rel(dbt, d64.sum(0)), dcol=rel(dcl, dx64.sum(0)))
        assert errs["dx"] < 3e-5 and rel(dxt, dx64) < tol_t and max(errs["dgamma"], errs["dbeta"], errs["dcol"]) < 2e-4, (tag, K, mode, errs)



def test_stream_share_hint_changes_grids_not_results():
    """mfvit_set_stream_share (include/mfvit.h; set to 2 by the two-stream CA model): at a small batch the row-complete GEMMs and the weight
    gradients launch about half as many, longer workgroups.  Row outputs are bit-identical (every row's products and LayerNorm are computed in
    the same order whatever the tile height); column sums and weight gradients agree at rounding level (different partial-sum order)."""
    from mfvit import _lib
    g = _gen(91)
    M = 16 * 197
    a, w, bias = sp(rn(g, M, 4 * D, sc=.3)), sp(rn(g, D, 4 * D, sc=.05)), rn(g, D).to(DEV)
    res, gamma, beta = rn(g, M, D).to(DEV), (1 + 0.1 * rn(g, D)).to(DEV), rn(g, D, sc=.1).to(DEV)
    dy, x = sp(rn(g, M, 3 * D, sc=.1)), sp(rn(g, M, D))
    out = {}
    try:
        for share in (1, 2):
            _lib.lib().mfvit_set_stream_share(share)
            xo, y, mean, rstd = ops.linear_res_ln_fwd(a, w, bias, res, gamma, beta, 1e-6, split=True)
            dw = ops.linear_wgrad(dy, x, split=True)
            out[share] = (xo.clone(), y.clone(), mean.clone(), rstd.clone(), dw.clone())
    finally:
        _lib.lib().mfvit_set_stream_share(1)
    for i in range(4):
        assert torch.equal(out[1][i], out[2][i]), i
    assert rel(out[2][4], out[1][4].double()) < 2e-6
