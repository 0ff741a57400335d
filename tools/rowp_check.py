"""Row-complete GEMMs: gemm_rowp.hip (MFVIT_ROWP=1) against gemm_nt_row (MFVIT_ROWP=0) and against f64 math; timing in one process.
   python3 tools/rowp_check.py [quick]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
D = 384


def sp(x):
    return ops.split_pack(x)


def timeit(mode, fn, n=20):
    os.environ["MFVIT_ROWP"] = str(mode)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


ok = True
for M, tag in [(128 * 197, "full"), (128 * 197 - 57, "ragged"), (4096 + 40, "small")][: 1 if quick else 3]:
    for K, name in ((D, "proj+LN"), (4 * D, "fc2+LN")):
        a32, w32 = torch.randn(M, K, device=dev), torch.randn(D, K, device=dev) * .05
        a, w = sp(a32), sp(w32)
        b, res = torch.randn(D, device=dev), torch.randn(M, D, device=dev)
        g, be = torch.rand(D, device=dev) + .5, torch.randn(D, device=dev)
        # f64 reference on the ROUNDED operands
        ar, wr = ops.split_unpack(a).double(), ops.split_unpack(w).double()
        x64 = ar @ wr.T + b.double() + res.double()
        mu = x64.mean(1, keepdim=True)
        var = ((x64 - mu) ** 2).mean(1, keepdim=True)
        y64 = (x64 - mu) / torch.sqrt(var + 1e-6) * g.double() + be.double()
        for y_f32 in (False, True):
            fn = lambda: ops.linear_res_ln_fwd(a, w, b, res, g, be, 1e-6, y_f32=y_f32, split=True)
            line = f"M={M:6d} {tag:6s} {name:8s} y_f32={int(y_f32)}"
            for mode in (0, 1):
                os.environ["MFVIT_ROWP"] = str(mode)
                x, y, mean, rstd = fn()
                torch.cuda.synchronize()
                yv = y if y_f32 else ops.split_unpack(y)
                ex, ey = rel(x, x64), rel(yv, y64)
                em, er = rel(mean, mu.squeeze(1)), rel(rstd, 1 / torch.sqrt(var + 1e-6).squeeze(1))
                good = ex < 2e-5 and ey < 3e-5 and em < 2e-5 and er < 2e-5
                ok &= good
                line += f" | {'rowp' if mode else 'row '}: x {ex:.1e} y {ey:.1e} mean {em:.1e} rstd {er:.1e}{'' if good else ' BAD'}"
            if tag == "full":
                line += f" | row {timeit(0, fn):7.1f} us  rowp {timeit(1, fn):7.1f} us"
            print(line, flush=True)
# ---- LayerNorm-backward row kernel: dx = LNbwd(dy @ wt.T; x) + dres, dgamma, dbeta, dcol
for M, tag in [(128 * 197, "full"), (128 * 197 - 57, "ragged"), (4096 + 40, "small")][: 1 if quick else 3]:
    for K, name in ((3 * D, "qkvd+LNb"), (4 * D, "fc1d+LNb")):
        dy32, wt32 = torch.randn(M, K, device=dev) * .1, torch.randn(D, K, device=dev) * .05
        dy, wt = sp(dy32), sp(wt32)
        x = torch.randn(M, D, device=dev) * 1.5 + .3
        mean = x.mean(1)
        rstd = 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
        g = torch.rand(D, device=dev) + .5
        dres = torch.randn(M, D, device=dev) * .1
        d64 = ops.split_unpack(dy).double() @ ops.split_unpack(wt).double().T
        h = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
        gg = d64 * g.double()
        dx64 = rstd.double()[:, None] * (gg - gg.mean(1, keepdim=True) - h * (gg * h).mean(1, keepdim=True)) + dres.double()
        dg64, db64, dc64 = (d64 * h).sum(0), d64.sum(0), dx64.sum(0)
        fn = lambda: ops.linear_dgrad_ln_bwd(dy, wt, x, mean, rstd, g, dres, split=True)
        line = f"M={M:6d} {tag:6s} {name:8s}"
        for mode in (0, 2):
            os.environ["MFVIT_ROWP"] = str(mode)
            dx, dxt, dgm, dbt, dcl = fn()
            torch.cuda.synchronize()
            e = [rel(dx, dx64), rel(ops.split_unpack(dxt), dx64), rel(dgm, dg64), rel(dbt, db64), rel(dcl, dc64)]
            good = max(e[:2]) < 3e-5 and max(e[2:]) < 2e-4
            ok &= good
            line += f" | {'rowp' if mode else 'row '}: dx {e[0]:.1e} dxT {e[1]:.1e} dgamma {e[2]:.1e} dbeta {e[3]:.1e} dcol {e[4]:.1e}{'' if good else ' BAD'}"
        if tag == "full":
            line += f" | row {timeit(0, fn):7.1f} us  rowp {timeit(2, fn):7.1f} us"
        print(line, flush=True)
print("ALL OK" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
