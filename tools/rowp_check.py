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
print("ALL OK" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
