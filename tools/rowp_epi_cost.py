"""Main loop vs epilogue of the row-complete kernel (csrc/gemm_rowp.hip): the same launches with MFVIT_ROWP_NOEPI=1 (no epilogue, results invalid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
D, M = 384, 128 * 197
sp = ops.split_pack


def timeit(fn, n=20):
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


os.environ["MFVIT_ROWP"] = "2"
for K, name in ((D, "proj+LN"), (4 * D, "fc2+LN")):
    a, w = sp(torch.randn(M, K, device=dev)), sp(torch.randn(D, K, device=dev) * .05)
    b, res = torch.randn(D, device=dev), torch.randn(M, D, device=dev)
    g, be = torch.rand(D, device=dev) + .5, torch.randn(D, device=dev)
    fn = lambda: ops.linear_res_ln_fwd(a, w, b, res, g, be, 1e-6, y_f32=False, split=True)
    os.environ["MFVIT_ROWP_NOEPI"] = "0"; t1 = timeit(fn)
    os.environ["MFVIT_ROWP_NOEPI"] = "1"; t0 = timeit(fn)
    print(f"fwd {name:9s} K={K:5d}: full {t1:6.1f} us, main loop only {t0:6.1f} us, epilogue {t1 - t0:5.1f} us", flush=True)
for K, name in ((D, "projd+LNb"), (3 * D, "qkvd+LNb"), (4 * D, "fc1d+LNb")):
    dy, wt = sp(torch.randn(M, K, device=dev) * .1), sp(torch.randn(D, K, device=dev) * .05)
    x = torch.randn(M, D, device=dev) * 1.5 + .3
    mean, rstd = x.mean(1), 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
    g, dres = torch.rand(D, device=dev) + .5, torch.randn(M, D, device=dev) * .1
    fn = lambda: ops.linear_dgrad_ln_bwd(dy, wt, x, mean, rstd, g, dres, split=True)
    os.environ["MFVIT_ROWP_NOEPI"] = "0"; t1 = timeit(fn)
    os.environ["MFVIT_ROWP_NOEPI"] = "1"; t0 = timeit(fn)
    os.environ["MFVIT_ROWP"] = "0"; told = timeit(fn); os.environ["MFVIT_ROWP"] = "2"
    print(f"bwd {name:9s} K={K:5d}: full {t1:6.1f} us, main loop only {t0:6.1f} us, epilogue {t1 - t0:5.1f} us   (gemm_nt_row {told:6.1f} us)", flush=True)
