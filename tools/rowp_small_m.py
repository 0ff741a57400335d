"""Row-complete kernels at small M: gemm_nt_row (MFVIT_ROWP=0) against the tall-tile kernel gemm_rowp (MFVIT_ROWP=2, MFVIT_ROWP_MINM=1) per
launch, split bf16, forward (fc2 + residual + LN, proj + LN) and backward (fc1-dgrad / qkv-dgrad + LN backward)."""
import os, sys
os.environ["MFVIT_AB_LIVE"] = "1"
os.environ["MFVIT_ROWP_MINM"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
D = 384


def timeit(fn, n=30):
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


sp = ops.split_pack
for M in [int(m) for m in os.environ.get("MS", "788,1576,2364,3152,3940,4728,6304,12608").split(",")]:
    row = f"M={M:6d} (B={M // 197:3d}):"
    for name, K, kind in (("proj+LN", D, "f"), ("fc2+LN", 4 * D, "f"), ("qkv-dgrad+LNbwd", 3 * D, "b"), ("fc1-dgrad+LNbwd", 4 * D, "b")):
        a, w = sp(torch.randn(M, K, device=dev)), sp(torch.randn(D, K, device=dev) * .05)
        b, res, g, be = torch.randn(D, device=dev), torch.randn(M, D, device=dev), torch.rand(D, device=dev) + .5, torch.randn(D, device=dev)
        x = torch.randn(M, D, device=dev)
        mean, rstd = x.mean(1), 1 / x.std(1)
        ts = []
        for mode in ("0", "2"):
            os.environ["MFVIT_ROWP"] = mode
            if kind == "f":
                fn = lambda: ops.linear_res_ln_fwd(a, w, b, res, g, be, 1e-6, split=True)
            else:
                fn = lambda: ops.linear_dgrad_ln_bwd(a, w, x, mean, rstd, g, res, split=True)
            ts.append(timeit(fn))
        row += f"  {name} {ts[0]:6.1f} -> {ts[1]:6.1f}"
    print(row, flush=True)
