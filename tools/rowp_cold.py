"""Row-complete GEMMs with COLD operands (rotating through distinct tensor sets larger than the Infinity Cache), as inside a training step.
   python3 tools/rowp_cold.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
M, D = 128 * 197, 384
NSET = 5
def sp(x): return ops.split_pack(x)
for K, name in ((D, "proj+LN"), (4 * D, "fc2+LN")):
    w, b = sp(torch.randn(D, K, device=dev) * .05), torch.randn(D, device=dev)
    g, be = torch.rand(D, device=dev) + .5, torch.randn(D, device=dev)
    sets = [(sp(torch.randn(M, K, device=dev)), torch.randn(M, D, device=dev)) for _ in range(NSET)]
    line = f"{name:8s} cold:"
    for mode in (0, 1, 0, 1):
        os.environ["MFVIT_ROWP"] = str(mode)
        for a, res in sets: ops.linear_res_ln_fwd(a, w, b, res, g, be, 1e-6, split=True)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for it in range(4):
            for a, res in sets: ops.linear_res_ln_fwd(a, w, b, res, g, be, 1e-6, split=True)
        e.record(); torch.cuda.synchronize()
        line += f"  {'rowp' if mode else 'row '} {s.elapsed_time(e) * 1e3 / (4 * NSET):7.1f} us"
    print(line, flush=True)
    del sets
