"""time one op through whatever library MFVIT_LIB names: python3 tools/pp_time.py <qkv|fc1|projd> <MFVIT_PP mode> [label]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
op, mode = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ""
M, D = 128 * 197, 384
N = {"qkv": 3 * D, "fc1": 4 * D, "projd": D}[op]
x = ops.split_pack(torch.randn(M, D, device=dev))
w = ops.split_pack(torch.randn(N, D, device=dev) * .05)
b = torch.randn(N, device=dev)
os.environ["MFVIT_PP"] = mode
fn = (lambda: ops.linear_fwd(x, w, b, gelu=True, split=True)) if op == "fc1" else (lambda: ops.linear_fwd(x, w, b if op == "qkv" else None, split=True))
for _ in range(5): fn()
torch.cuda.synchronize()
ts = []
for rep in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * 1e3 / 20)
print(f"{op} pp{mode} {label:40s} {min(ts):7.1f} us (median {sorted(ts)[1]:7.1f})", flush=True)
