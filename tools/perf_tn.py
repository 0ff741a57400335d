import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
M, D, F = 128 * 197, 384, 1536
def r(*s): return torch.randn(*s, device=dev).bfloat16()
def timeit(fn, flops, name, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / n
    print(f"{name:34s} {us:8.1f} us  {flops/us/1e6:7.1f} TFLOP/s", flush=True)
x384, x1536, dy1152, dy1536 = r(M, D), r(M, F), r(M, 3*D), r(M, F)
t = os.environ.get("MFVIT_TN_TARGET", "def")
for name, a, b, n, k in (("qkv", dy1152, x384, 3*D, D), ("fc1", dy1536, x384, F, D), ("fc2", x384, x1536, D, F), ("proj", x384, x384, D, D)):
    out = torch.zeros(n, k, device=dev)
    timeit(lambda: ops.linear_wgrad(a, b, out=out), 2.0*M*n*k, f"wgrad {name} target={t}")
