import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
M = 128 * 197
def r(*s, dt=torch.bfloat16, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(dt)
def timeit(fn, name, flops, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / n
    print(f"{name:34s} {us:8.1f} us {flops/us/1e6:8.1f} TF", flush=True)
for N in (384, 1152, 1536):
    b = r(N, dt=torch.float32)
    for K in (64, 384, 1536):
        x, w = r(M, K), r(N, K, sc=.05)
        timeit(lambda: ops.linear_fwd(x, w, b), f"tile N={N} K={K}", 2.0*M*N*K)
