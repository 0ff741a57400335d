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
x, w, b = r(M, 384), r(1152, 384, sc=.05), r(1152, dt=torch.float32)
y = torch.empty(M, 1152, device=dev, dtype=torch.bfloat16)
timeit(lambda: ops.linear_fwd(x, w, b), "qkv", 2.0*M*1152*384)
w2, b2 = r(1536, 384, sc=.05), r(1536, dt=torch.float32)
timeit(lambda: ops.linear_fwd(x, w2, b2, gelu=True), "fc1+gelu", 2.0*M*1536*384)

for mode in (True, "ws"):
    timeit(lambda: ops.linear_fwd(x, w, b, persistent=mode), f"qkv persistent={mode}", 2.0*M*1152*384)
    timeit(lambda: ops.linear_fwd(x, w2, b2, gelu=True, persistent=mode), f"fc1+gelu persistent={mode}", 2.0*M*1536*384)
y0 = ops.linear_fwd(x, w, b); y1 = ops.linear_fwd(x, w, b, persistent="ws")
print("ws == default:", torch.equal(y0, y1))
a0, g0 = ops.linear_fwd(x, w2, b2, gelu=True); a1, g1 = ops.linear_fwd(x, w2, b2, gelu=True, persistent="ws")
print("ws gelu == default:", torch.equal(a0, a1), torch.equal(g0, g1))
