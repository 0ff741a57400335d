import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
M, D = 128 * 197, 384
def r(*s, dt=torch.bfloat16, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(dt)
def timeit(fn, name, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:34s} {s.elapsed_time(e) * 1e3 / n:8.1f} us", flush=True)
res = r(M, D, dt=torch.float32); g, be = torch.ones(D, device=dev), torch.zeros(D, device=dev); b = r(D, dt=torch.float32)
for K in (64, 128, 384, 768, 1536):
    x, w = r(M, K), r(D, K, sc=.05)
    timeit(lambda: ops.linear_res_ln_fwd(x, w, b, res, g, be, 1e-6), f"row_fwd K={K}")
    timeit(lambda: ops.linear_res_ln_fwd(x, w, b, None, g, be, 1e-6), f"row_fwd K={K} no-res")
y = torch.empty(M, D, device=dev)
timeit(lambda: ops.layernorm_fwd(res, g, be, 1e-6, out_dtype=torch.bfloat16), "plain LN rows f32->bf16 (58 MB)")
timeit(lambda: torch.add(res, res, out=y), "torch add f32 (116 MB)")
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
for K in (64, 384, 1152, 1536):
    dy, wt = r(M, K), r(D, K, sc=.05)
    timeit(lambda: ops.linear_dgrad_ln_bwd(dy, wt, res, mean, rstd, g, res), f"row_bwd K={K}")
    timeit(lambda: ops.linear_dgrad_ln_bwd(dy, wt, res, mean, rstd, g, None, want_copy=False), f"row_bwd K={K} no-res no-copy")
