"""wgrad timing with COLD operands: rotate through enough distinct input / output sets that nothing is left in L2 / Infinity Cache."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
M, D, F = 128 * 197, 384, 1536
NSET = 6
def r(*s, sc=1.0): return (torch.randn(*s, device=dev) * sc).bfloat16()
for name, N, K in (("qkv 1152x384", 3 * D, D), ("fc1 1536x384", F, D), ("fc2 384x1536", D, F), ("proj 384x384", D, D)):
    sets = [(r(M, N), r(M, K), torch.zeros(N, K, device=dev)) for _ in range(NSET)]
    for dy, x, o in sets: ops.linear_wgrad(dy, x, out=o)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for it in range(4):
        for dy, x, o in sets: ops.linear_wgrad(dy, x, out=o)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / (4 * NSET)
    print(f"wgrad {name:14s} cold {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
    del sets
