"""Attention forward / backward of the bench shape (B = 128, T = 197, 12 heads x 32) per precision: time per launch, cold inputs excluded."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")


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


for prec in ("bf16x3", "bf16", "fp16"):
    x = torch.randn(128, 197, 1152, device=dev)
    d = torch.randn(128, 197, 384, device=dev)
    split = prec == "bf16x3"
    if split:
        qkv, do = ops.split_pack(x.view(-1, 1152)).view(128, 197, -1), ops.split_pack(d.view(-1, 384)).view(128, 197, -1)
    else:
        dt = torch.bfloat16 if prec == "bf16" else torch.float16
        qkv, do = x.to(dt), d.to(dt)
    o, lse = ops.attention_fwd(qkv, 12, split=split)
    tf = timeit(lambda: ops.attention_fwd(qkv, 12, split=split))
    tb = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, 12, want_dbias=False, split=split))
    print(f"{prec:7s} attention fwd {tf:6.1f} us   bwd {tb:6.1f} us", flush=True)
