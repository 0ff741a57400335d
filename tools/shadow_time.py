"""Weight-shadow refresh (f32 master weights -> operand type, straight + transposed copies) of two vit_small encoders: time per optimizer step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
shapes = [(1152, 384), (384, 384), (1536, 384), (384, 1536)]
ws = [torch.randn(s, device=dev) for s in shapes for _ in range(24)]
for name, dt, split in (("bf16x3", torch.bfloat16, True), ("fp16", torch.float16, False)):
    for w in ws:
        ops.cast_transpose(w, dt, split=split)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        for w in ws:
            ops.cast_transpose(w, dt, split=split)
    e.record()
    torch.cuda.synchronize()
    n = sum(w.numel() for w in ws)
    us = s.elapsed_time(e) * 1e3 / 10
    by = n * (4 + (8 if split else 4))
    print(f"shadow refresh {name}: {n / 1e6:.1f} M weights in {len(ws)} launches: {us:.1f} us  ({by / us / 1e6:.2f} TB/s)")
