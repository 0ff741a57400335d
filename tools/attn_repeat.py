"""Attention backward run-to-run repeatability (the kernel has no atomics: results must be bit-identical) and error against float64, bench shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
torch.manual_seed(3)
B, T, H, D = 16, 197, 12, 384
for prec in ("bf16x3", "fp16"):
    x = torch.randn(B, T, 3 * D, device=dev)
    d = torch.randn(B, T, D, device=dev)
    split = prec == "bf16x3"
    if split:
        qkv, do = ops.split_pack(x.view(-1, 3 * D)).view(B, T, -1), ops.split_pack(d.view(-1, D)).view(B, T, -1)
        xr, dr = ops.split_unpack(qkv.view(-1, 6 * D)).view(B, T, 3 * D).double(), ops.split_unpack(do.view(-1, 2 * D)).view(B, T, D).double()
    else:
        qkv, do = x.half(), d.half()
        xr, dr = qkv.double(), do.double()
    o, lse = ops.attention_fwd(qkv, H, split=split)
    outs = [ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=split)[0].clone() for _ in range(4)]
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0], t) for t in outs[1:])
    xr.requires_grad_(True)
    q, k, v = xr.view(B, T, 3, H, 32).permute(2, 0, 3, 1, 4)
    a = torch.softmax(q @ k.transpose(-1, -2) / 32 ** 0.5, -1) @ v
    a.permute(0, 2, 1, 3).reshape(B, T, D).backward(dr)
    got = ops.split_unpack(outs[0].view(-1, 6 * D)).view(B, T, 3 * D) if split else outs[0]
    err = float((got.double() - xr.grad).abs().max() / xr.grad.abs().max())
    print(f"{prec}: 4 runs bit-identical: {same}   dqkv vs f64: {err:.2e}", flush=True)
