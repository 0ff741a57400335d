"""Attention backward A/B at the bench shape: the register-prefetch kernel (MFVIT_ATTN_BWD_PP=0) against the producer-wave kernel (1),
interleaved rounds in ONE process (MFVIT_AB_LIVE=1); dqkv against float64 on the first images, the two kernels against each other."""
import os, sys
os.environ["MFVIT_AB_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
B, T, H, D = int(os.environ.get("AB_B", 128)), int(os.environ.get("AB_T", 197)), 12, 384
precs = sys.argv[1:] or ["bf16x3", "bf16", "fp16"]


def timeit(fn, n=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for prec in precs:
    torch.manual_seed(5)
    x = torch.randn(B, T, 3 * D, device=dev)
    d = torch.randn(B, T, D, device=dev)
    split = prec == "bf16x3"
    if split:
        qkv, do = ops.split_pack(x.view(-1, 3 * D)).view(B, T, -1), ops.split_pack(d.view(-1, D)).view(B, T, -1)
        xr, dr = ops.split_unpack(qkv.view(-1, 6 * D)).view(B, T, 3 * D), ops.split_unpack(do.view(-1, 2 * D)).view(B, T, D)
    else:
        dt = torch.bfloat16 if prec == "bf16" else torch.float16
        qkv, do = x.to(dt), d.to(dt)
        xr, dr = qkv.float(), do.float()
    o, lse = ops.attention_fwd(qkv, H, split=split)
    res = {}
    for sw in ("0", "1"):
        os.environ["MFVIT_ATTN_BWD_PP"] = sw
        g, _ = ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=split)
        res[sw] = g.clone()
    torch.cuda.synchronize()
    nb = min(B, 6)
    xd = xr[:nb].double().requires_grad_(True)
    q, k, v = xd.view(nb, T, 3, H, 32).permute(2, 0, 3, 1, 4)
    a = torch.softmax(q @ k.transpose(-1, -2) / 32 ** 0.5, -1) @ v
    a.permute(0, 2, 1, 3).reshape(nb, T, D).backward(dr[:nb].double())
    unp = (lambda t: ops.split_unpack(t.reshape(-1, 6 * D)).view(-1, T, 3 * D)) if split else (lambda t: t.float())
    for sw in ("0", "1"):
        g = unp(res[sw])
        e = float((g[:nb].double() - xd.grad).abs().max() / xd.grad.abs().max())
        print(f"{prec} pp={sw}: dqkv vs f64 {e:.2e}  finite {bool(torch.isfinite(g).all())}", flush=True)
    g0, g1 = unp(res["0"]), unp(res["1"])
    print(f"{prec} pp vs register-prefetch kernel over all {B} images: max |diff| / max = {float((g0 - g1).abs().max() / g0.abs().max()):.2e}", flush=True)
    ts = {"0": [], "1": []}
    for rnd in range(5):
        for sw in ("0", "1"):
            os.environ["MFVIT_ATTN_BWD_PP"] = sw
            ts[sw].append(timeit(lambda: ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=split)))
    for sw in ("0", "1"):
        t = sorted(ts[sw])
        print(f"{prec} pp={sw}: median {t[len(t) // 2]:6.1f} us  min {t[0]:6.1f} us", flush=True)
