"""Attention backward A/B at the bench shape: the two-phase register-prefetch kernel ("0") against
the single-pass kernel ("sp", MFVIT_ATTN_BWD_SP=1, the default), interleaved rounds in ONE process (MFVIT_AB_LIVE=1); dqkv against float64 on the first
images and the last one, the kernels against each other.   AB_KERNELS=0,sp python3 tools/attn_bwd_ab.py [bf16x3 bf16 fp16]"""
import os, sys
os.environ["MFVIT_AB_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
B, T, H, D = int(os.environ.get("AB_B", 128)), int(os.environ.get("AB_T", 197)), 12, 384
precs = sys.argv[1:] or ["bf16x3", "bf16", "fp16"]
KS = os.environ.get("AB_KERNELS", "0,sp").split(",")


def select(k):
    os.environ["MFVIT_ATTN_BWD_SP"] = "1" if k == "sp" else "0"



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
    for sw in KS:
        select(sw)
        g, _ = ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=split)
        res[sw] = g.clone()
    torch.cuda.synchronize()
    nb = min(B, 6)
    sel = list(range(nb - 1)) + [B - 1]                      # the first images and the last one (a workgroup's last pair)
    xd = xr[sel].double().requires_grad_(True)
    q, k, v = xd.view(nb, T, 3, H, 32).permute(2, 0, 3, 1, 4)
    a = torch.softmax(q @ k.transpose(-1, -2) / 32 ** 0.5, -1) @ v
    a.permute(0, 2, 1, 3).reshape(nb, T, D).backward(dr[sel].double())
    unp = (lambda t: ops.split_unpack(t.reshape(-1, 6 * D)).view(-1, T, 3 * D)) if split else (lambda t: t.float())
    for sw in KS:
        g = unp(res[sw])
        gs = g[sel].double()
        e = float((gs - xd.grad).abs().max() / xd.grad.abs().max())
        parts = [float((gs.view(nb, T, 3, D)[:, :, i] - xd.grad.view(nb, T, 3, D)[:, :, i]).abs().max() / xd.grad.view(nb, T, 3, D)[:, :, i].abs().max()) for i in range(3)]
        print(f"{prec} kernel {sw}: dqkv vs f64 {e:.2e} (dq {parts[0]:.1e} dk {parts[1]:.1e} dv {parts[2]:.1e})  finite {bool(torch.isfinite(g).all())}", flush=True)
    g0 = unp(res[KS[0]])
    for sw in KS[1:]:
        g1 = unp(res[sw])
        print(f"{prec} kernel {sw} vs {KS[0]} over all {B} images: max |diff| / max = {float((g0 - g1).abs().max() / g0.abs().max()):.2e}", flush=True)
    ts = {k: [] for k in KS}
    for rnd in range(5):
        for sw in KS:
            select(sw)
            ts[sw].append(timeit(lambda: ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=split)))
    for sw in KS:
        t = sorted(ts[sw])
        print(f"{prec} kernel {sw}: median {t[len(t) // 2]:6.1f} us  min {t[0]:6.1f} us", flush=True)
