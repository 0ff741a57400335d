"""Attention forward A/B at the bench shape (B = 128, T = 197, 12 heads x 32): the per-pair kernel (MFVIT_ATTN_FWD_RING=0) against the persistent
ring kernel (1), interleaved rounds in ONE process (MFVIT_AB_LIVE=1 makes the library re-read the switch at every launch); results compared
with each other and with float64 on the first images."""
import os, sys
os.environ["MFVIT_AB_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
B, T, H, D = int(os.environ.get("AB_B", 128)), int(os.environ.get("AB_T", 197)), 12, 384


def timeit(fn, n=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for prec in ("bf16x3", "bf16", "fp16"):
    torch.manual_seed(5)
    x = torch.randn(B, T, 3 * D, device=dev)
    split = prec == "bf16x3"
    if split:
        qkv = ops.split_pack(x.view(-1, 3 * D)).view(B, T, -1)
        xr = ops.split_unpack(qkv.view(-1, 6 * D)).view(B, T, 3 * D)
    else:
        qkv = x.to(torch.bfloat16 if prec == "bf16" else torch.float16)
        xr = qkv.float()
    res = {}
    for ring in ("0", "2"):
        os.environ["MFVIT_ATTN_FWD_RING"] = ring
        o, lse = ops.attention_fwd(qkv, H, split=split)
        res[ring] = (o.clone(), lse.clone())
    torch.cuda.synchronize()
    nb = min(B, 8)
    q, k, v = xr[:nb].double().view(nb, T, 3, H, 32).permute(2, 0, 3, 1, 4)
    a = (q @ k.transpose(-1, -2)) / 32 ** 0.5
    oref = (torch.softmax(a, -1) @ v).permute(0, 2, 1, 3).reshape(nb, T, D)
    lref = torch.logsumexp(a, -1)
    for ring in ("0", "2"):
        o, lse = res[ring]
        of = ops.split_unpack(o.view(-1, 2 * D)).view(B, T, D) if split else o.float()
        eo = float((of[:nb].double() - oref).abs().max() / oref.abs().max())
        el = float((lse[:nb].double() - lref).abs().max() / lref.abs().max())
        print(f"{prec} ring={ring}: out vs f64 {eo:.2e}  lse {el:.2e}  finite {bool(torch.isfinite(of).all())}", flush=True)
    o0 = ops.split_unpack(res["0"][0].view(-1, 2 * D)) if split else res["0"][0].float()
    o1 = ops.split_unpack(res["2"][0].view(-1, 2 * D)) if split else res["2"][0].float()
    print(f"{prec} ring vs per-pair over all {B} images: max |diff| / max |out| = {float((o0 - o1).abs().max() / o0.abs().max()):.2e}", flush=True)
    ts = {"0": [], "2": []}
    for rnd in range(5):
        for ring in ("0", "2"):
            os.environ["MFVIT_ATTN_FWD_RING"] = ring
            ts[ring].append(timeit(lambda: ops.attention_fwd(qkv, H, split=split)))
    for ring in ("0", "2"):
        t = sorted(ts[ring])
        print(f"{prec} ring={ring}: median {t[len(t) // 2]:6.1f} us  min {t[0]:6.1f} us", flush=True)
