import os, sys
os.environ["MFVIT_AB_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
B, T, H, D = 44, 197, 12, 384
torch.manual_seed(5)
x = torch.randn(B, T, 3 * D, device=dev)
d = torch.randn(B, T, D, device=dev)
qkv, do = ops.split_pack(x.view(-1, 3 * D)).view(B, T, -1), ops.split_pack(d.view(-1, D)).view(B, T, -1)
o, lse = ops.attention_fwd(qkv, H, split=True)
res = {}
for sw in ("0", "1"):
    os.environ["MFVIT_ATTN_BWD_PP"] = sw
    g, _ = ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=True)
    res[sw] = ops.split_unpack(g.reshape(-1, 6 * D)).view(B, T, 3, H, 32).clone()
torch.cuda.synchronize()
g0, g1 = res["0"], res["1"]
for i, name in enumerate("qkv"):
    a, b = g0[:, :, i], g1[:, :, i]
    bad = ~torch.isfinite(b)
    print(f"d{name}: nonfinite {int(bad.sum())} of {b.numel()}; tokens with nonfinite: {sorted(set(bad.any(-1).any(-1).nonzero()[:, 1].tolist()))[:40]}")
    ok = ~bad
    err = ((a - b).abs() * ok).amax(dim=(0, 2, 3)) / a.abs().max()
    print(f"   per-token max err (first 12): {[f'{v:.1e}' for v in err[:12].tolist()]} ... last 8: {[f'{v:.1e}' for v in err[-8:].tolist()]}")
    print(f"   worst tokens: {err.topk(8).indices.tolist()} {[f'{v:.1e}' for v in err.topk(8).values.tolist()]}")
    eb = ((a - b).abs() * ok).amax(dim=(1, 3)) / a.abs().max()
    print(f"   per (image, head) err: max {float(eb.max()):.2e}; pairs > 1e-3: {int((eb > 1e-3).sum())} of {eb.numel()}; first bad: {(eb > 1e-3).nonzero()[:6].tolist()}")
