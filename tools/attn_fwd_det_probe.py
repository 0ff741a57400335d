"""usage (GPU box): python tools/attn_fwd_det_probe.py [B]: where the persistent split-fp16 forward attention kernel differs from run to run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
T, D, H = 197, 384, 12
dev = "cuda:0"
g = torch.Generator(device=dev)
g.manual_seed(5)
qkv = ops.split_pack_f16(torch.randn(B, T, 3 * D, device=dev, generator=g))
runs = []
for it in range(6):
    o, l = ops.attention_fwd(qkv, H, split=True)
    torch.cuda.synchronize()
    runs.append((ops.split_unpack(o).double().view(B, T, H, D // H), l.clone()))
ref_o, ref_l = runs[0]
for it in range(1, 6):
    o, l = runs[it]
    dl = (l != ref_l)                                   # [B, H, T]
    do = ((o - ref_o).abs() > 0).any(-1).permute(0, 2, 1)   # [B, H, T]
    both = dl | do
    by_tile = [int(both[:, :, 32 * t:32 * t + 32].sum()) for t in range(7)]
    by_head = [int(both[:, h].sum()) for h in range(H)]
    by_img = both.sum((1, 2))
    print(f"run {it} vs 0: rows with a different lse {int(dl.sum())}, with a different out {int(do.sum())}, either {int(both.sum())} of {both.numel()}; by row tile {by_tile}; "
          f"by head {by_head}; images touched {int((by_img > 0).sum())} of {B}; max rows in one (image, head) {int(both.sum(-1).max())}", flush=True)
    prev_o, prev_l = runs[it - 1]
    print(f"   vs the run before: lse rows {int((l != prev_l).sum())}", flush=True)
# a few examples
o, l = runs[1]
idx = torch.nonzero(l != ref_l)[:10]
for b, h, t in idx.tolist():
    print(f"   (image {b}, head {h}, row {t}): lse {float(ref_l[b, h, t])!r} vs {float(l[b, h, t])!r}; out diffs {int(((o[b, t, h] - ref_o[b, t, h]).abs() > 0).sum())} of 32", flush=True)
