"""usage (GPU box): python tools/determinism_probe.py [B] [depth]: which 2-D weight gradients of the split-bf16 encoder are the same bits from run to run at batch B,
under the run-time switches that change which kernels run (MFVIT_AB_LIVE=1: read at every launch)."""
import os, sys
os.environ["MFVIT_AB_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import rng_tensor
from oracle import ref_vit
import vits

B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
m = vits.vit_small(num_classes=0, depth=depth, precision="bf16x3")
m.load_state_dict(ref_vit.seeded_params(621, num_classes=0, depth=depth), strict=False)
m = m.to("cuda:0")
x = rng_tensor(622, (B, 3, 224, 224)).to("cuda:0")
w = rng_tensor(623, (B, 197, 384)).to("cuda:0")


def run():
    m.zero_grad(set_to_none=True)
    f = m.features3D(x)
    (f * w).sum().backward()
    torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    g["features"] = f.detach().clone()
    return g


for env in ({}, {"MFVIT_PP": "0"}, {"MFVIT_ATTN_BWD_SP": "0"}, {"MFVIT_PP": "0", "MFVIT_ATTN_BWD_SP": "0"}, {"MFVIT_TN_PART": "0"}):
    for k in ("MFVIT_PP", "MFVIT_ATTN_BWD_SP", "MFVIT_TN_PART"):
        os.environ.pop(k, None)
    os.environ.update(env)
    run()
    a, b = run(), run()
    diff = [n for n in a if not torch.equal(a[n], b[n])]
    two_d = [n for n in diff if a[n].ndim >= 2 and n not in ("cls_token", "features")]
    print(f"B={B} depth={depth} {env or 'defaults'}: {len(a) - len(diff)} of {len(a)} tensors bit-identical; 2-D weight gradients that differ: {two_d or 'none'}; "
          f"others: {[n for n in diff if n not in two_d]}", flush=True)
