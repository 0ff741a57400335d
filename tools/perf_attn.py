import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
qkv = (torch.randn(128, 197, 1152, device=dev)).bfloat16()
do = torch.randn(128, 197, 384, device=dev).bfloat16()
o, lse = ops.attention_fwd(qkv, 12)
for _ in range(5):
    o, lse = ops.attention_fwd(qkv, 12)
    ops.attention_bwd(qkv, o, do, lse, 12, want_dbias=False)
torch.cuda.synchronize()
