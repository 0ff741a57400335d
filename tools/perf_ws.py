import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
M = 128 * 197
def r(*s, dt=torch.bfloat16, sc=1.0): return (torch.randn(*s, device=dev) * sc).to(dt)
x, w, b = r(M, 384), r(1152, 384, sc=.05), r(1152, dt=torch.float32)
w2, b2 = r(1536, 384, sc=.05), r(1536, dt=torch.float32)
for _ in range(25): ops.linear_fwd(x, w, b, persistent="ws")
for _ in range(25): ops.linear_fwd(x, w2, b2, gelu=True, persistent="ws")
torch.cuda.synchronize()
