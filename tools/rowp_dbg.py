import os, sys
os.environ["MFVIT_AB_LIVE"] = "1"; os.environ["MFVIT_ROWP_MINM"] = "1"; os.environ["MFVIT_ROWP"] = "2"
sys.path.insert(0, "multi-feature-vit_amd")
import torch
from mfvit import ops
dev = "cuda:0"
torch.manual_seed(0)
D = 384
for M, K in ((9456, 384), (3152, 384)):
    rpt = -(-M // 256)
    k0 = (torch.arange(M, device=dev) * 7) % K
    a32 = torch.zeros(M, K, device=dev)
    a32[torch.arange(M, device=dev), k0] = 1.0
    w32 = torch.randn(D, K, device=dev).bfloat16().float()          # exact in hi
    a, w = ops.split_pack(a32), ops.split_pack(w32)
    b, res = torch.zeros(D, device=dev), torch.zeros(M, D, device=dev)
    g, be = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    x, y, mean, rstd = ops.linear_res_ln_fwd(a, w, b, res, g, be, 1e-6, y_f32=True, split=True)
    ref = w32.T[k0]                                                   # [M][D]
    bad = ((x - ref).abs() > 1e-3).any(1).nonzero().flatten()
    print(M, "rpt", rpt, "bad rows", len(bad))
    for r in bad[:24].tolist():
        # which columns of w explain x[r]?  least squares over k is heavy: test single k'
        d = (w32.T - x[r][None, :]).abs().max(1).values              # [K]
        kk = int(d.argmin())
        nz = float(x[r].abs().max())
        print("  row", r, "in tile", r % rpt, "k0", int(k0[r]), "stage", int(k0[r]) // 32, "-> best single k'", kk, "stage", kk // 32, "resid", float(d[kk]), "|x|max", nz)
