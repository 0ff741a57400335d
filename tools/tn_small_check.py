import sys, os
sys.path.insert(0, "multi-feature-vit_amd")
import torch
from mfvit import ops
dev = "cuda:0"
for (M, N, K) in [(128, 4096, 4096), (128, 4096, 384), (128, 256, 4096), (128, 512, 512), (128, 1024, 2048)]:
    g = torch.Generator().manual_seed(1)
    dy, x = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
    for sw in ("1", "0"):
        os.environ["MFVIT_GEMM_SMALL"] = sw
        dw = ops.linear_wgrad(dy.to(dev), x.to(dev))
        ref = dy.double().t() @ x.double()
        e = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
        print((M, N, K), "small" if sw == "1" else "tile ", f"{e:.2e}", flush=True)
