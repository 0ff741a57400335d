"""Phase breakdown of the persistent GEMM (needs a -DMFVIT_PERS_TRACE build selected with MFVIT_LIB): wave 0 of every workgroup
accumulates s_memtime deltas per phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit._lib import check, lib, ptr, stream
dev = torch.device("cuda:0")
M, K = 128 * 197, 384
names = ["front MFMAs", "vmcnt wait", "lgkm wait", "barrier", "issue+frag prefetch", "last MFMAs", "epilogue", "refill after epilogue"]
for N in (1152, 1536, 384):
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * .05).bfloat16(); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tr = torch.zeros(512 * 8, device=dev)
    for _ in range(3):
        check(lib().mfvit_linear_fwd_persistent(0, ptr(x), K, ptr(w), K, ptr(b), ptr(y), N, ptr(tr), N, M, N, K, stream()), "x")
    torch.cuda.synchronize()
    t = tr.view(512, 8).cpu()
    used = t[t.sum(1) > 0]
    tot = used.sum(1)
    print(f"N={N}: {used.shape[0]} workgroups, cycles per workgroup mean {tot.mean():.0f} max {tot.max():.0f} (s_memtime ticks)")
    for i, n in enumerate(names):
        print(f"   {n:24s} {used[:, i].mean():10.0f}  {100 * used[:, i].mean() / tot.mean():5.1f} %")
