"""Row kernels at small M with and without K splits (MFVIT_ROWP_KSPLIT=0 / -1), isolated launches: fc2 + LN (K = 1536), fc1-dgrad + LayerNorm backward (K = 1536),
qkv-dgrad + LayerNorm backward (K = 1152); alone on the chip and with the half-chip hint of the two-stream model."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-feature-vit_amd")]
import torch
from mfvit import ops
from mfvit._lib import lib
dev = "cuda:0"
D, F = 384, 1536
g = torch.Generator(device=dev).manual_seed(1)
def rn(*s, sc=1.0): return torch.randn(*s, device=dev, generator=g) * sc
def sp(x): return ops.split_pack(x)
scratch = torch.empty(ops.ROWP_SCRATCH_FLOATS, device=dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for share in (1, 2):
    lib().mfvit_set_stream_share(share)
    for M in (1576, 3152, 6304, 12608):
        a, w2 = sp(rn(M, F)), sp(rn(D, F, sc=.05))
        res, gam, bet, b = rn(M, D), rn(D), rn(D), rn(D)
        dq, wq = sp(rn(M, 3 * D, sc=.1)), sp(rn(D, 3 * D, sc=.05))
        x = rn(M, D); mean = x.mean(1); rstd = 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
        line = f"share {share} M {M:6d}:"
        for name, fn in (("fc2+LN", lambda s: ops.linear_res_ln_fwd(a, w2, b, res, gam, bet, 1e-6, split=True, scratch=s)),
                         ("fc1-dgrad+LNbwd", lambda s: ops.linear_dgrad_ln_bwd(a, w2, x, mean, rstd, gam, res, split=True, scratch=s)),
                         ("qkv-dgrad+LNbwd", lambda s: ops.linear_dgrad_ln_bwd(dq, wq, x, mean, rstd, gam, res, split=True, scratch=s))):
            t0 = timeit(lambda: fn(None))
            t1 = timeit(lambda: fn(scratch))
            line += f"  {name} {t0:6.1f} -> {t1:6.1f} us"
        print(line, flush=True)
lib().mfvit_set_stream_share(1)
