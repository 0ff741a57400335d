"""Tile GEMM, N-wide 128 x 256 tile of 8 waves (MFVIT_NT_WIDE=1, one workgroup per CU) against the default 128 x 128 tile of 4 waves (two per
CU) at the bench shape (M = 25,216 rows): fc1 + GELU (N = 1536, K = 384) and the fc2 data gradient with the GELU backward (N = 1536, K = 384),
split bf16.  Interleaved rounds in one process (MFVIT_AB_LIVE=1), results compared bit for bit (same MFMA order per output element)."""
import os, sys
os.environ["MFVIT_AB_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
M, D, F = int(os.environ.get("AB_M", 25216)), 384, 1536


def timeit(fn, n=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


torch.manual_seed(3)
x = ops.split_pack(torch.randn(M, D)).to(dev)
w1 = ops.split_pack(torch.randn(F, D) * 0.05).to(dev)
b1 = torch.randn(F, device=dev)
dy = ops.split_pack(torch.randn(M, D)).to(dev)
w2t = ops.split_pack(torch.randn(F, D) * 0.05).to(dev)          # dX[M, F] = dY[M, D] W2 : the NT form takes W2^T [F][D]
cases = {"fc1 + GELU": lambda: ops.linear_fwd(x, w1, b1, gelu=True, split=True)}
os.environ["MFVIT_NT_WIDE"] = "0"
dact, act = ops.linear_fwd(x, w1, b1, gelu=True, split=True)
cases["fc2 dgrad x gelu'"] = lambda: ops.linear_dgrad_act(dy, w2t, dact, split=True)
res = {}
for sw in ("0", "1"):
    os.environ["MFVIT_NT_WIDE"] = sw
    res[sw] = {k: f() for k, f in cases.items()}
torch.cuda.synchronize()
for k in cases:
    a, b = res["0"][k], res["1"][k]
    a = a if isinstance(a, tuple) else (a,)
    b = b if isinstance(b, tuple) else (b,)
    print(f"{k}: wide tile bit-identical to the default: {all(torch.equal(u, v) for u, v in zip(a, b))}", flush=True)
ts = {(k, sw): [] for k in cases for sw in ("0", "1")}
for rnd in range(5):
    for k, f in cases.items():
        for sw in ("0", "1"):
            os.environ["MFVIT_NT_WIDE"] = sw
            ts[(k, sw)].append(timeit(f))
for k in cases:
    for sw in ("0", "1"):
        t = sorted(ts[(k, sw)])
        print(f"{k:20s} {'128 x 256, 8 waves, 1 per CU' if sw == '1' else '128 x 128, 4 waves, 2 per CU':30s} median {t[2]:6.1f} us  min {t[0]:6.1f} us", flush=True)
