"""Tile GEMM with loads / LDS stores interleaved between the MFMAs (MFVIT_NT_IL=1, gemm.cuh NtLoopDeep<..., IL>) against the default burst form: bit-exactness and
timing in ONE process.   python3 tools/tile_il_check.py [quick]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops

dev = torch.device("cuda:0")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def r(*s, sc=1.0):
    return ops.split_pack(torch.randn(*s, device=dev) * sc)


def run(mode, fn):
    os.environ["MFVIT_NT_IL"] = str(mode)
    out = fn()
    torch.cuda.synchronize()
    return out


def timeit(mode, fn, n=20):
    os.environ["MFVIT_NT_IL"] = str(mode)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


def same(a, b):
    if isinstance(a, (tuple, list)):
        return all(same(x, y) for x, y in zip(a, b))
    if a is None or b is None:
        return a is None and b is None
    return torch.equal(a, b)


def maxdiff(a, b):
    """largest difference of the VALUES (split tensors: hi + lo), relative to the largest value"""
    if isinstance(a, (tuple, list)):
        return max(maxdiff(x, y) for x, y in zip(a, b))
    if a is None:
        return 0.0
    if a.dtype == torch.bfloat16:
        a, b = ops.split_unpack(a), ops.split_unpack(b)
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


ok = True
shapes = [(128 * 197, "full"), (128 * 197 - 57, "ragged"), (4096 + 128, "small")]
if quick:
    shapes = shapes[:1]
for M, tag in shapes:
    D = 384
    x = r(M, D)
    cases = []
    wq, bq = r(3 * D, D, sc=.05), torch.randn(3 * D, device=dev)
    cases.append(("qkv   N=1152 bias", lambda: ops.linear_fwd(x, wq, bq, split=True), 2.0 * M * 3 * D * D))
    w1, b1 = r(4 * D, D, sc=.05), torch.randn(4 * D, device=dev)
    cases.append(("fc1   N=1536 gelu+grad", lambda: ops.linear_fwd(x, w1, b1, gelu=True, split=True), 2.0 * M * 4 * D * D))
    cases.append(("fc1   N=1536 gelu nograd", lambda: ops.linear_fwd(x, w1, b1, gelu=True, split=True, want_grad=False), 2.0 * M * 4 * D * D))
    wp = r(D, D, sc=.05)
    cases.append(("projd N=384 none", lambda: ops.linear_fwd(x, wp, None, split=True), 2.0 * M * D * D))
    w2t = r(4 * D, D, sc=.05)
    ag = (torch.rand(M, 4 * D, device=dev) * 1.2 - 0.1).to(torch.float16)
    cases.append(("fc2d  N=1536 gelu_bwd", lambda: ops.linear_dgrad_act(x, w2t, ag, split=True), 2.0 * M * 4 * D * D))
    for name, fn, flops in cases:
        ref = run(0, fn)
        line = f"M={M:6d} {tag:6s} {name:26s}"
        for mode in (1,):
            try:
                out = run(mode, fn)
                eq = same(ref, out)
                d = 0.0 if eq else maxdiff(ref, out)
                ok &= d < 2e-6          # with a bias: acc = bias + sum instead of sum + bias (one f32 rounding, then the hi / lo split)
                line += f" | il{mode}: {'bit-exact' if eq else 'rel %.1e' % d}"
            except Exception as ex:  # noqa: BLE001
                ok = False
                line += f" | il{mode}: ERROR {ex}"
        if tag == "full":
            t0 = timeit(0, fn)
            line += f" | tile {t0:7.1f} us ({flops / t0 / 1e6:5.0f} TF)"
            for mode in (1,):
                t = timeit(mode, fn)
                line += f" il{mode} {t:7.1f} us ({flops / t / 1e6:5.0f} TF)"
        print(line, flush=True)
print("ALL OK (bit-exact without bias, <= 2e-6 relative with)" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
