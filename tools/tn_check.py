"""Weight-gradient GEMM (csrc/gemm_tn2.hip): waves per workgroup (MFVIT_TN2_W8) x LDS-DMA issue interleaved with the MFMAs (MFVIT_TN2_IL),
every combination against f64 math on the rounded operands; timing in one process.   python3 tools/tn_check.py [quick]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
D, F = 384, 1536


MODES = [(0, 0), (0, 1), (1, 0), (1, 1)]           # (W8, IL); (0, 0) = the round-2 kernel


def setmode(mode):
    os.environ["MFVIT_TN2_W8"], os.environ["MFVIT_TN2_IL"] = str(mode[0]), str(mode[1])


def timeit(mode, fn, n=20):
    setmode(mode)
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


ok = True
for M, tag in [(128 * 197, "full"), (128 * 197 - 57, "ragged"), (4096 + 40, "small")][: 1 if quick else 3]:
    for name, n, k in (("qkv", 3 * D, D), ("fc1", F, D), ("fc2", D, F), ("proj", D, D)):
        for kind in ("split", "bf16"):
            a32, b32 = torch.randn(M, n, device=dev) * .1, torch.randn(M, k, device=dev)
            if kind == "split":
                a, b = ops.split_pack(a32), ops.split_pack(b32)
                ref = ops.split_unpack(a).double().T @ ops.split_unpack(b).double()
            else:
                a, b = a32.bfloat16(), b32.bfloat16()
                ref = a.double().T @ b.double()
            fn = lambda: ops.linear_wgrad(a, b, split=kind == "split")
            line = f"M={M:6d} {tag:6s} {name:5s} {kind:5s}"
            for mode in MODES:
                setmode(mode)
                out = fn()
                torch.cuda.synchronize()
                e = float((out.double() - ref).abs().max() / ref.abs().max())
                good = e < 2e-5
                ok &= good
                line += f" | w8={mode[0]} il={mode[1]}: {e:.1e}{'' if good else ' BAD'}"
            if tag == "full":
                out = torch.zeros(n, k, device=dev)
                fn2 = lambda: ops.linear_wgrad(a, b, out=out, split=kind == "split")
                fl = 2.0 * M * n * k
                line += " | us:" + "".join(f" [w8={m[0]} il={m[1]}] {timeit(m, fn2):6.1f}" for m in MODES)
            print(line, flush=True)
print("ALL OK" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
