"""Timing-only ablation of the ping-pong GEMM (trace library, see tools/build_trace_lib.sh): python3 tools/pp_ablate.py [qkv|fc1] [2|4]
   MFVIT_PP_ABLATE bits: 1 no LDS-DMA, 2 no MFMAs, 4 no global stores, 8 no epilogue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
op = sys.argv[1] if len(sys.argv) > 1 else "qkv"
mode = sys.argv[2] if len(sys.argv) > 2 else "4"
M, D = 128 * 197, 384
N = {"qkv": 3 * D, "fc1": 4 * D, "projd": D}[op]
x = ops.split_pack(torch.randn(M, D, device=dev))
w = ops.split_pack(torch.randn(N, D, device=dev) * .05)
b = torch.randn(N, device=dev)
os.environ["MFVIT_PP"] = mode
fn = (lambda: ops.linear_fwd(x, w, b, gelu=True, split=True)) if op == "fc1" else (lambda: ops.linear_fwd(x, w, b if op == "qkv" else None, split=True))
def t(bits, n=20):
    os.environ["MFVIT_PP_ABLATE"] = str(bits)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
for bits, what in ((0, "everything"), (4, "no global stores"), (8, "no epilogue"), (1, "no LDS-DMA"), (2, "no MFMAs"), (3, "no DMA, no MFMA"), (9, "no DMA, no epilogue"),
                   (10, "no MFMA, no epilogue"), (11, "skeleton: barriers + LDS reads only")):
    print(f"{op} pp{mode} bits {bits:2d} {what:36s} {t(bits):7.1f} us", flush=True)
