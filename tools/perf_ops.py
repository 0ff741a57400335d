"""Per-op timings at the bench shapes (M = 128 x 197 token rows), bf16.  Development aid.

CAUTION: every op is launched repeatedly on the SAME tensors, so its operands sit in the 256 MB Infinity Cache.  Variants that win
here can lose inside the training step, where operands come from HBM (it happened: LDS-DMA rings for the row kernels, the
persistent tile kernels).  Trust the per-class averages of bench.py's serialized pass (`roofline.serialized_pass.per_class`) or a
rotation over more than 256 MB of operands (tools/perf_tn_cold.py) before keeping a change."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch  # noqa: E402
from mfvit import ops  # noqa: E402

dev = torch.device("cuda:0")
M, D, F = 128 * 197, 384, 1536
bf = torch.bfloat16


def timeit(fn, flops, name, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / n
    print(f"{name:34s} {us:8.1f} us  {flops/us/1e6:7.1f} TFLOP/s", flush=True)


def r(*s, dt=bf, sc=1.0):
    return (torch.randn(*s, device=dev) * sc).to(dt)


x384, x1536 = r(M, D), r(M, F)
w_qkv, w_fc1, w_fc2, w_proj = r(3 * D, D, sc=.05), r(F, D, sc=.05), r(D, F, sc=.05), r(D, D, sc=.05)
b384, b1152, b1536 = r(D, dt=torch.float32), r(3 * D, dt=torch.float32), r(F, dt=torch.float32)
res = r(M, D, dt=torch.float32)
g, be = torch.ones(D, device=dev), torch.zeros(D, device=dev)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
tag = os.environ.get("MFVIT_ROW_VARIANT", "default")
timeit(lambda: ops.linear_res_ln_fwd(x384, w_proj, b384, res, g, be, 1e-6), 2.0 * M * D * D, f"row_fwd K=384  [v{tag}]")
timeit(lambda: ops.linear_res_ln_fwd(x1536, w_fc2, b384, res, g, be, 1e-6), 2.0 * M * D * F, f"row_fwd K=1536 [v{tag}]")
dy1152, dy1536 = r(M, 3 * D), r(M, F)
wt1152, wt1536 = r(D, 3 * D, sc=.05), r(D, F, sc=.05)
timeit(lambda: ops.linear_dgrad_ln_bwd(dy1152, wt1152, res, mean, rstd, g, res), 2.0 * M * D * 3 * D, f"row_bwd K=1152 [v{tag}]")
timeit(lambda: ops.linear_dgrad_ln_bwd(dy1536, wt1536, res, mean, rstd, g, res), 2.0 * M * D * F, f"row_bwd K=1536 [v{tag}]")
if tag in ("default", "2"):
    timeit(lambda: ops.linear_fwd(x384, w_qkv, b1152), 2.0 * M * D * 3 * D, "tile qkv (bias)")
    timeit(lambda: ops.linear_fwd(x384, w_fc1, b1536, gelu=True), 2.0 * M * D * F, "tile fc1 (bias+gelu)")
    out = torch.zeros(3 * D, D, device=dev)
    timeit(lambda: ops.linear_wgrad(dy1152, x384, out=out), 2.0 * M * D * 3 * D, "wgrad qkv  [1152x384]")
    out2 = torch.zeros(F, D, device=dev)
    timeit(lambda: ops.linear_wgrad(dy1536, x384, out=out2), 2.0 * M * D * F, "wgrad fc1  [1536x384]")
    out3 = torch.zeros(D, F, device=dev)
    timeit(lambda: ops.linear_wgrad(x384, x1536, out=out3), 2.0 * M * D * F, "wgrad fc2  [384x1536]")
    out4 = torch.zeros(D, D, device=dev)
    timeit(lambda: ops.linear_wgrad(x384, x384, out=out4), 2.0 * M * D * D, "wgrad proj [384x384]")
    scr = torch.empty(ops.WGRAD_SCRATCH_FLOATS, device=dev)
    timeit(lambda: ops.linear_wgrad(dy1152, x384, out=out, scratch=scr), 2.0 * M * D * 3 * D, "wgrad qkv  + scratch")
    timeit(lambda: ops.linear_wgrad(dy1536, x384, out=out2, scratch=scr), 2.0 * M * D * F, "wgrad fc1  + scratch")
    timeit(lambda: ops.linear_wgrad(x384, x1536, out=out3, scratch=scr), 2.0 * M * D * F, "wgrad fc2  + scratch")
    timeit(lambda: ops.linear_wgrad(x384, x384, out=out4, scratch=scr), 2.0 * M * D * D, "wgrad proj + scratch")
    qkv = r(128, 197, 3 * D)
    o, lse = ops.attention_fwd(qkv, 12)
    timeit(lambda: ops.attention_fwd(qkv, 12), 4.0 * 128 * 12 * 197 * 197 * 32, "attention fwd")
    do = r(128, 197, D)
    timeit(lambda: ops.attention_bwd(qkv, o, do, lse, 12, want_dbias=False), 8.0 * 128 * 12 * 197 * 197 * 32, "attention bwd")
