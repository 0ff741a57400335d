"""Is a kernel clock-/power-bound?  Same launch on random and on all-zero operands (zero data: far less switching, higher clock), first
repetitions (cool) against sustained ones.  python3 tools/power_probe.py [op ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
M, D = 128 * 197, 384
def mk(zero):
    f = (lambda *s: torch.zeros(*s, device=dev)) if zero else (lambda *s: torch.randn(*s, device=dev))
    x, w = ops.split_pack(f(M, D)), ops.split_pack(f(3 * D, D) * .05)
    b = f(3 * D)
    x4, w2 = ops.split_pack(f(M, 4 * D)), ops.split_pack(f(D, 4 * D) * .05)
    res, g, be = f(M, D), f(D), f(D)
    dy = ops.split_pack(f(M, 3 * D))
    out = torch.zeros(3 * D, D, device=dev)
    qkv = ops.split_pack(f(128, 197, 3 * D))
    return {
        "qkv": lambda: ops.linear_fwd(x, w, b, split=True),
        "row_fc2": lambda: ops.linear_res_ln_fwd(x4, w2, f(D) if False else be, res, g, be, 1e-6, split=True),
        "wgrad_qkv": lambda: ops.linear_wgrad(dy, x, out=out, split=True),
        "attn_fwd": lambda: ops.attention_fwd(qkv, 12, split=True),
    }
def t(fn, n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
want = sys.argv[1:] or ["qkv", "row_fc2", "wgrad_qkv", "attn_fwd"]
for zero in (False, True):
    fns = mk(zero)
    for name in want:
        fn = fns[name]
        fn(); torch.cuda.synchronize()
        import time; time.sleep(0.5)
        cool = t(fn, 5)
        hot = [t(fn, 50) for _ in range(4)]
        print(f"{name:10s} {'zeros ' if zero else 'random'}  first 5: {cool:7.1f} us   sustained: " + " ".join(f"{h:7.1f}" for h in hot), flush=True)
