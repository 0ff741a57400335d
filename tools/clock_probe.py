"""Shader clock and board power (sysfs hwmon of the card) sampled WHILE one kernel runs back to back for ~1.5 s: is the launch at the power cap, and at which
clock?  python3 tools/clock_probe.py [op ...]   (ops as tools/power_probe.py plus 'mfma_only': nothing but MFMAs, via tools/mfma_ceiling if built)"""
import glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")


def hwmon():
    out = []
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        f, p = os.path.join(h, "freq1_input"), None
        for cand in ("power1_average", "power1_input"):
            if os.path.exists(os.path.join(h, cand)):
                p = os.path.join(h, cand)
        cap = os.path.join(h, "power1_cap")
        if os.path.exists(f):
            out.append((f, p, cap if os.path.exists(cap) else None))
    return out


def read(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return -1


class Sampler(threading.Thread):
    def __init__(self, nodes):
        super().__init__(daemon=True)
        self.nodes, self.rows, self.stop = nodes, [], False

    def run(self):
        while not self.stop:
            self.rows.append([(read(f), read(p) if p else -1) for f, p, _ in self.nodes])
            time.sleep(0.02)


M, D = 128 * 197, 384
f = lambda *s: torch.randn(*s, device=dev)
z = lambda *s: torch.zeros(*s, device=dev)
def mk(g):
    x, w, b = ops.split_pack(g(M, D)), ops.split_pack(g(3 * D, D) * .05), g(3 * D)
    w1 = ops.split_pack(g(4 * D, D) * .05); b1 = g(4 * D)
    x4, w2 = ops.split_pack(g(M, 4 * D)), ops.split_pack(g(D, 4 * D) * .05)
    wp = ops.split_pack(g(D, D) * .05)
    w2t = ops.split_pack(g(4 * D, D) * .05)                         # W2^T: [1536][384] for the fc2 data gradient
    w1t = ops.split_pack(g(D, 4 * D) * .05)                         # W1^T: [384][1536] for the fc1 data gradient + LayerNorm backward
    dact = (g(M, 4 * D) * .5).half()
    res, gm, be = g(M, D), g(D), g(D)
    mean, rstd = g(M) * .1, g(M).abs() + .5
    dy = ops.split_pack(g(M, 3 * D))
    dy1 = ops.split_pack(g(M, D))
    out = torch.zeros(3 * D, D, device=dev)
    out1 = torch.zeros(4 * D, D, device=dev)
    qkv = ops.split_pack(g(128 * 197, 3 * D)).view(128, 197, -1)
    o, lse = ops.attention_fwd(qkv, 12, split=True)
    do = ops.split_pack(g(128 * 197, D)).view(128, 197, -1)
    G = 1e9
    return {"qkv": (lambda: ops.linear_fwd(x, w, b, split=True), 2 * M * D * 3 * D / G),
            "fc1_gelu": (lambda: ops.linear_fwd(x, w1, b1, gelu=True, split=True), 2 * M * D * 4 * D / G),
            "fc2_dgrad": (lambda: ops.linear_dgrad_act(dy1, w2t, dact, split=True), 2 * M * D * 4 * D / G),
            "proj_dgrad": (lambda: ops.linear_fwd(dy1, wp, None, split=True), 2 * M * D * D / G),
            "row_fc2": (lambda: ops.linear_res_ln_fwd(x4, w2, be, res, gm, be, 1e-6, split=True), 2 * M * D * 4 * D / G),
            "row_proj": (lambda: ops.linear_res_ln_fwd(x, wp, be, res, gm, be, 1e-6, split=True), 2 * M * D * D / G),
            "lnbwd_fc1": (lambda: ops.linear_dgrad_ln_bwd(x4, w1t, res, mean, rstd, gm, res, split=True), 2 * M * D * 4 * D / G),
            "wgrad_qkv": (lambda: ops.linear_wgrad(dy, x, out=out, split=True), 2 * M * D * 3 * D / G),
            "wgrad_fc1": (lambda: ops.linear_wgrad(x4, x, out=out1, split=True), 2 * M * D * 4 * D / G),
            "attn_fwd": (lambda: ops.attention_fwd(qkv, 12, split=True), 4 * 128 * 12 * 197 * 197 * 32 / G),
            "attn_bwd": (lambda: ops.attention_bwd(qkv, o, do, lse, 12, want_dbias=False, split=True), 8 * 128 * 12 * 197 * 197 * 32 / G)}
nodes = hwmon()
print("hwmon nodes:", [(n[0], read(n[2]) if n[2] else None) for n in nodes], flush=True)
want = sys.argv[1:] or ["qkv", "fc1_gelu", "fc2_dgrad", "proj_dgrad", "row_fc2", "row_proj", "lnbwd_fc1", "wgrad_qkv", "wgrad_fc1", "attn_fwd", "attn_bwd"]
for label, g in (("random", f), ("zeros ", z)):
    fns = mk(g)
    for name in want:
        fn, gflop = fns[name]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s = Sampler(nodes); s.start()
        t0 = time.time(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < 1.5:
            for _ in range(50):
                fn()
            n += 50
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        s.stop = True; s.join()
        us = e0.elapsed_time(e1) * 1e3 / n
        rows = s.rows[len(s.rows) // 3:]                       # sustained part
        best = max(range(len(nodes)), key=lambda i: sum(r[i][1] for r in rows)) if nodes else 0
        mhz = [r[best][0] / 1e6 for r in rows if r[best][0] > 0]
        pw = [r[best][1] / 1e6 for r in rows if r[best][1] > 0]
        pavg = sum(pw) / max(len(pw), 1)
        print(f"{name:10s} {label}: {us:7.1f} us per launch   sclk {sum(mhz) / max(len(mhz), 1):6.0f} MHz (min {min(mhz, default=0):.0f})   "
              f"power {pavg:5.0f} W (max {max(pw, default=0):.0f})   {gflop / us * 1e3:6.1f} algorithmic TFLOP/s   "
              f"{pavg * us * 1e-6 / (3 * gflop * 1e9) * 1e12:5.2f} pJ per MFMA flop   [{len(rows)} samples]", flush=True)
