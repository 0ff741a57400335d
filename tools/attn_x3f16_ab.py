"""Attention core at the bench shape (B = 128, T = 197, 12 heads x 32), split bf16 qkv (round 4) against split FP16 qkv (MFVIT_X3F16, round 5)
with P / dS in one or two fp16 parts: error of out / dqkv against float64 on the first images, and time per launch (interleaved rounds in
ONE process; MFVIT_AB_LIVE=1 makes the library re-read the switches at every launch)."""
import os, sys
os.environ["MFVIT_AB_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
B, T, H, D = int(os.environ.get("AB_B", 128)), int(os.environ.get("AB_T", 197)), 12, 384


def timeit(fn, n=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


torch.manual_seed(5)
x = torch.randn(B, T, 3 * D)
d = torch.randn(B, T, D) * float(os.environ.get("AB_DSCALE", "1e-3"))
q_b = ops.split_pack(x.view(-1, 3 * D)).view(B, T, -1).to(dev)
q_h = ops.split_pack_f16(x.view(-1, 3 * D)).view(B, T, -1).to(dev)
do = ops.split_pack(d.view(-1, D)).view(B, T, -1).to(dev)
nb = min(B, 6)


def ref(xr):
    xd = xr[:nb].double().requires_grad_(True)
    q, k, v = xd.view(nb, T, 3, H, 32).permute(2, 0, 3, 1, 4)
    a = (q @ k.transpose(-1, -2)) / 32 ** 0.5
    o = (torch.softmax(a, -1) @ v).permute(0, 2, 1, 3).reshape(nb, T, D)
    o.backward(ops.split_unpack(do.cpu().view(-1, 2 * D)).view(B, T, D)[:nb].double())
    return o.detach(), xd.grad


def f16_rounded(x):
    hi = x.half()
    return hi.double() + (x - hi.float()).half().double()


refs = {"bf16": ref(ops.split_unpack(q_b.cpu().view(-1, 6 * D)).view(B, T, 3 * D)), "f16": ref(f16_rounded(x))}
# (MFVIT_ATTN_PF, the one-part forward, existed until the end of round 5: 2.2e-4 on the output for 46 instead of 48 us - profiles/r05_attention_x3f16_ab.txt)
variants = [("split bf16 qkv (round 4)", q_b, "bf16", None, None), ("split fp16, P 2 parts / dS 2 parts", q_h, "f16", "2", "2"),
            ("split fp16, P 2 parts / dS 1 part", q_h, "f16", "2", "1")]
outs = {}
for name, qkv, rk, pf, pb in variants:
    if pf:
        os.environ["MFVIT_ATTN_PF"], os.environ["MFVIT_ATTN_PB"] = pf, pb
    o, lse = ops.attention_fwd(qkv, H, split=True)
    dq, _ = ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=True)
    torch.cuda.synchronize()
    outs[name] = (o, lse)
    of = ops.split_unpack(o.cpu().view(-1, 2 * D)).view(B, T, D)[:nb].double()
    dqf = ops.split_unpack(dq.cpu().view(-1, 6 * D)).view(B, T, 3 * D)[:nb].double()
    oref, gref = refs[rk]
    eo = float((of - oref).abs().max() / oref.abs().max())
    g, r = dqf.view(nb, T, 3, D), gref.view(nb, T, 3, D)
    es = [float((g[:, :, i] - r[:, :, i]).abs().max() / r[:, :, i].abs().max()) for i in range(3)]
    print(f"{name:38s} out {eo:.2e}  dq {es[0]:.2e} dk {es[1]:.2e} dv {es[2]:.2e}  finite {bool(torch.isfinite(dqf).all())}", flush=True)
tf = {n: [] for n, *_ in variants}
tb = {n: [] for n, *_ in variants}
for rnd in range(5):
    for name, qkv, rk, pf, pb in variants:
        if pf:
            os.environ["MFVIT_ATTN_PF"], os.environ["MFVIT_ATTN_PB"] = pf, pb
        o, lse = outs[name]
        tf[name].append(timeit(lambda: ops.attention_fwd(qkv, H, split=True)))
        tb[name].append(timeit(lambda: ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=True)))
for name, *_ in variants:
    a, b = sorted(tf[name]), sorted(tb[name])
    print(f"{name:38s} fwd median {a[2]:6.1f} us (min {a[0]:6.1f})   bwd median {b[2]:6.1f} us (min {b[0]:6.1f})", flush=True)
