"""Run ONE op of the hot path a few times (for rocprofv3 --pmc passes): python3 tools/one_op.py <op> [n]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
from mfvit import ops
dev = torch.device("cuda:0")
M, D = 128 * 197, 384
op = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"      # bf16 | bf16x3 | fp16
SPLIT = prec == "bf16x3"
def r(*s, dt=None, sc=1.0):
    x = torch.randn(*s, device=dev) * sc
    if dt is not None: return x.to(dt)
    if SPLIT: return ops.split_pack(x)
    return x.to(torch.float16 if prec == "fp16" else torch.bfloat16)
_lf, _wg, _rl, _db, _af, _ab = ops.linear_fwd, ops.linear_wgrad, ops.linear_res_ln_fwd, ops.linear_dgrad_ln_bwd, ops.attention_fwd, ops.attention_bwd
if SPLIT:
    import functools
    ops.linear_fwd, ops.linear_wgrad = functools.partial(_lf, split=True), functools.partial(_wg, split=True)
    ops.linear_res_ln_fwd, ops.linear_dgrad_ln_bwd = functools.partial(_rl, split=True), functools.partial(_db, split=True)
    ops.attention_fwd, ops.attention_bwd = functools.partial(_af, split=True), functools.partial(_ab, split=True)
    ops.linear_dgrad_act = functools.partial(ops.linear_dgrad_act, split=True)
if op == "qkv":
    x, w, b = r(M, D), r(3 * D, D, sc=.05), r(3 * D, dt=torch.float32)
    fn = (lambda: ops.linear_fwd(x, w, b, qkv_f16=True)) if SPLIT else (lambda: ops.linear_fwd(x, w, b))   # bf16x3: the encoder's split-fp16 epilogue
elif op == "fc2_dgrad":     # dX[M, 1536] = (dY[M, 384] W2) * gelu'(pre): the tile GEMM with the GELU-backward epilogue
    x, w, b = r(M, D), r(4 * D, D, sc=.05), r(4 * D, dt=torch.float32)
    dact, _ = ops.linear_fwd(x, w, b, gelu=True)
    dy, wt = r(M, D), r(4 * D, D, sc=.05)
    fn = lambda: ops.linear_dgrad_act(dy, wt, dact)
elif op == "fc1":
    x, w, b = r(M, D), r(4 * D, D, sc=.05), r(4 * D, dt=torch.float32)
    fn = lambda: ops.linear_fwd(x, w, b, gelu=True)
elif op == "dgrad_fc1":   # dY[M,1536] @ W[384,1536]^T... plain tile GEMM N=384? (runs on the row kernel in the encoder)
    x, w = r(M, 4 * D), r(D, 4 * D, sc=.05)
    fn = lambda: ops.linear_fwd(x, w, None)
elif op == "wgrad_fc1":
    dy, x = r(M, 4 * D), r(M, D)
    out = torch.zeros(4 * D, D, device=dev)
    fn = lambda: ops.linear_wgrad(dy, x, out=out)
elif op == "wgrad_qkv":
    dy, x = r(M, 3 * D), r(M, D)
    out = torch.zeros(3 * D, D, device=dev)
    fn = lambda: ops.linear_wgrad(dy, x, out=out)
elif op in ("row_proj", "row_fc2"):
    K = D if op == "row_proj" else 4 * D
    a, w, b = r(M, K), r(D, K, sc=.05), r(D, dt=torch.float32)
    res, g, be = r(M, D, dt=torch.float32), r(D, dt=torch.float32), r(D, dt=torch.float32)
    fn = lambda: ops.linear_res_ln_fwd(a, w, b, res, g, be, 1e-6)
elif op in ("rowb_proj", "rowb_fc1"):   # dgrad + LN backward: dy [M,K] @ wt[384,K]^T
    K = 3 * D if op == "rowb_proj" else 4 * D
    dy, wt = r(M, K), r(D, K, sc=.05)
    x = r(M, D, dt=torch.float32); mean = x.mean(1); rstd = 1 / x.std(1)
    g, dres = r(D, dt=torch.float32), r(M, D, dt=torch.float32)
    fn = lambda: ops.linear_dgrad_ln_bwd(dy, wt, x, mean, rstd, g, dres)
elif op in ("attn_fwd", "attn_bwd"):
    qkv = r(128, 197, 3 * D)
    if SPLIT and not os.environ.get("ONEOP_QKV_BF16"):   # what the encoder hands the attention core in bf16x3 mode: split FP16 (MFVIT_X3F16)
        qkv = ops.split_pack_f16(torch.randn(128, 197, 3 * D)).to(dev)
    o, lse = ops.attention_fwd(qkv, 12)
    do = r(128, 197, D)
    fn = (lambda: ops.attention_fwd(qkv, 12)) if op == "attn_fwd" else (lambda: ops.attention_bwd(qkv, o, do, lse, 12, want_dbias=False))
else:
    raise SystemExit("unknown op")
for _ in range(n):
    fn()
torch.cuda.synchronize()
