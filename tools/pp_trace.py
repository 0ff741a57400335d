"""Phase shares of the ping-pong GEMM (trace build: MFVIT_DEFS=-DMFVIT_PP_TRACE python3 __graft_entry__.py --force):
   python3 tools/pp_trace.py [qkv|fc1|projd] [2|4]"""
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
buf = torch.zeros(256, 8, 8, device=dev)
os.environ["MFVIT_PP"] = mode
os.environ["MFVIT_PP_TRACE_PTR"] = hex(buf.data_ptr())
fn = (lambda: ops.linear_fwd(x, w, b, gelu=True, split=True)) if op == "fc1" else (lambda: ops.linear_fwd(x, w, b if op == "qkv" else None, split=True))
for _ in range(5):
    fn()
torch.cuda.synchronize()
t = buf.cpu()
names = ["c:prod1+2+waits", "c:barrier", "c:frags+prod3", "a:dma issue", "a:epilogue", "a:vmcnt wait", "a:barrier", "life"]
for g, nm in ((0, "group 0 (waves 0-3)"), (1, "group 1 (waves 4-7)")):
    v = t[:, 4 * g:4 * g + 4, :].reshape(-1, 8)
    life = v[:, 7].mean()
    print(f"{op} pp{mode} {nm}: life {life:9.0f} cycles")
    for i in range(7):
        print(f"    {names[i]:18s} {v[:, i].mean():9.0f}  {100 * v[:, i].mean() / life:5.1f} %   (min {v[:, i].min():8.0f} max {v[:, i].max():8.0f})")
