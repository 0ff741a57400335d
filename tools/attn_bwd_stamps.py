"""Where a pair of the producer-wave attention backward spends its cycles (diagnostic library of tools/build_attn_stamp_lib.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MFVIT_LIB"] = os.path.join(ROOT, "multi-feature-vit_amd", "build", "libmfvit_attnstamp.so")
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import ctypes
import torch
from mfvit import ops, _lib
dev = torch.device("cuda:0")
B, T, H, D = 128, 197, 12, 384
x = torch.randn(B, T, 3 * D, device=dev)
d = torch.randn(B, T, D, device=dev)
qkv, do = ops.split_pack(x.view(-1, 3 * D)).view(B, T, -1), ops.split_pack(d.view(-1, D)).view(B, T, -1)
o, lse = ops.attention_fwd(qkv, H, split=True)
for _ in range(3):
    ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=True)
buf = torch.zeros(2 * 16 * 8 * 32, dtype=torch.int64, device=dev)
f = _lib.lib().mfvit_debug_attn_stamps
f.argtypes = [ctypes.c_void_p]
torch.cuda.synchronize()
assert f(buf.data_ptr()) == 0
ops.attention_bwd(qkv, o, do, lse, H, want_dbias=False, split=True)
torch.cuda.synchronize()
assert f(None) == 0
s = buf.cpu().view(2, 16, 8, 32)
names = {0: "X wait", 1: "phase A / prod A", 3: "Z wait", 4: "phase B / prod B"}
order = [10, 0, 1, 3, 4]
for blk in range(2):
    t0 = int(s[blk, 0, :, 10][s[blk, 0, :, 10] > 0].min())
    print(f"--- workgroup {'0' if blk == 0 else '37'}")
    for r in range(16):
        if int(s[blk, r, :, 10].max()) == 0:
            break
        for w in range(8):
            row = s[blk, r, w]
            pts = [(i, int(row[i])) for i in order if int(row[i]) != 0]
            txt = f"pair {r} wave {w}: start {pts[0][1] - t0:7d} |"
            for (i0, v0), (i1, v1) in zip(pts[:-1], pts[1:]):
                txt += f" {names[i1]} {v1 - v0:6d}"
            print(txt + f" | total {pts[-1][1] - pts[0][1]:7d}")
