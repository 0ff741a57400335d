"""Where a round of the persistent attention forward spends its cycles: s_memtime stamps of workgroups 0 and 37 (diagnostic library built by
tools/build_attn_stamp_lib.sh, loaded through MFVIT_LIB).  Prints per round and wave the cycles between consecutive stamp points."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MFVIT_LIB"] = os.path.join(ROOT, "multi-feature-vit_amd", "build", "libmfvit_attnstamp.so")
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import ctypes
import torch
from mfvit import ops, _lib
dev = torch.device("cuda:0")
B, T, H, D = 128, 197, 12, 384
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
x = torch.randn(B, T, 3 * D, device=dev)
split = prec == "bf16x3"
qkv = ops.split_pack(x.view(-1, 3 * D)).view(B, T, -1) if split else x.to(torch.bfloat16 if prec == "bf16" else torch.float16)
for _ in range(5):
    ops.attention_fwd(qkv, H, split=split)
buf = torch.zeros(2 * 16 * 8 * 32, dtype=torch.int64, device=dev)
f = _lib.lib().mfvit_debug_attn_stamps
f.argtypes = [ctypes.c_void_p]
assert f(buf.data_ptr()) == 0
torch.cuda.synchronize()
ops.attention_fwd(qkv, H, split=split)
torch.cuda.synchronize()
assert f(None) == 0
s = buf.cpu().view(2, 16, 8, 32)
names = {0: "barrier->", 1: "Q-issue", 2: "S(0)", 3: "7 tile steps", 4: "stage+waits", 9: "stores", 10: "top"}
order = [10, 0, 1, 2, 3, 4, 9]
fine = list(range(11, 23))
fnames = {12: "QK0+ex3", 13: "QK1+ex3", 14: "QK2+ex2", 15: "QK3+pk0", 16: "12 reads", 17: "QK4+ex3", 18: "QK5+ex3", 19: "PV0+ex2", 20: "PV1+pk1", 21: "PV2..5", 22: "mask+max"}
for blk in range(2):
    t0 = int(s[blk, 0, :, 10].min())
    print(f"--- workgroup {'0' if blk == 0 else '37'}: cycles since the first stamp; per point: delta to the previous point")
    for r in range(16):
        if int(s[blk, r, :, 10].max()) == 0:
            break
        for w in range(8):
            row = s[blk, r, w]
            pts = [(i, int(row[i])) for i in order if int(row[i]) != 0]
            txt = f"round {r} wave {w}: start {pts[0][1] - t0:7d} |"
            for (i0, v0), (i1, v1) in zip(pts[:-1], pts[1:]):
                txt += f" {names[i1]} {v1 - v0:6d}"
            txt += f" | total {pts[-1][1] - pts[0][1]:7d}"
            print(txt)
            fp = [(i, int(row[i])) for i in fine if int(row[i]) != 0]
            if len(fp) > 1:
                print("      step 4:" + "".join(f" {fnames[i1]} {v1 - v0:5d}" for (i0, v0), (i1, v1) in zip(fp[:-1], fp[1:])) + f" | {fp[-1][1] - fp[0][1]}")
