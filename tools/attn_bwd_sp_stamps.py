"""Where a pair of the single-pass attention backward (attn_bwd_sp_kernel) spends its cycles: per wave, the time from barrier to barrier of the
seven steps (work + wait in front of the step's barrier).  Diagnostic library of tools/build_attn_stamp_lib.sh; MFVIT_ATTN_BWD_SP=1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MFVIT_LIB"] = os.path.join(ROOT, "multi-feature-vit_amd", "build", "libmfvit_attnstamp.so")
os.environ["MFVIT_ATTN_BWD_SP"] = "1"
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import ctypes
import torch
from mfvit import ops, _lib
dev = torch.device("cuda:0")
B, T, H, D = 128, 197, 12, 384
x = torch.randn(B, T, 3 * D, device=dev)
d = torch.randn(B, T, D, device=dev)
qkv, do = ops.split_pack(x.view(-1, 3 * D)).view(B, T, -1), ops.split_pack(d.view(-1, D)).view(B, T, -1)
if os.environ.get("STAMP_QKV", "f16") == "f16":     # the encoder's format in bf16x3 mode (MFVIT_X3F16); STAMP_QKV=bf16: the split-bf16 kernels
    qkv = ops.split_pack_f16(x.view(-1, 3 * D).cpu()).view(B, T, -1).to(dev)
print("qkv", qkv.dtype, "parts bwd", os.environ.get("MFVIT_ATTN_PB", "default"))
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
for blk in range(1):
    t0 = int(s[blk, 0, :, 10][s[blk, 0, :, 10] > 0].min())
    print(f"--- workgroup {'0' if blk == 0 else '37'}: per step  work (stamp to arrival at the barrier) + wait (in the barrier)")
    for r in range(16):
        if int(s[blk, r, :, 10].max()) == 0:
            break
        for w in range(8):
            row = [int(v) for v in s[blk, r, w]]
            txt = f"pair {r} wave {w}: top {row[10] - t0:7d} | X {row[0] - row[10]:5d} | Z {row[1] - row[0]:5d} |"
            prev = row[1]
            for t in range(7):
                txt += f" {row[12 + t] - prev:5d}+{row[2 + t] - row[12 + t]:<5d}"
                prev = row[2 + t]
            end = row[11] if row[11] else prev
            print(txt + f" | tail {end - prev:5d} | total {end - row[10]:6d}")
