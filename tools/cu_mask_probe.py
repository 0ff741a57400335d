"""usage (GPU box): python tools/cu_mask_probe.py: the two-stream CA step with the two encoder streams created by hipExtStreamCreateWithCUMask - each encoder on its own
half of the CUs (contiguous halves / interleaved) - against the plain pool streams, fresh process each, same box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, os, sys, time
sys.path[:0] = [%r, os.path.join(%r, "multi-feature-vit_amd")]
import torch
mode = sys.argv[1]
hip = ctypes.CDLL("libamdhip64.so")
def masked(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=torch.device("cuda:0"))
torch.zeros(1, device="cuda:0")
LO, HI = [0xffffffff] * 4 + [0] * 4, [0] * 4 + [0xffffffff] * 4
EV, OD = [0x55555555] * 8, [0xaaaaaaaa] * 8
import bench
sys.argv = sys.argv[:1]
args = bench.parse()
run = bench.CaRun(args, torch.device("cuda:0"), 0, "bf16x3", "T")
main = None
if mode == "side_lo": run.model._side = masked(LO)
elif mode == "side_ev": run.model._side = masked(EV)
elif mode == "both_halves": run.model._side = masked(LO); main = masked(HI)
elif mode == "both_evod": run.model._side = masked(EV); main = masked(OD)
elif mode == "main_full_side_full": run.model._side = masked([0xffffffff] * 8)
def steps(n):
    for _ in range(n): run.step()
ctx = torch.cuda.stream(main) if main is not None else torch.cuda.stream(torch.cuda.current_stream())
with ctx:
    steps(6)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps(20)
    torch.cuda.synchronize()
print("RESULT", mode, round((time.perf_counter() - t0) / 20 * 1e3, 3), flush=True)
''' % (ROOT, ROOT)
for mode in (sys.argv[1:] or ["plain", "side_lo", "side_ev", "both_halves", "both_evod", "main_full_side_full", "plain"]):
    r = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True)
    out = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print(out[-1] if out else ("FAILED " + mode + " " + r.stderr[-400:].replace("\n", " | ")), flush=True)
