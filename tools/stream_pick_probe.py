"""usage (GPU box): python tools/stream_pick_probe.py: does the step time of the two-stream CA step depend on WHICH HIP stream the second encoder gets?  The model takes
the next stream of torch's pool; here k throw-away streams are taken first (k = 0 .. 5), each in a fresh process, same box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path[:0] = [%r, os.path.join(%r, "multi-feature-vit_amd")]
import torch, time
k = int(sys.argv[1]); prio = int(sys.argv[2])
keep = [torch.cuda.Stream() for _ in range(k)]
for s in keep:
    with torch.cuda.stream(s):
        torch.zeros(1, device="cuda:0")
import bench
sys.argv = sys.argv[:1]
args = bench.parse()
run = bench.CaRun(args, torch.device("cuda:0"), 0, "bf16x3", "T")
if prio:
    run.model._side = torch.cuda.Stream(priority=-1)
for _ in range(6): run.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): run.step()
torch.cuda.synchronize()
print("RESULT", k, prio, round((time.perf_counter() - t0) / 20 * 1e3, 3), flush=True)
''' % (ROOT, ROOT)
for k, prio in [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (0, 1), (0, 0)]:
    r = subprocess.run([sys.executable, "-c", CHILD, str(k), str(prio)], capture_output=True, text=True)
    out = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print(out[-1] if out else ("FAILED " + r.stderr[-300:]), flush=True)
