"""Which torch ops put fill / copy kernels into a CA train step (67 fills + 50 copies per step): torch.profiler over 3 steps, grouped by
the Python source line that issued them."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-feature-vit_amd")]
import torch
import bench
from torch.profiler import profile, ProfilerActivity
args = bench.parse() if hasattr(bench, "parse") else None
args.batch = int(os.environ.get("CENSUS_B", 16))
run = bench.CaRun(args, torch.device("cuda:0"), 0, "bf16x3", "T")
for _ in range(3):
    run.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        run.step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::add_", "aten::contiguous") and ev.device_type == torch.autograd.DeviceType.CPU:
        st = [s for s in (ev.stack or []) if "site-packages" not in s and "dist-packages" not in s and ".py" in s]
        cnt[(ev.name, st[0] if st else "(no python frame: autograd engine / C++)")] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n / 3:6.1f} per step  {name:18s} {where}")

# the same three steps at the KERNEL level: launches per step by kernel name (what rocprofv3 --stats counts over the whole process, model construction included)
kc = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kc[ev.name.split("(")[0][:100]] += 1
print("---- device kernels / memory operations per step (3 profiled steps)")
tot = 0
for name, n in kc.most_common():
    tot += n
    if any(t in name for t in ("Fill", "fill", "copy", "Copy", "Memset", "Memcpy", "elementwise")) or n / 3 >= 20:
        print(f"{n / 3:7.1f} per step  {name}")
print(f"{tot / 3:7.1f} per step  ALL")
