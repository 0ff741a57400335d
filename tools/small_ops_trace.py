"""Which host-side torch calls launch the small fill / copy / add kernels of one CA step: bench.py's CaRun.step under torch.profiler with
stacks; prints per (op, innermost python frame) the number of calls per step and the GPU time."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multi-feature-vit_amd"))
import torch
import bench
sys.argv = ["bench.py", "--no-extras", "--no-cpu-baseline"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda:0")
run = bench.MocoRun(args, dev, 0, args.precision) if args.workload == "moco" else bench.CaRun(args, dev, 0, args.precision, args.mode)
for _ in range(3):
    run.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        run.step()
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.name in (
            "aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_",
            "aten::to", "aten::_to_copy", "aten::cat", "aten::sum", "aten::mul", "aten::empty_like", "aten::new_zeros"):
        st = [s for s in (ev.stack or []) if ("repo" in s and "small_ops_trace" not in s)]
        where = st[0] if st else (ev.stack[0] if ev.stack else "?")
        k = (ev.name, where[-110:])
        acc[k][0] += 1
        acc[k][1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
for (name, where), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{n / N:7.1f} / step  {t / N:9.1f} us  {name:18s} {where}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
