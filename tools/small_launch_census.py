"""usage (GPU box): python tools/small_launch_census.py [batch]: which Python lines of the package issue the small torch kernels of a CA train step - fills
(torch.zeros / zero_), device-to-device copies (copy_ / clone / contiguous / cat) and elementwise adds - from torch.profiler with stacks, three steps."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-feature-vit_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv.pop(1)) if len(sys.argv) > 1 else 16
args = bench.parse()
args.batch = B
run = bench.CaRun(args, torch.device("cuda:0"), 0, "bf16x3", "T")
for _ in range(4):
    run.step()
torch.cuda.synchronize()
NSTEP = 3
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(NSTEP):
        run.step()
    torch.cuda.synchronize()
WATCH = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::clone", "aten::sum", "aten::div", "aten::mul_", "aten::_foreach")
cnt = Counter()
for e in prof.events():
    if not any(e.name.startswith(w) for w in WATCH):
        continue
    frames = [f for f in (e.stack or []) if "multi-feature-vit_amd" in f or "bench.py" in f]
    where = frames[0].split("multi-feature-vit_amd/")[-1] if frames else "(no package frame)"
    cnt[(e.name, where)] += 1
print(f"per step (batch {args.batch}), torch ops by the innermost package frame:")
for (name, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n / NSTEP:7.1f}  {name:18s} {where}")
