"""Host-side cost of one CA train step at a small batch (HOSTPROF_B, default 16), where the host - not the GPU - sets the step time:
(1) wall time per phase without synchronisation, (2) cProfile of the same loop, sorted by own time and by cumulative time.
The GPU queue never fills at this batch, so the host times are launch-path costs and not waits."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-feature-vit_amd")]
import torch  # noqa: E402
import bench  # noqa: E402

args = bench.parse()
args.batch = int(os.environ.get("HOSTPROF_B", 16))
run = bench.CaRun(args, torch.device("cuda:0"), 0, "bf16x3", "T")
for _ in range(5):
    run.step()
torch.cuda.synchronize()

N = int(os.environ.get("HOSTPROF_N", 50))
acc = [0.0] * 6


def step():
    t0 = time.perf_counter()
    run.opt.zero_grad(set_to_none=True); t1 = time.perf_counter()
    fused, x_c, x_e = run.model(run.backs[0], run.backs[1], run.x, run.xe); t2 = time.perf_counter()
    loss, _ = run._ce(fused + x_c + x_e, run.target); t3 = time.perf_counter()
    loss.backward(); t4 = time.perf_counter()
    run.sync.reduce_grads(run.small); t5 = time.perf_counter()
    run.opt.step(); t6 = time.perf_counter()
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        acc[i] += d


T0 = time.perf_counter()
for _ in range(N):
    step()
Th = time.perf_counter() - T0
torch.cuda.synchronize()
Tg = time.perf_counter() - T0
print(f"B = {args.batch}: host loop {1e3 * Th / N:.3f} ms/step, with the final sync {1e3 * Tg / N:.3f} ms/step")
for n, v in zip(("zero_grad", "forward", "loss", "backward", "reduce_grads", "opt.step"), acc):
    print(f"  {n:12s} {1e3 * v / N:7.3f} ms/step (host)")

pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    run.step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    print(f"---- cProfile, {N} steps, by {key}")
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats(key).print_stats(28)
