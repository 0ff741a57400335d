"""Concurrency analysis of a rocprofv3 --kernel-trace CSV of bench.py: how many kernels are in flight over time, and how long each
kernel class takes when it shares the GPU compared with its serialized duration.
usage: python3 tools/timeline.py <dir with *_kernel_trace.csv> [warmup steps]"""
import csv, glob, os, sys, collections

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows]
ev.sort()
# the timed region = from the end of optimizer launch number <warmup> to the end of launch <warmup + steps> (one adam launch per step)
warm, steps = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 8)
ad_all = [e for e in ev if "adam_kernel" in e[2]]
per_step = int(os.environ.get("ADAM_PER_STEP", "3"))      # optimizer launches per step = param groups (bench.py CaRun: fusion + heads, encoder 1, encoder 2)
ad = ad_all[per_step - 1::per_step]                        # the LAST optimizer launch of every step
t0, t1 = ad[warm - 1][1], ad[warm + steps - 1][1]
ev = [e for e in ev if e[0] >= t0 and e[1] <= t1]
pts = []
for s, e, n, q in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
hist = collections.Counter(); cur = 0; last = t0
for t, dlt in pts:
    hist[cur] += t - last; last = t; cur += dlt
tot = t1 - t0
print(f"window {tot/1e6:.2f} ms = {steps} steps of {tot/1e6/steps:.2f} ms, {len(ev)} kernels")
for k in sorted(hist): print(f"  {k} kernels in flight: {100*hist[k]/tot:5.1f} %")
def cls(n):
    if "gemm_rowp_kernel" in n:
        return "gemm_rowp_kernel" + ("<bwd>" if "<1>" in n else "<fwd>" if "<0>" in n else "<tile>")
    for key in ("gemm_nt_row_kernel", "gemm_nt_tile_kernel", "gemm_tn_glds_kernel", "gemm_tn_kernel", "attn_bwd", "attn_fwd", "adam", "x_stream"):
        if key in n:
            if key == "gemm_nt_row_kernel": return key + ("<bwd>" if ", 1, " in n.split("(")[0] else "<fwd>")
            return key
    return "other"
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in ev:
    a = agg[cls(n)]; a[0] += 1; a[1] += e - s
print("class: launches, avg us (while sharing the GPU), total ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"  {k:28s} {c:6d} {t/c/1e3:9.1f} {t/1e6:9.2f}")
print(f"sum of kernel durations / window = {sum(t for c, t in agg.values())/tot:.2f}")
qs = collections.Counter(q for *_, q in ev); print("queues:", dict(qs))
# the idle gaps (no kernel in flight): which kernels sit on either side of the longest ones
if len(sys.argv) > 4:
    ev2 = sorted(ev)
    gaps = []
    cur_end = ev2[0][1]; last_name = ev2[0][2]
    for s0, e0, n0, q0 in ev2[1:]:
        if s0 > cur_end:
            gaps.append((s0 - cur_end, last_name[:60], n0[:60]))
        if e0 > cur_end:
            cur_end = e0; last_name = n0
    gaps.sort(reverse=True)
    print(f"idle gaps: {len(gaps)} totalling {sum(g[0] for g in gaps)/1e6:.2f} ms; the longest:")
    for g in gaps[:int(sys.argv[4])]:
        print(f"  {g[0]/1e3:8.1f} us  after {g[1]}  before {g[2]}")
    import collections as _c
    agg2 = _c.Counter()
    for g in gaps: agg2[(g[1][:40], g[2][:40])] += g[0]
    print("by (previous kernel, next kernel), ms:")
    for k, v in agg2.most_common(12): print(f"  {v/1e6:7.3f}  {k[0]}  ->  {k[1]}")
