"""GPU idle analysis of a rocprofv3 kernel trace: union of kernel intervals vs wall time, largest gaps and what surrounds them."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:46]) for r in rows]
lo, hi = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25, float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
t0, t1 = iv[int(len(iv) * lo)][0], iv[int(len(iv) * hi)][0]
gaps, cur_e, last, busy = [], None, None, 0
for s, e, n in iv:
    if e < t0 or s > t1:
        continue
    if cur_e is not None and s > cur_e:
        gaps.append((s - cur_e, last, n))
    if cur_e is None or e > cur_e:
        busy += min(e, t1) - max(s if cur_e is None else max(s, cur_e), t0)
        cur_e, last = e, n
print(f"window {(t1 - t0) / 1e6:.2f} ms, busy fraction {busy / (t1 - t0):.3f}, {len(gaps)} gaps totalling {sum(g[0] for g in gaps) / 1e6:.2f} ms")
c = collections.Counter()
for g in gaps:
    c[(g[1][:34], g[2][:34])] += g[0]
for k, v in c.most_common(10):
    print(f"  {v / 1e6:7.3f} ms  after {k[0]:36s} before {k[1]}")
