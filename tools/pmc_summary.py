import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "mfvit" not in k:
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void mfvit::", "")[:72]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
for k, cs in acc.items():
    lines.append(k)
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c in sorted(m):
        lines.append(f"   {c:28s} {m[c]:16.0f}")
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in m:
                lines.append(f"   {c}/WAVE_CYCLES = {m[c] / wc:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        lines.append(f"   MFMA_BUSY/BUSY_CYCLES = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES']:.3f}")
    if "FETCH_SIZE" in m:
        lines.append(f"   HBM read MB (FETCH_SIZE KB x2 gfx950) = {m['FETCH_SIZE'] * 2 / 1024:.1f}   write MB = {m.get('WRITE_SIZE', 0) / 1024:.1f}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
