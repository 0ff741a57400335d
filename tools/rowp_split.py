import csv, glob, statistics as st
for m in (1,):
    f = glob.glob(f"gpurun_out/ks_rowp{m}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for nm in ("gemm_rowp_kernel<0>", "gemm_rowp_kernel<1>"):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if nm in r["Kernel_Name"]]
        d = d[len(d) // 2:]
        print(nm, "first-of-pair median %.1f us, second-of-pair median %.1f us (n=%d)" % (st.median(d[0::2]), st.median(d[1::2]), len(d)))
