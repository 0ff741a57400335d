"""Summarise the rocprofv3 outputs of tools/profile_bench.sh: kernel_stats.csv + HBM traffic per launch / per mfvit_prof class."""
import collections, csv, glob, json, os, re, shutil, sys

out = sys.argv[1]
precision = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multi-feature-vit_amd"))
from mfvit._lib import source_hash  # noqa: E402


def klass(name):
    if "gemm_tn_kernel" in name or "gemm_tn_glds_kernel" in name:
        return "gemm_tn_wgrad"
    if "gemm_rowp_kernel" in name or "gemm_rowp_mixed_kernel" in name:
        # the tall-tile row kernel: <0> residual + LayerNorm forward, <1> LayerNorm backward, >= 10 the (opt-in) plain linears
        m = re.search(r"gemm_rowp_(?:mixed_)?kernel(?:ILi|<)(\d+)", name)
        mode = int(m.group(1)) if m else 0
        return "gemm_nt_row_res_ln" if mode == 0 else "gemm_nt_row_lnbwd" if mode == 1 else "gemm_nt_tile"
    if "gemm_nt_row_kernel" in name:
        # first template argument = element type, second = the epilogue (0: residual + LayerNorm forward, 1: LayerNorm backward)
        return "gemm_nt_row_res_ln" if re.search(r"gemm_nt_row_kernelI(DF16b|DF16_|f|NS_5sbf16E)Li0E", name) or re.search(r"gemm_nt_row_kernel<[^,]+, 0,", name) else "gemm_nt_row_lnbwd"
    if "gemm_nt_tile_kernel" in name or "gemm_nt_pp_kernel" in name:
        return "gemm_nt_tile"
    if "attn_fwd" in name:
        return "attention_fwd"
    if "attn_bwd" in name:
        return "attention_bwd"
    if "x_stream_fwd" in name:
        return "xattn_stream_fwd"
    if "x_stream_bwd" in name:
        return "xattn_stream_bwd"
    return None


for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(out, "kernel_stats.csv"))
vals = {}
for ctr in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    vals[ctr] = acc
per_launch, by_class = {}, collections.defaultdict(lambda: [0.0, 0])
for name in sorted(set(vals["fetch"]) | set(vals["write"])):
    fv, wv = vals["fetch"].get(name, []), vals["write"].get(name, [])
    if not fv or not wv:
        continue
    fetch_kb, write_kb = sum(fv) / len(fv), sum(wv) / len(wv)
    hbm = (2 * fetch_kb + write_kb) * 1024      # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md)
    per_launch[name[:160]] = dict(launches=len(fv), fetch_size_kb_raw=round(fetch_kb, 1), write_size_kb=round(write_kb, 1),
                                  hbm_mb_per_launch=round(hbm / 1e6, 2))
    c = klass(name)
    if c:
        by_class[c][0] += hbm * len(fv)
        by_class[c][1] += len(fv)
src = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`, "
       "MI355X; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section)")
json.dump(dict(source=src, per_kernel=per_launch), open(os.path.join(out, "hbm_traffic_per_launch.json"), "w"), indent=1)
json.dump(dict(source=src, source_hash=source_hash(), precision=precision, per_class={c: dict(hbm_bytes_per_launch=int(b / n), launches_sampled=n) for c, (b, n) in by_class.items()}),
          open(os.path.join(out, "hbm_traffic_by_class.json"), "w"), indent=1)
print(json.dumps({c: round(b / n / 1e6, 1) for c, (b, n) in by_class.items()}))
