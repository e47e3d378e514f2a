"""profiles/<tag>_* from a gpurun_out/<round> directory written by scripts/round2_a.sh (+ scripts/pmc_round.sh):
bench JSON lines, per-op tables, rocprofv3 --kernel-trace --stats summaries (our kernels only) and HBM traffic per launch
from the separate FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE x 2: gfx950 tallies wide coalesced reads at half their
bytes, MI355X_MICROARCH.md; both counters are reported in KiB).

    python scripts/collect_profiles.py gpurun_out/r2e r02
"""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")


def copy(name, out):
    p = os.path.join(src, name)
    if os.path.exists(p):
        text = open(p).read()
        text = "\n".join(l for l in text.splitlines() if "amdgpu.ids" not in l and not l.startswith("bench exit")) + "\n"
        open(os.path.join(dst, out), "w").write(text)


def ours(name):
    return "rpn::" in name


def short(name):
    return name.replace("void ", "").replace("rpn::", "").split("(")[0]


def stats(csv_path, out):
    if not os.path.exists(csv_path):
        return
    rows = [r for r in csv.DictReader(open(csv_path)) if ours(r["Name"])]
    with open(os.path.join(dst, out), "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]), r["MinNs"], r["MaxNs"]])


def traffic(prefix, out, note):
    acc = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": [], "dur": []})
    for cname in ("fetch", "write"):
        for path in glob.glob(os.path.join(src, "%s_%s" % (prefix, cname), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if ours(r["Kernel_Name"]) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    k = short(r["Kernel_Name"])
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    acc[k]["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    res = {}
    for k, d in acc.items():
        if d["FETCH_SIZE"] and d["WRITE_SIZE"]:
            fetch = 2.0 * 1024 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
            write = 1024.0 * sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
            dur = sum(d["dur"]) / len(d["dur"])
            res[k] = {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_corrected_x2": round(fetch),
                      "write_bytes": round(write), "avg_duration_us_under_pmc": round(dur / 1e3, 2),
                      "hbm_GBps": round((fetch + write) / dur, 1), "frac_of_8TBps": round((fetch + write) / dur / 8000.0, 4),
                      "launches_sampled": len(d["FETCH_SIZE"])}
    if res:
        json.dump({"source": note, "kernels": res}, open(os.path.join(dst, out), "w"), indent=1)


copy("bench_default.json", "%s_f16x3_bench_default.json" % tag)
copy("bench_default_layers.txt", "%s_f16x3_bench_layers.txt" % tag)
for cfg in ("c4", "c5", "mn8"):
    copy("bench_%s.json" % cfg, "%s_%s_bench.json" % (tag, cfg))
    copy("bench_%s_layers.txt" % cfg, "%s_%s_bench_layers.txt" % (tag, cfg))
copy("bbox_c3.json", "%s_c3_bbox_kernels.json" % tag)
stats(os.path.join(src, "stats", "stats_kernel_stats.csv"), "%s_f16x3_bench_kernel_stats.csv" % tag)   # scripts/round2_final.sh
stats(os.path.join(src, "c3_stats", "c3_kernel_stats.csv"), "%s_c3_kernel_stats.csv" % tag)
stats(os.path.join(src, "mn8_stats", "mn8_kernel_stats.csv"), "%s_mn8_kernel_stats.csv" % tag)
stats(os.path.join(src, "c5_stats", "c5_kernel_stats.csv"), "%s_c5_kernel_stats.csv" % tag)
traffic("c3", "%s_c3_traffic.json" % tag,
        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) of scripts/bench_bbox.py: B=64, A=8649 (vgg16) and 9216 (mobilenet_v2), G=42")
traffic("mn8", "%s_mn8_traffic.json" % tag,
        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --backbone mobilenet_v2 --steps 5 (B=8, 500x500)")
print(sorted(os.listdir(dst)))
