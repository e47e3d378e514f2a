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


def bench_name(k):
    """rocprofv3 kernel name -> the kernel name bench.py prints for the op (both POOL instantiations together)."""
    import re
    prec = lambda f16: "f16x3" if f16 == "true" else "bf16x3"
    m = re.search(r"conv3x3_split16_dma_kernel<(true|false), (true|false), (\d+)(?:, (?:true|false))?>", k)
    if m: return "conv3x3_split16_dma<%s,%s>" % (prec(m.group(1)), m.group(3))
    m = re.search(r"conv3x3_split16_kernel<(\d+), (\d+), (\d+), (true|false), (true|false)>", k)
    if m: return "conv3x3_split16<%s,%d>" % (prec(m.group(4)), 64 * int(m.group(2)))
    m = re.search(r"conv3x3_split_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false), true>", k)
    if m: return "vgg_block1<%s>" % prec(m.group(5))
    m = re.search(r"conv3x3_split_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)(, false)?>", k)
    if m: return "conv3x3_split<%s>" % prec(m.group(5))
    m = re.search(r"pw_x3_kernel<(\d+), (\d+), (\d+)>", k)
    if m: return "pw_f16x3<%s,576>" % m.group(1)
    m = re.search(r"conv_igemm_f32_dma<(\d+), (\d+), (\d+), (\d+)(?:, (?:true|false))*>", k)        # (, POOL since round 5, LEAN since round 6)
    if m: return "conv_igemm_f32_dma<128x%d>" % (32 * int(m.group(2)) * int(m.group(4)))
    m = re.search(r"conv_igemm_f32<(\d+), (\d+), (\d+), (\d+), (true|false)(?:, (?:true|false))?>", k)
    if m: return "conv_igemm_f32<128x%d%s>" % (32 * int(m.group(2)) * int(m.group(4)), ",generic" if m.group(5) == "true" else "")
    if "conv3x3_wino4n_f32_kernel" in k: return "conv3x3_wino4_f32<16x16x128>"
    if "conv3x3_wino4_f32_kernel" in k: return "conv3x3_wino4_f32<16x32x64>"
    if "conv3x3_wino_f32_kernel" in k: return "conv3x3_wino_f32<16x16x64>"
    return short(k)


def traffic(prefix, out, note, workload=None, subdirs=None, rename=False):
    """subdirs: (fetch dir, write dir) relative to src (default: <prefix>_fetch / <prefix>_write); rename: key the kernels by
    bench.py's op-table names (bench.py reads `hbm_bytes_per_launch` back under those); workload: what the passes ran."""
    acc = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": [], "dur": []})
    dirs = subdirs or ("%s_fetch" % prefix, "%s_write" % prefix)
    for d in dirs:
        for path in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if ours(r["Kernel_Name"]) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    k = bench_name(r["Kernel_Name"]) if rename else short(r["Kernel_Name"])
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
        doc = {"source": note, "kernels": res}
        if workload:
            doc["workload"] = workload
        json.dump(doc, open(os.path.join(dst, out), "w"), indent=1)


copy("bench_default.json", "%s_f16x3_bench_default.json" % tag)
copy("bench_default_layers.txt", "%s_f16x3_bench_layers.txt" % tag)
for cfg in ("c1", "c4", "c5", "mn8", "f32", "f32w"):
    copy("bench_%s.json" % cfg, "%s_%s_bench.json" % (tag, cfg))
    copy("bench_%s_layers.txt" % cfg, "%s_%s_bench_layers.txt" % (tag, cfg))
copy("bbox_c3.json", "%s_c3_bbox_kernels.json" % tag)
for part in ("A", "B"):
    copy("device_%s.txt" % part, "%s_device_%s.txt" % (tag, part))
copy("bw_probe.txt", "%s_bw_probe.txt" % tag)
copy("bench_forcedist.json", "%s_forcedist_bench.json" % tag)          # (round 6) `multi_gpu_configs`: configs[3] / [4] through pack + all-gather
copy("wn_stamps.txt", "%s_f32w_wave_stamps.txt" % tag)                  # (round 6) scripts/wn_stamp_probe.py, -DRPN_STAMP build
copy("mfma_f32_rate.txt", "%s_mfma_f32_rate.txt" % tag)                 # (round 6) scripts/micro/mfma_f32_rate.hip
stats(os.path.join(src, "c3_stats", "c3_kernel_stats.csv"), "%s_c3_kernel_stats.csv" % tag)
traffic("c3", "%s_c3_traffic.json" % tag,
        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) of scripts/bench_bbox.py: B=64, A=8649 (vgg16) and 9216 (mobilenet_v2), G=42")
# counter passes of scripts/pmc_passes.sh: <run>/{mfma,wait,lds,insts,fetch,write,stats}
import subprocess
for run, name, wl in (("vgg_f16x3", "f16x3", {"backbone": "vgg16", "img_size": 500, "batch": 8, "precision": "f16x3"}),
                      ("vgg_f32", "f32", {"backbone": "vgg16", "img_size": 500, "batch": 8, "precision": "f32"}),
                      ("vgg_f32w", "f32w", {"backbone": "vgg16", "img_size": 500, "batch": 8, "precision": "f32w"}),
                      ("mn8", "mn8", {"backbone": "mobilenet_v2", "img_size": 500, "batch": 8, "precision": "f16x3"}),
                      ("c5", "c5", {"backbone": "mobilenet_v2", "img_size": 1024, "batch": 1, "precision": "f16x3"}),
                      ("c1", "c1", {"backbone": "mobilenet_v2", "img_size": 500, "batch": 1, "precision": "f16x3"})):
    d = os.path.join(src, run)
    if not os.path.isdir(d):
        continue
    stats(os.path.join(d, "stats", "stats_kernel_stats.csv"), "%s_%s_bench_kernel_stats.csv" % (tag, name))
    traffic(run, "%s_%s_traffic.json" % (tag, name),
            "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) of bench.py --steps 3 --warmup 1 on this workload",
            workload=wl, subdirs=(os.path.join(run, "fetch"), os.path.join(run, "write")), rename=True)
    summ = os.path.join(d, "summary.txt")
    if os.path.exists(summ):
        table = subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_table.py"), summ], capture_output=True, text=True).stdout
        head = open(os.path.join(root, "scripts", "pmc_table.py")).read().split('"""')[1]
        open(os.path.join(dst, "%s_%s_pmc.txt" % (tag, name)), "w").write(
            "rocprofv3 --pmc passes (separate, kernel-trace only) of bench.py --steps 3 --warmup 1, workload %s\n%s\n%s" % (json.dumps(wl), head, table))
print(sorted(f for f in os.listdir(dst) if f.startswith(tag)))
