"""Regenerates the numbers of DESIGN.md section 5 from the committed evidence (profiles/<tag>_*) and from the code objects of the
shipped library, between the `BEGIN GENERATED` / `END GENERATED` markers.  Nothing is typed by hand:

    python scripts/design_tables.py r05            # rewrites DESIGN.md in place
"""
import csv, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import codeobj  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (TAG, name))
PEAK = {"f16x3": 2500.0 / 3.0, "bf16x3": 2500.0 / 3.0, "f32": 157.3, "f32w": 157.3}
out = []
w = out.append


def layers(path):
    rows = []
    if not os.path.exists(path):
        return rows
    for line in open(path):
        m = re.match(r"(\S+)\s+(\S+)\s+([\d.]+) ms(?:\s+([\d.]+) TF/s\s+([\d.]+) GB/s)?", line)
        if m:
            rows.append((m.group(1), m.group(2), float(m.group(3)), float(m.group(4) or 0), float(m.group(5) or 0)))
    return rows


line = json.load(open(P("f16x3_bench_default.json")))
w("### 5.1 The driver's line (`profiles/%s_f16x3_bench_default.json`; one MI355X (gfx950) of the pool: devices differ by +-4 %%, `%s_device_A.txt`)\n" % (TAG, TAG))
r = line["roofline"]
w("| leg | value | ms per step | dominant kernel | achieved | peak | frac | note |")
w("|---|---|---|---|---|---|---|---|")
w("| **configs[1] headline** (`dtype` %s, K = %d) | **%.1f images/s** | %.4f | `%s` | %.1f TF/s | %.1f | **%.3f** | %d launches per step, %.4f ms each; HBM traffic per launch %s MB; conv stack %.3f ms, decode+NMS %.4f ms (overlapped) |"
  % (line["dtype"], line["steps"], line["value"], line["ms_per_step"], r["kernel"], r["achieved"], r["peak"], r["frac"],
     r["launches_per_step"], r["avg_launch_ms"], ("%.1f" % (r["traffic"] / 1e6)) if r.get("traffic") else "n/a", r["conv_stack_ms"], r["decode_nms_ms"]))
whole = 156.552e9 * 8 / (line["ms_per_step"] * 1e-3) / 1e12
w("| whole step | | | all kernels | %.1f TF/s | %.1f | %.3f | 8 x 156.552 GF per step |" % (whole, r["peak"], whole / r["peak"]))
if line.get("sustained"):
    s = line["sustained"]
    w("| sustained loop | %.1f images/s | | | | | | %d steps over %.2f s |" % (s["images_per_s"], s["steps"], s["seconds"]))
for key, label in (("exact_f32", "exact float32 (parity-clean arithmetic)"), ("f32_winograd", "float32 Winograd (`f32w`, a precision of its own)")):
    if key in line:
        e = line[key]
        er = e["roofline"]
        diff = e.get("max_abs_diff_vs_headline") or e.get("max_abs_diff_vs_exact_f32")
        note = "head outputs vs %s: reg %.1e / cls %.1e" % ("the headline's" if key == "exact_f32" else "exact f32's", diff["reg"], diff["cls"]) if diff else ""
        if "effective_tflops" in er:
            note += "; effective (direct-conv flops) %.1f TF/s" % er["effective_tflops"]
        w("| %s | %.1f images/s | %.4f | `%s` | %.1f TF/s | %.1f | %.3f | %s |" % (label, e["value"], e["ms_per_step"], er["kernel"], er["achieved"], er["peak"], er["frac"], note))
cb = line.get("cpu_baseline")
if cb:
    w("| `cpu_baseline` (kind %s) | %.2f %s | | | | | | %s cores; %s |" % (cb["kind"], cb["value"], cb["unit"], cb["cores"], cb.get("sample", "")[:160]))
ck = line["checks"]
w("\n`checks`: ok = %s, float16 range word %s, valid_min %s, head outputs vs exact f32 reg %.1e / cls %.1e (tolerance %g)%s.\n"
  % (ck["ok"], "set" if ck["f16_range"] else "clear", ck["valid_min"], ck["max_abs_diff_vs_exact_f32"]["reg"], ck["max_abs_diff_vs_exact_f32"]["cls"],
     ck["tolerance"], (", f32w vs exact f32 reg %.1e / cls %.1e" % (ck["f32_winograd_max_abs_diff_vs_exact_f32"]["reg"], ck["f32_winograd_max_abs_diff_vs_exact_f32"]["cls"]))
     if ck.get("f32_winograd_max_abs_diff_vs_exact_f32") else ""))

# ---- per-layer table
w("### 5.2 VGG16 + head per layer, batch 8 (`profiles/%s_{f16x3,f32,f32w}_bench_layers.txt`; per-op HIP events)\n" % TAG)
L16, L32, L32w = layers(P("f16x3_bench_layers.txt")), layers(P("f32_bench_layers.txt")), layers(P("f32w_bench_layers.txt"))
d32 = {n: (k, ms, tf) for n, k, ms, tf, _ in L32}
d32w = {n: (k, ms, tf) for n, k, ms, tf, _ in L32w}
w("| layer | f16x3 kernel | ms | TF/s | frac of 833 | f32 ms | frac of 157.3 | f32w ms | executed frac of 157.3 |")
w("|---|---|---|---|---|---|---|---|---|")
for n, k, ms, tf, _ in L16:
    if n == "decode+nms":
        w("| decode+nms | `%s` | %.3f | | | | | | |" % (k, ms))
        continue
    a = d32.get(n)
    b = d32w.get(n)
    a_s = "%.3f | %.3f" % (a[1], a[2] / 157.3) if a else " | "
    if b:
        executed = b[2] / (4.0 if "wino4" in b[0] else (2.25 if "wino" in b[0] else 1.0))
        b_s = "%.3f | %.3f" % (b[1], executed / 157.3)
    else:
        b_s = " | "
    w("| %s | `%s` | %.3f | %.1f | %.3f | %s | %s |" % (n, k, ms, tf, tf / PEAK["f16x3"], a_s, b_s))
if L32:
    extra = [n for n, *_ in L32 if n not in {x[0] for x in L16}]
    if extra:
        w("\n(the f32 / f32w graphs run block 1 as two layers: " + ", ".join("%s %.3f / %s ms" % (n, d32[n][1], ("%.3f" % d32w[n][1]) if n in d32w else "-") for n in extra) + ")")
w("")

# ---- box path
c3 = line.get("c3")
if c3:
    w("### 5.3 Box path, configs[2] (B = 64, A = %d, G = %d; event-timed, algorithmic bytes against 8 TB/s)\n" % (c3["A"], c3["G"]))
    w("| kernel | us | GB/s | frac | measured HBM traffic per launch |")
    w("|---|---|---|---|---|")
    for key in ("decode", "iou_map", "decode_nms_iou0.7", "decode_nms_iou0.5"):
        v = c3[key]
        extra = " (%.2f G boxes/s)" % (v["boxes_per_sec"] / 1e9) if "boxes_per_sec" in v else ""
        w("| %s | %.2f | %.1f | %.4f | %s |" % (key + extra, v["us"], v["GBps"], v["frac"], ("%.1f MB" % (v["traffic"] / 1e6)) if v.get("traffic") else "n/a"))
    if os.path.exists(P("bw_probe.txt")):
        first = [l for l in open(P("bw_probe.txt")) if "93 MB" in l]
        if first:
            w("\nWrite-only rate of the device at the IoU map's size (`%s_bw_probe.txt`): %s" % (TAG, first[0].strip()))
    w("")

# ---- other configs
oc = line.get("other_configs")
if oc:
    w("### 5.4 The other BASELINE configs (per-GPU shapes, same line)\n")
    w("| leg | workload | ms per step | images/s | two pipelines in flight |")
    w("|---|---|---|---|---|")
    for k, v in oc.items():
        two = v.get("pipelines_in_flight_2")
        w("| %s | %s | %.4f | %.1f | %s |" % (k, v.get("workload", ""), v["ms_per_step"], v["value"],
                                            ("%.4f ms, %.1f images/s" % (two["ms_per_step"], two["value"])) if two else "-"))
    w("")

# ---- rocprof stats of the dominant kernels
w("### 5.5 rocprofv3 `--kernel-trace --stats` of the same commands (`profiles/%s_*_bench_kernel_stats.csv`; our kernels)\n" % TAG)
w("| workload | kernel | calls | average us |")
w("|---|---|---|---|")
for wl in ("f16x3", "f32", "f32w", "mn8", "c5", "c1", "c3"):
    path = P("%s_bench_kernel_stats.csv" % wl) if wl != "c3" else P("c3_kernel_stats.csv")
    if not os.path.exists(path):
        continue
    rows = sorted((r_ for r_ in csv.DictReader(open(path)) if "stream_spin" not in r_["Kernel"]), key=lambda r_: -float(r_["TotalDurationNs"]))[:4]
    for r_ in rows:
        w("| %s | `%s` | %s | %.1f |" % (wl, r_["Kernel"][:80], r_["Calls"], float(r_["AverageNs"]) / 1e3))
w("\nCounter passes (six separate `--pmc` passes + FETCH / WRITE per workload): `profiles/%s_{f16x3,f32,f32w,mn8,c5,c1}_pmc.txt` and `_traffic.json`.\n" % TAG)

# ---- counters of the dominant kernels
w("### 5.5b Counters of the heaviest kernels (`profiles/%s_*_pmc.txt`: separate `--pmc` passes; columns as defined in those files)\n" % TAG)
w("| workload | kernel | launches | us | matrix pipe busy % | waves parked % | issue-stalled % | waves / SIMD | LDS conflict % | GHz | fetch MB | write MB |")
w("|---|---|---|---|---|---|---|---|---|---|---|---|")
for wl in ("f16x3", "f32", "f32w", "mn8", "c5", "c1"):
    path = P("%s_pmc.txt" % wl)
    if not os.path.exists(path):
        continue
    rows = []
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m and "stream_spin" not in m.group(1):
            rows.append(m.groups())
    rows.sort(key=lambda g_: -int(g_[1]) * float(g_[2]))
    for g_ in rows[:3]:
        w("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (wl, g_[0].strip()[:70], g_[1], g_[2], g_[3], g_[6], g_[7], g_[8], g_[9], g_[10], g_[11], g_[12]))
w("")

# ---- code objects
w("### 5.6 Code objects of the shipped library (`tests/codeobj.py`; budgets held by `tests/test_host.py::test_kernel_register_budgets`)\n")
tab = codeobj.table(os.path.join(ROOT, "tf_rpn_amd", "csrc", "librpn_hip.so"))
keys = [r"conv3x3_split16_dma_kernel<true,(false|true),128,false>", r"conv3x3_split16_dma_kernel<true,false,64,(false|true)>", r"conv3x3_split_kernel<8,1,2,4,true,true,true>",
        r"conv_igemm_f32_dma<2,2,2,2,(false|true),true>", r"conv_igemm_f32<2,2,2,2,false,false,true>", r"conv_igemm_f32<4,1,1,2,false,false,true>", r"conv3x3_wino4n_f32_kernel", r"conv3x3_wino4_f32_kernel", r"conv3x3_wino_f32_kernel", r"conv_cin3_f32_mfma_kernel<true>",
        r"nms_kernel<true>", r"iou_map_rows_kernel<true,4,true,64>", r"stem_block_kernel<8>", r"ir_block_hrx3_kernel<16,96,32,24,2,false,1,1>",
        r"ir_block_x3_kernel<64,384,64,true>", r"pw_x3_kernel<96,96,8>", r"rpn_head_kernel<3,2,16>", r"decode_kernel"]
w("| kernel | VGPRs | LDS bytes | threads | workgroups per CU (registers / LDS) | SGPR spills | scratch |")
w("|---|---|---|---|---|---|---|")
for pat in keys:
    for name, (vg, ss, vs, scr, lds, wg) in sorted(tab.items()):
        if re.fullmatch(pat, name):
            waves = max(1, wg // 64)
            alloc = (vg + 7) // 8 * 8
            by_reg = (512 // alloc) * 4 // waves if alloc else 0
            by_lds = (160 * 1024) // lds if lds else 99
            w("| `%s` | %d | %d | %d | %s / %s | %d | %d |" % (name, vg, lds, wg, by_reg, by_lds if lds else "-", ss, scr))
n_k = len(tab)
w("\n%d kernels in the library; none uses scratch memory (`test_nms_kernels_use_no_scratch_memory`).  (`nms_kernel`'s LDS is dynamic: up to 160 KB, one workgroup per CU.)\n" % n_k)

# ---- tests
def count(marker):
    try:
        t = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "--collect-only", "-q", "-m", marker, "-p", "no:cacheprovider"],
                           capture_output=True, text=True, timeout=300, cwd=ROOT).stdout
        m = re.search(r"(\d+)(?:/\d+)? tests? (?:collected|selected)", t) or re.search(r"(\d+) selected", t)
        return int(m.group(1)) if m else None
    except Exception:
        return None
g, c = count("gpu"), count("not gpu")
w("### 5.7 Tests\n")
w("`pytest -m gpu`: %s tests (all through the C ABI; one of them, the 2-rank RCCL test, runs only where two devices are visible); `pytest -m \"not gpu\"`: %s tests.\n" % (g, c))

text = open(os.path.join(ROOT, "DESIGN.md")).read()
a, b = text.index("<!-- BEGIN GENERATED (scripts/design_tables.py) -->"), text.index("<!-- END GENERATED -->")
text = text[:a] + "<!-- BEGIN GENERATED (scripts/design_tables.py) -->\n" + "\n".join(out) + "\n" + text[b:]
open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
print("DESIGN.md section 5 regenerated from profiles/%s_* (%d lines)" % (TAG, len(out)))
