"""Aggregate rocprofv3 --pmc CSVs: mean counter value per dispatch, grouped by kernel name."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row.get("Kernel_Name", "")[:110]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for path in glob.glob(os.path.join(root, "*", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row.get("Kernel_Name", "")[:110]
        dur[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
for k in sorted(agg, key=lambda k: -sum(dur.get(k, [0]))):
    if not any(s in k for s in ("rpn::", "conv", "nms", "iou")):
        continue
    d = dur.get(k, [])
    print("%s\n   dispatches/pass %d  mean duration %.1f us" % (k, len(d) // max(1, len(glob.glob(os.path.join(root, '*', '**', '*kernel_trace.csv'), recursive=True))), (sum(d) / len(d) / 1e3) if d else 0))
    for c, v in sorted(agg[k].items()):
        print("   %-32s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
