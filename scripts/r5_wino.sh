#!/bin/bash
# float32 Winograd conv: parity tests, then kernel durations of the direct and the Winograd kernel on VGG16 layer shapes (rocprofv3 stats)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT; shift
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_wino.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
export TMPDIR=/tmp; cd /tmp
for L in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$L -o w -- python $GRAFT_REPO_ROOT/scripts/wino_probe.py $L > $OUT/probe_$L.txt 2> $OUT/probe_$L.err
  cat $OUT/probe_$L.txt
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof_$L/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wino" in r["Name"] or "igemm" in r["Name"]:
            print("   %-70s calls %s avg %.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
