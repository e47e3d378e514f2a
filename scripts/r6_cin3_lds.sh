#!/bin/bash
# first-layer float32-MFMA kernel: conv tests, then an LDS counter pass of the f32w bench
OUT=gpurun_out/r6_cin3_lds; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider -x -k "first_layer or exact_on_integers or fused" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/$OUT/lds -o pmc -- python $R/bench.py --precision f32w --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --sustained-seconds 0 > $R/$OUT/lds.json 2> $R/$OUT/lds.log
cd $R; python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob('$OUT/lds/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:5]:
    print('%-70s conf%% %.1f  lds-active %.3g' % (k, 100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1), v.get('SQ_LDS_IDX_ACTIVE', 0)))
PY
