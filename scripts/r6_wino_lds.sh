#!/bin/bash
# f32w parity + A/B + one LDS counter pass of the f32w bench
TAG=$1; A=$2; B=$3
bash scripts/r6_wino.sh $TAG $A $B 2>&1 | grep -v "^  MFMA\|^  staging\|last arriver\|arrival\|workgroup 0\|^layer\|period\|per workgroup"
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=gpurun_out/$TAG; cd /tmp
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
