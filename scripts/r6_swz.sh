#!/bin/bash
# round 6: halo swizzle A/B (two libraries alternated in one call) + the conv parity tests + one LDS counter pass
OUT=gpurun_out/r6_swz; mkdir -p $OUT
bash scripts/lib_ab.sh r6_swz ab/librpn_r5.so tf_rpn_amd/csrc/librpn_hip.so --no-extra-legs --sustained-seconds 0 2>&1 | tee $OUT/ab.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/$OUT/lds -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --sustained-seconds 0 > $R/$OUT/lds.json 2> $R/$OUT/lds.log
cd $R; python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r6_swz/lds/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        acc[r['Kernel_Name'][:90]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:12]:
    print('%-90s conf%% %.1f' % (k, 100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
