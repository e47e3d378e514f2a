#!/bin/bash
TAG=${1:-b1ab2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_$name.json 2> $OUT/layers_$name.txt
}
for R in 1 2; do
  run unfused_$R RPN_B1_FUSE=0
  run fused_$R RPN_B1_FUSE=1
  run prio_$R RPN_NMS_PRIO=1
  run hs_$R RPN_NMS_HANDSHAKE=1
  run priohs_$R RPN_NMS_PRIO=1 RPN_NMS_HANDSHAKE=1
  env RPN_B1_FUSE=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --serial-nms > $OUT/bench_serial_$R.json 2> /dev/null
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/b1ab2/bench_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
grep -h block2_conv1 $OUT/layers_*_1.txt
