#!/bin/bash
# MobileNetV2: parity tests, then bench lines of the three MobileNetV2 shapes for each value of one A/B knob.
# Usage: gpurun -- bash scripts/mn_round.sh tag KNOB
TAG=${1:-mn}; KNOB=${2:-RPN_MN_HR}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x -k "mobilenet or mnv2 or c5 or C5" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
for R in 1 2; do for V in ${VALS:-0 1}; do
  for C in "--backbone mobilenet_v2" "--config c5" "--backbone mobilenet_v2 --batch 1"; do
    echo -n "$KNOB=$V $C: " >> $OUT/res.txt
    env $KNOB=$V timeout -k 10 200 python bench.py $C --no-cpu-baseline --no-extra-legs --layers 2> $OUT/layers_${V}.tmp | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $OUT/res.txt
    [ "$R" = 1 ] && head -3 $OUT/layers_${V}.tmp | tail -1 | cut -c1-78 >> $OUT/res.txt
  done
done; done
cat $OUT/res.txt
