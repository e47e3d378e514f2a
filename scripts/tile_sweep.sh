#!/bin/bash
# A/B the split-conv tile variants on the bench workload (env knobs change speed only).
TAG=${1:-sweep}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider -k "split" > $OUT/pytest.log 2>&1
tail -2 $OUT/pytest.log
for CFG in "0 0" "82 1" "42 1" "42 2" "41 1" "41 2"; do
  set -- $CFG
  RPN_SPLIT_TILE=$1 RPN_SPLIT_BBUF=$2 timeout 300 python bench.py --steps 10 --warmup 2 --layers --precision f16x3 --no-cpu-baseline > $OUT/bench_$1_$2.json 2> $OUT/layers_$1_$2.txt
  echo "tile=$1 bbuf=$2: $(python -c "import json;d=json.load(open('$OUT/bench_$1_$2.json'));print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")"
  grep -E "block2_conv2|block3_conv2|block4_conv2|block5_conv2|block1_conv2" $OUT/layers_$1_$2.txt | awk '{print "   ", $1, $3, $4, $5, $6}'
done
