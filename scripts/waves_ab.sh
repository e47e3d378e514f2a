#!/bin/bash
OUT=gpurun_out/${1:-waves}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_default.log 2>&1; tail -1 $OUT/pytest_default.log
RPN_SPLIT_WAVES=8 timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider -k "split" > $OUT/pytest_w8.log 2>&1; tail -1 $OUT/pytest_w8.log
for W in 0 8 0 8; do
  RPN_SPLIT_WAVES=$W timeout 300 python bench.py --steps 20 --warmup 3 --layers --precision f16x3 --no-cpu-baseline > $OUT/bench_w$W.json 2> $OUT/layers_w$W.txt
  echo "waves=$W: $(python -c "import json;d=json.load(open('$OUT/bench_w$W.json'));print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")"
  grep -E "block1_conv1|block2_conv2|block3_conv2|block4_conv2|block4_conv1" $OUT/layers_w$W.txt | awk '{print "   ", $1, $3, $4, $5, $6}'
done
