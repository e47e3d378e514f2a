#!/bin/bash
OUT=gpurun_out/${1:-m16}
mkdir -p $OUT
RPN_SPLIT_MFMA16=2 timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -k "split or propose" > $OUT/pytest_m16.log 2>&1; tail -3 $OUT/pytest_m16.log
for M in 1 2 1 2; do
  RPN_SPLIT_MFMA16=$M timeout 300 python bench.py --steps 20 --warmup 3 --layers --no-cpu-baseline > $OUT/bench_$M.json 2> $OUT/layers_$M.txt
  echo "mfma16=$M: $(python -c "import json;d=json.load(open('$OUT/bench_$M.json'));print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")"
  grep -E "block3_conv2|block4_conv2|block5_conv2|rpn_conv" $OUT/layers_$M.txt | awk '{print "   ", $1, $3, $4, $5, $6}'
done
