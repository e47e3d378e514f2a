#!/bin/bash
OUT=gpurun_out/${1:-small}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider -k "split" > $OUT/pytest.log 2>&1; tail -1 $OUT/pytest.log
for M in 0 1 0 1; do
  RPN_SPLIT_SMALL=$M timeout 300 python bench.py --steps 20 --warmup 3 --layers --no-cpu-baseline > $OUT/bench_$M.json 2> $OUT/layers_$M.txt
  echo "small=$M: $(python -c "import json;d=json.load(open('$OUT/bench_$M.json'));print(d['value'], d['ms_per_step'])")  $(grep -E 'block5_conv2' $OUT/layers_$M.txt | awk '{print $1, $3, $4, $5, $6}')"
done
