#!/bin/bash
OUT=gpurun_out/${1:-c3}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for M in 0 1 0 1; do
  RPN_CIN3_MFMA=$M timeout 300 python bench.py --steps 20 --warmup 3 --layers --no-cpu-baseline > $OUT/bench_$M.json 2> $OUT/layers_$M.txt
  echo "cin3_mfma=$M: $(python -c "import json;d=json.load(open('$OUT/bench_$M.json'));print(d['value'], d['ms_per_step'])")  $(grep -E 'block1_conv1' $OUT/layers_$M.txt | awk '{print $1, $2, $3, $4, $5, $6}')"
done
RPN_CIN3_MFMA=1 timeout 300 python bench.py --backbone mobilenet_v2 --layers --no-cpu-baseline 2>&1 | grep -E "^Conv1|value" | cut -c1-160
