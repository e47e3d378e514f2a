#!/bin/bash
# round 6 headline: conv / config / pipeline parity with the new library, then A/B of two libraries (alternated), per-layer tables
TAG=$1; A=$2; B=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
for rep in 1 2 3; do for L in $A $B; do
  n=$(basename $L .so)
  RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python bench.py --steps 30 --warmup 3 --layers --no-cpu-baseline --no-extra-legs --sustained-seconds 0 > $OUT/bench_$n.json 2> $OUT/layers_$n.txt
  echo "$n: $(python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done; done
paste <(awk '{print $1, $3}' $OUT/layers_$(basename $A .so).txt) <(awk '{print $3}' $OUT/layers_$(basename $B .so).txt) | grep -v amdgpu
