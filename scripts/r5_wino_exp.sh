#!/bin/bash
# f32w timing experiments (wrong results on purpose): which part of the slice loop costs what
OUT=gpurun_out/$1; mkdir -p $OUT; shift
for rep in 1 2; do for L in "$@"; do
  n=$(basename $L .so)
  RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python bench.py --precision f32w --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_$n.json 2> $OUT/layers_$n.txt
  echo "[$n] $(python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print(d['value'], d['ms_per_step'])") $(grep -E 'block3_conv2|block5_conv1 ' $OUT/layers_$n.txt | awk '{printf "%s %s  ", $1, $3}')"
done; done
