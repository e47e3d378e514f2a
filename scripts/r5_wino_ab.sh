#!/bin/bash
# f32w: A/B of two libraries in one call (bench --precision f32w, per-layer tables)
TAG=$1; A=$2; B=$3; EXTRA=$4      # EXTRA: more bench.py flags, e.g. "--batch 2"
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do for L in $A $B; do
  n=$(basename $L .so)
  RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python bench.py --precision f32w --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --layers $EXTRA > $OUT/bench_$n.json 2> $OUT/layers_$n.txt
  echo "[$n] $(python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['checks']['ok'])")"
done; done
paste <(awk '{print $1, $3}' $OUT/layers_$(basename $A .so).txt) <(awk '{print $3}' $OUT/layers_$(basename $B .so).txt) | grep -v amdgpu
