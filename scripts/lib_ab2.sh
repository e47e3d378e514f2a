#!/bin/bash
# A/B two builds of librpn_hip.so in one GPU-box call (timing only).  Usage: gpurun -- bash scripts/lib_ab2.sh tag libA.so libB.so [bench args]
TAG=$1; A=$2; B=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do for L in $A $B; do
  n=$(basename $L .so)
  RPN_HIP_LIB=$PWD/$L timeout 300 python bench.py --steps 30 --warmup 3 --layers --no-cpu-baseline --no-extra-legs "$@" > $OUT/bench_$n.json 2> $OUT/layers_$n.txt
  echo "$n: $(python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done; done
paste <(awk '{print $1, $3}' $OUT/layers_$(basename $A .so).txt) <(awk '{print $3}' $OUT/layers_$(basename $B .so).txt) | grep -v amdgpu
