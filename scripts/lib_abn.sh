#!/bin/bash
# Timing of N builds of librpn_hip.so in one GPU-box call.  Usage: gpurun -- bash scripts/lib_abn.sh tag lib1.so lib2.so ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for L in "$@"; do
  n=$(basename $L .so)
  RPN_HIP_LIB=$PWD/$L timeout 300 python bench.py --steps 30 --warmup 3 --layers --no-cpu-baseline --no-extra-legs > $OUT/bench_$n.json 2> $OUT/layers_$n.txt
  echo "$n: $(python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done
first=$(basename $1 .so)
cmd="paste <(awk '{print \$1, \$3}' $OUT/layers_$first.txt)"
shift
for L in "$@"; do cmd="$cmd <(awk '{print \$3}' $OUT/layers_$(basename $L .so).txt)"; done
eval "$cmd" | grep -v amdgpu
