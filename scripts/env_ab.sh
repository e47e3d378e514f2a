#!/bin/bash
# A/B environment knobs of one build in one GPU-box call.  Usage: gpurun -- bash scripts/env_ab.sh tag "A=1 B=0" "A=0" ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do i=0; for E in "$@"; do i=$((i+1))
  env $E timeout 300 python bench.py --steps 30 --warmup 3 --layers --no-cpu-baseline > $OUT/bench_$i.json 2> $OUT/layers_$i.txt
  echo "[$E]: $(python -c "import json;d=json.load(open('$OUT/bench_$i.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done; done
paste <(awk '{print $1, $3}' $OUT/layers_1.txt) $(for j in $(seq 2 $#); do echo "<(awk '{print \$3}' $OUT/layers_$j.txt)"; done | xargs -I{} echo {} ) 2>/dev/null | grep -v amdgpu || true
for j in $(seq 1 $#); do echo "--- $j"; grep -v amdgpu $OUT/layers_$j.txt | awk '{print $1, $2, $3}' | grep -E "block1_conv2|block5_conv1|block2_conv1|rpn_conv"; done
