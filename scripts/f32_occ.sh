#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-f32occ}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for R in 1 2; do for V in 1 2 3; do
  RPN_F32_OCC=$V timeout 300 python bench.py --precision f32 --steps 5 --warmup 2 --layers --no-cpu-baseline --no-extra-legs > $OUT/bench_$V.json 2> $OUT/layers_$V.txt
  echo "occ$V: $(python -c "import json;d=json.load(open('$OUT/bench_$V.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done; done
paste <(awk '{print $1, $2, $3}' $OUT/layers_1.txt) <(awk '{print $2, $3}' $OUT/layers_2.txt) <(awk '{print $2, $3}' $OUT/layers_3.txt) | grep -v amdgpu
