#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-f32occ2}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for V in 1 2; do for C in "--backbone mobilenet_v2" "--config c5" "--backbone mobilenet_v2 --batch 1" "--backbone mobilenet_v2 --precision f32"; do
  echo -n "occ$V $C: "
  RPN_F32_OCC=$V timeout 300 python bench.py $C --layers --no-cpu-baseline --no-extra-legs 2> $OUT/l.txt | python -c "import sys,json;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'])"
  grep -E "block_13_expand" $OUT/l.txt
done; done
