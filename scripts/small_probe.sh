#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-smallp}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for V in "RPN_S16_DMA=1" "RPN_S16_DMA=0"; do
  for C in "--config c5" "--backbone mobilenet_v2 --batch 1" "--backbone vgg16 --batch 1"; do
    echo "== $V $C" >> $OUT/res.txt
    env $V timeout -k 10 200 python bench.py $C --no-cpu-baseline --no-extra-legs --layers 2> $OUT/layers.tmp | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $OUT/res.txt
    grep -E "rpn_conv|block5|rpn_head|block_13|decode" $OUT/layers.tmp >> $OUT/res.txt
  done
done
cat $OUT/res.txt
