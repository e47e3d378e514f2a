#!/bin/bash
TAG=${1:-iouab}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_bbox.py -m gpu -q --tb=short -p no:cacheprovider -x -k "iou" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
for R in 1 2; do
  for V in "RPN_IOU_FAST=0" "RPN_IOU_FAST=1" "RPN_IOU_WAVES=2" "RPN_IOU_PERSIST=1" "RPN_IOU_PERSIST=1 RPN_IOU_WAVES=2"; do
    echo -n "$V  " >> $OUT/bbox.txt
    env $V timeout -k 10 100 python scripts/iou_probe.py 2>/dev/null >> $OUT/bbox.txt
  done
done
cat $OUT/bbox.txt
