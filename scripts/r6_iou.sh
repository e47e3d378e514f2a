#!/bin/bash
# round 6: the IoU-map row kernel's laboratory knobs re-measured (store rate depends on the waves in flight: scripts/micro/store_rate.hip)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_iou; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; rm -f $OUT/bbox.txt
export RPN_HIP_LIB=$PWD/tf_rpn_amd/csrc/librpn_hip_lab.so
for R in 1 2; do
  for V in "RPN_IOU_NT=1" "RPN_IOU_NT=0" "RPN_IOU_HALF=1" "RPN_IOU_WAVES=2" "RPN_IOU_PERSIST=1"; do
    echo -n "$V  " >> $OUT/bbox.txt
    env $V timeout -k 10 100 python scripts/iou_probe.py 2>/dev/null >> $OUT/bbox.txt
  done
done
cat $OUT/bbox.txt
