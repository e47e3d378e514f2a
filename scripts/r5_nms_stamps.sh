#!/bin/bash
# NMS phase stamps (ab/nmsstamp.so = a -DRPN_NMS_STAMP build) at configs[2] for both IoU thresholds, and on the bench model's head outputs
# usage: r5_nms_stamps.sh tag ["ENV=.. ENV=.."]
OUT=gpurun_out/$1; mkdir -p $OUT
export RPN_HIP_LIB=$PWD/ab/nmsstamp.so
for thr in 0.7 0.5; do
  env $2 timeout -k 10 200 python scripts/nms_stamp_probe.py perm $thr > $OUT/perm_$thr.txt 2>&1; tail -4 $OUT/perm_$thr.txt
done
env $2 timeout -k 10 200 python scripts/nms_stamp_probe.py model 0.7 > $OUT/model_0.7.txt 2>&1; tail -4 $OUT/model_0.7.txt
