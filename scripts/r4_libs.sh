#!/bin/bash
# time several builds of the library on the MobileNetV2 shapes in one GPU-box call: gpurun -- bash scripts/r4_libs.sh TAG lib1.so lib2.so ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do for L in "$@"; do
  echo "-- $L" | tee -a $OUT/times.txt
  RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/mn_time.py --ops ${MN_ARGS} 2>/dev/null | tee -a $OUT/times.txt
done; done
