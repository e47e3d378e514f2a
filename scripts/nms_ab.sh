#!/bin/bash
# NMS: parity tests with the in-tree build, then the NMS timings of several builds (scripts/nms_phases.py) at IoU
# thresholds 0.7 and 0.5.  Usage: gpurun -- bash scripts/nms_ab.sh tag lib1.so lib2.so ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_bbox.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x -k "nms or propose or decode" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for R in 1 2; do for L in "$@"; do for T in 0.7 0.5; do
  echo "== $(basename $L .so)"; NMS_THR=$T RPN_HIP_LIB=$PWD/$L python scripts/nms_phases.py 2>/dev/null
done; done; done
