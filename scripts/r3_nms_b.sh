#!/bin/bash
# round 3: NMS parity with the in-tree build, the two-rank test, then N builds timed side by side + phase stamps
# usage: bash scripts/r3_nms_b.sh tag lib1.so lib2.so ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_bbox.py tests/test_gpu_pipeline.py tests/test_gpu_distributed.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for L in "$@"; do for T in 0.7 0.5; do
  echo "== $(basename $L .so)"; NMS_THR=$T RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_phases.py 2>/dev/null
done; done
for L in "$@"; do echo "== $(basename $L .so)"; RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_c5_time.py 2>/dev/null; done
if [ -f ab/nmsstamp.so ]; then
for K in perm; do for T in 0.7 0.5; do echo "== stamps $K $T"; RPN_HIP_LIB=$PWD/ab/nmsstamp.so timeout -k 10 300 python scripts/nms_stamp_probe.py $K $T 2>/dev/null | cut -c1-1500; done; done
fi
