#!/bin/bash
# round 3: area-pruned NMS -- parity (in-tree build), then timings with the pruning on / off in the same build, then stamps
OUT=gpurun_out/prune; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests/test_gpu_bbox.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for R in 1 2; do for V in 1 0; do
  echo "== RPN_NMS_PRUNE=$V"
  RPN_NMS_PRUNE=$V timeout -k 10 300 python scripts/nms_c5_time.py 2>/dev/null
  for T in 0.7 0.5; do NMS_THR=$T RPN_NMS_PRUNE=$V timeout -k 10 300 python scripts/nms_phases.py 2>/dev/null; done
done; done
if [ -f ab/nmsstamp.so ]; then
for V in 1 0; do for T in 0.7 0.5; do echo "== stamps perm $T PRUNE=$V"; RPN_NMS_PRUNE=$V RPN_HIP_LIB=$PWD/ab/nmsstamp.so timeout -k 10 300 python scripts/nms_stamp_probe.py perm $T 2>/dev/null | cut -c1-1500; done; done
fi
