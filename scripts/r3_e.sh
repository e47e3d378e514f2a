#!/bin/bash
# round 3, call E: NMS parity (parallel histogram find, shared suffix sums, LDS prefix for the cluster gather), timings, stamps
OUT=gpurun_out/r3e; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests/test_gpu_bbox.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for L in ab/cluster2.so ab/scan.so; do for T in 0.7 0.5; do
  echo "== $(basename $L .so)"; NMS_THR=$T RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_phases.py 2>/dev/null
done; done
for L in ab/cluster2.so ab/scan.so; do echo "== $(basename $L .so)"; RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_c5_time.py 2>/dev/null; done
for K in perm model_c5; do echo "== stamps $K 0.7"; RPN_HIP_LIB=$PWD/ab/nmsstamp.so timeout -k 10 300 python scripts/nms_stamp_probe.py $K 0.7 2>/dev/null | cut -c1-1200; done
