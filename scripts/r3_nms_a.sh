#!/bin/bash
# round 3, call A: NMS parity with the spill-free build, A/B against the round-2 library, phase stamps
OUT=gpurun_out/r3a; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_bbox.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for L in ab/r2.so ab/nospill.so; do for T in 0.7 0.5; do
  echo "== $(basename $L .so)"; NMS_THR=$T RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_phases.py 2>/dev/null
done; done
for L in ab/r2.so ab/nospill.so; do echo "== $(basename $L .so)"; RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_c5_time.py 2>/dev/null; done
for K in perm model_c5; do for T in 0.7 0.5; do echo "== stamps $K $T"; RPN_HIP_LIB=$PWD/ab/nmsstamp.so timeout -k 10 300 python scripts/nms_stamp_probe.py $K $T 2>/dev/null; done; done
