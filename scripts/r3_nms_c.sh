#!/bin/bash
# round 3: cluster-mode NMS -- parity (in-tree build), then timings of builds side by side and phase stamps at C5
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests/test_gpu_bbox.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for L in "$@"; do echo "== $(basename $L .so)"; RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_c5_time.py 2>/dev/null; done
for L in "$@"; do echo "== $(basename $L .so) CLUSTER=1"; RPN_NMS_CLUSTER=1 RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_c5_time.py 2>/dev/null; done
for T in 0.7; do echo "== stamps model_c5 $T"; RPN_HIP_LIB=$PWD/ab/nmsstamp.so timeout -k 10 300 python scripts/nms_stamp_probe.py model_c5 $T 2>/dev/null | cut -c1-1500; done
