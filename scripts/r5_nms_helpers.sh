#!/bin/bash
# round 5: NMS ahead mode (helper workgroups): parity tests in the automatic mode and with packages forced at every chunk, then timings
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_bbox.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x -k "nms or propose or decode" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for E in "$@"; do for T in 0.7 0.5; do
  echo "== [$E] thr=$T"; env $E NMS_THR=$T timeout -k 10 300 python scripts/nms_phases.py 2>/dev/null
done; done
