#!/bin/bash
# iou map A/B (laboratory library, each argument = one environment) + its parity tests under each, one GPU-box call
OUT=gpurun_out/${1:-r5iou}; mkdir -p $OUT; shift
export RPN_HIP_LIB=$PWD/tf_rpn_amd/csrc/librpn_hip_lab.so
i=0; for E in "$@"; do i=$((i+1))
  env $E timeout -k 10 600 python -m pytest tests/test_gpu_bbox.py -m gpu -q --tb=short -p no:cacheprovider -x -k "iou" > $OUT/pytest_$i.log 2>&1; echo "[$E]: $(tail -1 $OUT/pytest_$i.log)"
done
for rep in 1 2 3; do for E in "$@"; do
  echo -n "[$E]: "; env $E timeout -k 10 300 python scripts/iou_probe.py 2>/dev/null
done; done
