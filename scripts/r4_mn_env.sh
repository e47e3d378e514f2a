#!/bin/bash
# A/B of laboratory knobs on the MobileNetV2 shapes in ONE GPU-box call (devices differ between calls):
# usage: gpurun -- bash scripts/r4_mn_env.sh TAG "RPN_X=1" "RPN_X=2 RPN_Y=0" ...   (each argument = one environment; "" = defaults)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export RPN_HIP_LIB=$PWD/tf_rpn_amd/csrc/librpn_hip_lab.so
for E in "$@"; do
  echo "== parity [$E]" | tee -a $OUT/times.txt
  env $E timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x -k "${PYTEST_K:-mobilenet_v2 and (full_size or batch_invariance or c5)}" > $OUT/pytest_$(echo "$E" | tr ' =' '__').log 2>&1
  tail -1 $OUT/pytest_$(echo "$E" | tr ' =' '__').log | tee -a $OUT/times.txt
done
for rep in 1 2; do for E in "$@"; do
  echo "-- [$E]" | tee -a $OUT/times.txt
  env $E timeout -k 10 300 python scripts/mn_time.py --ops ${MN_ARGS} 2>/dev/null | tee -a $OUT/times.txt
done; done
