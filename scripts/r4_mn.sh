#!/bin/bash
# Round 4 MobileNetV2 loop: parity tests of the MobileNetV2 path, then step times (product lib, optionally a second lib to A/B).
# usage: gpurun -- bash scripts/r4_mn.sh TAG [other_lib.so]
TAG=${1:-r4mn}; OTHER=$2
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x -k "mobilenet or mnv2 or c5 or C5 or ir_block or ksplit" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for rep in 1 2; do
  echo "-- product lib" | tee -a $OUT/times.txt
  timeout -k 10 300 python scripts/mn_time.py --ops 2>/dev/null | tee -a $OUT/times.txt
  if [ -n "$OTHER" ]; then echo "-- $OTHER" | tee -a $OUT/times.txt; RPN_HIP_LIB=$PWD/$OTHER timeout -k 10 300 python scripts/mn_time.py --ops 2>/dev/null | tee -a $OUT/times.txt; fi
done
