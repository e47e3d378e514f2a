#!/bin/bash
# round 3, call I: batch-invariant K-split of the fused MobileNetV2 f16x3 blocks: parity, then A/B (RPN_MN_KSPLIT=1 = never split)
OUT=gpurun_out/r3i; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x -k "mobilenet or mnv2 or c5 or model or propose or bench or cli or invariance" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for R in 1 2; do for KS in 0 1; do
  for CFG in "--backbone mobilenet_v2 --batch 1" "--config c5" "--backbone mobilenet_v2"; do
  echo "== KSPLIT=$KS $CFG"; RPN_MN_KSPLIT=$KS RPN_HIP_LIB=$PWD/ab/lab.so timeout -k 10 300 python bench.py --steps 30 --warmup 3 --layers --no-cpu-baseline --no-extra-legs $CFG 2> $OUT/layers_${KS}_$(echo $CFG | tr -d ' -').txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done
for CFG in "--backbone mobilenet_v2 --batch 1" "--config c5"; do T=$(echo $CFG | tr -d ' -'); paste <(awk '{print $1, $2, $3}' $OUT/layers_0_$T.txt) <(awk '{print $3}' $OUT/layers_1_$T.txt) | grep -v amdgpu; done
