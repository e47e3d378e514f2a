#!/bin/bash
# round 3, call F: MobileNetV2 blocks 1-3 on the 16-bit MFMA (ir_block_hrx3_kernel): parity, then A/B against the f32 form
OUT=gpurun_out/r3f; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x -k "mobilenet or mnv2 or c5 or model or propose or bench or cli" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for R in 1 2; do for HR in 1 0; do
  for CFG in "--backbone mobilenet_v2" "--config c5" "--backbone mobilenet_v2 --batch 1"; do
  echo "== HRX3=$HR $CFG"; RPN_MN_HRX3=$HR RPN_HIP_LIB=$PWD/ab/lab.so timeout -k 10 300 python bench.py --steps 30 --warmup 3 --layers --no-cpu-baseline --no-extra-legs $CFG 2> $OUT/layers_${HR}_$(echo $CFG | tr -d ' -').txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done
for CFG in "--backbone mobilenet_v2" "--config c5"; do T=$(echo $CFG | tr -d ' -'); paste <(awk '{print $1, $2, $3}' $OUT/layers_1_$T.txt) <(awk '{print $3}' $OUT/layers_0_$T.txt) | grep -v amdgpu; done
