#!/bin/bash
# round 3, call J: K-split with sc1 hand-off: invariance tests, then A/B (RPN_MN_KSPLIT: 0 automatic, 1 never, 3, 6)
OUT=gpurun_out/r3j; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider -x -k "mobilenet or invariance" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for R in 1 2; do for KS in 0 1 2; do
  for CFG in "--backbone mobilenet_v2 --batch 1" "--config c5"; do
  echo "== KSPLIT=$KS $CFG"; RPN_MN_KSPLIT=$KS RPN_HIP_LIB=$PWD/ab/lab.so timeout -k 10 300 python bench.py --steps 30 --warmup 3 --layers --no-cpu-baseline --no-extra-legs $CFG 2> $OUT/layers_${KS}_$(echo $CFG | tr -d ' -').txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done
for CFG in "--backbone mobilenet_v2 --batch 1" "--config c5"; do T=$(echo $CFG | tr -d ' -'); paste <(awk '{print $1, $2, $3}' $OUT/layers_1_$T.txt) <(awk "{print \$3}" $OUT/layers_2_$T.txt) <(awk '{print $3}' $OUT/layers_0_$T.txt) | grep -v amdgpu; done
