#!/bin/bash
# round 3: rpn_conv as a K tree -- parity, then the configs with the tree on (default) and off
OUT=gpurun_out/ktree; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for R in 1 2; do for K in 1 0; do
  echo "== RPN_KSPLIT=$K"
  for ARGS in "--backbone mobilenet_v2 --batch 1" "--config c5" "--backbone mobilenet_v2" "" "--config c4 --steps 5"; do
    RPN_KSPLIT=$K timeout 300 python bench.py $ARGS --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ARGS', d['value'], d['ms_per_step'])"
  done
done; done
