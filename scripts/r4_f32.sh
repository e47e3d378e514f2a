#!/bin/bash
# exact-float32 path: parity tests, then A/B of laboratory knobs (each argument = one environment) in one GPU-box call
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x -k "vgg16 or conv2d or pool or f32 or pipeline or propose" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
export RPN_HIP_LIB=$PWD/tf_rpn_amd/csrc/librpn_hip_lab.so
for rep in 1 2; do for E in "$@"; do
  env $E timeout -k 10 300 python bench.py --precision f32 --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench.json 2> $OUT/layers_$(echo "$E" | tr ' =' '__').txt
  echo "[$E] $(python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['checks']['ok'])")"
done; done
