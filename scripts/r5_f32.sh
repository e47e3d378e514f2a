#!/bin/bash
# round 5: exact-float32 path after POOL became a template parameter -- parity tests, then an A/B of TWO LIBRARIES (never an
# environment knob on one binary) in one GPU-box call.  Usage: gpurun -- bash scripts/r5_f32.sh tag libA.so libB.so
TAG=$1; A=$2; B=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x -k "vgg16 or conv2d or pool or f32 or pipeline or propose" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for rep in 1 2 3; do for L in $A $B; do
  n=$(basename $L .so)
  RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python bench.py --precision f32 --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_$n.json 2> $OUT/layers_$n.txt
  echo "[$n] $(python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['checks']['ok'])")"
done; done
paste <(awk '{print $1, $3}' $OUT/layers_$(basename $A .so).txt) <(awk '{print $3}' $OUT/layers_$(basename $B .so).txt) | grep -v amdgpu
