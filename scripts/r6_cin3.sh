#!/bin/bash
# first layer of the float32 graphs on the matrix pipe (conv_cin3_f32_mfma_kernel): parity tests, then f32w and f32 against HEAD's library
OUT=gpurun_out/r6_cin3; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider -x  > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
grep -q passed $OUT/pytest.log || exit 1
for P in f32w f32; do for rep in 1 2; do for L in tf_rpn_amd/csrc/librpn_hip.so ab/librpn_head.so; do
  n=$(basename $L .so)
  RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python bench.py --precision $P --steps 20 --warmup 3 --layers --no-cpu-baseline --no-extra-legs > $OUT/bench_${P}_$n.json 2> $OUT/layers_${P}_$n.txt
  echo "$P $n: $(python -c "import json;d=json.load(open('$OUT/bench_${P}_$n.json'));print(d['value'], d['ms_per_step'], d['checks']['ok'])")"
done; done; done
grep -h "conv_cin3\|block1_conv1" $OUT/layers_*.txt
