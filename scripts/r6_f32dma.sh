#!/bin/bash
# exact f32: the DMA kernel on every tile width (laboratory knob RPN_F32_DMA=2) against the default rule (128-wide tiles only), one call
OUT=gpurun_out/r6_f32dma; mkdir -p $OUT
for rep in 1 2; do for K in 1 2; do
  RPN_HIP_LIB=$PWD/tf_rpn_amd/csrc/librpn_hip_lab.so RPN_F32_DMA=$K timeout -k 10 300 python bench.py --precision f32 --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_$K.json 2> $OUT/layers_$K.txt
  echo "[dma=$K] $(python -c "import json;d=json.load(open('$OUT/bench_$K.json'));print(d['value'], d['ms_per_step'], d['checks']['ok'])")"
done; done
paste <(awk '{print $1, $3}' $OUT/layers_1.txt) <(awk '{print $3}' $OUT/layers_2.txt) | grep -v amdgpu
