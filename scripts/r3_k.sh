#!/bin/bash
# round 3, call K: the one-wave-per-SIMD persistent conv (epilogue under the next tile's taps): parity, then A/B (RPN_S16_W4=0: 8 waves)
OUT=gpurun_out/r3k; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider -x -k "split or vgg or model or persistent or invariance or block1" > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for R in 1 2; do for W4 in 0 1; do
  echo "== W4=$W4"; RPN_S16_W4=$W4 RPN_HIP_LIB=$PWD/ab/lab.so timeout -k 10 300 python bench.py --steps 30 --warmup 3 --layers --no-cpu-baseline --no-extra-legs 2> $OUT/layers_$W4.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
done; done
paste <(awk '{print $1, $2, $3}' $OUT/layers_0.txt) <(awk '{print $3}' $OUT/layers_1.txt) | grep -v amdgpu
