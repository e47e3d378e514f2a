#!/bin/bash
# round 6: full -m gpu suite + a headline A/B of the round-5 library against the current one
OUT=gpurun_out/r6_check; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
for rep in 1 2; do for L in ab/librpn_r5.so tf_rpn_amd/csrc/librpn_hip.so; do
  n=$(basename $L .so)
  RPN_HIP_LIB=$PWD/$L timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --sustained-seconds 0 > $OUT/bench_$n.json 2> $OUT/err_$n.txt
  echo "$n: $(python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done; done
