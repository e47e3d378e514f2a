#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-f32ab}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -p no:cacheprovider -x -k "not subprocess" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for R in 1 2; do for L in ab/nobr.so ab/f32new.so; do
  n=$(basename $L .so)
  RPN_HIP_LIB=$PWD/$L timeout 300 python bench.py --precision f32 --steps 5 --warmup 2 --layers --no-cpu-baseline --no-extra-legs > $OUT/bench_$n.json 2> $OUT/layers_$n.txt
  echo "$n: $(python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
done; done
paste <(awk '{print $1, $3}' $OUT/layers_nobr.txt) <(awk '{print $3}' $OUT/layers_f32new.txt) | grep -v amdgpu
