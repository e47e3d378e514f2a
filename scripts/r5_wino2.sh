#!/bin/bash
# f32w integrated: tests, then the bench step under f32 / f32w with per-layer tables
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_wino.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for P in f32 f32w; do
  timeout -k 10 300 python bench.py --precision $P --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_$P.json 2> $OUT/layers_$P.txt
  echo "[$P] $(python -c "import json;d=json.load(open('$OUT/bench_$P.json'));print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['checks']['ok'])")"
done
paste <(awk '{print $1, $3}' $OUT/layers_f32.txt) <(awk '{print $3}' $OUT/layers_f32w.txt) | grep -v amdgpu
