#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-headab}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x -k "not bench_prints" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
for V in "RPN_KSPLIT=0" "RPN_KSPLIT=1"; do
  for C in "--config c5" "--backbone mobilenet_v2 --batch 1" "--backbone vgg16 --batch 1"; do
    echo "== $V $C" >> $OUT/res.txt
    env $V timeout -k 10 200 python bench.py $C --no-cpu-baseline --no-extra-legs --layers 2> $OUT/layers.tmp | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $OUT/res.txt
    grep -E "rpn_head|rpn_conv" $OUT/layers.tmp >> $OUT/res.txt
  done
done
cat $OUT/res.txt
