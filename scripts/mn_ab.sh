#!/bin/bash
TAG=${1:-mnab}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x -k "mobilenet or mnv2 or c5 or C5" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
for R in 1 2; do
  for V in "RPN_MN_STEM8=0" "RPN_MN_STEM8=1"; do
    for C in "--backbone mobilenet_v2" "--config c5" "--backbone mobilenet_v2 --batch 1"; do
      echo -n "$V $C: " >> $OUT/res.txt
      env $V timeout -k 10 200 python bench.py $C --no-cpu-baseline --no-extra-legs --layers 2> $OUT/layers.tmp | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $OUT/res.txt
      head -3 $OUT/layers.tmp | tail -2 >> $OUT/res.txt
    done
  done
done
cat $OUT/res.txt
