#!/bin/bash
# A/B inside one GPU call: VGG16 block 1 as one launch (default) vs layer by layer (RPN_B1_FUSE=0).
TAG=${1:-b1ab}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x -k "vgg or split or f16 or propose or pipeline" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for R in 1 2; do
  RPN_B1_FUSE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_unfused_$R.json 2> $OUT/bench_unfused_layers_$R.txt
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_fused_$R.json 2> $OUT/bench_fused_layers_$R.txt
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/%s/bench_*.json' % os.environ.get('TAG','b1ab'))):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
head -4 $OUT/bench_fused_layers_1.txt $OUT/bench_unfused_layers_1.txt
