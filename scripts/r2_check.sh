#!/bin/bash
# GPU call: parity tests, smoke, default bench line (+ MobileNetV2 / C5 lines).
TAG=${1:-r2b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocminfo | grep -E "gfx" | head -2 > $OUT/device.txt 2>&1
timeout -k 10 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
tail -2 $OUT/smoke.log
timeout -k 10 300 python bench.py --layers > $OUT/bench_default.json 2> $OUT/bench_default_layers.txt
echo "bench exit $?" >> $OUT/bench_default_layers.txt
cat $OUT/bench_default.json
timeout -k 10 200 python bench.py --backbone mobilenet_v2 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_mn8.json 2> $OUT/bench_mn8_layers.txt
timeout -k 10 200 python bench.py --config c5 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_c5.json 2> $OUT/bench_c5_layers.txt
cat $OUT/bench_mn8.json $OUT/bench_c5.json
