#!/bin/bash
# Round-2 GPU call A: parity tests (incl. the C4 / C5 configs), smoke, the full default bench line, the other
# configs' bench lines, and rocprofv3 stats + FETCH/WRITE passes for the box kernels (configs[2]) and MobileNetV2.
TAG=${1:-r2a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocminfo | grep -E "gfx" | head -2 > $OUT/device.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s -x > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py --layers > $OUT/bench_default.json 2> $OUT/bench_default_layers.txt
echo "bench exit $?" >> $OUT/bench_default_layers.txt
timeout 600 python bench.py --config c4 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_c4.json 2> $OUT/bench_c4_layers.txt
timeout 600 python bench.py --config c5 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_c5.json 2> $OUT/bench_c5_layers.txt
timeout 600 python bench.py --backbone mobilenet_v2 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_mn8.json 2> $OUT/bench_mn8_layers.txt
timeout 300 python scripts/bench_bbox.py > $OUT/bbox_c3.json 2> $OUT/bbox_c3.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o c3 -- python $GRAFT_REPO_ROOT/scripts/bench_bbox.py > $OUT/c3_stats.json 2> $OUT/c3_stats.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/c3_write -o c3 -- python $GRAFT_REPO_ROOT/scripts/bench_bbox.py > /dev/null 2> $OUT/c3_write.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/c3_fetch -o c3 -- python $GRAFT_REPO_ROOT/scripts/bench_bbox.py > /dev/null 2> $OUT/c3_fetch.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mn8_stats -o mn8 -- python $GRAFT_REPO_ROOT/bench.py --backbone mobilenet_v2 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/mn8_stats.json 2> $OUT/mn8_stats.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/mn8_write -o mn8 -- python $GRAFT_REPO_ROOT/bench.py --backbone mobilenet_v2 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $OUT/mn8_write.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/mn8_fetch -o mn8 -- python $GRAFT_REPO_ROOT/bench.py --backbone mobilenet_v2 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $OUT/mn8_fetch.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -o c5 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/c5_stats.json 2> $OUT/c5_stats.log
cd $GRAFT_REPO_ROOT
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; tail -2 $OUT/smoke.log; cat $OUT/bench_default.json; cat $OUT/bench_c4.json $OUT/bench_c5.json $OUT/bench_mn8.json
