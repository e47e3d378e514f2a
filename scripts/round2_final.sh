#!/bin/bash
# Round-2 evidence call (one device for everything): default bench line (all legs + CPU baseline) and per-op table,
# rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only) of the same VGG16
# command, of the box kernels (configs[2]) and of MobileNetV2 at batch 8.
TAG=${1:-r2f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocminfo | grep -E "gfx" | head -2 > $OUT/device.txt 2>&1
timeout -k 10 400 python bench.py --layers > $OUT/bench_default.json 2> $OUT/bench_default_layers.txt
echo "bench exit $?" >> $OUT/bench_default_layers.txt
timeout -k 10 200 python bench.py --config c4 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_c4.json 2> $OUT/bench_c4_layers.txt
timeout -k 10 200 python bench.py --config c5 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_c5.json 2> $OUT/bench_c5_layers.txt
timeout -k 10 200 python bench.py --backbone mobilenet_v2 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_mn8.json 2> $OUT/bench_mn8_layers.txt
timeout -k 10 200 python scripts/bench_bbox.py > $OUT/bbox_c3.json 2> $OUT/bbox_c3.err
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.json 2> $OUT/stats.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- $BENCH > /dev/null 2> $OUT/fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- $BENCH > /dev/null 2> $OUT/write.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o c3 -- python $GRAFT_REPO_ROOT/scripts/bench_bbox.py > /dev/null 2> $OUT/c3_stats.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/c3_write -o c3 -- python $GRAFT_REPO_ROOT/scripts/bench_bbox.py > /dev/null 2> $OUT/c3_write.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/c3_fetch -o c3 -- python $GRAFT_REPO_ROOT/scripts/bench_bbox.py > /dev/null 2> $OUT/c3_fetch.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mn8_stats -o mn8 -- $BENCH --backbone mobilenet_v2 > /dev/null 2> $OUT/mn8_stats.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/mn8_write -o mn8 -- $BENCH --backbone mobilenet_v2 > /dev/null 2> $OUT/mn8_write.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/mn8_fetch -o mn8 -- $BENCH --backbone mobilenet_v2 > /dev/null 2> $OUT/mn8_fetch.log
cd $GRAFT_REPO_ROOT
cat $OUT/bench_default.json | cut -c1-400
ls $OUT
