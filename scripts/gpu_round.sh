#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel stats.  Usage: gpurun -- bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $OUT/device.txt 2>&1
lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/device.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 600 python bench.py --steps 10 --warmup 2 --layers > $OUT/bench.json 2> $OUT/bench_layers.txt
echo "bench exit $?" >> $OUT/bench_layers.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.log)
tail -5 $OUT/pytest.log; cat $OUT/smoke.log | tail -3; cat $OUT/bench.json
