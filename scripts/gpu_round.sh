#!/bin/bash
# One GPU-box round: parity tests, smoke, bench (all precisions), rocprof kernel stats.
# Usage: gpurun -- bash scripts/gpu_round.sh [tag] [pytest-args]
TAG=${1:-r01}
PYTEST_ARGS=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -E "gfx" | head -2 > $OUT/device.txt 2>&1
lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/device.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s $PYTEST_ARGS > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
for P in f32 bf16x3 f16x3; do
  timeout 600 python bench.py --steps 20 --warmup 3 --layers --precision $P --no-cpu-baseline > $OUT/bench_$P.json 2> $OUT/bench_layers_$P.txt
  echo "bench exit $?" >> $OUT/bench_layers_$P.txt
done
timeout 300 python scripts/nms_probe.py > $OUT/nms_probe.txt 2>&1
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; tail -2 $OUT/smoke.log; cat $OUT/bench_*.json; cat $OUT/nms_probe.txt
