#!/bin/bash
# f32w: parity tests with the in-tree library, then an A/B of two libraries
TAG=$1; A=$2; B=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_wino.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
bash scripts/r5_wino_ab.sh $TAG $A $B
