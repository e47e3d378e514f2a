#!/bin/bash
# round 6: the whole -m gpu suite with the current library
OUT=gpurun_out/r6_full; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
