#!/bin/bash
# round 6: the whole -m gpu suite with the current library, then the driver's smoke()
OUT=gpurun_out/r6_full; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
