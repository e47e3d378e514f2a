#!/bin/bash
# round 6 f32w: parity tests with the new library, then A/B of two libraries (alternated), per-layer tables, then the stamp probe
TAG=$1; A=$2; B=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_wino.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
bash scripts/r5_wino_ab.sh $TAG $A $B
if [ -f tf_rpn_amd/csrc/librpn_hip_wnstamp.so ]; then RPN_HIP_LIB=$PWD/tf_rpn_amd/csrc/librpn_hip_wnstamp.so timeout -k 10 200 python scripts/wn_stamp_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps.txt; fi
