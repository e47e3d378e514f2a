#!/bin/bash
# laboratory library: NMS timings under several environments (each argument = one environment)
export RPN_HIP_LIB=$PWD/tf_rpn_amd/csrc/librpn_hip_lab.so
for E in "$@"; do for T in ${THRS:-0.7 0.5}; do
  echo "== [$E] thr=$T"; env $E NMS_THR=$T timeout -k 10 300 python scripts/nms_phases.py 2>/dev/null
done; done
