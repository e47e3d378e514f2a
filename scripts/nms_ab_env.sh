#!/bin/bash
# NMS timings of one build under several settings of an environment variable.
# Usage: gpurun -- bash scripts/nms_ab_env.sh lib.so VAR v1 v2 ...
LIB=$1; VAR=$2; shift 2
for R in 1 2; do for V in "$@"; do
  echo "== $VAR=$V"; env RPN_HIP_LIB=$PWD/$LIB $VAR=$V python scripts/nms_phases.py 2>/dev/null
done; done
