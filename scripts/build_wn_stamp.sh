#!/bin/bash
# Build the -DRPN_STAMP variant of the Winograd kernels next to the product library (tf_rpn_amd/csrc/librpn_hip_wnstamp.so), then restore the product build.
set -e
cd "$(dirname "$0")/../tf_rpn_amd/csrc"
make EXTRA_conv_wino_kernels="-DRPN_STAMP $1" -B _build/conv_wino_kernels.o librpn_hip.so 2>&1 | grep -E "error" || true
cp librpn_hip.so librpn_hip_wnstamp.so
make -B _build/conv_wino_kernels.o librpn_hip.so 2>&1 | grep -E "error" || true
