#!/bin/bash
# -DRPN_STAMP variant of the fused MobileNetV2 block kernel next to the product library (librpn_hip_irstamp.so).
set -e
cd "$(dirname "$0")/../tf_rpn_amd/csrc"
make EXTRA_mnv2_block_kernels="-DRPN_STAMP -mllvm -amdgpu-mfma-vgpr-form" -B _build/mnv2_block_kernels.o librpn_hip.so 2>&1 | grep -E "error|warning" || true
cp librpn_hip.so librpn_hip_irstamp.so
make -B _build/mnv2_block_kernels.o librpn_hip.so 2>&1 | grep -E "error|warning" || true
