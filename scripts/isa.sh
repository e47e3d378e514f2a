#!/bin/bash
# device ISA of one csrc file: scripts/isa.sh conv_wino_kernels [extra flags]  ->  /tmp/isa/<name>.s
N=$1; shift
mkdir -p /tmp/isa
cd /root/repo/tf_rpn_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include --cuda-device-only -S $N.hip -o /tmp/isa/$N.s "$@" 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^\|generated" | head -30
