#!/bin/bash
# wide Winograd form, per-wave stamps of three builds: as shipped | every filter request hits slice 0 (L2-hot filters) | no staging work
OUT=gpurun_out/r6_wn_decomp; mkdir -p $OUT
for L in tf_rpn_amd/csrc/librpn_hip_wnstamp.so ab/librpn_wnstamp_u0.so ab/librpn_wnstamp_nostage.so; do
  echo "== $L"
  RPN_HIP_LIB=$PWD/$L timeout -k 10 200 python scripts/wn_stamp_probe.py 8,125,256,256 8,250,128,128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps.txt
done
