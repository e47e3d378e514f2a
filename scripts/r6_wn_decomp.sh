#!/bin/bash
# wide Winograd form, per-wave stamps of timing-experiment builds (scripts/build_wn_stamp.sh "<defines>"; wrong results by design):
# as shipped | every filter request hits slice 0 | no staging work | the GEMM at the f16x3 rate + hi / lo split in the staging | both
OUT=gpurun_out/r6_wn_decomp; mkdir -p $OUT; rm -f $OUT/stamps.txt
for L in tf_rpn_amd/csrc/librpn_hip_wnstamp.so ab/librpn_wnstamp_u0.so ab/librpn_wnstamp_nostage.so ab/librpn_wnstamp_f16rate.so ab/librpn_wnstamp_f16rate_nostage.so; do
  [ -f $L ] || continue
  echo "== $L" | tee -a $OUT/stamps.txt
  RPN_HIP_LIB=$PWD/$L timeout -k 10 200 python scripts/wn_stamp_probe.py 8,125,256,256 8,250,128,128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps.txt
done
