OUT=gpurun_out/r6_cin3; mkdir -p $OUT
timeout -k 10 250 python scripts/r6_cin3_diag.py 2>&1 | grep -v amdgpu | grep "shape\|!=" 
