#!/bin/bash
# iou map A/B (laboratory library, knobs from the environment) + its parity tests, one GPU-box call
OUT=gpurun_out/${1:-r4iou}; mkdir -p $OUT
export RPN_HIP_LIB=$PWD/tf_rpn_amd/csrc/librpn_hip_lab.so
for H in 1 0; do
  RPN_IOU_HALF=$H timeout -k 10 600 python -m pytest tests/test_gpu_bbox.py -m gpu -q --tb=short -p no:cacheprovider -x -k "iou" > $OUT/pytest_$H.log 2>&1; echo "HALF=$H: $(tail -1 $OUT/pytest_$H.log)"
done
for rep in 1 2 3; do for H in 1 0; do
  echo -n "HALF=$H: "; RPN_IOU_HALF=$H timeout -k 10 300 python scripts/bench_bbox.py 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print({k.split()[0]: round(v['iou_map']['us'],2) for k,v in d.items()})"
done; done
