#!/bin/bash
# round 3, call D: NMS parity (cluster barriers reworked), C5 NMS timing + stamps, staggered-start A/B of the persistent conv
OUT=gpurun_out/r3d; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests/test_gpu_bbox.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for L in ab/merge_nolicm.so ab/cluster2.so; do echo "== $(basename $L .so)"; RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_c5_time.py 2>/dev/null; done
echo "== cluster2 CLUSTER=1"; RPN_NMS_CLUSTER=1 RPN_HIP_LIB=$PWD/ab/cluster2.so timeout -k 10 300 python scripts/nms_c5_time.py 2>/dev/null
for T in 0.7; do echo "== stamps model_c5 $T"; RPN_HIP_LIB=$PWD/ab/nmsstamp.so timeout -k 10 300 python scripts/nms_stamp_probe.py model_c5 $T 2>/dev/null | cut -c1-1500; done
echo "== stamps model_c5 CLUSTER=1"; RPN_NMS_CLUSTER=1 RPN_HIP_LIB=$PWD/ab/nmsstamp.so timeout -k 10 300 python scripts/nms_stamp_probe.py model_c5 0.7 2>/dev/null | cut -c1-1500
for R in 1 2; do for SK in 0 128 256 512 1024; do
  echo "== skew $SK"; RPN_S16_SKEW=$SK RPN_HIP_LIB=$PWD/ab/lab.so timeout -k 10 300 python bench.py --steps 30 --warmup 3 --layers --no-cpu-baseline --no-extra-legs 2> $OUT/layers_$SK.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
paste <(awk '{print $1, $3}' $OUT/layers_0.txt) <(awk '{print $3}' $OUT/layers_128.txt) <(awk '{print $3}' $OUT/layers_256.txt) <(awk '{print $3}' $OUT/layers_512.txt) <(awk '{print $3}' $OUT/layers_1024.txt) | grep -v amdgpu
