#!/bin/bash
# round 3, call G: NMS with the LDS score cache: parity + A/B + stamps; PMC passes of the exact-f32 VGG16 run
OUT=gpurun_out/r3g; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests/test_gpu_bbox.py tests/test_gpu_pipeline.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for L in ab/scan.so ab/cache.so; do for T in 0.7 0.5; do
  echo "== $(basename $L .so)"; NMS_THR=$T RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_phases.py 2>/dev/null
done; done
for L in ab/scan.so ab/cache.so; do echo "== $(basename $L .so)"; RPN_HIP_LIB=$PWD/$L timeout -k 10 300 python scripts/nms_c5_time.py 2>/dev/null; done
for K in perm; do for T in 0.7 0.5; do echo "== stamps $K $T"; RPN_HIP_LIB=$PWD/ab/nmsstamp.so timeout -k 10 300 python scripts/nms_stamp_probe.py $K $T 2>/dev/null | cut -c1-900; done; done
bash scripts/pmc_passes.sh r3_f32_pmc --precision f32 > $OUT/f32_pmc.log 2>&1
python scripts/pmc_table.py gpurun_out/r3_f32_pmc/summary.txt
