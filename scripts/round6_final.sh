#!/bin/bash
# Round-6 evidence (one gpurun call per part; each part records its device).  Part A: the default bench line (all legs + CPU
# baseline) and the per-op tables of every config and precision, the box kernels (configs[2]) with kernel stats and FETCH / WRITE
# traffic, and the counter passes of the VGG16 f16x3 run.  Part B: counter passes of the exact-f32 and f32w runs and of MobileNetV2
# (batch 8, configs[4] and configs[0]).
#   gpurun -- bash scripts/round6_final.sh A     |     gpurun -- bash scripts/round6_final.sh B
PART=${1:-A}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6final; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocminfo | grep -E "Marketing Name|gfx9" | head -4 > $OUT/device_$PART.txt 2>&1
if [ "$PART" = "A" ]; then
  timeout -k 10 700 python bench.py --layers > $OUT/bench_default.json 2> $OUT/bench_default_layers.txt
  echo "bench exit $?" >> $OUT/bench_default_layers.txt
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench_k20.json 2> /dev/null
  timeout -k 10 200 python bench.py --config c4 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_c4.json 2> $OUT/bench_c4_layers.txt
  timeout -k 10 200 python bench.py --config c5 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_c5.json 2> $OUT/bench_c5_layers.txt
  timeout -k 10 200 python bench.py --backbone mobilenet_v2 --batch 1 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_c1.json 2> $OUT/bench_c1_layers.txt
  timeout -k 10 200 python bench.py --backbone mobilenet_v2 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_mn8.json 2> $OUT/bench_mn8_layers.txt
  timeout -k 10 200 python bench.py --precision f32 --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_f32.json 2> $OUT/bench_f32_layers.txt
  timeout -k 10 200 python bench.py --precision f32w --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --layers > $OUT/bench_f32w.json 2> $OUT/bench_f32w_layers.txt
  timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --no-extra-legs --sustained-seconds 0 > $OUT/bench_forcedist.json 2> /dev/null
  [ -f tf_rpn_amd/csrc/librpn_hip_wnstamp.so ] && RPN_HIP_LIB=$PWD/tf_rpn_amd/csrc/librpn_hip_wnstamp.so timeout -k 10 200 python scripts/wn_stamp_probe.py 8,125,256,256 8,250,128,128 8,250,64,128 2>&1 | grep -v amdgpu.ids > $OUT/wn_stamps.txt
  [ -x ab/mfma_rate ] && timeout -k 5 60 ./ab/mfma_rate > $OUT/mfma_f32_rate.txt 2>&1
  timeout -k 10 200 python scripts/bench_bbox.py > $OUT/bbox_c3.json 2> $OUT/bbox_c3.err
  timeout -k 10 100 python scripts/bw_probe.py > $OUT/bw_probe.txt 2> /dev/null
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o c3 -- python $GRAFT_REPO_ROOT/scripts/bench_bbox.py > /dev/null 2> $OUT/c3_stats.log
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/c3_write -o c3 -- python $GRAFT_REPO_ROOT/scripts/bench_bbox.py > /dev/null 2> $OUT/c3_write.log
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/c3_fetch -o c3 -- python $GRAFT_REPO_ROOT/scripts/bench_bbox.py > /dev/null 2> $OUT/c3_fetch.log
  cd $GRAFT_REPO_ROOT
  bash scripts/pmc_passes.sh r6final/vgg_f16x3 > $OUT/pmc_vgg.log 2>&1
else
  bash scripts/pmc_passes.sh r6final/vgg_f32 --precision f32 > $OUT/pmc_f32.log 2>&1
  bash scripts/pmc_passes.sh r6final/vgg_f32w --precision f32w > $OUT/pmc_f32w.log 2>&1
  bash scripts/pmc_passes.sh r6final/mn8 --backbone mobilenet_v2 > $OUT/pmc_mn8.log 2>&1
  bash scripts/pmc_passes.sh r6final/c5 --config c5 > $OUT/pmc_c5.log 2>&1
  bash scripts/pmc_passes.sh r6final/c1 --backbone mobilenet_v2 --batch 1 > $OUT/pmc_c1.log 2>&1
fi
cut -c1-300 $OUT/bench_default.json 2>/dev/null; ls $OUT
