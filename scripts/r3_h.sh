#!/bin/bash
# round 3, call H: full GPU suite on the in-tree build; exact-f32 conv with the register bound (A/B against the build before)
OUT=gpurun_out/r3h; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
for R in 1 2; do
for V in "ab/cache.so 2" "ab/lab.so 2" "ab/lab.so 3" "ab/lab.so 4"; do set -- $V
  echo "== $(basename $1 .so) F32_OCC=$2"; RPN_F32_OCC=$2 RPN_HIP_LIB=$PWD/$1 timeout -k 10 300 python bench.py --precision f32 --steps 8 --warmup 2 --layers --no-cpu-baseline --no-extra-legs 2> $OUT/layers_$(basename $1 .so)_$2.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
paste <(awk '{print $1, $2, $3}' $OUT/layers_cache_2.txt) <(awk '{print $3}' $OUT/layers_lab_2.txt) <(awk '{print $3}' $OUT/layers_lab_3.txt) <(awk '{print $3}' $OUT/layers_lab_4.txt) | grep -v amdgpu
