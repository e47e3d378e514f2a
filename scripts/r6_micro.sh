#!/bin/bash
# round 6: the micro benchmarks behind DESIGN 4.8 / NOTES, outputs kept under profiles/ (build: the header line of each scripts/micro/*.hip)
OUT=gpurun_out/r6_micro; mkdir -p $OUT
for m in store_hazard mfma_hazard store_rate store_overlap; do
  [ -x ab/$m ] && timeout -k 10 200 ./ab/$m > $OUT/$m.txt 2>&1; echo "$m rc $?"
done
