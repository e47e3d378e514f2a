#!/bin/bash
# PMC passes over the IoU-map probe (separate passes, kernel-trace only).
TAG=${1:-ioupmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python $GRAFT_REPO_ROOT/scripts/iou_probe.py > $OUT/$name.txt 2> $OUT/$name.log
  echo "$name exit $?" >> $OUT/status.txt
}
run busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR
run write WRITE_SIZE
cat $OUT/status.txt
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A25 "iou_map" $OUT/summary.txt | head -60
