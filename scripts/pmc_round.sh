#!/bin/bash
# PMC passes over a short bench run (separate passes, kernel-trace only, as MI355X_MICROARCH.md prescribes).
# Usage: gpurun -- bash scripts/pmc_round.sh [tag] [precision]
TAG=${1:-pmc}
PREC=${2:-f16x3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --precision $PREC > $OUT/$name.json 2> $OUT/$name.log
  echo "$name exit $?" >> $OUT/status.txt
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
(cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --precision $PREC > $OUT/stats.json 2> $OUT/stats.log)
cat $OUT/status.txt; ls $OUT/*/ | head -30
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
