#!/bin/bash
# rocprofv3 counter passes over a short bench.py run: separate passes, kernel-trace only (MI355X_MICROARCH.md), + a
# --stats pass.  Usage (inside one gpurun call): bash scripts/pmc_passes.sh <out dir under gpurun_out> <bench.py args...>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --sustained-seconds 0 $*"
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.json 2> $OUT/$name.log
  echo "$name exit $?" >> $OUT/status.txt
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES
run insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run fetch FETCH_SIZE
run write WRITE_SIZE
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.json 2> $OUT/stats.log
cat $OUT/status.txt
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; head -c 6000 $OUT/summary.txt
