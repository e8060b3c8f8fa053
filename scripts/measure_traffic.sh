#!/bin/bash
# usage (on the GPU box): bash scripts/measure_traffic.sh  -- rocprofv3 evidence of the default bench
# workloads (S=500 and S=100): kernel-trace stats + SQ / FETCH_SIZE / WRITE_SIZE counters of the screen
# kernel, each counter group in its own pass (no trace domains mixed with --pmc).
ROUND=${WCX_PROF_ROUND:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$ROUND
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WCX_BENCH_SPINUP_STEPS=0
T="timeout 300"   # a counter pass that hangs must not eat the GPU budget
for S in 500 100; do
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-extras --concurrent-passes 0 --samples $S"
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_S$S -o t -- $CMD > $OUT/trace_S$S.log 2>&1
  $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1_S$S -o p -- $CMD > $OUT/pmc1_S$S.log 2>&1
  $T rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2_S$S -o p -- $CMD > $OUT/pmc2_S$S.log 2>&1
  $T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3_S$S -o p -- $CMD > $OUT/pmc3_S$S.log 2>&1
  $T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4_S$S -o p -- $CMD > $OUT/pmc4_S$S.log 2>&1
done
find $OUT -name "*.csv" | wc -l
