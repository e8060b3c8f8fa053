#!/bin/bash
# dev helper (GPU box): cache / issue counters of the batched normalize kernel (config 5)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_predict
rm -rf $OUT; mkdir -p $OUT
CMD="python scripts/bench_predict.py --cbs-samples 0"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/p3 -o p -- $CMD > $OUT/p3.log 2>&1
