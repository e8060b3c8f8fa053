#!/bin/bash
# dev helper (GPU box): rocprofv3 kernel stats of two default bench steps (S=500)
export TMPDIR=/tmp
rm -rf gpurun_out/stats_step
WCX_BENCH_SPINUP_STEPS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_step -o t -- python bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1 > gpurun_out/stats_step.log 2>&1
