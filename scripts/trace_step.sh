#!/bin/bash
# dev helper (GPU box): kernel trace of two bench steps (S=100) for a timeline of the predict tail
export TMPDIR=/tmp
WCX_BENCH_SPINUP_STEPS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_step -o t -- python bench.py --no-cpu-baseline --no-secondary --samples 100 --steps 2 --warmup 1 > gpurun_out/trace_step.log 2>&1
