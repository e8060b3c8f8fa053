#!/bin/bash
# dev helper (GPU box): per-kernel times of the CBS of 1 and 16 synthetic 15 kb samples
export TMPDIR=/tmp
for n in 1 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cbs$n -o cbs -- python scripts/prof_cbs.py $n > gpurun_out/prof_cbs$n.log 2>&1
done
