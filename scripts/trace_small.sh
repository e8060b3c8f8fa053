# usage (on the GPU box): bash scripts/trace_small.sh -- kernel timeline of the 100 kb x 100 samples step
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/trace_small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ts -o t -- python $R/bench.py --binsize 100000 --samples 100 --steps 4 --warmup 2 --no-secondary --no-extras --no-cpu-baseline --no-verify --concurrent-passes ${CP:-1} > $OUT/bench.log 2>&1 || tail -5 $OUT/bench.log
tail -1 $OUT/bench.log | cut -c1-300
cp $(find /tmp/ts -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
cp $(find /tmp/ts -name "*memory_copy_trace.csv" | head -1) $OUT/memcpy_trace.csv 2>/dev/null || true
