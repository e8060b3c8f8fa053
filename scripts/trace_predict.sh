# usage (on the GPU box): bash scripts/trace_predict.sh [bench args] -- kernel timeline of a step (predict tail analysis)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/trace_predict
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -o t -- python $R/bench.py --steps 3 --warmup 1 --no-secondary --no-extras --no-cpu-baseline --no-verify "$@" > $OUT/bench.log 2>&1 || tail -5 $OUT/bench.log
cp $(find /tmp/tp -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
