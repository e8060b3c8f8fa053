#!/bin/bash
# dev helper: bash scripts/gpu_iter.sh <tag> "<tiles S=100>" "<tiles S=500>" -- parity + bench variants on the GPU box
TAG=$1
T100=${2:-"2,1,4,3,0"}
T500=${3:-"1,1,4,2,0"}
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 2400 -- "python -m pytest tests/test_gpu_newref.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4; for t in $T100; do echo tile \$t; WCX_SCREEN_TILE=\$t python -m pytest tests/test_gpu_newref.py -x -q -k 'k_step_boundaries and (108 or 109)' 2>&1 | tail -1; WCX_SCREEN_TILE=\$t python bench.py --no-cpu-baseline --samples 100 --steps 5 2>&1 | tail -1 > gpurun_out/${TAG}_100_\$t.json; WCX_SCREEN_TILE=\$t python bench.py --no-cpu-baseline --samples 100 --steps 3 --debug-flags 4 2>&1 | tail -1 > gpurun_out/${TAG}_100_\${t}_prof.json; done; for t in $T500; do echo tile \$t; WCX_SCREEN_TILE=\$t python -m pytest tests/test_gpu_newref.py -x -q -k 'k_step_boundaries and (508 or 500)' 2>&1 | tail -1; WCX_SCREEN_TILE=\$t python bench.py --no-cpu-baseline --samples 500 --steps 5 2>&1 | tail -1 > gpurun_out/${TAG}_500_\$t.json; WCX_SCREEN_TILE=\$t python bench.py --no-cpu-baseline --samples 500 --steps 3 --debug-flags 4 2>&1 | tail -1 > gpurun_out/${TAG}_500_\${t}_prof.json; done" > gpurun_out/run_${TAG}.log 2>&1
tail -40 gpurun_out/run_${TAG}.log | grep -v "^\[gpurun\] sending"
python scripts/show_bench.py gpurun_out/${TAG}_*.json
