#!/bin/bash
# dev helper: bash scripts/gpu_iter.sh <tag> [extra bench variants...] -- runs parity + benches on the GPU box
TAG=$1
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 1500 -- "python -m pytest tests/test_gpu_newref.py -x -q 2>&1 | tail -3; python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/${TAG}.json; python bench.py --no-cpu-baseline --samples 500 2>&1 | tail -1 | tee gpurun_out/${TAG}_500.json; python bench.py --no-cpu-baseline --debug-flags 4 2>&1 | tail -1 | tee gpurun_out/${TAG}_prof.json" > gpurun_out/run_${TAG}.log 2>&1
grep -h "passed\|failed\|rror" gpurun_out/run_${TAG}.log | head
python - <<PY
import json
for f in ("gpurun_out/${TAG}.json","gpurun_out/${TAG}_500.json"):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1]); r=d["roofline"]
        print(f, "%.2f"%d["ms_per_step"], {k:(round(r[k],3) if isinstance(r[k],float) else r[k]) for k in ("kernel_ms","frac","prep_ms","refine_ms","compactions","appends","null_ratios_ms")})
    except Exception as e: print(f, "ERR", e)
try:
    d=json.loads(open("gpurun_out/${TAG}_prof.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("prof run: kernel_ms", r["kernel_ms"])
    pc=r.get("phase_cycles",[0]); tot=sum(pc)
    names=["issue loads+MFMA","wait loads+ds_write","barrier","MFMA done+sign+OR","appends","maintenance"]
    for n,c in zip(names,pc): print("%-24s %6.1f%%  %.3g"%(n,100*c/tot,c))
except Exception as e: print("prof ERR", e)
PY
