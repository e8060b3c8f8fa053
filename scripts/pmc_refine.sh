#!/bin/bash
# usage (on the GPU box): bash scripts/pmc_refine.sh -- L2 hit / miss and fabric fetch of the refine
# kernels (one-wave-per-row vs sliced), counters in their own passes.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_refine
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WCX_BENCH_SPINUP_STEPS=0
T="timeout 300"
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-extras --concurrent-passes 0"
for SL in 0 1; do
  export WCX_REFINE_SLICED=$SL
  $T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_$SL -o p -- $CMD > $OUT/f_$SL.log 2>&1
  $T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/h_$SL -o p -- $CMD > $OUT/h_$SL.log 2>&1
  $T rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/t_$SL -o p -- $CMD > $OUT/t_$SL.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_refine"
for d in sorted(glob.glob(out + "/*_[01]")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-28:]
            if "refine" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(os.path.basename(d), k, {c: (v, n[(k, c)]) for c, v in acc[k].items()})
PY
