#!/bin/bash
# usage (on the GPU box): bash scripts/pmc_screen.sh <tag> <samples> [tile]  -- SQ counters of the screen kernel
TAG=${1:-x}; S=${2:-100}; TILE=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WCX_BENCH_SPINUP_STEPS=0
[ -n "$TILE" ] && export WCX_SCREEN_TILE=$TILE
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --samples $S"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -o p -- $CMD > $OUT/pmc3.log 2>&1
python - <<PY
import csv, collections, glob, json
res = {}
for n in (1, 2, 3):
    for f in glob.glob("$OUT/pmc%d/**/*counter_collection.csv" % n, recursive=True):
        agg = collections.defaultdict(float); disp = set()
        for r in csv.DictReader(open(f)):
            if "k_screenILi" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        res.update(agg); res["dispatches_pass%d" % n] = len(disp)
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
