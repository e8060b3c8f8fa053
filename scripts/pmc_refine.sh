#!/bin/bash
# dev helper (GPU box): cache / issue counters of k_refine at 15 kb (bench_refine.py).  usage: pmc_refine.sh S CHUNK
S=${1:-500}; CH=${2:-100000}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_refine_S${S}_c${CH}
rm -rf $OUT; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/scripts/bench_refine.py $S $CH"
cd /tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/p3 -o p -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/p4 -o p -- $CMD > $OUT/p4.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/p5 -o p -- $CMD > $OUT/p5.log 2>&1
python - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in tot:
    if "refine" in k or "screen<" in k:
        print(k, json.dumps({c: v for c, v in sorted(tot[k].items())}))
PY
