OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_nr
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for r in 4 1; do
  WCX_NR_DIRECT_RATIO=$r timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/r$r -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify > $OUT/r$r.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for r in (4,1):
    f=glob.glob("gpurun_out/pmc_nr/r%d/*counter_collection.csv"%r)
    if not f: print("no file", r); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for row in csv.DictReader(open(f[0])):
        n=row["Kernel_Name"]
        if "k_null_ratios" not in n: continue
        key="hi" if "null_ratios_hi" in n else ("dummy" if "dummy" in n else "rank")
        agg[key][row["Counter_Name"]]+=float(row["Counter_Value"])
        if row["Counter_Name"]=="SQ_WAVE_CYCLES": cnt[key]+=1
    for k,v in agg.items():
        print("ratio",r,k,"launches",cnt[k],{a:"%.3g"%b for a,b in v.items()})
PY
