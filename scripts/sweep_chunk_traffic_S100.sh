#!/bin/bash
# A-pass screen: HBM-side fetch (rocprofv3 --pmc FETCH_SIZE, x2 per MI355X_MICROARCH.md) and sweep time
# against the candidate chunk size of its launches (WCX_SCREEN_CHUNK_KB; default 8192 at K = 512).
OUT=$GRAFT_REPO_ROOT/gpurun_out/chunk_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kb in 1536 3072 6144 12288 24576; do
  export WCX_SCREEN_CHUNK_KB=$kb
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c$kb -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --samples 100 > $OUT/c$kb.log 2>&1
  timeout 200 python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-verify --samples 100 > $OUT/b$kb.json 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json
out = {}
for kb in (1536, 3072, 6144, 12288, 24576):
    f = glob.glob("gpurun_out/chunk_traffic/c%d/*counter_collection.csv" % kb)
    fetch = 0.0
    n = 0
    if f:
        for row in csv.DictReader(open(f[0])):
            if "k_screen<7" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
                fetch += float(row["Counter_Value"]); n += 1
    d = json.load(open("gpurun_out/chunk_traffic/b%d.json" % kb))
    r = d["roofline"]
    out[kb] = {"launches_per_sweep": n // 2, "fetch_GB_per_sweep_x2": round(fetch * 1024 * 2 / 2 / 1e9, 2),
               "sweep_ms": round(r["kernel_ms"], 2), "step_ms": round(d["ms_per_step"], 2)}
    print(kb, out[kb])
json.dump(out, open("gpurun_out/chunk_traffic/summary.json", "w"), indent=1)
PY
