#!/bin/bash
# Symmetric sweep of the A pass (15 kb x 500): HBM-side fetch (rocprofv3 --pmc FETCH_SIZE, x2 per
# MI355X_MICROARCH.md) and sweep time against the size of the streamed chunks (WCX_SYM_CHUNK_KB; the
# workgroups running at any time stream the same one or two chunks: a chunk that fits an XCD's 4 MB L2 is
# fetched once per XCD, a larger one cycles through it).
OUT=$GRAFT_REPO_ROOT/gpurun_out/sym_chunk_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kb in 1536 2048 3072 4096 8192; do
  export WCX_SYM_CHUNK_KB=$kb
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c$kb -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-verify --no-extras --concurrent-passes 0 > $OUT/c$kb.log 2>&1
  timeout 200 python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-verify --no-extras > $OUT/b$kb.json 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json
out = {}
for kb in (1536, 2048, 3072, 4096, 8192):
    f = glob.glob("gpurun_out/sym_chunk_traffic/c%d/**/*counter_collection.csv" % kb, recursive=True)
    fetch = 0.0
    if f:
        for row in csv.DictReader(open(f[0])):
            if ("k_screen_sym<" in row["Kernel_Name"] or "k_screen_count<" in row["Kernel_Name"]) and row["Counter_Name"] == "FETCH_SIZE":
                fetch += float(row["Counter_Value"])
    try:
        d = json.loads(open("gpurun_out/sym_chunk_traffic/b%d.json" % kb).read().strip().splitlines()[-1])
        r = d["roofline"]
        out[kb] = {"fetch_GB_per_sweep_x2": round(fetch * 1024 * 2 / 2 / 1e9, 2), "sweep_ms": round(r["kernel_ms"], 2),
                   "step_ms": round(d["ms_per_step"], 2), "fallback_rows": r["fallback_rows"]}
    except Exception as e:
        out[kb] = {"error": repr(e)}
    print(kb, out[kb])
json.dump(out, open("gpurun_out/sym_chunk_traffic/summary.json", "w"), indent=1)
PY
find $OUT -name "*.csv" -size +5M -delete
