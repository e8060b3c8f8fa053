# usage (on the GPU box): bash scripts/trace_prep.sh -- every kernel of the last bench step in launch order
# (passes one after another) with its duration: what the preparation of a pass consists of
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/trace_prep
rm -rf $OUT /tmp/tp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WCX_BENCH_SPINUP_STEPS=0
rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -o t -- python $R/bench.py --steps 2 --warmup 1 --no-secondary --no-extras --no-cpu-baseline --no-verify --concurrent-passes 0 > $OUT/bench.log 2>&1 || tail -5 $OUT/bench.log
python - <<'PY'
import csv, glob, os
f = glob.glob("/tmp/tp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
marks = [i for i, r in enumerate(rows) if "k_col_sum" in r[2]]
a = marks[-3]            # A pass of the last step (three k_col_sum per step)
out = open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/trace_prep/last_step.txt", "w")
t0 = rows[a][0]
for s, e, n in rows[a:]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    out.write("{:9.3f} ms  {:8.1f} us  {}\n".format((s - t0) / 1e6, (e - s) / 1e3, n))
PY
