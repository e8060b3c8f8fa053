#!/bin/bash
# dev helper (GPU box): rocprofv3 kernel trace of one bench step per sample count; per-dispatch durations
export TMPDIR=/tmp
for s in ${1:-500 100}; do
  rm -rf gpurun_out/trace_sym_$s
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_sym_$s -o t -- python bench.py --no-cpu-baseline --no-secondary --no-verify --samples $s --steps 1 --warmup 1 > gpurun_out/trace_sym_$s.log 2>&1
  f=$(find gpurun_out/trace_sym_$s -name '*kernel_trace.csv' | head -1)
  python - "$f" > gpurun_out/trace_sym_$s.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step only: find the last k_col_sum of the A pass (grid big) -- print all screen-related dispatches after the last 'k_sym_hist'
idx = max(i for i, r in enumerate(rows) if "k_sym_hist" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx - 12: idx + 140]:
    n = r["Kernel_Name"]
    n = n[:60]
    print("%9.1f %8.1f us  grid %8s wg %4s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), n))
PY
  rm -rf gpurun_out/trace_sym_$s
done
