# usage (on the GPU box): bash scripts/prof_rank.sh -- kernel durations of the null-ratio ranking + selection alone
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpk -o rp -- python $R/scripts/prof_rank.py "$@" > /tmp/rpk.log 2>&1 || tail -5 /tmp/rpk.log
tail -1 /tmp/rpk.log
f=$(find /tmp/rpk -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'k_rank' in r['Name'] or 'k_null_ratios' in r['Name']:
        print("%-60s %4s %9.1f"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
