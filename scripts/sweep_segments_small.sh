for s in 0 2 3 4 6 8; do
  if [ $s = 0 ]; then unset WCX_SCREEN_SEGMENTS; else export WCX_SCREEN_SEGMENTS=$s; fi
  python bench.py --binsize 100000 --samples 100 --steps 30 --warmup 5 --no-secondary --no-extras --no-cpu-baseline > /tmp/o.json 2>/dev/null
  python - $s <<'PY'
import json,sys
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
r=d["roofline"]
print("seg",sys.argv[1],"step %.3f sweep %.3f refine %.3f prep %.3f nr %.3f fb %s mism %s"%(d["ms_per_step"],r["kernel_ms"],r["refine_ms"],r["prep_ms"],r["null_ratios_ms"],r["fallback_rows"],d.get("verified",{}).get("mismatches_all_passes")))
PY
done
