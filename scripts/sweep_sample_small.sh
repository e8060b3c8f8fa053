# usage (on the GPU box): bash scripts/sweep_sample_small.sh -- the 100 kb x 100 step against the sampling factor of the pre-pass
for s in 0 4 8 16; do
  export WCX_SCREEN_SAMPLE=$s
  python bench.py --binsize 100000 --samples 100 --steps 30 --warmup 5 --no-secondary --no-extras --no-cpu-baseline > /tmp/o.json 2>/dev/null
  python - $s <<'PY'
import json,sys
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
r=d["roofline"]
print("sample",sys.argv[1],"step %.3f sweep %.3f pre %.3f refine %.3f appends %d compactions %d fb %s mism %s"%(d["ms_per_step"],r["kernel_ms"],r["pre_ms"],r["refine_ms"],r["appends"],r["compactions"],r["fallback_rows"],d.get("verified",{}).get("mismatches_all_passes")))
PY
done
