mkdir -p gpurun_out/shard
for S in 500 100; do
for ratio in 0 2 4 8; do
  WCX_NR_DIRECT_RATIO=$ratio timeout 400 python scripts/bench_shard.py $S > gpurun_out/shard/S${S}_r$ratio.json 2> gpurun_out/shard/S${S}_r$ratio.err
  python - <<PY
import json
d=json.load(open("gpurun_out/shard/S${S}_r$ratio.json"))
for n in (2,4,8):
    v=d["N%d_segauto"%n]
    print("S=$S ratio=$ratio N=%d rows=%d wall=%.2f topk=%.2f nr=%.2f"%(n,v["rows"],v["shard_wall_ms"],v["topk_ms"],v["null_ratios_ms"]))
PY
done; done
