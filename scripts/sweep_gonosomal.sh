#!/bin/bash
# F / M pass search time of the bench step against the chunk size of their launches
# (WCX_SCREEN_CHUNK_KB_SMALL) and the number of candidate segments (WCX_SCREEN_SEGMENTS_SMALL)
mkdir -p gpurun_out/gono
for cfg in "24576 4" "49152 4" "98304 4" "131072 4" "49152 3" "49152 6"; do
  set -- $cfg
  WCX_SCREEN_CHUNK_KB_SMALL=$1 WCX_SCREEN_SEGMENTS_SMALL=$2 timeout 300 python bench.py --steps 6 --warmup 2 --no-secondary --no-verify --no-cpu-baseline > gpurun_out/gono/c$1_s$2.json 2> gpurun_out/gono/c$1_s$2.err
  python - <<PY
import json
d=json.load(open("gpurun_out/gono/c$1_s$2.json"))
g=d["roofline"]["gonosomal_passes"]
print("chunk $1 seg $2 step %.2f A_screen %.2f F topk %.2f screen %.2f | M topk %.2f screen %.2f"%(d["ms_per_step"], d["roofline"]["kernel_ms"], g["F"]["topk_ms"],g["F"]["screen_ms"],g["M"]["topk_ms"],g["M"]["screen_ms"]))
PY
done
