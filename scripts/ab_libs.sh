#!/bin/bash
# Same-box comparison of builds of the library: wisecondorx_amd/libwcx_hip_<tag>.so for every tag given
# (the tracked build first and last).  Prints step / sweep / refine / verification per build.
mkdir -p gpurun_out
B="python bench.py --steps 6 --warmup 2 --no-extras --no-secondary --no-cpu-baseline"
sum() { python - "$1" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[1],'step',round(d['ms_per_step'],2),'sweep',round(r['kernel_ms'],2),'refine',round(r['refine_ms'],2),'nr',round(r['null_ratios_ms'],2),
      'mism',d['verified']['mismatches_all_passes'],'fb',r['fallback_rows'])
P
}
cp wisecondorx_amd/libwcx_hip.so /tmp/base.so
$B > gpurun_out/ab_base1.json 2>/dev/null; sum gpurun_out/ab_base1.json
for t in "$@"; do
  cp wisecondorx_amd/libwcx_hip_$t.so wisecondorx_amd/libwcx_hip.so
  $B > gpurun_out/ab_$t.json 2>/dev/null; sum gpurun_out/ab_$t.json
done
cp /tmp/base.so wisecondorx_amd/libwcx_hip.so
$B > gpurun_out/ab_base2.json 2>/dev/null; sum gpurun_out/ab_base2.json
