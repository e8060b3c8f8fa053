#!/bin/bash
# Same-box A / B of two builds of libwcx_hip.so (wisecondorx_amd/libwcx_hip_old.so = the build to compare
# against, made from an older tree): step, sweep, S = 100 block, F / M passes, 100 kb; then the sweep tests.
mkdir -p gpurun_out
B="python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline"
sum() { python - "$1" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; s=d.get('secondary',{}); g=r['gonosomal_passes']
print(sys.argv[1],'step',round(d['ms_per_step'],2),'sweep',round(r['kernel_ms'],2),'frac',round(r['frac'],3),'refine',round(r['refine_ms'],2),
      'mism',d['verified']['mismatches_all_passes'],'fb',r['fallback_rows'],
      '| S100 step',round(s.get('ms_per_step',0),2),'sweep',round(s.get('roofline',{}).get('kernel_ms',0),2),'frac',round(s.get('roofline',{}).get('frac',0),3),'mism',s.get('verified',{}).get('mismatches_all_passes'),
      '| F',round(g['F']['screen_ms'],2),round(g['F']['topk_ms'],2),'M',round(g['M']['screen_ms'],2),round(g['M']['topk_ms'],2))
P
}
$B > gpurun_out/ab_new1.json 2>/dev/null; sum gpurun_out/ab_new1.json
if [ -f wisecondorx_amd/libwcx_hip_old.so ]; then
cp wisecondorx_amd/libwcx_hip.so /tmp/new.so; cp wisecondorx_amd/libwcx_hip_old.so wisecondorx_amd/libwcx_hip.so
$B > gpurun_out/ab_old1.json 2>/dev/null; sum gpurun_out/ab_old1.json
cp /tmp/new.so wisecondorx_amd/libwcx_hip.so
$B > gpurun_out/ab_new2.json 2>/dev/null; sum gpurun_out/ab_new2.json
fi
python scripts/sweep_hub1.py 100kb 2>&1 | grep -E "sampled  |hub   " | head -3
timeout 800 python -m pytest tests/test_gpu_sym.py tests/test_gpu_newref.py -x -q 2>&1 | tail -3
