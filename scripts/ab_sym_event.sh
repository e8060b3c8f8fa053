set -x
mkdir -p gpurun_out
B="python bench.py --steps 8 --warmup 3 --no-secondary --no-extras --no-cpu-baseline"
sum() { python -c "
import json,sys; d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[1], 'step',round(d['ms_per_step'],2),'sweep',round(r['kernel_ms'],2),'frac',round(r['frac'],3),'refine',round(r['refine_ms'],2),'mism',d['verified']['mismatches_all_passes'],'fb',r['fallback_rows'],'app',r['appends'])" $1; }
$B > gpurun_out/ab_new1.json 2>/dev/null; sum gpurun_out/ab_new1.json
cp wisecondorx_amd/libwcx_hip.so /tmp/new.so; cp wisecondorx_amd/libwcx_hip_old.so wisecondorx_amd/libwcx_hip.so
$B > gpurun_out/ab_old1.json 2>/dev/null; sum gpurun_out/ab_old1.json
cp /tmp/new.so wisecondorx_amd/libwcx_hip.so
$B > gpurun_out/ab_new2.json 2>/dev/null; sum gpurun_out/ab_new2.json
timeout 600 python -m pytest tests/test_gpu_sym.py -x -q 2>&1 | tail -3
