set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from wisecondorx_amd import npz_io, synth
os.makedirs("/tmp/cli", exist_ok=True)
co = synth.Cohort(100000, female_y=0.1)
samples, g = co.cohort(30)
for i, s in enumerate(samples):
    npz_io.save_sample("/tmp/cli/s%02d.npz" % i, s, 100000)
npz_io.save_sample("/tmp/cli/test.npz", co.sample(99, "F", cnv=[(3, 100, 400, 1.5)]), 100000)
PY
time python -m wisecondorx_amd.main --loglevel warning newref /tmp/cli/s*.npz /tmp/cli/ref.npz --binsize 100000 --yfrac 0.004
time python -m wisecondorx_amd.main --loglevel warning predict /tmp/cli/test.npz /tmp/cli/ref.npz /tmp/cli/out --bed --seed 1
python -m wisecondorx_amd.main gender /tmp/cli/test.npz /tmp/cli/ref.npz
head -3 /tmp/cli/out_aberrations.bed; wc -l /tmp/cli/out_segments.bed
python -c "import numpy as np; r=np.load('/tmp/cli/ref.npz', allow_pickle=True); print(sorted(r.files)[:8], r['indexes'].shape)"
