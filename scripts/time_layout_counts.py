import numpy as np, time, sys, os
sys.path.insert(0,'.')
import wisecondorx_amd.predict_tools as P
from wisecondorx_amd.synth import bins_per_chr
import torch
rng=np.random.default_rng(0)
bpc=[int(v) for v in bins_per_chr(15000)]
ref={"bins_per_chr":np.array(bpc)}
samples=[{str(c+1): rng.integers(0,200,bpc[c]).astype(np.int32) for c in range(24)} for _ in range(96)]
out=torch.empty((96,sum(bpc)),dtype=torch.int32,pin_memory=True).numpy()
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
for rep in range(3):
    t0=time.perf_counter(); a=P.sample_counts_matrix(samples,ref,"",out=out); print("native %.2f ms"%((time.perf_counter()-t0)*1e3))
a=a.copy(); orig=P._layout_counts_native; P._layout_counts_native=lambda *x: False
for rep in range(2):
    t0=time.perf_counter(); b=P.sample_counts_matrix(samples,ref,"",out=out); print("python %.2f ms equal %s"%((time.perf_counter()-t0)*1e3,np.array_equal(a,b)))
