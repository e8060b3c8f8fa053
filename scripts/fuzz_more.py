import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from test_gpu_fuzz import _case
from oracle import c_oracle as CO
from wisecondorx_amd import newref_tools as nt
bad = 0
for seed in range(64, 400):
    X, cum, k, s, e = _case(seed)
    oi, od = CO.get_reference_rows(np.ascontiguousarray(np.asarray(X).T), cum, s, e, k)
    for mode in (2, 0):
        idx, dist = nt.get_ref_for_rows(X, cum, k, s, e, mode=mode)
        if not (np.array_equal(idx, oi) and np.array_equal(dist, od)):
            bad += 1; print("MISMATCH seed", seed, "mode", mode)
print("checked seeds 64..399, mismatches:", bad)
