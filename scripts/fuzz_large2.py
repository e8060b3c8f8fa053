#!/usr/bin/env python3
"""dev helper (GPU box): more pathological inputs for the production search path (B >= 32768):
constant data, exact duplicates of a few prototypes, huge dynamic range, B just above the sampling
threshold, one dominant chromosome."""
import os
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import c_oracle as CO
from wisecondorx_amd import _lib, newref_tools as nt

bad = 0
for kind in range(5):
    rng = np.random.default_rng(900 + kind)
    S, k = 80, 300
    mb = rng.integers(5000, 9000, 6)
    if kind == 3:
        mb = np.array([8000, 8000, 8000, 8800])            # B = 32800, just above the threshold
    if kind == 4:
        mb = np.array([45000, 300, 200, 500])
    cum = np.cumsum(mb).tolist()
    B = cum[-1]
    if kind == 0:
        X = np.full((B, S), 1.25)
        X[::7, 3] += 1e-9
    elif kind == 1:
        proto = 1.0 + 0.05 * rng.standard_normal((50, S))
        X = proto[rng.integers(0, 50, B)].copy()
        nz = rng.random(B) < 0.1
        X[nz] += 0.01 * rng.standard_normal((int(nz.sum()), S))
    elif kind == 2:
        X = (1.0 + 0.05 * rng.standard_normal((B, S))) * 10.0 ** rng.integers(-6, 7, B)[:, None]
    else:
        X = 1.0 + 0.05 * rng.standard_normal((B, S))
    X = np.asfortranarray(X)
    nt.get_ref_for_rows(X, cum, k, 0, min(B, 4096), mode=2)
    t0 = time.perf_counter()
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B, mode=2)
    dt = time.perf_counter() - t0
    fb = _lib.default_context().topk_stats()["fallback_rows"]
    Xs = np.ascontiguousarray(np.asarray(X).T)
    ok = True
    for t in rng.choice(B, 80, replace=False):
        oi, od = CO.get_reference_rows(Xs, cum, int(t), int(t) + 1, k)
        if not (np.array_equal(idx[t], oi[0]) and np.array_equal(dist[t], od[0])):
            ok = False
    bad += 0 if ok else 1
    print("kind", kind, "B", B, "flagged rows", fb, "exact:", ok, "search s %.3f" % dt)
print("mismatching cases:", bad)
