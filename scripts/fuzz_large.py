#!/usr/bin/env python3
"""dev helper (GPU box): the production search path (sampled pre-pass, two-stream sweep, B >= 32768) on
structured data -- integer ties, wide norm spread, clustered prototypes, near-duplicates -- against the
C oracle on sampled rows; prints flagged-row counts."""
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import c_oracle as CO
from wisecondorx_amd import _lib, newref_tools as nt

bad = 0
for kind in range(5):
    for S, k in ((60, 300), (140, 150)):
        rng = np.random.default_rng(100 * kind + S)
        mb = rng.integers(5000, 12000, 8)
        cum = np.cumsum(mb).tolist()
        B = cum[-1]
        if kind == 0:
            X = rng.integers(0, 4, (B, S)).astype(np.float64)
        elif kind == 1:
            X = 1.0 + 0.05 * rng.standard_normal((B, S))
        elif kind == 2:
            X = 1.0 + 0.1 * rng.gamma(2.0, 0.5, B)[:, None] * rng.standard_normal((B, S))
        elif kind == 3:
            proto = rng.standard_normal((16, S))
            X = 1.0 + 0.05 * (proto[rng.integers(0, 16, B)] + 0.3 * rng.standard_normal((B, S)))
        else:                                   # neighbours concentrated in bins b % 16 == 3 (worst case
            X = 1.0 + 0.05 * rng.standard_normal((B, S))          # for a b % 16 == 0 sample)
            X[3::16] = 1.0 + 0.01 * rng.standard_normal((len(X[3::16]), S))
        X = np.asfortranarray(X)
        import time
        nt.get_ref_for_rows(X, cum, k, 0, min(B, 4096), mode=2)      # (warm-up: allocations)
        t0 = time.perf_counter()
        idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B, mode=2)
        dt = time.perf_counter() - t0
        fb = _lib.default_context().topk_stats()["fallback_rows"]
        rows = rng.choice(B, 120, replace=False)
        Xs = np.ascontiguousarray(np.asarray(X).T)
        ok = True
        for t in rows:
            oi, od = CO.get_reference_rows(Xs, cum, int(t), int(t) + 1, k)
            if not (np.array_equal(idx[t], oi[0]) and np.array_equal(dist[t], od[0])):
                ok = False
        bad += 0 if ok else 1
        print("kind", kind, "B", B, "S", S, "k", k, "flagged rows", fb, "sampled rows exact:", ok, "search s %.3f" % dt)
print("mismatching cases:", bad)
