#!/usr/bin/env python3
"""dev helper (GPU box): a longer hunt with the generator of tests/test_gpu_fuzz.py -- the search (MFMA
screen + refine, and auto mode) against the C oracle, the null ratios (whole range: rank path or, for
few rows, the no-ranking path) against the NumPy oracle.  usage: fuzz_more.py [first_seed [last_seed]]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from test_gpu_fuzz import _case
from oracle import c_oracle as CO
from oracle import wcx_oracle as O
from wisecondorx_amd import newref_tools as nt

first = int(sys.argv[1]) if len(sys.argv) > 1 else 64
last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
bad = 0
for seed in range(first, last):
    X, cum, k, s, e = _case(seed)
    oi, od = CO.get_reference_rows(np.ascontiguousarray(np.asarray(X).T), cum, s, e, k)
    for mode in (2, 0):
        idx, dist = nt.get_ref_for_rows(X, cum, k, s, e, mode=mode)
        if not (np.array_equal(idx, oi) and np.array_equal(dist, od)):
            bad += 1
            print("MISMATCH search seed", seed, "mode", mode)
    S = np.asarray(X).shape[1]
    ids = np.random.default_rng(seed).permutation(S)[:min(S, 100)].tolist()
    with np.errstate(all="ignore"):
        want = O.null_ratios(X, oi, s, e, ids)
    got = nt.get_null_ratios(X, oi, s, e, ids)
    if not np.allclose(got, want, rtol=1e-12, atol=1e-13, equal_nan=True):
        bad += 1
        print("MISMATCH null ratios seed", seed)
print("checked seeds {}..{}, mismatches: {}".format(first, last - 1, bad))
