#!/usr/bin/env python3
"""dev helper (GPU box): a hunt for the symmetric sweep (csrc/screen_sym.h) -- ALL rows of random
matrices (shapes, K, refsize, data families with ties / outliers / hubs / NaN) through the forced
symmetric path with random work-item geometry (chunk size, splits, fill), indices and distances bit for
bit against the threaded C oracle.  usage: fuzz_sym.py [first_seed [last_seed]]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import c_oracle as CO
from wisecondorx_amd import _lib, newref_tools as nt

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 120
os.environ["WCX_SCREEN_SYM"] = "2"
bad = ran = 0
for seed in range(first, last):
    rng = np.random.default_rng(10_000 + seed)
    n_chr = int(rng.integers(3, 12))
    mb = rng.integers(0 if seed % 7 == 0 else 200, 2200, n_chr)
    mb[int(rng.integers(0, n_chr))] += 2100                   # one big chromosome (B >= 2048)
    cum = np.cumsum(mb).tolist()
    B = cum[-1]
    S = int(rng.choice([12, 24, 40, 64, 100, 112, 130, 200, 256, 300, 384, 500, 508]))
    k = int(min(rng.choice([20, 60, 100, 150, 300, 400]), max(4, (B - int(mb.max())) // 3)))
    fam = seed % 6
    base = rng.standard_normal((B, S))
    if fam == 0:
        X = 1.0 + 0.05 * base
    elif fam == 1:                                            # hubs: a low-norm cluster everybody likes
        scale = np.where(rng.random(B) < 0.03, 0.01, 0.08)
        X = 1.0 + scale[:, None] * base
    elif fam == 2:                                            # heavy ties
        X = rng.integers(0, 5, (B, S)).astype(np.float64)
    elif fam == 3:                                            # wide norm spread
        X = 1.0 + 0.1 * rng.gamma(2.0, 0.5, B)[:, None] * base
    elif fam == 4:                                            # low-rank structure + noise (PCA-like residue)
        X = 1.0 + 0.05 * (rng.standard_normal((B, 3)) @ rng.standard_normal((3, S))) + 0.02 * base
    else:                                                     # specials
        X = 1.0 + 0.05 * base
        X[rng.integers(0, B, 3)] = np.nan
        X[int(rng.integers(0, B)), int(rng.integers(0, S))] = np.inf
        X[rng.integers(0, B, 4)] *= 100.0
        d = rng.integers(0, B, 6)
        X[d[1::2]] = X[d[0::2]]                               # duplicate rows
    X = np.asfortranarray(X)
    os.environ["WCX_SCREEN_SAMPLE"] = str(int(rng.choice([4, 8, 16])))
    os.environ["WCX_SYM_CHUNK_KB"] = str(int(rng.choice([64, 160, 512, 2048, 8192])))
    os.environ["WCX_SYM_SPLIT"] = str(int(rng.choice([0, 0, 2, 3])))
    os.environ["WCX_SYM_FILL"] = str(int(rng.choice([0, 1, 1])))
    oi, od = CO.get_reference_rows_threaded(np.ascontiguousarray(X.T), cum, 0, B, k)
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B, mode=2)
    st = _lib.default_context().topk_stats()
    ran += 1 if st["sym_gates"] > 0 else 0
    ok = np.array_equal(idx, oi) and np.array_equal(dist, od)
    if not ok:
        bad += 1
        rows = np.flatnonzero((idx != oi).any(axis=1) | (dist != od).any(axis=1))
        print("MISMATCH seed", seed, "B", B, "S", S, "k", k, "fam", fam, "rows", rows[:8], len(rows),
              {k_: os.environ[k_] for k_ in ("WCX_SCREEN_SAMPLE", "WCX_SYM_CHUNK_KB", "WCX_SYM_SPLIT", "WCX_SYM_FILL")})
    if seed % 10 == 0:
        print("seed", seed, "B", B, "S", S, "k", k, "fam", fam, "fallback", st["fallback_rows"], "ok", ok, flush=True)
print("checked seeds {}..{}: symmetric sweep ran in {} cases, mismatches: {}".format(first, last - 1, ran, bad))
