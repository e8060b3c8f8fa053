#!/usr/bin/env python3
"""dev helper (GPU box): robustness fuzz of wcx_cbs / wcx_cbs_batch -- constant series, all-NA and
tiny chromosomes, long NA runs, huge / tiny values and weights, spikes; checks structure (segments
ordered, inside the chromosome, no overlap), determinism, batch == single."""
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from wisecondorx_amd import predict_tools as pt

def make(rng):
    n_chr = 23
    n = [int(rng.choice([1, 2, 3, 5, 40, 250, 900, 3000])) for _ in range(n_chr)]
    r, w = [], []
    for c in range(n_chr):
        kind = rng.integers(0, 8)
        x = rng.normal(0, 0.08, n[c])
        if kind == 0: x[:] = 0.0                                  # all NA
        elif kind == 1: x[:] = 0.37                               # constant
        elif kind == 2 and n[c] > 20: x[n[c] // 3: n[c] // 3 + max(2, n[c] // 5)] += rng.choice([-1, 1]) * 0.5
        elif kind == 3: x[rng.random(n[c]) < 0.6] = 0.0           # many NA
        elif kind == 4 and n[c] > 60: x[10:10 + n[c] // 2] = 0.0  # one long NA run
        elif kind == 5: x *= 1e6
        elif kind == 6: x *= 1e-9
        elif kind == 7 and n[c] > 5: x[rng.integers(0, n[c])] += 5.0   # spike
        ww = rng.uniform(0.5, 2.0, n[c])
        if rng.random() < 0.2: ww[rng.random(n[c]) < 0.1] = 0.0   # weight 0 -> 1
        if rng.random() < 0.1: ww *= 1e4
        r.append(x); w.append(ww)
    return {"results_r": r, "results_w": w}

rng = np.random.default_rng(0)
cases = [make(rng) for _ in range(60)]
bad = 0
single = []
for i, res in enumerate(cases):
    segs = pt.run_cbs(res, "F", 1e-4, 100000, 5)
    again = pt.run_cbs(res, "F", 1e-4, 100000, 5)
    if segs != again:
        bad += 1; print("non-deterministic", i)
    last = {}
    for c, s, e, ratio in segs:
        n = len(res["results_r"][c])
        ok = 0 <= s < e <= n and s >= last.get(c, 0) and (np.isfinite(ratio) or np.isnan(ratio))
        if not ok:
            bad += 1; print("bad segment", i, c, s, e, ratio, n)
        last[c] = e
    single.append(segs)
for b0 in range(0, 60, 12):                              # batches of samples with the SAME layout only
    pass
same = [dict(results_r=[x.copy() for x in cases[0]["results_r"]], results_w=cases[0]["results_w"]) for _ in range(5)]
for j, res in enumerate(same):
    res["results_r"][7] = res["results_r"][7] + 0.01 * j
got = pt.run_cbs_batch(same, "F", 1e-4, 100000, 5)
for j, res in enumerate(same):
    if got[j] != pt.run_cbs(res, "F", 1e-4, 100000, 5):
        bad += 1; print("batch != single", j)
print("cases", len(cases), "problems", bad)
