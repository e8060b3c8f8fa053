#!/usr/bin/env python3
"""dev helper: batched normalise (config 5) kernel time, lanes path on/off, + batch == single check."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from wisecondorx_amd import _lib, newref_tools, predict_tools as pt
co, p, _ = bench.make_workload(15000, 100)
X = p["X"]; cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
ctx = _lib.default_context(0)
idx, dist = newref_tools.get_ref_for_rows(X, cum, 300, 0, cum[-1], ctx)
ref = dict(p); ref.update({"indexes": idx, "distances": dist})
rng = np.random.default_rng(1)
xs = np.asarray(X)[:, 7][None, :] * (1.0 + 0.02 * rng.standard_normal((96, cum[-1])))
xs[5, 1000:1040] = 0.0
cache = {}
cutoff = pt.get_optimal_cutoff(ref, 5, cache)
for rep in range(3):
    z, r, n, mlr, mz = pt.normalize_repeat_batch(xs, ref, cutoff, 0, 0, "", cache)
    print("batch normalize kernels ms", ctx.kernel_ms("normalize"))
bad = 0
for i in (0, 5, 17, 48, 95):
    z1, r1, n1, mlr1, mz1 = pt.normalize_repeat(xs[i], ref, cutoff, 0, 0, "", cache)
    same = np.array_equal(r1, r[i], equal_nan=True) and np.array_equal(n1, n[i]) and mlr1 == mlr[i]
    with np.errstate(all="ignore"):
        zerr = np.nanmax(np.abs(z1 - z[i]) / np.maximum(1.0, np.abs(z1)))
    zsame = np.array_equal(z1, z[i], equal_nan=True)
    print(i, "batch == single (r, n, m_lr):", same, "n diffs", int(np.sum(n1 != n[i])), "r diffs",
          int(np.sum(~((r1 == r[i]) | (np.isnan(r1) & np.isnan(r[i]))))), "z bitwise", zsame, "z max |dz| / max(1, |z|)", zerr,
          "m_z", mz1, mz[i])
    bad += not same or not (zerr < 1e-11)
print("BAD" if bad else "OK")
# where the largest z deviations of the batch path sit (round 6: incremental statistics)
i = 5
z1, r1, n1, _, _ = pt.normalize_repeat(xs[i], ref, cutoff, 0, 0, "", cache)
with np.errstate(all="ignore"):
    dz = np.abs(z1 - z[i])
    worst = np.argsort(np.nan_to_num(dz))[-8:]
for b in worst:
    print("bin", int(b), "x", xs[i, b], "z single", z1[b], "z batch", z[i][b], "|dz|", dz[b], "n", n1[b], "r", r1[b])
