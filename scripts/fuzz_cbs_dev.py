#!/usr/bin/env python3
"""Hunt: wcx_cbs_batch_dev (series compacted and kept on the device) against wcx_cbs_batch (host copies)
on random batches -- NA runs of random lengths (zeros / NaN), zero weights, planted change-points, a few
+-inf, random bin sizes (the NA-run limit of CBS.R:84-113) -- segments and means bit for bit.
usage: python scripts/fuzz_cbs_dev.py [first_seed=0] [n_seeds=40]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from wisecondorx_amd import _lib  # noqa: E402


def one(seed, ctx):
    rng = np.random.default_rng(seed)
    n_chr = int(rng.integers(1, 25))
    n_per_chr = rng.integers(5, 900, n_chr)
    ns = int(rng.integers(1, 9))
    off = np.concatenate(([0], np.cumsum(n_per_chr))).astype(np.int64)
    n_bins = int(off[-1]) + int(rng.integers(0, 50))
    r = rng.normal(0, 0.05, (ns, n_bins))
    w = rng.uniform(0.5, 2.0, (ns, n_bins))
    for s in range(ns):
        for _ in range(int(rng.integers(0, 6))):           # change-points
            c = int(rng.integers(0, n_chr)); a = int(off[c] + rng.integers(0, n_per_chr[c]))
            r[s, a:min(a + int(rng.integers(3, 200)), int(off[c + 1]))] += rng.normal(0, 0.5)
        for _ in range(int(rng.integers(0, 12))):          # NA runs
            a = int(rng.integers(0, n_bins)); L = int(rng.integers(1, 80))
            r[s, a:a + L] = 0 if rng.random() < 0.6 else np.nan
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, n_bins)); w[s, a:a + int(rng.integers(1, 30))] = 0
        if rng.random() < 0.2:
            c = int(rng.integers(0, n_chr)); r[s, off[c]:off[c + 1]] = 0
    with_inf = rng.random() < 0.25
    if with_inf:
        for _ in range(int(rng.integers(1, 4))):
            r[int(rng.integers(0, ns)), int(rng.integers(0, n_bins))] = np.inf if rng.random() < 0.5 else -np.inf
    binsize = int(rng.choice([15000, 100000, 500000, 1000000, 3000000]))
    alpha = float(rng.choice([1e-2, 1e-3, 1e-4]))
    off_a, off_p = _lib.i64_array(off)
    cap = 2048
    out = []
    for dev in (False, True):
        seg = np.empty((ns, cap, 4)); cnt = np.zeros(ns, dtype=np.int32)
        if dev:
            d_r = torch.from_numpy(r).cuda(); d_w = torch.from_numpy(w).cuda(); torch.cuda.synchronize()
            _lib.check(ctx.lib.wcx_cbs_batch_dev(ctx.h, d_r.data_ptr(), d_w.data_ptr(), ns, n_bins, off_p, n_chr,
                                                 alpha, binsize, seed, _lib.ptr(seg), cap, _lib.ptr(cnt)))
        else:
            _lib.check(ctx.lib.wcx_cbs_batch(ctx.h, _lib.ptr(np.ascontiguousarray(r)), _lib.ptr(np.ascontiguousarray(w)),
                                             ns, n_bins, off_p, n_chr, alpha, binsize, seed, _lib.ptr(seg), cap,
                                             _lib.ptr(cnt)))
        out.append([seg[i, :cnt[i]].copy() for i in range(ns)])
    ok = all(a.shape == b.shape and np.array_equal(a, b, equal_nan=True) for a, b in zip(*out))
    return ok, sum(len(a) for a in out[0]), with_inf


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ctx = _lib.default_context()
    bad = 0
    segs = 0
    infs = 0
    for seed in range(first, first + n):
        ok, ns_, wi = one(seed, ctx)
        segs += ns_
        infs += wi
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, flush=True)
    print("seeds {}..{}: {} mismatches, {} segments, {} cases with inf".format(first, first + n - 1, bad, segs, infs))


if __name__ == "__main__":
    main()
