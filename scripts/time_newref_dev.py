#!/usr/bin/env python3
"""Where the wall-clock of one device-resident newref pass goes (prep.DeviceCounts -> prepare_dev ->
get_reference_dev), call by call, on Poisson counts of BASELINE.json's shape.  Diagnostic only."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wisecondorx_amd import _lib, newref_tools, prep, synth  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    bpc = synth.bins_per_chr(15000)
    rng = np.random.default_rng(1)
    lam = rng.uniform(60, 140, int(np.sum(bpc)))
    lam[rng.random(lam.size) < 0.05] = 0.0
    samples = []
    for i in range(S):
        c = rng.poisson(lam * rng.uniform(0.8, 1.2)).astype(np.int32)
        off = np.concatenate(([0], np.cumsum(bpc)))
        samples.append({str(k + 1): c[off[k]:off[k + 1]] for k in range(24)})
    ctx = _lib.default_context(0)
    lib = ctx.lib
    T = {}

    def tick(name, t0):
        ctx.sync()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0

    for rep in range(2):
        T.clear()
        t0 = time.perf_counter(); dc = prep.DeviceCounts(ctx, samples); tick("DeviceCounts", t0)
        t0 = time.perf_counter(); mask, b = dc.get_mask(); tick("get_mask", t0)
        t0 = time.perf_counter()
        p = prep.prepare_dev(dc, np.arange(S), "A", mask, b)
        tick("prepare_dev", t0)
        cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
        ids = list(range(min(S, 100)))
        B, k, m = cum[-1], 300, len(ids)
        t0 = time.perf_counter()
        idx = np.empty((B, k), np.int32); dist = np.empty((B, k)); nr = np.empty((B, m))
        tick("np.empty", t0)
        bufs = [C.c_void_p() for _ in range(3)]
        t0 = time.perf_counter()
        for bb, a in zip(bufs, (idx, dist, nr)):
            _lib.check(lib.wcx_malloc(ctx.h, a.nbytes, C.byref(bb)))
        tick("wcx_malloc x3", t0)
        dX = C.c_void_p()
        _lib.check(lib.wcx_pca_corrected_dev(ctx.h, C.byref(dX)))
        cum_a, cum_p = _lib.i64_array(cum)
        ids_a, ids_p = _lib.i32_array(ids)
        t0 = time.perf_counter()
        _lib.check(lib.wcx_null_rank_prepare_dev(ctx.h, dX, B, S, ids_p, m)); tick("rank_prepare", t0)
        t0 = time.perf_counter()
        _lib.check(lib.wcx_newref_topk_dev(ctx.h, dX, B, S, cum_p, len(cum), 0, B, k, 0, bufs[0], bufs[1]))
        tick("topk", t0)
        t0 = time.perf_counter()
        _lib.check(lib.wcx_null_ratios_dev(ctx.h, dX, B, S, bufs[0], 0, B, k, ids_p, m, bufs[2]))
        tick("null_ratios", t0)
        for bb, a, nm in zip(bufs, (idx, dist, nr), ("idx", "dist", "nr")):
            t0 = time.perf_counter()
            _lib.check(lib.wcx_memcpy_d2h(ctx.h, _lib.ptr(a), bb, a.nbytes))
            tick("d2h " + nm + " ({} MB)".format(a.nbytes >> 20), t0)
        t0 = time.perf_counter()
        for bb in bufs:
            lib.wcx_free(ctx.h, bb)
        tick("wcx_free x3", t0)
        t0 = time.perf_counter()
        r = newref_tools.get_reference_dev(ctx, S, cum, k, ids)
        tick("get_reference_dev (whole)", t0)
        assert np.array_equal(r[0], idx) and np.array_equal(r[1], dist)
        dc.close()
        lib.wcx_pca_end(ctx.h)
        print("rep", rep, {k_: round(v, 4) for k_, v in T.items()})


if __name__ == "__main__":
    main()
