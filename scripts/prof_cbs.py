#!/usr/bin/env python3
"""dev helper: CBS of one synthetic 15 kb sample (planted CNV), timed; run under rocprofv3 --stats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wisecondorx_amd import _lib, predict_tools as pt
from wisecondorx_amd.synth import bins_per_chr
rng = np.random.default_rng(0)
bpc = bins_per_chr(15000)[:22]
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res_list = []
for s in range(ns):
    r = [rng.normal(0, 0.08, n) for n in bpc]
    w = [rng.uniform(0.5, 2.0, n) for n in bpc]
    r[2 + s % 5][100:2767] += 0.58
    for c in range(22):
        r[c][rng.random(len(r[c])) < 0.05] = 0
    res_list.append({"results_r": r, "results_w": w})
ctx = _lib.default_context(0)
if os.environ.get('CBS_LAPS'):
    ctx.lib.wcx_debug_flags(ctx.h, 8)
for rep in range(3):
    t = time.perf_counter()
    segs = pt.run_cbs_batch(res_list, "F", 1e-4, 15000, 1, ctx)
    dt = time.perf_counter() - t
    print("samples", ns, "wall ms", 1e3 * dt, "kernel-span ms", ctx.kernel_ms("cbs"), "segments", [len(x) for x in segs][:8])
