#!/usr/bin/env python3
"""The ranking and selection kernels of the null ratios ALONE on the device (in bench.py the ranking
runs beside the refine / the sweep on the auxiliary stream, and rocprofv3 then reports durations
stretched by the kernel it shares the chip with).  Run under rocprofv3 --kernel-trace --stats.
usage: prof_rank.py [bins] [samples] [null samples]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from wisecondorx_amd import _lib, newref_tools as nt
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 182000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    n_ids = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    rng = np.random.default_rng(0)
    X = np.abs(rng.normal(1.0, 0.1, (B, S)))
    idx = rng.integers(0, B, (B, 300)).astype(np.int32)
    ctx = _lib.default_context(0)
    for _ in range(3):
        nt.get_null_ratios(X, idx, 0, B, list(range(n_ids)), ctx=ctx)
    print("null_ratios timer (ranking + selection): %.3f ms" % ctx.kernel_ms("null_ratios"))


if __name__ == "__main__":
    main()
