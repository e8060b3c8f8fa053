#!/usr/bin/env python3
"""dev helper: time the CBS oracle on the fuzz cases of tests/test_gpu_cbs_oracle.py (CPU only) and
show which decision paths the cases reach."""
import os
import sys
import time
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
from concurrent.futures import ProcessPoolExecutor

import numpy as np

if __name__ == "__main__":
    import test_gpu_cbs_oracle as T
    tab = [int(v) for v in np.load(os.path.join(ROOT, "tests/golden/cbs_bdry.npz"))["table"]]
    cases = T._cases()
    which = [int(a) for a in sys.argv[1:]] or range(len(cases))
    with ProcessPoolExecutor(max_workers=max(2, (os.cpu_count() or 4)), mp_context=mp.get_context("spawn")) as ex:
        for ci in which:
            cseed, alpha, sizes, binsize, seed = cases[ci]
            res = T._case(cseed, sizes)
            t = time.time()
            segs, tr = T._oracle(ex, res, alpha, binsize, seed, tab)
            print(cseed, alpha, "time", round(time.time() - t, 1), "segs", len(segs), "tests", len(tr),
                  dict(Counter(r.get("why") for r in tr)), "sigperm",
                  sum(1 for r in tr if r.get("why") == "perm" and r.get("significant")), "np",
                  sorted([r.get("np", 0) for r in tr if r.get("why") == "perm"])[-6:], "edges",
                  [r["edge"] for r in tr if "edge" in r][:6], flush=True)
