#!/usr/bin/env python3
"""Is the 100 kb x 100 samples step bound by the HOST issuing its ~280 launches?  Host timestamps
around the calls of bench.Workload.step (no synchronisation added) beside the step's wall time.
usage: host_issue_small.py [binsize] [samples]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from wisecondorx_amd import dist as wd
    binsize = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    args = argparse.Namespace(gpus=1, steps=20, warmup=5, binsize=binsize, samples=S, refsize=300, replicas=False,
                              debug_flags=0, concurrent_passes=int(os.environ.get("CP", "1")))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = bench.Workload(args, S, torch, dev, 0, 0, 1)
    marks = []

    def wrap(name):
        f = getattr(wd, name)

        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            marks.append((name, t0, time.perf_counter()))
            return r
        setattr(wd, name, g)
    for n in ("newref_sharded", "newref_gonosomal_sharded", "gather_reference3", "predict_full_dev"):
        wrap(n)
    for _ in range(8):
        w.step(False)
    torch.cuda.synchronize()
    rows = []
    for _ in range(20):
        marks.clear()
        t0 = time.perf_counter()
        w.step(False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append([1e3 * (t1 - t0), 1e3 * (t2 - t0)] + [1e3 * (b - a) for _, a, b in marks] +
                    [1e3 * (marks[-1][1] - t0)])
    names = ["step returns", "step + drain"] + [m[0] for m in marks] + ["predict call starts at"]
    med = np.median(np.array(rows), axis=0)
    for n, v in zip(names, med):
        print("%-28s %7.3f ms" % (n, v))


if __name__ == "__main__":
    main()
