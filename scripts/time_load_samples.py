#!/usr/bin/env python3
"""dev helper: wall-clock of importing N sample files (npz_io.load_sample + scale_sample + read totals, what
newref's loader threads do) against the number of loader threads.  usage: time_load_samples.py [n_files]"""
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WCX_NO_TORCH_PRELOAD", "1")
from wisecondorx_amd import npz_io, synth                      # noqa: E402
from wisecondorx_amd.main import _read_totals                  # noqa: E402
from wisecondorx_amd.overall_tools import scale_sample         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d = tempfile.mkdtemp(prefix="wcx_load_")
co = synth.Cohort(15000, female_y=0.1)
base = [co.sample(i, "F") for i in range(8)]
files = []
for i in range(n):
    f = os.path.join(d, "s{}.npz".format(i))
    npz_io.save_sample(f, base[i % 8], 15000)
    files.append(f)
if os.environ.get("WCX_MALLOPT"):
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    print("mallopt", libc.mallopt(-3, 256 << 20), libc.mallopt(-1, 1 << 30), flush=True)   # M_MMAP_THRESHOLD, M_TRIM_THRESHOLD
print("cpu budget", npz_io._cpu_budget(), "os.cpu_count", os.cpu_count(), flush=True)


def load_one(f):
    s, b = npz_io.load_sample(f)
    s = scale_sample(s, b, 15000)
    return s, _read_totals(s)


for nt in (4, 8, 12, 16, 24, 32, 48, 64):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(nt) as ex:
            list(ex.map(load_one, files))
        best = min(best, time.perf_counter() - t0)
    print("threads {:3d}: {:.3f} s = {:.2f} ms per file".format(nt, best, 1e3 * best / n), flush=True)
