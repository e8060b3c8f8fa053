"""ctypes wrapper of oracle/libwcx_oracle.so (C restatement of newref_tools.py:255-278).
TEST INFRASTRUCTURE ONLY -- see oracle/wcx_oracle.c."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwcx_oracle.so")
_SO_TILED = os.path.join(_HERE, "libwcx_oracle_tiled.so")
_lib = None
_lib_tiled = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def host_threads():
    """Worker threads worth starting: the CPUs this process may use, capped by the cgroup CPU quota
    (the GPU boxes expose 256 CPUs under a 16-CPU quota; 254 busy threads are throttled to a third
    of the throughput of 32)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, 2 * int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.wcxo_topk_rows.restype = C.c_int
        _lib.wcxo_topk_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64,
                                        C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


def lib_tiled():
    global _lib_tiled
    if _lib_tiled is None:
        if not os.path.exists(_SO_TILED):
            build()
        _lib_tiled = C.CDLL(_SO_TILED)
        _lib_tiled.wcxo_topk_rows_tiled.restype = C.c_int
        _lib_tiled.wcxo_topk_rows_tiled.argtypes = lib().wcxo_topk_rows.argtypes
    return _lib_tiled


def topk_rows(Xs, cs, ce, row_begin, row_end, k):
    """Xs float64[S][B] C-contiguous; rows [row_begin,row_end) of chromosome [cs,ce)."""
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    S, B = Xs.shape
    n = row_end - row_begin
    idx = np.empty((n, k), dtype=np.int32)
    dist = np.empty((n, k), dtype=np.float64)
    rc = lib().wcxo_topk_rows(Xs.ctypes.data, B, S, cs, ce, row_begin, row_end, k,
                              idx.ctypes.data, dist.ctypes.data)
    if rc:
        raise MemoryError("wcxo_topk_rows failed")
    return idx, dist


def get_reference_rows(Xs, chr_cum, row_begin, row_end, k):
    """All target rows [row_begin,row_end) with the gonosomal dummy rule
    (newref_tools.py:186-191) -- C analogue of the search part of get_reference."""
    n_chr = len(chr_cum)
    out_i = np.zeros((row_end - row_begin, k), dtype=np.int32)
    out_d = np.ones((row_end - row_begin, k), dtype=np.float64)
    for c in range(n_chr):
        cs = chr_cum[c - 1] if c else 0
        ce = chr_cum[c]
        lo, hi = max(cs, row_begin), min(ce, row_end)
        if lo >= hi:
            continue
        if n_chr > 22 and c != 22 and c != 23:
            continue
        i, d = topk_rows(Xs, cs, ce, lo, hi, k)
        out_i[lo - row_begin:hi - row_begin] = i
        out_d[lo - row_begin:hi - row_begin] = d
    return out_i, out_d


def get_reference_rows_threaded(Xs, chr_cum, row_begin, row_end, k, threads=None, rows_per_task=64):
    """get_reference_rows for MANY rows: the cache-tiled C restatement (oracle/wcx_oracle_tiled.c,
    asserted bit-identical to the per-row oracle/wcx_oracle.c in tests/test_oracle_golden.py) on a
    pool of host threads (ctypes releases the GIL), row blocks of one chromosome per task.  This
    is what lets the GPU tests compare ALL 182 k rows of the 15 kb problems instead of a sample."""
    from concurrent.futures import ThreadPoolExecutor
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    S, B = Xs.shape
    n_chr = len(chr_cum)
    threads = threads or host_threads()
    out_i = np.zeros((row_end - row_begin, k), dtype=np.int32)
    out_d = np.ones((row_end - row_begin, k), dtype=np.float64)
    fn = lib_tiled().wcxo_topk_rows_tiled
    tasks = []
    for c in range(n_chr):
        cs = int(chr_cum[c - 1]) if c else 0
        ce = int(chr_cum[c])
        lo, hi = max(cs, row_begin), min(ce, row_end)
        if lo >= hi or (n_chr > 22 and c != 22 and c != 23):
            continue
        for a in range(lo, hi, rows_per_task):
            tasks.append((cs, ce, a, min(a + rows_per_task, hi)))

    def work(t):
        cs, ce, a, b = t
        i = out_i[a - row_begin:b - row_begin]
        d = out_d[a - row_begin:b - row_begin]
        rc = fn(Xs.ctypes.data, B, S, cs, ce, a, b, k, i.ctypes.data, d.ctypes.data)
        if rc:
            raise MemoryError("wcxo_topk_rows_tiled failed")
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(work, tasks))
    return out_i, out_d


def topk_row_blocks_threaded(Xs, chr_cum, starts, rows_per_block, k, threads=None):
    """Search of scattered row blocks [s, s + rows_per_block) (clipped at their chromosome's end) with
    the tiled C oracle on host threads.  Returns (rows int64[n], idx int32[n][k], dist float64[n][k])
    -- bench.py's out-of-timed-region self-check."""
    from concurrent.futures import ThreadPoolExecutor
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    S, B = Xs.shape
    cum = np.asarray(chr_cum, dtype=np.int64)
    fn = lib_tiled().wcxo_topk_rows_tiled
    tasks, total = [], 0
    for s0 in starts:
        c = int(np.searchsorted(cum, s0, side="right"))
        cs, ce = (int(cum[c - 1]) if c else 0), int(cum[c])
        e0 = min(int(s0) + rows_per_block, ce)
        tasks.append((cs, ce, int(s0), e0, total))
        total += e0 - int(s0)
    rows = np.empty(total, dtype=np.int64)
    idx = np.empty((total, k), dtype=np.int32)
    dist = np.empty((total, k), dtype=np.float64)

    def work(t):
        cs, ce, a, b, o = t
        rows[o:o + b - a] = np.arange(a, b)
        i, d = idx[o:o + b - a], dist[o:o + b - a]
        if fn(Xs.ctypes.data, B, S, cs, ce, a, b, k, i.ctypes.data, d.ctypes.data):
            raise MemoryError("wcxo_topk_rows_tiled failed")
    with ThreadPoolExecutor(max_workers=threads or host_threads()) as ex:
        list(ex.map(work, tasks))
    return rows, idx, dist
