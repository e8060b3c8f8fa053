"""ctypes wrapper of oracle/libwcx_oracle.so (C restatement of newref_tools.py:255-278).
TEST INFRASTRUCTURE ONLY -- see oracle/wcx_oracle.c."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwcx_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


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
