"""Host-side mirror of the reference's newref numerical layer (newref_tools.py) for the hot
path: same function names, argument meaning and return values, with the per-bin search and
the null-ratio loop executed by libwcx_hip.so on the MI355X.

Reference seam (newref_control.py:136-143):
    indexes, distances, null_ratios = get_reference(pca_corrected_data, masked_bins_per_chr,
                                                    masked_bins_per_chr_cum, ref_size=...,
                                                    part=..., split_parts=...)
"""
import logging
import random

import numpy as np

from . import _lib


def _get_part(partnum, outof, bincount):
    """Row range of part `partnum` (0-based) of `outof` -- newref_tools.py:244-247."""
    start_bin = int(bincount / float(outof) * partnum)
    end_bin = int(bincount / float(outof) * (partnum + 1))
    return start_bin, end_bin


def _split_by_chr(start, end, chr_bin_sums):
    """Per-chromosome regions [chr_idx, start, end] of a row range -- same results as
    newref_tools.py:227-241.  Kept for callers that want the reference's region list; the
    GPU path derives its own per-chromosome workgroups from the cumulative bin counts."""
    areas = []
    cur = [0, start, 0]
    for i, val in enumerate(chr_bin_sums):
        cur[0] = i
        if val >= end:
            break
        if start < val < end:
            cur[2] = val
            areas.append(cur)
            cur = [i, val, 0]
        cur[1] = val
    cur[2] = end
    areas.append(cur)
    return areas


def as_sample_major(pca_corrected_data):
    """(B,S) matrix -> C-contiguous float64 [S][B] sharing memory when the input is the
    Fortran-ordered array train_pca returns (newref_tools.py:147)."""
    X = np.asarray(pca_corrected_data)
    if X.ndim != 2:
        raise ValueError("pca_corrected_data must be 2-D (bins x samples)")
    Xs = X.T
    if Xs.dtype != np.float64 or not Xs.flags["C_CONTIGUOUS"]:
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    return Xs


def get_ref_for_rows(pca_corrected_data, masked_bins_per_chr_cum, ref_size, start, end,
                     ctx=None, mode=0):
    """Reference bins of target rows [start,end): the GPU form of get_ref_for_bins
    (newref_tools.py:255-278) + the region loop of get_reference (:176-206).  Returns
    (int32[n,k] indexes in own-chromosome-excluded index space, float64[n,k] distances)."""
    ctx = ctx or _lib.default_context()
    Xs = as_sample_major(pca_corrected_data)
    S, B = Xs.shape
    cum, cum_p = _lib.i64_array(masked_bins_per_chr_cum)
    n = end - start
    idx = np.empty((n, ref_size), dtype=np.int32)
    dist = np.empty((n, ref_size), dtype=np.float64)
    _lib.check(ctx.lib.wcx_newref_topk(ctx.h, _lib.ptr(Xs), B, S, cum_p, len(cum), start, end,
                                       int(ref_size), int(mode), _lib.ptr(idx), _lib.ptr(dist)))
    return idx, dist


def get_null_ratios(pca_corrected_data, index_array, start_num, end_num, sample_ids, ctx=None):
    """The null-ratio table of newref_tools.py:210-223 for already chosen sample ids."""
    ctx = ctx or _lib.default_context()
    Xs = as_sample_major(pca_corrected_data)
    S, B = Xs.shape
    idx = np.ascontiguousarray(index_array, dtype=np.int32)
    ids, ids_p = _lib.i32_array(sample_ids)
    out = np.zeros((end_num - start_num, len(ids)), dtype=np.float64)
    if out.size:
        _lib.check(ctx.lib.wcx_null_ratios(ctx.h, _lib.ptr(Xs), B, S, _lib.ptr(idx), start_num,
                                           end_num, idx.shape[1], ids_p, len(ids),
                                           _lib.ptr(out)))
    return out


def get_reference(pca_corrected_data, masked_bins_per_chr, masked_bins_per_chr_cum, ref_size,
                  part, split_parts, ctx=None, mode=0):
    """Within-sample reference of one row part -- newref_tools.py:155-224.

    Same contract as the reference: `part` is 1-based; returns (index_array int32[n,k],
    distance_array float64[n,k], null_ratio_array float64[n,min(S,100)]).  The null samples
    are drawn with random.sample exactly like newref_tools.py:214-217, so seeding `random`
    reproduces the reference's choice."""
    bincount = masked_bins_per_chr_cum[-1]
    start_num, end_num = _get_part(part - 1, split_parts, bincount)
    logging.info("Working on thread {} of {}, meaning bins {} up to {}".format(
        part, split_parts, start_num, end_num))
    index_array, distance_array = get_ref_for_rows(
        pca_corrected_data, masked_bins_per_chr_cum, ref_size, start_num, end_num, ctx, mode)
    n_samples = np.asarray(pca_corrected_data).shape[1]
    sample_ids = random.sample(range(n_samples), min(n_samples, 100))
    null_ratio_array = get_null_ratios(pca_corrected_data, index_array, start_num, end_num,
                                       sample_ids, ctx)
    return index_array, distance_array, null_ratio_array


def get_reference_parts(pca_corrected_data, masked_bins_per_chr_cum, ref_size, n_parts,
                        sample_ids, contexts=None, mode=0):
    """All `n_parts` row parts (the reference's --cpus split, newref_control.py:92-98), part p on
    device contexts[p % len(contexts)], driven from host threads (the C-ABI is re-entrant per
    context).  Returns [(indexes, distances, null_ratios)] in part order."""
    from concurrent.futures import ThreadPoolExecutor
    contexts = contexts or [_lib.default_context()]
    bincount = masked_bins_per_chr_cum[-1]

    def work(p, ctx):
        s, e = _get_part(p, n_parts, bincount)
        idx, dist = get_ref_for_rows(pca_corrected_data, masked_bins_per_chr_cum, ref_size, s, e,
                                     ctx, mode)
        nr = get_null_ratios(pca_corrected_data, idx, s, e, sample_ids, ctx)
        return idx, dist, nr

    if n_parts == 1:
        return [work(0, contexts[0])]

    def worker(t):          # a context is only ever used by ONE thread: parts t, t + T, ...
        return [(p, work(p, contexts[t])) for p in range(t, n_parts, len(contexts))]
    out = [None] * n_parts
    with ThreadPoolExecutor(max_workers=len(contexts)) as ex:
        for part in ex.map(worker, range(len(contexts))):
            for p, res in part:
                out[p] = res
    return out


def get_reference_dev(ctx, n_samples, masked_bins_per_chr_cum, ref_size, sample_ids, mode=0):
    """The whole pass (every row: indexes, distances, null ratios) on ONE device from the corrected
    matrix the PCA stage left in HBM (wcx_pca_corrected_dev; prep.prepare_dev): no host copy of X, only
    the three result tables come back.  Same device sequence as dist.GpuBackend.search -- the
    gonosomal passes' autosomal rows are the reference's dummies (newref_tools.py:186-191)."""
    import ctypes as C
    lib = ctx.lib
    cum, cum_p = _lib.i64_array(masked_bins_per_chr_cum)
    ids, ids_p = _lib.i32_array(sample_ids)
    B, S, k, m = int(cum[-1]), int(n_samples), int(ref_size), len(ids)
    dX = C.c_void_p()
    _lib.check(lib.wcx_pca_corrected_dev(ctx.h, C.byref(dX)))
    ct = int(cum[21]) if len(cum) > 22 else 0
    # a gonosomal pass: the autosomal target rows of indexes / distances are the reference's dummies (0 / 1,
    # newref_tools.py:186-191) -- 0.65 GB of constants per pass at 15 kb.  Only the gonosomal rows come back
    # from the device; the tables travel on as npz_io.PrefixConst (the .npz writer stores the constant rows
    # as a few hundred KB of deflate blocks, np.asarray() materialises them)
    r0 = ct if 0 < ct < B else 0
    idx = np.empty((B - r0, k), dtype=np.int32)
    dist = np.empty((B - r0, k), dtype=np.float64)
    nr = np.empty((B, m), dtype=np.float64)
    # the result tables are 0.8 GB of fresh host pages at 15 kb: worker threads touch them (page
    # faults, the expensive part of a device -> pageable-host copy) while the device searches
    from concurrent.futures import ThreadPoolExecutor
    from .npz_io import _THREADS
    ex = ThreadPoolExecutor(max_workers=_THREADS)
    touched = []
    for a in (idx, dist, nr):
        flat = a.reshape(-1).view(np.uint8)
        touched += [ex.submit(flat[o:o + (32 << 20)].fill, 0) for o in range(0, flat.size, 32 << 20)]
    try:
        d_idx, d_dist, d_nr = ctx.buffers([B * k * 4, B * k * 8, nr.nbytes])       # kept between passes
        if ct < B:
            _lib.check(lib.wcx_null_rank_prepare_dev(ctx.h, dX, B, S, ids_p, m))
        _lib.check(lib.wcx_newref_topk_dev(ctx.h, dX, B, S, cum_p, len(cum), 0, B, k, int(mode),
                                           d_idx, d_dist))
        if ct > 0:
            _lib.check(lib.wcx_null_ratios_dummy_dev(ctx.h, dX, B, S, 0, min(B, ct), ids_p, m, d_nr))
        if ct < B:
            _lib.check(lib.wcx_null_ratios_dev(ctx.h, dX, B, S, d_idx + ct * k * 4, ct, B, k, ids_p, m,
                                               d_nr + ct * m * 8))
        for f in touched:
            f.result()
        for b, a in zip((d_idx + r0 * k * 4, d_dist + r0 * k * 8, d_nr), (idx, dist, nr)):
            if a.nbytes:
                _lib.check(lib.wcx_memcpy_d2h(ctx.h, _lib.ptr(a), b, a.nbytes))
    finally:
        ex.shutdown(wait=True)
    if r0:
        from .npz_io import PrefixConst
        return PrefixConst(r0, 0, idx), PrefixConst(r0, 1.0, dist), nr
    return idx, dist, nr
