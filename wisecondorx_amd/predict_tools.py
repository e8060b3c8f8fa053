"""Host-side mirror of the reference's predict numerical layer (predict_tools.py,
predict_control.py) -- same function names and argument meaning; the O(B*k) work runs in
libwcx_hip.so on the MI355X, the O(B) glue stays in NumPy.
"""
import os

import numpy as np

from . import _lib


class DeviceReference:
    """indexes{ap}/distances{ap} of a reference .npz resident in HBM (uploaded once per
    batch; SURVEY.md §8b `wcx_ref_upload`)."""

    def __init__(self, ref_file, ap="", ctx=None):
        self.ctx = ctx or _lib.default_context()
        idx = np.ascontiguousarray(ref_file["indexes{}".format(ap)], dtype=np.int32)
        dist = np.ascontiguousarray(ref_file["distances{}".format(ap)], dtype=np.float64)
        cum, cum_p = _lib.i64_array(ref_file["masked_bins_per_chr_cum{}".format(ap)])
        self.B, self.k = idx.shape
        self.chr_cum = cum
        h = _lib.vp()
        _lib.check(self.ctx.lib.wcx_ref_upload(self.ctx.h, _lib.ptr(idx), _lib.ptr(dist), self.B,
                                               self.k, cum_p, len(cum), _lib.C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.wcx_ref_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _dev(ref_file, ap, cache):
    key = ("devref", ap)
    if cache is not None and key in cache:
        return cache[key]
    d = DeviceReference(ref_file, ap)
    if cache is not None:
        cache[key] = d
    return d


def coverage_normalize_and_mask(sample, ref_file, ap):
    """The sample as one vector over the reference's bins (predict_tools.py:32-48): each
    chromosome truncated or zero-padded to the reference's bin count, divided by the total read
    count, masked bins dropped.  One preallocated vector, filled through the chromosome offsets."""
    bpc = np.asarray(ref_file["bins_per_chr{}".format(ap)], dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(bpc)))
    depth = np.zeros(int(starts[-1]), dtype=np.float64)
    for c, n_ref in enumerate(bpc):
        counts = sample[str(c + 1)]
        n = min(int(n_ref), len(counts))
        depth[starts[c]:starts[c] + n] = counts[:n]
    depth /= depth.sum()
    return depth[ref_file["mask{}".format(ap)]]


def _layout_counts_native(samples, bpc, out):
    """The native form of sample_counts_matrix's loop (csrc/tables.hip: wcx_layout_counts, host threads
    over the samples): the 79 MB of a 96-sample batch at 15 kb in under a millisecond instead of 10.
    Needs int32, C-contiguous count vectors and output (what npz_io loads); returns False otherwise."""
    n_chr = len(bpc)
    if out.dtype != np.int32 or not out.flags.c_contiguous or len(samples) == 0:
        return False
    ptrs = np.empty(len(samples) * n_chr, dtype=np.uintp)
    lens = np.empty(len(samples) * n_chr, dtype=np.int64)
    keys = [str(c + 1) for c in range(n_chr)]
    j = 0
    for sample in samples:
        for key in keys:
            a = sample[key]
            if type(a) is not np.ndarray or a.dtype != np.int32 or not a.flags.c_contiguous:
                return False
            ptrs[j] = a.__array_interface__["data"][0]
            lens[j] = a.shape[0]
            j += 1
    b = np.asarray(bpc, dtype=np.int64)
    lib = _lib.load()
    _lib.check(lib.wcx_layout_counts(ptrs.ctypes.data, lens.ctypes.data, len(samples), n_chr, b.ctypes.data,
                                     out.ctypes.data, min(16, os.cpu_count() or 1)))
    return True


def sample_counts_matrix(samples, ref_file, ap, out=None):
    """The bin counts of a batch of samples laid out over the reference's bins (each chromosome
    truncated or zero-padded to bins_per_chr{ap}, predict_tools.py:36-44) as int32 [ns][n_bins]: the
    input of prepare_batch_dev.  out: a preallocated int32 [>= ns][n_bins] array to fill (e.g. a view
    of pinned memory)."""
    bpc = [int(v) for v in ref_file["bins_per_chr{}".format(ap)]]
    starts = np.concatenate(([0], np.cumsum(bpc))).astype(np.int64)
    ns, n_bins = len(samples), int(starts[-1])
    if out is None:
        out = np.empty((ns, n_bins), dtype=np.int32)
    out = out[:ns]
    if _layout_counts_native(samples, bpc, out):
        return out

    # one concatenate per sample straight into its row (a thread pool over the samples is SLOWER here:
    # ~2 300 small slice copies fight for the GIL -- measured 70 ms against 14 ms for 96 samples)
    for i, sample in enumerate(samples):
        pieces = []
        for c, n_ref in enumerate(bpc):
            counts = sample[str(c + 1)]
            n = min(n_ref, len(counts))
            pieces.append(counts if n == len(counts) else counts[:n])
            if n < n_ref:
                pieces.append(np.zeros(n_ref - n, dtype=np.int32))
        np.concatenate(pieces, out=out[i], casting="unsafe")
    return out


def batch_counts_dev(samples, ref_file, aps, device, cache=None):
    """sample_counts_matrix of a batch for each reference suffix in `aps`, ON THE DEVICE (torch int32
    [ns][n_bins]): laid out straight into a pinned staging buffer (kept in `cache` between batches;
    a pageable 79 MB matrix at 15 kb x 96 samples costs 8 ms to upload, a pinned one 1.6 ms) and
    uploaded once per distinct bin layout -- the autosomal and the gonosomal reference of one file
    share bins_per_chr, so their matrices are the same tensor."""
    import torch
    cache = cache if cache is not None else {}
    done, out = {}, []
    for ap in aps:
        bpc = tuple(int(v) for v in ref_file["bins_per_chr{}".format(ap)])
        if bpc not in done:
            ns, n_bins = len(samples), int(sum(bpc))
            key = ("pinned_counts", len(done))
            slot = cache.get(key)
            if slot is None or slot[0].shape[0] < ns or slot[0].shape[1] != n_bins:
                slot = [torch.empty((ns, n_bins), dtype=torch.int32, pin_memory=True), None]
                cache[key] = slot
            if slot[1] is not None:
                slot[1].synchronize()                    # (the previous batch's upload has left the buffer)
            host = slot[0][:ns]
            sample_counts_matrix(samples, ref_file, ap, out=host.numpy())
            d = host.to(device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
            done[bpc] = d
        out.append(done[bpc])
    return out


def prepare_batch_dev(d_counts, ref_file, ap, ctx, cache=None):
    """coverage_normalize_and_mask + project_pc of a batch ON THE DEVICE (wcx_predict_prep_dev).
    d_counts: torch int32 tensor [ns][n_bins] on the device (sample_counts_matrix, uploaded).  Returns
    the projected vectors as a torch float64 tensor [ns][B].  The reference-side operands (mask
    positions, PCA mean and components) are uploaded once per (reference, suffix) into `cache`."""
    import torch
    cache = cache if cache is not None else {}
    key = ("prep", ap)
    if key not in cache:
        dev = d_counts.device
        mask = np.asarray(ref_file["mask{}".format(ap)], dtype=bool)
        cache[key] = (torch.from_numpy(np.flatnonzero(mask).astype(np.int32)).to(dev),
                      torch.from_numpy(np.ascontiguousarray(ref_file["pca_mean{}".format(ap)], dtype=np.float64)).to(dev),
                      torch.from_numpy(np.ascontiguousarray(ref_file["pca_components{}".format(ap)], dtype=np.float64)).to(dev))
    pos, mean, comps = cache[key]
    ns, n_bins = d_counts.shape
    B = int(pos.numel())
    x = torch.empty((ns, B), dtype=torch.float64, device=d_counts.device)
    _lib.check(ctx.lib.wcx_predict_prep_dev(ctx.h, d_counts.data_ptr(), int(ns), int(n_bins), pos.data_ptr(), B,
                                            mean.data_ptr(), comps.data_ptr(), int(comps.shape[0]), x.data_ptr()))
    return x


def project_pc(sample_data, ref_file, ap):
    """predict_tools.py:56-65 with the scikit-learn<=1.4.2 transform the reference pins
    (setup.cfg:42): x / (((x - mean) . C^T) . C + mean)."""
    comp = np.asarray(ref_file["pca_components{}".format(ap)])       # (n_comp, B)
    mean = ref_file["pca_mean{}".format(ap)]
    # (skinny products without BLAS: a threaded GEMV on 5 x B operands costs tens of ms in thread
    # wake-ups; einsum + five axpys take ~1 ms)
    t = np.einsum("cb,b->c", comp, np.asarray(sample_data) - mean)
    reconstructed = np.array(mean, dtype=float, copy=True)
    for c in range(comp.shape[0]):
        reconstructed += t[c] * comp[c]
    return sample_data / reconstructed


def get_optimal_cutoff(ref_file, repeats, cache=None):
    """predict_tools.py:74-82: always on the autosomal `distances`."""
    d = _dev(ref_file, "", cache)
    out = _lib.C.c_double()
    _lib.check(d.ctx.lib.wcx_cutoff(d.ctx.h, d.h, int(repeats), _lib.C.byref(out)))
    return out.value


def get_weights(ref_file, ap, cache=None):
    """predict_tools.py:152-155."""
    d = _dev(ref_file, ap, cache)
    out = np.empty(d.B, dtype=np.float64)
    _lib.check(d.ctx.lib.wcx_weights(d.ctx.h, d.h, _lib.ptr(out)))
    return out


def normalize_repeat(test_data, ref_file, optimal_cutoff, ct, cp, ap, cache=None):
    """predict_tools.py:94-108: returns (results_z, results_r, ref_sizes, m_lr, m_z)."""
    z, r, n, mlr, mz = normalize_repeat_batch(np.asarray(test_data)[None, :], ref_file,
                                              optimal_cutoff, ct, cp, ap, cache)
    return z[0], r[0], n[0], mlr[0], mz[0]


def normalize_repeat_batch(test_batch, ref_file, optimal_cutoff, ct, cp, ap, cache=None):
    """Batched normalize_repeat: test_batch float64[n_samples][B]."""
    d = _dev(ref_file, ap, cache)
    x = np.ascontiguousarray(test_batch, dtype=np.float64)
    ns, B = x.shape
    if B != d.B:
        raise ValueError("sample vector length {} != reference bins {}".format(B, d.B))
    Bp = B - int(ct)
    z = np.empty((ns, Bp))
    r = np.empty((ns, Bp))
    n = np.empty((ns, Bp))
    mlr = np.empty(ns)
    mz = np.empty(ns)
    _lib.check(d.ctx.lib.wcx_predict_normalize(
        d.ctx.h, d.h, _lib.ptr(x), ns, float(optimal_cutoff), int(ct), int(cp), _lib.ptr(z),
        _lib.ptr(r), _lib.ptr(n), _lib.ptr(mlr), _lib.ptr(mz)))
    return z, r, n, mlr, mz


def normalize(args, sample, ref_file, ref_gender, cache=None):
    """predict_control.py:21-39."""
    if ref_gender == "A":
        ap, cp, ct = "", 0, 0
    else:
        ap = ".{}".format(ref_gender)
        cp = 22
        ct = int(ref_file["masked_bins_per_chr_cum{}".format(ap)][cp - 1])
    sample = coverage_normalize_and_mask(sample, ref_file, ap)
    sample = project_pc(sample, ref_file, ap)
    results_w = get_weights(ref_file, ap, cache)[ct:]
    optimal_cutoff = get_optimal_cutoff(ref_file, args.maskrepeats, cache)
    results_z, results_r, ref_sizes, m_lr, m_z = normalize_repeat(
        sample, ref_file, optimal_cutoff, ct, cp, ap, cache)
    return results_r, results_z, results_w, ref_sizes, m_lr, m_z


# --------------------------------------------------------------------------- a14-a17 (host glue)
def merge_autosomes_gonosomes(res_a, res_g):
    """main.py:242-257: append gonosomal to autosomal results, centre z, cross-scale and
    renormalise the weights (all-ones fallback when not finite; the last return value tells the
    caller whether the weights were usable, main.py:252-256)."""
    results_r, results_z, results_w, ref_sizes, m_lr, m_z = res_a
    results_r_2, results_z_2, results_w_2, ref_sizes_2, _, _ = res_g
    with np.errstate(all="ignore"):
        r = np.append(results_r, results_r_2)
        z = np.append(results_z, results_z_2) - m_z
        w = np.append(results_w * np.nanmean(results_w_2), results_w_2 * np.nanmean(results_w))
        w = w / np.nanmean(w)
    weights_ok = bool(np.isfinite(w).all())
    if not weights_ok:
        w = np.ones(len(w))
    return r, z, w, np.append(ref_sizes, ref_sizes_2), weights_ok


def inflate_results(results, rem_input):
    """predict_tools.py:163-170 (vectorised): masked-out bins become 0."""
    mask = np.asarray(rem_input["mask"], dtype=bool)
    results = np.asarray(results)
    out = np.zeros((len(mask),) + results.shape[1:], dtype=float)
    out[mask] = results
    return out


def get_post_processed_result(args, result, ref_sizes, rem_input):
    """predict_control.py:49-63: zero bins with < minrefbins reference bins, inflate to the
    unmasked length, split per chromosome.  Returns a list of per-chromosome arrays."""
    result = np.array(result, dtype=float, copy=True)
    result[np.asarray(ref_sizes) < args.minrefbins] = 0
    inflated = inflate_results(result, rem_input)
    off = np.concatenate(([0], np.cumsum(rem_input["bins_per_chr"]))).astype(int)
    return [inflated[off[c]:off[c + 1]] for c in range(len(rem_input["bins_per_chr"]))]


def log_trans(results, log_r_median):
    """predict_tools.py:180-193 (vectorised, in place)."""
    for c in range(len(results["results_r"])):
        with np.errstate(all="ignore"):
            r = np.log2(np.asarray(results["results_r"][c], dtype=float))
        bad = ~np.isfinite(r)
        r[bad] = 0
        z = np.array(results["results_z"][c], dtype=float)
        w = np.array(results["results_w"][c], dtype=float)
        z[bad] = 0
        w[bad] = 0
        nz = r != 0
        r[nz] = r[nz] - log_r_median
        results["results_r"][c], results["results_z"][c], results["results_w"][c] = r, z, w


def post_process_fused(args, r, z, w, ref_sizes, log_r_median, rem_input):
    """get_post_processed_result x3 + log_trans (predict_control.py:49-63, predict_tools.py:163-193)
    in one vectorised pass over the masked vectors: same values, but the per-chromosome arrays are
    views of three contiguous vectors (so run_cbs / get_z_score need no concatenation).
    r, z, w, ref_sizes: masked-length vectors (z already shifted by m_z, w already scaled)."""
    mask = np.asarray(rem_input["mask"], dtype=bool)
    idx = rem_input.get("_mask_idx")
    if idx is None or len(idx) != len(r):
        idx = np.flatnonzero(mask)
        rem_input["_mask_idx"] = idx
    keep = np.asarray(ref_sizes) >= args.minrefbins
    with np.errstate(all="ignore"):
        lr = np.log2(np.where(keep, r, 0.0))
    good = np.isfinite(lr)                       # log2 of 0 / negative / inf / nan ratios -> bin zeroed
    lr = np.where(good, lr, 0.0)
    nz = lr != 0
    lr[nz] -= log_r_median
    good &= keep
    off = np.concatenate(([0], np.cumsum(rem_input["bins_per_chr"]))).astype(int)
    n_chr = len(rem_input["bins_per_chr"])
    results = {}
    for key, vec in (("results_r", lr), ("results_z", np.where(good, z, 0.0)),
                     ("results_w", np.where(good, w, 0.0))):
        full = np.zeros(len(mask))
        full[idx] = vec
        results[key] = [full[off[c]:off[c + 1]] for c in range(n_chr)]
    return results


_BED_CHR = {"X": 23, "Y": 24}


def _blacklist_spans(path, binsize):
    """(chromosome index, first bin, one-past-last bin) of every blacklist row
    (predict_tools.py:217-233: `chr` prefix optional, X/Y = 23/24, end bin inclusive)."""
    spans = []
    with open(path) as fh:
        for row in fh:
            name, start, end = row.strip().split("\t")
            name = name[3:] if name.startswith("chr") else name
            c = _BED_CHR.get(name) or int(name)
            spans.append((c - 1, int(int(start) / binsize), int(int(end) / binsize) + 1))
    return spans


def apply_blacklist(rem_input, results):
    """predict_tools.py:202-214: zero ratio, z and weight of the blacklisted bins (slice
    assignment per span; chrY rows are ignored when the result holds 23 chromosomes)."""
    n_chr = len(results["results_r"])
    for c, first, last in _blacklist_spans(rem_input["args"].blacklist, rem_input["binsize"]):
        if c >= n_chr or (n_chr < 24 and c == 23):
            continue
        lo, hi = max(first, 0), min(last, len(results["results_r"][c]))
        if lo >= hi:
            continue
        for key in ("results_r", "results_z", "results_w"):
            results[key][c][lo:hi] = 0


def blacklist_bin_indices(rem_input):
    """The flat (unmasked, concatenated per chromosome) bin indexes apply_blacklist zeroes -- for the
    device-resident predict (dist.predict_batch_dev), which holds r / z / w as [samples][n_bins]."""
    bpc = [int(v) for v in rem_input["bins_per_chr"]]
    off = np.concatenate(([0], np.cumsum(bpc))).astype(np.int64)
    n_chr = len(bpc)
    idx = []
    for c, first, last in _blacklist_spans(rem_input["args"].blacklist, rem_input["binsize"]):
        if c >= n_chr or (n_chr < 24 and c == 23):
            continue
        lo, hi = max(first, 0), min(last, bpc[c])
        if lo < hi:
            idx.append(np.arange(off[c] + lo, off[c] + hi, dtype=np.int64))
    return np.unique(np.concatenate(idx)) if idx else np.zeros(0, dtype=np.int64)


def _flatten(results, key, n_chr=None):
    """Per-chromosome arrays -> one contiguous float64 vector.  When the pieces already are
    consecutive views of one contiguous array (post_process_fused builds them that way) that array
    is returned as is: no copy."""
    parts = results[key][:n_chr] if n_chr is not None else results[key]
    base = getattr(parts[0], "base", None) if len(parts) else None
    if (isinstance(base, np.ndarray) and base.ndim == 1 and base.dtype == np.float64
            and base.flags.c_contiguous):
        base_addr = base.__array_interface__["data"][0]
        addr = parts[0].__array_interface__["data"][0]
        total = 0
        for part in parts:
            if (not isinstance(part, np.ndarray) or part.base is not base or part.dtype != np.float64
                    or part.ndim != 1 or part.strides != (8,)
                    or part.__array_interface__["data"][0] != addr + 8 * total):
                break
            total += len(part)
        else:
            start = (addr - base_addr) // 8
            return base[start:start + total]
    return np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=float) for c in parts]))


def _chr_offsets(results):
    n = [len(c) for c in results["results_r"]]
    return np.concatenate(([0], np.cumsum(n))).astype(np.int64)


def _null_matrix(results):
    """Per-bin null ratios -> dense [n_bins][m].  Accepts per-chromosome 2-D arrays or the
    reference's ragged lists (autosomal rows have min(S,100) columns, gonosomal rows
    min(S_gender,100); masked bins hold the int 0): short rows are padded with NaN, which the
    kernel ignores exactly like the reference ignores non-finite entries."""
    chrs = results["results_nr"]
    if all(isinstance(c, np.ndarray) and c.ndim == 2 for c in chrs):
        m = max(c.shape[1] for c in chrs)
        return np.vstack([c if c.shape[1] == m else
                          np.hstack([c, np.full((c.shape[0], m - c.shape[1]), np.nan)])
                          for c in chrs])
    rows = [row for c in chrs for row in c]
    m = max([len(row) for row in rows if np.ndim(row) > 0] + [1])
    out = np.full((len(rows), m), np.nan)
    for i, row in enumerate(rows):
        if np.ndim(row) > 0:
            out[i, :len(row)] = row
    return out


def attach_null_matrix(results_nr, ctx=None):
    """Upload the (inflated, per-chromosome) null ratios once; subsequent get_z_score calls with
    results["results_nr"] = ATTACHED use the device copy.  Bins whose ratio is 0 are skipped by
    the kernel exactly like overall_tools.py:98-100 skips them, so the per-sample zeroing of
    null rows (predict_control.py:50-52) has no effect on the result and is not needed."""
    ctx = ctx or _lib.default_context()
    nr = np.ascontiguousarray(_null_matrix({"results_nr": results_nr}), dtype=np.float64)
    _lib.check(ctx.lib.wcx_set_null_matrix(ctx.h, _lib.ptr(nr), nr.shape[0], nr.shape[1]))
    return nr.shape


def attach_null_matrix_dev(d_nr, mask, ctx=None):
    """attach_null_matrix for a null-ratio table that is already in HBM (torch tensor or any object
    with data_ptr()/shape: rows of the MASKED bins): inflated on the device, no host round trip."""
    ctx = ctx or _lib.default_context()
    mask8 = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
    _lib.check(ctx.lib.wcx_set_null_matrix_dev(ctx.h, _lib.ptr(d_nr.data_ptr()), int(d_nr.shape[0]),
                                               int(d_nr.shape[1]), _lib.ptr(mask8), len(mask8)))


ATTACHED = "attached"


def get_z_score(results_c, results, ctx=None):
    """overall_tools.py:88-119: per-segment z against the null ratios; "nan" string where the
    null mean/sd is undefined, exactly like the reference."""
    ctx = ctx or _lib.default_context()
    if not len(results_c):
        return []
    r, w = _flatten(results, "results_r"), _flatten(results, "results_w")
    off, off_p = _lib.i64_array(_chr_offsets(results))
    seg = np.ascontiguousarray([[s[0], s[1], s[2], s[3]] for s in results_c], dtype=np.float64)
    z = np.empty(len(seg))
    nn = np.empty(len(seg))
    if isinstance(results["results_nr"], str) and results["results_nr"] == ATTACHED:
        nr_p, m = None, 0
    else:
        nr = np.ascontiguousarray(_null_matrix(results), dtype=np.float64)
        nr_p, m = _lib.ptr(nr), nr.shape[1]
    _lib.check(ctx.lib.wcx_segment_z(ctx.h, _lib.ptr(r), _lib.ptr(w), nr_p, m, off_p, len(off) - 1,
                                     _lib.ptr(seg), len(seg), _lib.ptr(z), _lib.ptr(nn)))
    return ["nan" if nn[i] == 0 else float(z[i]) for i in range(len(seg))]


def run_cbs(results, ref_gender, alpha, binsize, seed, ctx=None):
    """The Rscript CBS.R call of exec_cbs (predict_tools.py:245-257) on the GPU.  Returns the
    reference's JSON rows as [chr0, s, e, r]."""
    ctx = ctx or _lib.default_context()
    n_chr = 24 if ref_gender == "M" else 23            # CBS.R:30-34
    n_chr = min(n_chr, len(results["results_r"]))
    r = _flatten(results, "results_r", n_chr)
    w = _flatten(results, "results_w", n_chr)
    n = [len(results["results_r"][c]) for c in range(n_chr)]
    off, off_p = _lib.i64_array(np.concatenate(([0], np.cumsum(n))))
    cap = 4096
    seg = np.empty((cap, 4))
    cnt = _lib.C.c_int()
    seed_v = 0 if seed is None else int(seed)
    _lib.check(ctx.lib.wcx_cbs(ctx.h, _lib.ptr(r), _lib.ptr(w), off_p, n_chr, float(alpha),
                               int(binsize), seed_v, _lib.ptr(seg), cap, _lib.C.byref(cnt)))
    return [[int(s[0]), int(s[1]), int(s[2]), float(s[3])] for s in seg[:cnt.value]]


def exec_cbs(rem_input, results, ctx=None):
    """predict_tools.py:242-263: CBS + segment z, rows [chr0, start, end, z, ratio]."""
    results_c = run_cbs(results, rem_input["ref_gender"], rem_input["args"].alpha,
                        rem_input["binsize"], rem_input["args"].seed, ctx)
    segment_z = get_z_score(results_c, results, ctx)
    return [results_c[i][:3] + [segment_z[i]] + [results_c[i][3]] for i in range(len(results_c))]


def exec_cbs_batch_dev(rem_input, d_r, d_w, ctx=None):
    """exec_cbs for a BATCH on DEVICE-resident per-bin vectors (torch tensors [n_samples][n_bins] as
    wcx_post_process_merge_dev leaves them; the null matrix attached): one level-synchronous CBS pass
    over all samples (wcx_cbs_batch_dev) and one segment-z call for all their segments
    (wcx_segment_z_batch_dev) -- no NumPy hop.  Returns per sample the rows [chr0, start, end, z, ratio]."""
    ctx = ctx or _lib.default_context()
    ns, n_bins = int(d_r.shape[0]), int(d_r.shape[1])
    bpc = [int(v) for v in rem_input["bins_per_chr"]]
    n_chr = min(24 if rem_input["ref_gender"] == "M" else 23, len(bpc))      # CBS.R:30-34
    off, off_p = _lib.i64_array(np.concatenate(([0], np.cumsum(bpc[:n_chr]))))
    cap = 4096
    seg = np.empty((ns, cap, 4))
    cnt = np.zeros(ns, dtype=np.int32)
    seed = rem_input["args"].seed
    _lib.check(ctx.lib.wcx_cbs_batch_dev(ctx.h, d_r.data_ptr(), d_w.data_ptr(), ns, n_bins, off_p, n_chr,
                                         float(rem_input["args"].alpha), int(rem_input["binsize"]),
                                         0 if seed is None else int(seed), _lib.ptr(seg), cap,
                                         _lib.ptr(cnt)))
    flat = np.ascontiguousarray(np.concatenate([seg[i, :cnt[i]] for i in range(ns)]))
    z = np.empty(len(flat))
    nn = np.empty(len(flat))
    # segment z sees every chromosome of the result vectors (24 for the male reference's lists)
    off_all, off_all_p = _lib.i64_array(np.concatenate(([0], np.cumsum(bpc))))
    if len(flat):
        _lib.check(ctx.lib.wcx_segment_z_batch_dev(ctx.h, d_r.data_ptr(), d_w.data_ptr(), ns, off_all_p,
                                                   len(bpc), _lib.ptr(flat), _lib.ptr(cnt), _lib.ptr(z),
                                                   _lib.ptr(nn)))
    out, o = [], 0
    for i in range(ns):
        out.append([[int(s[0]), int(s[1]), int(s[2]), "nan" if nn[o + j] == 0 else float(z[o + j]), float(s[3])]
                    for j, s in enumerate(flat[o:o + cnt[i]])])
        o += int(cnt[i])
    return out


def exec_cbs_dev(rem_input, d_r, d_w, ctx=None):
    """One sample of exec_cbs_batch_dev (d_r, d_w: [n_bins])."""
    return exec_cbs_batch_dev(rem_input, d_r.view(1, -1), d_w.view(1, -1), ctx)[0]


def run_cbs_batch(results_list, ref_gender, alpha, binsize, seed, ctx=None):
    """CBS of a batch of samples in ONE library call (wcx_cbs_batch): all chromosomes of all
    samples advance together on the device.  results_list: results dicts with the same chromosome
    layout.  Returns [[chr0, s, e, r], ...] per sample."""
    ctx = ctx or _lib.default_context()
    if not results_list:
        return []
    n_chr = 24 if ref_gender == "M" else 23            # CBS.R:30-34
    n_chr = min(n_chr, len(results_list[0]["results_r"]))
    n = [len(results_list[0]["results_r"][c]) for c in range(n_chr)]
    off, off_p = _lib.i64_array(np.concatenate(([0], np.cumsum(n))))
    n_bins = int(off[-1])
    ns = len(results_list)
    r = np.empty((ns, n_bins))
    w = np.empty((ns, n_bins))
    for i, res in enumerate(results_list):
        r[i] = _flatten(res, "results_r", n_chr)
        w[i] = _flatten(res, "results_w", n_chr)
    cap = 4096
    seg = np.empty((ns, cap, 4))
    cnt = np.zeros(ns, dtype=np.int32)
    seed_v = 0 if seed is None else int(seed)
    _lib.check(ctx.lib.wcx_cbs_batch(ctx.h, _lib.ptr(r), _lib.ptr(w), ns, n_bins, off_p, n_chr,
                                     float(alpha), int(binsize), seed_v, _lib.ptr(seg), cap,
                                     _lib.ptr(cnt)))
    return [[[int(s[0]), int(s[1]), int(s[2]), float(s[3])] for s in seg[i, :cnt[i]]]
            for i in range(ns)]


def segment_batch(results_list, rem_input, contexts, post=None):
    """Config 5: CBS + segment z of a batch of samples.  The segmentation of the
    whole batch is one level-synchronous device pass (run_cbs_batch) on contexts[0]; the segment
    z-scores are striped over `contexts` (each needs its null matrix attached) from host threads.
    results_list[i] is either a finished results dict or the argument of `post(i)` -> dict.
    Returns [results_c, ...] in input order (rows [chr, start, end, z, ratio])."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(results_list)

    def prep(t):
        return [(i, post(i) if post is not None else results_list[i])
                for i in range(t, n, len(contexts))]
    res = [None] * n
    with ThreadPoolExecutor(max_workers=len(contexts)) as ex:     # host post-processing in parallel
        for part in ex.map(prep, range(len(contexts))):
            for i, rr in part:
                res[i] = rr
    segs = run_cbs_batch(res, rem_input["ref_gender"], rem_input["args"].alpha, rem_input["binsize"],
                         rem_input["args"].seed, contexts[0])

    def zwork(t):
        ctx = contexts[t]
        out = []
        for i in range(t, n, len(contexts)):
            zs = get_z_score(segs[i], res[i], ctx)
            out.append((i, [[s[0], s[1], s[2], zs[j], s[3]] for j, s in enumerate(segs[i])]))
        return out
    results = [None] * n
    with ThreadPoolExecutor(max_workers=len(contexts)) as ex:
        for part in ex.map(zwork, range(len(contexts))):
            for i, rows in part:
                results[i] = rows
    return results
