"""Input producers of the newref hot path (host NumPy):
bin mask, depth normalisation, PCA correction and the PCA-distance bin filter.

Same function names / argument meaning as the reference (newref_tools.py:77-147,
newref_control.py:24-80) so a reference user finds them where expected.  Differences, by
design: the PCA is an exact (deterministic) thin SVD instead of scikit-learn's unseeded
randomised solver (the reference is not reproducible run to run), and the
per-sample loops are vectorised.
"""
import logging

import numpy as np


def _stack(samples, chrs):
    by_chr = []
    for c in chrs:
        max_len = max(len(s[str(c)]) for s in samples)
        this_chr = np.zeros((max_len, len(samples)), dtype=float)
        for i, s in enumerate(samples):
            v = s[str(c)]
            this_chr[:len(v), i] = v
        by_chr.append(this_chr)
    return by_chr


def get_mask(samples):
    """newref_tools.py:77-102: bins with > 5 % of the median (non-zero) summed coverage."""
    by_chr = _stack(samples, range(1, 25))
    bins_per_chr = [a.shape[0] for a in by_chr]
    all_data = np.concatenate(by_chr, axis=0)
    all_data = all_data / np.sum(all_data, 0)
    sum_per_bin = np.sum(all_data, 1)
    median_cov = np.median(sum_per_bin[sum_per_bin > 0])
    return sum_per_bin > (0.05 * median_cov), bins_per_chr


def normalize_and_mask(samples, chrs, mask):
    """newref_tools.py:110-129: (bins x samples), each column divided by its sum over the
    chromosomes of this pass BEFORE masking, masked rows kept."""
    all_data = np.concatenate(_stack(samples, chrs), axis=0)
    all_data = all_data / np.sum(all_data, 0)
    return all_data[mask, :]


def normalize_and_mask_t(samples, chrs, mask):
    """normalize_and_mask in the layout the PCA stage wants: (samples x masked bins), sample-major.
    Bit-identical values (each count divided by the sample's total over the chromosomes of the
    pass -- an exact integer sum in any order -- masked bins dropped), but every sample is written
    as one contiguous row: no (bins x samples) matrix with strided column writes, no transpose
    copy afterwards.  0.3 s instead of 1 s per pass at 15 kb x 500 samples."""
    chrs = list(chrs)
    lens = [max(len(s[str(c)]) for s in samples) for c in chrs]
    off = np.concatenate(([0], np.cumsum(lens))).astype(int)
    mask = np.asarray(mask[:off[-1]], dtype=bool)
    keep = [np.flatnonzero(mask[off[j]:off[j + 1]]) for j in range(len(chrs))]
    moff = np.concatenate(([0], np.cumsum([len(k) for k in keep]))).astype(int)
    out = np.empty((len(samples), moff[-1]), dtype=float)
    for i, s in enumerate(samples):
        total = 0.0
        row = out[i]
        for j, c in enumerate(chrs):
            v = np.asarray(s[str(c)])
            total += float(v.sum(dtype=np.float64))
            k = keep[j]
            nv = len(v)
            if nv >= lens[j]:
                row[moff[j]:moff[j + 1]] = v[k]
            else:                                   # shorter than the longest sample: zero padded
                inside = k < nv
                row[moff[j]:moff[j + 1]] = 0.0
                row[moff[j]:moff[j + 1]][inside] = v[k[inside]]
        row /= total
    return out


class PCAModel:
    """The two attributes predict needs (reference .npz keys pca_components / pca_mean)."""

    def __init__(self, components, mean):
        self.components_ = components
        self.mean_ = mean


_LAPACKE = {}


def _lapacke_dsyevx():
    """LAPACKE_dsyevx of the OpenBLAS numpy itself has loaded (ILP64 build, symbols prefixed scipy_ /
    suffixed 64_), or None: importing scipy.linalg for the same routine costs ~0.3-1 s of start-up."""
    if "fn" not in _LAPACKE:
        fn = None
        try:
            import ctypes as C
            import glob
            import os
            libdir = os.path.join(os.path.dirname(np.__file__), os.pardir, "numpy.libs")
            for path in sorted(glob.glob(os.path.join(libdir, "libscipy_openblas64_*.so*"))):
                lib = C.CDLL(path)
                f = getattr(lib, "scipy_LAPACKE_dsyevx64_", None)
                if f is not None:
                    i64, dbl, vp = C.c_int64, C.c_double, C.c_void_p
                    f.restype = i64
                    f.argtypes = [C.c_int, C.c_char, C.c_char, C.c_char, i64, vp, i64, dbl, dbl, i64, i64, dbl,
                                  vp, vp, vp, i64, vp]
                    fn = f
                    break
        except Exception:           # any surprise in the private library layout: numpy's own eigh
            fn = None
        _LAPACKE["fn"] = fn
    return _LAPACKE["fn"]


def top_eigh(gram, k):
    """The k LARGEST eigenpairs of the symmetric matrix `gram` (S x S), eigenvalues descending: (w[k],
    v[S][k]).  Only pcacomp = 5 of the S pairs are ever used (newref_tools.py:138-147), so LAPACK's
    dsyevx (tridiagonal reduction, bisection, inverse iteration for the selected pairs; abstol = 2 x
    safe minimum = its most accurate setting) replaces the full divide-and-conquer solve of
    numpy.linalg.eigh: 4 ms instead of 45 ms at S = 500, five times per reference build.  Falls back to
    numpy.linalg.eigh when the routine is not reachable or reports a failure."""
    S = gram.shape[0]
    k = min(int(k), S)
    fn = _lapacke_dsyevx() if S > k else None
    if fn is not None and k > 0:
        a = np.array(gram, dtype=np.float64, order="F", copy=True)        # (destroyed by the routine)
        w = np.empty(S)
        z = np.empty((S, k), order="F")
        m = np.zeros(1, dtype=np.int64)
        ifail = np.zeros(S, dtype=np.int64)
        info = fn(102, b"V", b"I", b"U", S, a.ctypes.data, S, 0.0, 0.0, S - k + 1, S, 2.0 * np.finfo(float).tiny,
                  m.ctypes.data, w.ctypes.data, z.ctypes.data, S, ifail.ctypes.data)
        if info == 0 and int(m[0]) == k and np.all(np.isfinite(w[:k])) and np.all(np.isfinite(z)):
            return w[:k][::-1].copy(), np.ascontiguousarray(z[:, ::-1])
    w, v = np.linalg.eigh(gram)
    order = np.argsort(w)[::-1][:k]
    return w[order], v[:, order]


def train_pca(ref_data, pcacomp=5):
    """newref_tools.py:138-147: X = t / inverse_transform(transform(t)), returned as the
    Fortran-ordered (bins x samples) view of the sample-major matrix, + the PCA model."""
    t_data = np.ascontiguousarray(ref_data.T)                 # (S, B)
    mean = t_data.mean(axis=0)
    centred = t_data - mean
    # thin SVD through the S x S Gram matrix: exact, deterministic, O(S^2 B)
    gram = centred @ centred.T
    w, u = top_eigh(gram, pcacomp)
    sv = np.sqrt(np.maximum(w, 0.0))
    comps = (u.T @ centred) / np.where(sv > 0, sv, 1.0)[:, None]    # (pcacomp, B)
    # sklearn's svd_flip convention: largest |loading| of each component is positive
    signs = np.sign(comps[np.arange(comps.shape[0]), np.argmax(np.abs(comps), axis=1)])
    signs[signs == 0] = 1.0
    comps = comps * signs[:, None]
    transformed = centred @ comps.T
    inversed = transformed @ comps + mean
    corrected = t_data / inversed
    return corrected.T, PCAModel(comps, mean)


def _pca_finish(ctx, S, B, mean, gram, pcacomp, want_dist, want_X):
    """The host half of the device PCA: the S x S symmetric eigenproblem (LAPACK) between
    wcx_pca_begin* and wcx_pca_finish, then scikit-learn's sign convention."""
    from . import _lib
    w, u = top_eigh(gram, pcacomp)
    sv = np.ascontiguousarray(np.sqrt(np.maximum(w, 0.0)))
    u = np.ascontiguousarray(u)                                       # (S, pcacomp)
    comps = np.empty((pcacomp, B))
    Xs = np.empty((S, B)) if want_X else None
    d2m = np.empty(B) if want_dist else None
    _lib.check(ctx.lib.wcx_pca_finish(ctx.h, _lib.ptr(u), _lib.ptr(sv), pcacomp, _lib.ptr(comps),
                                      _lib.ptr(Xs), _lib.ptr(d2m)))
    # sklearn's svd_flip convention: largest |loading| of each component is positive
    signs = np.sign(comps[np.arange(pcacomp), np.argmax(np.abs(comps), axis=1)])
    signs[signs == 0] = 1.0
    comps = comps * signs[:, None]
    return Xs, PCAModel(comps, mean), d2m


def train_pca_gpu(ref_data, ctx, pcacomp=5, want_dist=False, sample_major=False):
    """train_pca on the MI355X (libwcx_hip.so: wcx_pca_begin / wcx_pca_finish): Gram matrix,
    components, reconstruction and ratio on the device, the S x S eigenproblem here (LAPACK).
    Same return values as train_pca (+ the filter's dist_to_med profile if want_dist).
    sample_major: ref_data already is the (S, B) matrix of normalize_and_mask_t."""
    from . import _lib
    if sample_major:
        t_data = np.ascontiguousarray(ref_data, dtype=np.float64)
    else:
        t_data = np.ascontiguousarray(ref_data.T, dtype=np.float64)      # (S, B) sample-major
    S, B = t_data.shape
    mean = np.empty(B)
    gram = np.empty((S, S))
    _lib.check(ctx.lib.wcx_pca_begin(ctx.h, _lib.ptr(t_data), B, S, _lib.ptr(mean), _lib.ptr(gram)))
    try:
        Xs, pca, d2m = _pca_finish(ctx, S, B, mean, gram, pcacomp, want_dist, True)
    finally:
        ctx.lib.wcx_pca_end(ctx.h)
    out = (Xs.T, pca)                                                     # X: F-ordered (B, S) view
    return out + (d2m,) if want_dist else out


class DeviceCounts:
    """The cohort's bin counts resident in HBM: int32 [S][n_bins], every sample's chromosomes 1..24
    laid out over the longest sample's bins per chromosome, zero padded (what the reference's
    np.zeros + copy loops build, newref_tools.py:80-91).  The masks (get_mask) and every pass's
    normalised matrix (prepare_dev) are computed from it on the device: the 729 MB float matrix of
    a 15 kb x 500 cohort is never built on the host."""

    def __init__(self, ctx, samples):
        from . import _lib
        import ctypes as C
        self.ctx = ctx
        self.S = len(samples)
        self.bins_per_chr = [max(len(s[str(c)]) for s in samples) for c in range(1, 25)]
        off = np.concatenate(([0], np.cumsum(self.bins_per_chr))).astype(np.int64)
        self.off = off
        self.n_bins = int(off[-1])
        # int32 vectors (what `convert` writes and npz_io loads): laid out by host threads
        # (wcx_layout_counts, csrc/tables.hip) -- 412 MB at 15 kb x 500 in ~10 ms instead of 12 000 slice
        # copies under the GIL; anything else takes the checked loop below
        from .predict_tools import _layout_counts_native
        counts = np.empty((self.S, self.n_bins), dtype=np.int32)
        native = _layout_counts_native(list(samples), self.bins_per_chr, counts)
        if not native:
            counts.fill(0)
        for i, s in enumerate(samples if not native else ()):
            row = counts[i]
            for c in range(24):
                v = np.asarray(s[str(c + 1)])
                if not np.issubdtype(v.dtype, np.integer):
                    raise TypeError("bin counts must be integers (sample {}, chromosome {}: {})".format(
                        i, c + 1, v.dtype))
                if v.dtype.itemsize > 4 or v.dtype == np.uint32:        # (convert writes int32)
                    if len(v) and (v.max() > 2 ** 31 - 1 or v.min() < -2 ** 31):
                        raise ValueError("bin count outside int32 (sample {}, chromosome {})".format(
                            i, c + 1))
                row[off[c]:off[c] + len(v)] = v
        self.d = C.c_void_p()
        _lib.check(ctx.lib.wcx_malloc(ctx.h, counts.nbytes, C.byref(self.d)))
        _lib.check(ctx.lib.wcx_memcpy_h2d(ctx.h, self.d, _lib.ptr(counts), counts.nbytes))

    def close(self):
        if self.d is not None and self.d.value:
            self.ctx.lib.wcx_free(self.ctx.h, self.d)
        self.d = None

    def get_mask(self, sel=None, want_sums=False):
        """get_mask (newref_tools.py:77-102) of the samples `sel` (default: all) on the device.
        want_sums: also return the per-bin coverage sums the threshold was applied to."""
        from . import _lib
        sel, sel_p = _lib.i32_array(range(self.S) if sel is None else sel)
        mask = np.empty(self.n_bins, dtype=np.uint8)
        sums = np.empty(self.n_bins) if want_sums else None
        _lib.check(self.ctx.lib.wcx_prep_mask_dev(self.ctx.h, self.d, self.n_bins, sel_p, len(sel),
                                                  _lib.ptr(mask), _lib.ptr(sums)))
        out = (mask.astype(bool), list(self.bins_per_chr))
        return out + (sums,) if want_sums else out

    def pca_begin(self, sel, last_chr, mask):
        """normalize_and_mask of the pass (chromosomes 1..last_chr, kept bins of `mask`) written into
        the PCA stage's device buffer + its Gram step.  Returns (S, B, mean, gram)."""
        from . import _lib
        sel, sel_p = _lib.i32_array(sel)
        n_pass = int(self.off[last_chr])
        pos, pos_p = _lib.i32_array(np.flatnonzero(mask[:n_pass]))
        S, B = len(sel), len(pos)
        mean = np.empty(B)
        gram = np.empty((S, S))
        _lib.check(self.ctx.lib.wcx_pca_begin_counts_dev(self.ctx.h, self.d, self.n_bins, sel_p, S, n_pass,
                                                         pos_p, B, _lib.ptr(mean), _lib.ptr(gram)))
        return S, B, mean, gram


def filter_from_dist(dist_to_med):
    """newref_control.py:42-47 on a precomputed distance profile."""
    med = np.median(dist_to_med)
    mad = np.median(np.abs(dist_to_med - med))
    cutoff = max(med + 10 * mad, 5.0)
    return dist_to_med > cutoff, cutoff


def pca_distance_filter(pca_corrected_data):
    """newref_control.py:38-47: bins far from the median profile.  Returns bad-bin mask."""
    med_prof = np.median(pca_corrected_data, axis=0)
    dist_to_med = np.sum((pca_corrected_data - med_prof) ** 2, axis=1)
    mad = np.median(np.abs(dist_to_med - np.median(dist_to_med)))
    cutoff = max(np.median(dist_to_med) + 10 * mad, 5.0)
    return dist_to_med > cutoff, cutoff


def prepare(samples, gender, mask, bins_per_chr, ctx=None, frozen=0):
    """The numerical part of newref_control.tool_newref_prep (newref_control.py:24-66).
    NOTE: like the reference, this mutates `mask` IN PLACE when the PCA-distance filter
    fires (newref_control.py:48-54).  ctx: a device context -> the PCA stage runs on the GPU.
    frozen > 0: the first `frozen` bins (the autosomes, in a gonosomal pass) keep their mask -- the
    filter of that pass may only drop later bins.  The reference lets the F / M pass drop
    autosomal bins the finished A reference still holds and then misaligns silently at predict
    time (main.py:242-275 inflates the merged vector with mask.F / mask.M); frozen = 0 reproduces
    that."""
    last_chr = {"A": 22, "F": 23}.get(gender, 24)
    bins_per_chr = list(bins_per_chr[:last_chr])
    mask = mask[:int(np.sum(bins_per_chr))]
    if ctx is not None:
        masked_t = normalize_and_mask_t(samples, range(1, last_chr + 1), mask)
        X, pca, d2m = train_pca_gpu(masked_t, ctx, want_dist=True, sample_major=True)
        bad, cutoff = filter_from_dist(d2m)
    else:
        masked_data = normalize_and_mask(samples, range(1, last_chr + 1), mask)
        X, pca = train_pca(masked_data)
        bad, cutoff = pca_distance_filter(X)
    if np.any(bad) and frozen:
        kept = bad & (np.where(mask)[0] < frozen)
        if np.any(kept):
            logging.info("Keeping {} anomalous autosomal bins (PCA distance): the autosomal reference "
                         "of this file already holds them".format(int(np.sum(kept))))
            bad = bad & ~kept
    if np.any(bad):
        logging.info("Removing {} anomalous bins based on PCA distance (cutoff={:.4f})".format(
            int(np.sum(bad)), cutoff))
        mask[np.where(mask)[0][bad]] = False
        if ctx is not None:
            masked_t = normalize_and_mask_t(samples, range(1, last_chr + 1), mask)
            X, pca = train_pca_gpu(masked_t, ctx, sample_major=True)
        else:
            X, pca = train_pca(normalize_and_mask(samples, range(1, last_chr + 1), mask))
    off = np.concatenate(([0], np.cumsum(bins_per_chr)))
    masked_bins_per_chr = [int(np.sum(mask[off[i]:off[i + 1]])) for i in range(len(bins_per_chr))]
    masked_bins_per_chr_cum = np.cumsum(masked_bins_per_chr).tolist()
    return {
        "X": X, "mask": mask.copy(), "bins_per_chr": np.array(bins_per_chr),
        "masked_bins_per_chr": np.array(masked_bins_per_chr),
        "masked_bins_per_chr_cum": np.array(masked_bins_per_chr_cum),
        "pca_components": pca.components_, "pca_mean": pca.mean_, "gender": gender,
    }


def prepare_dev(dc, sel, gender, mask, bins_per_chr, frozen=0, want_host_X=False):
    """prepare() from device-resident counts (DeviceCounts): same steps, same in-place mask update,
    same return value -- except that "X" is None unless want_host_X: the corrected matrix stays in
    HBM (wcx_pca_corrected_dev) for the search of this pass.  The caller ends the PCA stage
    (wcx_pca_end) when the last pass is done."""
    ctx = dc.ctx
    last_chr = {"A": 22, "F": 23}.get(gender, 24)
    bins_per_chr = list(bins_per_chr[:last_chr])
    mask = mask[:int(np.sum(bins_per_chr))]
    S, B, mean, gram = dc.pca_begin(sel, last_chr, mask)
    Xs, pca, d2m = _pca_finish(ctx, S, B, mean, gram, 5, True, False)
    bad, cutoff = filter_from_dist(d2m)
    if np.any(bad) and frozen:
        kept = bad & (np.where(mask)[0] < frozen)
        if np.any(kept):
            logging.info("Keeping {} anomalous autosomal bins (PCA distance): the autosomal reference "
                         "of this file already holds them".format(int(np.sum(kept))))
            bad = bad & ~kept
    if np.any(bad):
        logging.info("Removing {} anomalous bins based on PCA distance (cutoff={:.4f})".format(
            int(np.sum(bad)), cutoff))
        mask[np.where(mask)[0][bad]] = False
        S, B, mean, gram = dc.pca_begin(sel, last_chr, mask)
        Xs, pca, _ = _pca_finish(ctx, S, B, mean, gram, 5, False, False)
    if want_host_X:
        from . import _lib
        import ctypes as C
        dX = C.c_void_p()
        _lib.check(ctx.lib.wcx_pca_corrected_dev(ctx.h, C.byref(dX)))
        Xs = np.empty((S, B))
        _lib.check(ctx.lib.wcx_memcpy_d2h(ctx.h, _lib.ptr(Xs), dX, Xs.nbytes))
    off = np.concatenate(([0], np.cumsum(bins_per_chr)))
    masked_bins_per_chr = [int(np.sum(mask[off[i]:off[i + 1]])) for i in range(len(bins_per_chr))]
    return {
        "X": Xs.T if want_host_X else None, "n_samples": S, "mask": mask.copy(),
        "bins_per_chr": np.array(bins_per_chr),
        "masked_bins_per_chr": np.array(masked_bins_per_chr),
        "masked_bins_per_chr_cum": np.array(np.cumsum(masked_bins_per_chr).tolist()),
        "pca_components": pca.components_, "pca_mean": pca.mean_, "gender": gender,
    }
