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


def train_pca(ref_data, pcacomp=5):
    """newref_tools.py:138-147: X = t / inverse_transform(transform(t)), returned as the
    Fortran-ordered (bins x samples) view of the sample-major matrix, + the PCA model."""
    t_data = np.ascontiguousarray(ref_data.T)                 # (S, B)
    mean = t_data.mean(axis=0)
    centred = t_data - mean
    # thin SVD through the S x S Gram matrix: exact, deterministic, O(S^2 B)
    gram = centred @ centred.T
    w, v = np.linalg.eigh(gram)
    order = np.argsort(w)[::-1][:pcacomp]
    sv = np.sqrt(np.maximum(w[order], 0.0))
    u = v[:, order]
    comps = (u.T @ centred) / np.where(sv > 0, sv, 1.0)[:, None]    # (pcacomp, B)
    # sklearn's svd_flip convention: largest |loading| of each component is positive
    signs = np.sign(comps[np.arange(comps.shape[0]), np.argmax(np.abs(comps), axis=1)])
    signs[signs == 0] = 1.0
    comps = comps * signs[:, None]
    transformed = centred @ comps.T
    inversed = transformed @ comps + mean
    corrected = t_data / inversed
    return corrected.T, PCAModel(comps, mean)


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
        w, v = np.linalg.eigh(gram)
        order = np.argsort(w)[::-1][:pcacomp]
        sv = np.ascontiguousarray(np.sqrt(np.maximum(w[order], 0.0)))
        u = np.ascontiguousarray(v[:, order])                             # (S, pcacomp)
        comps = np.empty((pcacomp, B))
        Xs = np.empty((S, B))
        d2m = np.empty(B) if want_dist else None
        _lib.check(ctx.lib.wcx_pca_finish(ctx.h, _lib.ptr(u), _lib.ptr(sv), pcacomp, _lib.ptr(comps),
                                          _lib.ptr(Xs), _lib.ptr(d2m)))
    finally:
        ctx.lib.wcx_pca_end(ctx.h)
    # sklearn's svd_flip convention: largest |loading| of each component is positive
    signs = np.sign(comps[np.arange(pcacomp), np.argmax(np.abs(comps), axis=1)])
    signs[signs == 0] = 1.0
    comps = comps * signs[:, None]
    out = (Xs.T, PCAModel(comps, mean))                                   # X: F-ordered (B, S) view
    return out + (d2m,) if want_dist else out


def filter_from_dist(dist_to_med):
    """newref_control.py:42-47 on a precomputed distance profile."""
    mad = np.median(np.abs(dist_to_med - np.median(dist_to_med)))
    cutoff = max(np.median(dist_to_med) + 10 * mad, 5.0)
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
