"""Rows a1 / f1 on the device (prep.hip): the bin mask and the depth normalisation of a pass from
HBM-resident counts, the PCA stage fed from them, and the search on the corrected matrix that never
leaves HBM -- against the host functions of prep.py (pinned against the reference's in
tests/test_host.py) and the host-buffer entry points.  Bit-exact: same integer counts, same fp64
divisions, same order of the per-bin sums."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cohort(n=36, binsize=2000000, ragged=True):
    from wisecondorx_amd.overall_tools import gender_correct
    from wisecondorx_amd.synth import Cohort
    co = Cohort(binsize, struct_seed=5, female_y=0.1)
    samples, genders = co.cohort(n, reads=3e6)
    if ragged:       # samples whose chromosomes are shorter than the cohort's longest (zero padded)
        for i in (1, 7):
            for c in ("3", "23"):
                samples[i][c] = samples[i][c][:-2]
    samples = np.array([gender_correct(s, g) for s, g in zip(samples, genders)])
    return samples, np.array(genders)


def test_mask_on_device_equals_host():
    from wisecondorx_amd import _lib, prep
    samples, g = _cohort()
    dc = prep.DeviceCounts(_lib.default_context(0), samples)
    try:
        for sel in (np.arange(len(samples)), np.flatnonzero(g == "F"), np.flatnonzero(g == "M")):
            mh, bh = prep.get_mask(samples[sel])
            md, bd, sums = dc.get_mask(sel, want_sums=True)
            # the per-bin sums carry NumPy's own (pairwise) summation order: same bits as the host's
            stacked = np.concatenate(prep._stack(samples[sel], range(1, 25)), axis=0)
            host_sums = np.sum(stacked / np.sum(stacked, 0), 1)
            off_s = np.concatenate(([0], np.cumsum([a.shape[0] for a in prep._stack(samples[sel], range(1, 25))])))
            # (a subset's own longest chromosome may be shorter than the cohort's: the device mask
            # is laid out over the cohort's bins, the extra bins hold zeros -> masked out)
            assert bd == dc.bins_per_chr and sum(bd) >= sum(bh)
            off_h = np.concatenate(([0], np.cumsum(bh)))
            off_d = np.concatenate(([0], np.cumsum(bd)))
            for c in range(24):
                n = bh[c]
                assert np.array_equal(md[off_d[c]:off_d[c] + n], mh[off_h[c]:off_h[c + 1]]), c
                assert not md[off_d[c] + n:off_d[c + 1]].any()
                assert np.array_equal(sums[off_d[c]:off_d[c] + n], host_sums[off_s[c]:off_s[c + 1]]), c
            assert md.sum() > 100
    finally:
        dc.close()


def test_counts_must_be_integers():
    from wisecondorx_amd import _lib, prep
    samples, _ = _cohort(12, ragged=False)
    samples[3]["5"] = samples[3]["5"].astype(float)
    with pytest.raises(TypeError):
        prep.DeviceCounts(_lib.default_context(0), samples)


@pytest.mark.parametrize("gender", ["A", "F", "M"])
def test_prepare_and_search_from_device_counts(gender):
    """prepare_dev + get_reference_dev == prepare(ctx) + get_reference_parts, every bit: mask, PCA
    mean / components, X, indexes, distances, null ratios."""
    from wisecondorx_amd import _lib, newref_tools, prep
    samples, g = _cohort()
    ctx = _lib.default_context(0)
    sel = np.arange(len(samples)) if gender == "A" else np.flatnonzero(g == gender)
    total_mask, bpc = prep.get_mask(samples)
    frozen = 0 if gender == "A" else int(np.sum(bpc[:22]))
    mh = total_mask.copy()
    ph = prep.prepare(samples[sel], gender, mh, bpc, ctx=ctx, frozen=frozen)
    dc = prep.DeviceCounts(ctx, samples)
    try:
        md = total_mask.copy()
        pd = prep.prepare_dev(dc, sel, gender, md, bpc, frozen=frozen, want_host_X=True)
        assert np.array_equal(md, mh)
        for k in ("mask", "bins_per_chr", "masked_bins_per_chr", "masked_bins_per_chr_cum",
                  "pca_mean", "pca_components"):
            assert np.array_equal(pd[k], ph[k]), k
        assert np.array_equal(pd["X"], ph["X"])
        cum = [int(v) for v in ph["masked_bins_per_chr_cum"]]
        S = len(sel)
        ids = random.Random(3).sample(range(S), min(S, 100))
        hi, hd, hn = newref_tools.get_reference_parts(ph["X"], cum, 40, 1, ids, [ctx])[0]
        # the PCA buffers of prepare_dev are still in place (prepare()'s train_pca_gpu frees its own)
        pd2 = prep.prepare_dev(dc, sel, gender, total_mask.copy(), bpc, frozen=frozen)
        assert pd2["X"] is None and pd2["n_samples"] == S
        di, dd, dn = newref_tools.get_reference_dev(ctx, S, cum, 40, ids)
        assert np.array_equal(di, hi) and np.array_equal(dd, hd)
        assert np.array_equal(dn, hn, equal_nan=True)
        if gender != "A":
            assert np.all(di[:cum[21]] == 0) and np.all(dd[:cum[21]] == 1.0)
    finally:
        dc.close()
        ctx.lib.wcx_pca_end(ctx.h)


@pytest.mark.parametrize("nbytes", [1, 4097, (32 << 20) - 1, 32 << 20, (32 << 20) + 1, (96 << 20) + 12345])
def test_staged_host_copies_round_trip(nbytes):
    """wcx_memcpy_h2d / _d2h: from 32 MB on, the copy runs through two pinned halves with host threads
    moving the other half (api.hip: staged_copy) -- every byte must arrive, whatever the size's
    relation to the 32 MB chunk and to the threads' 4 KB-aligned pieces."""
    import ctypes as C
    from wisecondorx_amd import _lib
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(nbytes % 1000)
    src = rng.integers(0, 256, nbytes, dtype=np.uint8)
    dst = np.zeros(nbytes + 64, dtype=np.uint8)                     # (guard bytes after the end)
    d = C.c_void_p()
    _lib.check(ctx.lib.wcx_malloc(ctx.h, nbytes, C.byref(d)))
    try:
        _lib.check(ctx.lib.wcx_memcpy_h2d(ctx.h, d, _lib.ptr(src), nbytes))
        _lib.check(ctx.lib.wcx_memcpy_d2h(ctx.h, _lib.ptr(dst), d, nbytes))
    finally:
        ctx.lib.wcx_free(ctx.h, d)
    assert np.array_equal(dst[:nbytes], src) and not dst[nbytes:].any()


def test_counts_entry_points_reject_bad_arguments():
    from wisecondorx_amd import _lib, prep
    samples, _ = _cohort(12, ragged=False)
    ctx = _lib.default_context(0)
    dc = prep.DeviceCounts(ctx, samples)
    try:
        mask = np.ones(dc.n_bins, dtype=bool)
        sel, sel_p = _lib.i32_array([0, 1, 2])
        pos, pos_p = _lib.i32_array(np.arange(10))
        mean, gram = np.empty(10), np.empty((3, 3))
        lib = ctx.lib
        # more kept bins than the pass has bins; a pass longer than the matrix; one sample only
        assert lib.wcx_pca_begin_counts_dev(ctx.h, dc.d, dc.n_bins, sel_p, 3, 5, pos_p, 10, _lib.ptr(mean),
                                            _lib.ptr(gram)) != 0
        assert lib.wcx_pca_begin_counts_dev(ctx.h, dc.d, dc.n_bins, sel_p, 3, dc.n_bins + 1, pos_p, 10,
                                            _lib.ptr(mean), _lib.ptr(gram)) != 0
        assert lib.wcx_pca_begin_counts_dev(ctx.h, dc.d, dc.n_bins, sel_p, 1, dc.n_bins, pos_p, 10,
                                            _lib.ptr(mean), _lib.ptr(gram)) != 0
        assert b"bad sizes" in lib.wcx_last_error()
        out = np.empty(dc.n_bins, dtype=np.uint8)
        assert lib.wcx_prep_mask_dev(ctx.h, dc.d, dc.n_bins, sel_p, 0, _lib.ptr(out), None) != 0
        assert mask.all()
    finally:
        dc.close()
