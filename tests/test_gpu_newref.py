"""GPU parity tests of the newref hot path: HIP (through the C-ABI) vs the oracle / golden
fixtures.  Bit-exact for indices AND distances; null ratios to 1e-12 (log2 is not correctly
rounded on either side)."""
import random

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import wcx_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nt():
    from wisecondorx_amd import newref_tools
    return newref_tools


def _X(g, key="Xs", nb=None):
    X = g[key].T
    return X if nb is None else np.asfortranarray(X[:nb])


@pytest.mark.parametrize("tag", ["A", "A1", "F", "M", "M2"])
def test_get_reference_golden(nt, g_search, tag):
    g = g_search
    mb = g[tag + "_mb"].tolist()
    cum = np.cumsum(mb).tolist()
    part, parts = g[tag + "_part"].tolist()
    X = _X(g, nb=cum[-1])
    # same seed as tests/golden/make_golden.py so random.sample picks the same null samples
    random.seed(1000 + len(mb) + part)
    idx, dist, nr = nt.get_reference(X, mb, cum, 40, part, parts)
    assert idx.dtype == np.int32 and dist.dtype == np.float64
    assert np.array_equal(idx, g[tag + "_idx"])
    assert np.array_equal(dist, g[tag + "_dist"])
    np.testing.assert_allclose(nr, g[tag + "_nr"], rtol=1e-12, atol=1e-13, equal_nan=True)


@pytest.mark.parametrize("tag,cs,ce,k", [("tie", 20, 50, 25), ("few", 10, 18, 40),
                                         ("nan", 20, 30, 45)])
def test_edge_cases_golden(nt, g_search, tag, cs, ce, k):
    """ties (stable by index), fewer than k candidates (-1/1e10 padding), NaN/inf/>=1e10
    candidates never admitted, NaN target row -> all padding."""
    X = _X(g_search, tag + "_Xs")
    B = X.shape[0]
    idx, dist = nt.get_ref_for_rows(X, [cs, ce, B], k, cs, ce)
    assert np.array_equal(idx, g_search[tag + "_idx"])
    assert np.array_equal(dist, g_search[tag + "_dist"])


@pytest.mark.parametrize("S,k,seed", [(100, 300, 0), (500, 300, 1), (33, 64, 2), (7, 1, 3)])
def test_seeded_vs_c_oracle(nt, S, k, seed):
    from wisecondorx_amd.synth import corrected_matrix
    rng = np.random.default_rng(seed)
    mb = rng.integers(40, 400, 24).tolist()
    mb[5] = 0                      # an empty chromosome
    X, mbpc, cum = corrected_matrix(mb, S, seed=seed)
    B = cum[-1]
    for n_chr in (22, 24):
        Xp = np.asfortranarray(X[:cum[n_chr - 1]])
        c = cum[:n_chr]
        s, e = O.get_part(1, 3, c[-1]) if n_chr == 22 else (0, c[-1])
        idx, dist = nt.get_ref_for_rows(Xp, c, k, s, e)
        oi, od = CO.get_reference_rows(np.ascontiguousarray(Xp.T), c, s, e, k)
        assert np.array_equal(idx, oi)
        assert np.array_equal(dist, od)


def test_empty_row_range(nt):
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([30, 30, 30], 8, seed=5)
    idx, dist = nt.get_ref_for_rows(X, cum, 10, 17, 17)
    assert idx.shape == (0, 10) and dist.shape == (0, 10)


def test_config1_size_properties(nt):
    """BASELINE config[1] shape (100 kb bins -- the reference's default bin size, main.py:377-380 --, S=100,
    k=300): size-independent properties on all rows, EVERY row bit for bit against the tiled C oracle,
    and the null ratios of a tenth of the rows against the NumPy oracle."""
    from wisecondorx_amd import _lib
    from wisecondorx_amd.synth import bins_per_chr, corrected_matrix
    bpc = [int(b * 0.93) for b in bins_per_chr(100000)[:22]]
    X, mbpc, cum = corrected_matrix(bpc, 100, seed=11)
    B, k = cum[-1], 300
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B)
    st = _lib.default_context().topk_stats()
    assert st["rows"] == B and st["fallback_rows"] == 0
    assert (np.diff(dist, axis=1) >= 0).all()                      # ascending
    assert (idx >= 0).all()
    own = np.repeat(np.array(mbpc), np.array(mbpc))
    assert (idx < (B - own)[:, None]).all()                        # chr-excluded index space
    srt = np.sort(idx, axis=1)
    assert (np.diff(srt, axis=1) > 0).all()                        # no duplicates
    Xs = np.ascontiguousarray(X.T)
    oi, od = CO.get_reference_rows_threaded(Xs, cum, 0, B, k)
    bad = np.flatnonzero((idx != oi).any(axis=1) | (dist != od).any(axis=1))
    assert bad.size == 0, "{} of {} rows differ (first {})".format(bad.size, B, bad[:5])
    ids = [3, 97, 41, 0, 58, 12, 77, 5]
    for lo in (0, B // 3, B - B // 30):
        hi = lo + B // 30
        nr = nt.get_null_ratios(X, idx[lo:hi], lo, hi, ids)
        with np.errstate(all="ignore"):
            enr = O.null_ratios(X, oi[lo:hi], lo, hi, ids)
        np.testing.assert_allclose(nr, enr, rtol=1e-12, atol=1e-13)


def test_config1_bench_cohort_all_rows(nt):
    """What bench.py's `config2_100kb` block runs: the PCA-corrected matrices of the 100-sample cohort at
    100 kb -- the A pass, every row, and the gonosomal rows of the F and M passes -- through the default
    policy, bit for bit against the tiled C oracle."""
    import bench
    from wisecondorx_amd import _lib
    passes = bench.make_full_workload(100000, 100)[1]
    for tag in ("A", "F", "M"):
        p = passes[tag]
        X = p["X"]
        cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
        B, k = cum[-1], 300
        r0 = 0 if tag == "A" else cum[21]
        idx, dist = nt.get_ref_for_rows(X, cum, k, r0, B, mode=0)
        st = _lib.default_context().topk_stats()
        assert st["rows"] == B - r0 and st["fallback_rows"] == 0, (tag, st)
        oi, od = CO.get_reference_rows_threaded(np.ascontiguousarray(np.asarray(X).T), cum, r0, B, k)
        bad = np.flatnonzero((idx != oi).any(axis=1) | (dist != od).any(axis=1))
        assert bad.size == 0, "{} pass: {} of {} rows differ (first {})".format(tag, bad.size, B - r0, bad[:5])


def test_null_ratio_index_space_quirk(nt):
    """The reference applies chr-excluded indices to the FULL vector (newref_tools.py:219-221)."""
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([50, 40, 30], 12, seed=9)
    idx, dist = nt.get_ref_for_rows(X, cum, 20, 0, cum[-1])
    ids = list(range(12))
    nr = nt.get_null_ratios(X, idx, 0, cum[-1], ids)
    onr = O.null_ratios(X, idx, 0, cum[-1], ids)
    np.testing.assert_allclose(nr, onr, rtol=1e-12, atol=1e-13)
    # padding index -1 wraps to the last bin like NumPy
    idx2 = idx.copy()
    idx2[:, -3:] = -1
    np.testing.assert_allclose(nt.get_null_ratios(X, idx2, 0, cum[-1], ids),
                               O.null_ratios(X, idx2, 0, cum[-1], ids), rtol=1e-12, atol=1e-13)


# ---------------------------------------------------------------- MFMA screen path (mode 2)
def _check_vs_c(nt, X, cum, k, s, e, mode):
    idx, dist = nt.get_ref_for_rows(X, cum, k, s, e, mode=mode)
    oi, od = CO.get_reference_rows(np.ascontiguousarray(np.asarray(X).T), cum, s, e, k)
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)


@pytest.mark.parametrize("S,k,seed", [(100, 300, 0), (33, 64, 2), (128, 512, 4), (16, 7, 5),
                                      (500, 300, 6), (200, 100, 7), (300, 40, 8)])
@pytest.mark.parametrize("mode", [1, 2])
def test_screen_and_exact_modes_vs_c_oracle(nt, S, k, seed, mode):
    from wisecondorx_amd.synth import corrected_matrix
    rng = np.random.default_rng(seed)
    mb = rng.integers(60, 420, 24).tolist()
    mb[7] = 0
    X, mbpc, cum = corrected_matrix(mb, S, seed=seed)
    _check_vs_c(nt, np.asfortranarray(X[:cum[21]]), cum[:22], k, 100, cum[21] - 37, mode)
    _check_vs_c(nt, X, cum, k, 0, cum[-1], mode)      # gonosomal pass (24 chromosomes)


def test_screen_ties_nan_inf_outliers(nt):
    """Screen path on data with exact ties (integers), NaN/inf rows, huge outliers and
    duplicated rows: the refine must reproduce the reference's order exactly."""
    rng = np.random.default_rng(12)
    mb = [700, 650, 600, 500, 450]
    cum = np.cumsum(mb).tolist()
    B, S, k = cum[-1], 24, 40
    X = np.asfortranarray(rng.integers(0, 4, (B, S)).astype(np.float64))
    X[5, 3] = np.nan
    X[900, 0] = np.inf
    X[1500, 2] = -np.inf
    X[2000, 1] = 3e5                 # d ~ 9e10 -> never admitted
    X[2100] = X[10]                  # duplicate rows -> zero distances
    X[2101] = X[10]
    X[800] = np.nan                  # NaN target row
    for mode in (1, 2):
        _check_vs_c(nt, X, cum, k, 0, B, mode)


def test_screen_real_valued_outlier_scale(nt):
    """A few rows 100x larger than the rest inflate the error budget (bigger shortlists) but
    must not change the result."""
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([900, 800, 700, 600], 64, seed=21)
    X = np.array(X, order="F")
    X[[3, 1000, 2500]] *= 100.0
    X[[7, 1200]] = 1.0               # zero-norm rows after centring
    _check_vs_c(nt, X, cum, 100, 0, cum[-1], 2)


@pytest.mark.parametrize("k", [700, 1500])
def test_large_refsize(nt, k):
    """Large refsizes on a small matrix (beyond the one-directional screen's 512, and beyond the
    symmetric sweep's 1024): the all-fp64 search; + the 16/32-entries-per-lane median of the null ratios."""
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([900, 700, 500, 300], 20, seed=k)
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, cum[-1])
    oi, od = CO.get_reference_rows(np.ascontiguousarray(np.asarray(X).T), cum, 0, cum[-1], k)
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
    ids = list(range(20))
    np.testing.assert_allclose(nt.get_null_ratios(X, idx, 0, cum[-1], ids),
                               O.null_ratios(X, idx, 0, cum[-1], ids), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("S", [12, 13, 28, 29, 108, 109, 124, 125, 156, 253, 444, 508,
                               509, 636, 637, 764, 765, 892, 893, 1020])
def test_screen_k_step_boundaries(nt, S):
    """Every instantiated K = 16*NK of the screen kernel, at the S values where the four augmented
    columns (norm / threshold) just fit or spill into the next k-step."""
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([800, 700, 600, 500], S, seed=S)
    k = 50
    rows = [(0, 130), (790, 930), (cum[-1] - 140, cum[-1])]
    Xc = np.ascontiguousarray(np.asarray(X).T)
    for s, e in rows:
        idx, dist = nt.get_ref_for_rows(X, cum, k, s, e, mode=2)
        oi, od = CO.get_reference_rows(Xc, cum, s, e, k)
        assert np.array_equal(idx, oi)
        assert np.array_equal(dist, od)


def test_beyond_screen_sample_limit_falls_back_to_exact(nt):
    """More than 1020 samples (K = 16 NK > 1024: the target fragments no longer fit a wave's registers)
    and a refsize beyond 1024 go to the all-fp64 search; mode 2 (screen required) refuses them."""
    from wisecondorx_amd import _lib
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([900, 700, 500], 1021, seed=3)
    idx, dist = nt.get_ref_for_rows(X, cum, 30, 100, 200)          # mode 0: auto
    oi, od = CO.get_reference_rows(np.ascontiguousarray(np.asarray(X).T), cum, 100, 200, 30)
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
    with pytest.raises(_lib.WcxError):
        nt.get_ref_for_rows(X, cum, 30, 100, 200, mode=2)


@pytest.mark.parametrize("k,screened", [(300, True), (512, None), (513, False)])
def test_one_directional_refsize_limit(nt, k, screened):
    """A row shard (one-directional sweep, shortlists of 1024 entries) screens refsizes up to 512 (where
    the lists of a small, noisy problem may overflow: those rows are redone exactly); one more goes to
    the exact kernel as a whole.  Same bits either way."""
    from wisecondorx_amd import _lib
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([1500, 1300, 1200, 1000], 40, seed=k)
    idx, dist = nt.get_ref_for_rows(X, cum, k, 200, 700)
    st = _lib.default_context().topk_stats()
    oi, od = CO.get_reference_rows(np.ascontiguousarray(np.asarray(X).T), cum, 200, 700, k)
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
    if screened is not None:
        assert (st["refined"] > 0) == screened       # (pairs the refine re-evaluated: the screen ran)
    if screened:
        assert st["fallback_rows"] <= 5


def test_null_ratios_nan_duplicates_and_ties(nt):
    """Rank-based medians: NaN in a null sample propagates like np.median, index rows made of one
    or two repeated bins (more than 64 equal ranks in a bucket), tied values, -0.0."""
    rng = np.random.default_rng(5)
    B, S, k = 700, 11, 90
    X = np.asfortranarray(1.0 + 0.1 * rng.standard_normal((B, S)))
    X[17, 3] = np.nan
    X[40:60, 5] = 1.25                 # ties inside a sample
    X[61, 6] = -0.0
    X[62, 6] = 0.0
    idx = rng.integers(0, B - 50, (B, k)).astype(np.int32)
    idx[0, :] = 17                     # all NaN for sample 3, constant otherwise
    idx[1, :] = 5                      # one bin repeated
    idx[2, :45] = 7
    idx[2, 45:] = 9                    # two bins, half / half: median = mean of the two values
    idx[3, :70] = -1                   # padding wraps to the last bin
    idx[4, :] = np.arange(40, 40 + k)  # covers the tie block
    idx[5, :] = np.where(np.arange(k) % 2 == 0, 61, 62)
    ids = list(range(S))
    nr = nt.get_null_ratios(X, idx, 0, B, ids)
    with np.errstate(all="ignore"):
        onr = O.null_ratios(X, idx, 0, B, ids)
    np.testing.assert_allclose(nr, onr, rtol=1e-12, atol=1e-13, equal_nan=True)
    assert np.isnan(nr[0, 3]) and not np.isnan(nr[0, 2])
    # few target rows (rows x 4 <= bins): selection on the high halves of the values' keys, settled on
    # the full doubles -- no ranking of every bin of every null sample; same bits as the rank path
    few = nt.get_null_ratios(X, idx[:50], 0, 50, ids)
    assert np.array_equal(few, nr[:50], equal_nan=True)
    X[5, :] = -0.0
    idx[6, :] = 5                      # np.median of negative zeros is +0 (its mean starts from 0.0)
    for rows in (B, 40):
        with np.errstate(all="ignore"):
            got = nt.get_null_ratios(X, idx[:rows], 0, rows, ids)
            want = O.null_ratios(X, idx[:rows], 0, rows, ids)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13, equal_nan=True)


@pytest.mark.parametrize("B,S,k,n_ids,kind", [
    (70, 5, 7, 5, "real"),            # fewer bins than sampled splitters, one sample group
    (4097, 17, 65, 16, "counts"),     # one bin past a bucket / scatter tile; integer data = heavy ties
    (5000, 9, 64, 9, "constant"),     # a constant null sample: every key equal but for the bin
    (12000, 130, 90, 128, "counts"),  # the most null samples the API takes (16 groups of 8)
    (9000, 12, 300, 11, "nan"),       # NaNs rank last; a group that is not full
])
def test_null_sample_ranking_shapes(nt, B, S, k, n_ids, kind):
    """The ranking of the null samples (k_rank_*: sample sort, buckets ranked by counting, equal values
    sharing a rank, one sample per XCD) on shapes around its tile and group sizes, all rows as targets so
    that the rank path runs (not the few-rows selection on key halves)."""
    rng = np.random.default_rng(B + n_ids)
    if kind == "counts":
        X = rng.poisson(6.0, (B, S)).astype(np.float64) / 4.0 + 0.25
    else:
        X = 1.0 + 0.1 * rng.standard_normal((B, S))
    if kind == "constant":
        X[:, 3] = 0.75
    if kind == "nan":
        X[rng.integers(0, B, 40), 2] = np.nan
        X[:, 7] = np.nan
    X = np.asfortranarray(X)
    idx = rng.integers(0, B, (B, k)).astype(np.int32)
    idx[1, :] = idx[1, 0]                                   # one bin repeated k times
    ids = [int(i) for i in rng.permutation(S)[:n_ids]]
    got = nt.get_null_ratios(X, idx, 0, B, ids)
    with np.errstate(all="ignore"):
        want = O.null_ratios(X, idx, 0, B, ids)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13, equal_nan=True)


def test_null_ratios_direct_kernel_equals_rank_path(nt):
    """Both selection paths on a realistic shape (k = 300, 100 null samples): the rows of a small
    shard (high-key kernel: bucket selection on hi32 of the values' keys, ties settled on the doubles)
    against the same rows taken from the whole-matrix call (rank path) -- identical bits; and against
    the oracle."""
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([2600, 2400, 2200, 2000, 1800], 120, seed=23)
    X = np.array(X, order="F")
    B, k = cum[-1], 300
    rng = np.random.default_rng(2)
    idx = rng.integers(0, B, (B, k)).astype(np.int32)
    idx[7, 250:] = -1
    X[123, 4] = np.nan
    idx[9, 17] = 123
    ids = rng.choice(120, 100, replace=False).tolist()
    full = nt.get_null_ratios(X, idx, 0, B, ids)
    for r0, r1 in ((0, 900), (5000, 5003), (B - 700, B)):
        part = nt.get_null_ratios(X, idx[r0:r1], r0, r1, ids)
        assert np.array_equal(part, full[r0:r1], equal_nan=True), (r0, r1)
    np.testing.assert_allclose(full[:300], O.null_ratios(X, idx[:300], 0, 300, ids), rtol=1e-12, atol=1e-13,
                               equal_nan=True)
    assert np.isnan(full[9, ids.index(4)]) if 4 in ids else True


@pytest.mark.parametrize("segments", ["2", "4"])
def test_candidate_segments(nt, segments, monkeypatch):
    """Row shards of a multi-GPU build have few target blocks: the sweep is then split over
    candidate segments with separate shortlists that are merged before the refine."""
    from wisecondorx_amd.synth import corrected_matrix
    monkeypatch.setenv("WCX_SCREEN_SEGMENTS", segments)
    X, mbpc, cum = corrected_matrix([900, 800, 700, 600, 500], 40, seed=17)
    X = np.array(X, order="F")
    X[11] = np.nan
    _check_vs_c(nt, X, cum, 120, 0, cum[-1], 2)
    _check_vs_c(nt, X, cum, 120, 850, 1900, 2)


def test_config2_full_size_15kb(nt):
    """BASELINE config[2] shape (15 kb bins, ~180 k rows, S=100, k=300) on the MFMA screen path:
    ALL rows bit-exact (indices, distances) vs the C oracle; null ratios of >= 10 % of the rows."""
    from wisecondorx_amd import _lib
    from wisecondorx_amd.synth import bins_per_chr, corrected_matrix
    bpc = [int(b * 0.95) for b in bins_per_chr(15000)[:22]]
    X, mbpc, cum = corrected_matrix(bpc, 100, seed=15)
    B, k = cum[-1], 300
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B, mode=2)
    st = _lib.default_context().topk_stats()
    assert st["rows"] == B and st["fallback_rows"] == 0
    assert (np.diff(dist, axis=1) >= 0).all()                      # ascending
    assert (idx >= 0).all()
    own = np.repeat(np.array(mbpc), np.array(mbpc))
    assert (idx < (B - own)[:, None]).all()                        # chr-excluded index space
    Xs = np.ascontiguousarray(X.T)
    # EVERY row against the C oracle (cache-tiled, on the host's cores)
    oi, od = CO.get_reference_rows_threaded(Xs, cum, 0, B, k)
    bad = np.flatnonzero((idx != oi).any(axis=1) | (dist != od).any(axis=1))
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert bad.size == 0, "{} of {} rows differ (first {}); sha256 idx {} dist {}".format(
        bad.size, B, bad[:5], sha(idx), sha(dist))
    print("config 2: {} / {} rows bit-exact; sha256 idx {} dist {}".format(B, B, sha(idx)[:16], sha(dist)[:16]))
    ids = list(range(0, 100, 7))
    nr_all = nt.get_null_ratios(X, idx, 0, B, ids)
    starts = sorted(set(list(range(0, B - 64, 512)) + [int(c) - 32 for c in cum[:-1]]))
    for r0 in starts:
        np.testing.assert_allclose(nr_all[r0:r0 + 64], O.null_ratios(X, idx[r0:r0 + 64], r0, r0 + 64, ids),
                                   rtol=1e-12, atol=1e-13)
    assert 64 * len(starts) >= 0.1 * B


def test_reference_parts_from_threads(nt):
    """The CLI's --gpus path: row parts driven from host threads, one context per thread (here two
    contexts on the same device), five parts."""
    from wisecondorx_amd import _lib
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([800, 700, 600, 500], 30, seed=31)
    k, ids = 50, [0, 5, 9, 17, 29]
    ctxs = [_lib.Context(0), _lib.Context(0)]
    parts = nt.get_reference_parts(X, cum, k, 5, ids, ctxs, mode=2)
    idx = np.concatenate([p[0] for p in parts])
    dist = np.concatenate([p[1] for p in parts])
    nr = np.concatenate([p[2] for p in parts])
    oi, od = CO.get_reference_rows(np.ascontiguousarray(np.asarray(X).T), cum, 0, cum[-1], k)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    np.testing.assert_allclose(nr, O.null_ratios(X, oi, 0, cum[-1], ids), rtol=1e-12, atol=1e-13)


def test_large_bin_count_5kb(nt):
    """5 kb bins (~590 k rows): index arithmetic beyond 2^19 rows / 2^32 bytes -- properties on all
    rows, oracle agreement on sampled rows, null ratios on a block."""
    from wisecondorx_amd.synth import bins_per_chr, corrected_matrix
    bpc = [int(b * 0.95) for b in bins_per_chr(5000)[:22]]
    S, k = 24, 60
    X, mbpc, cum = corrected_matrix(bpc, S, seed=5)
    B = cum[-1]
    assert B > 520000
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B, mode=2)
    assert (np.diff(dist, axis=1) >= 0).all() and (idx >= 0).all()
    own = np.repeat(np.array(mbpc), np.array(mbpc))
    assert (idx < (B - own)[:, None]).all()
    Xs = np.ascontiguousarray(np.asarray(X).T)
    rng = np.random.default_rng(2)
    rows = np.concatenate([[0, B - 1], rng.integers(0, B, 30)])
    for t in rows:
        c = int(np.searchsorted(cum, t, side="right"))
        cs = cum[c - 1] if c else 0
        oi, od = CO.topk_rows(Xs, cs, cum[c], int(t), int(t) + 1, k)
        assert np.array_equal(idx[t], oi[0]) and np.array_equal(dist[t], od[0])
    r0 = B - 64
    ids = [0, 7, 23]
    nr = nt.get_null_ratios(X, idx[r0:], r0, B, ids)
    np.testing.assert_allclose(nr, O.null_ratios(X, idx[r0:], r0, B, ids), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("segments", ["1", "3"])
def test_failed_threshold_estimates_are_redone_exactly(nt, segments, monkeypatch):
    """The sampled pre-pass hands every target an ESTIMATED threshold; a wrong (too tight) estimate
    must be caught by the final cut and the row redone by the exact kernel -- on the device, no host
    round trip.  Forced here with an absurd sample rank (r = 2 at a 1/4 sample): most rows fail,
    every result is still bit-identical to the C oracle."""
    from wisecondorx_amd import _lib
    from wisecondorx_amd.synth import corrected_matrix
    monkeypatch.setenv("WCX_SCREEN_SAMPLE", "4")
    monkeypatch.setenv("WCX_SCREEN_CUT_R", "2")
    monkeypatch.setenv("WCX_SCREEN_SEGMENTS", segments)
    X, mbpc, cum = corrected_matrix([1500, 1300, 1100, 900, 700, 500], 36, seed=23)
    k = 80
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, cum[-1], mode=2)
    st = _lib.default_context().topk_stats()
    assert st["fallback_rows"] > 100                    # the estimates did fail ...
    oi, od = CO.get_reference_rows(np.ascontiguousarray(np.asarray(X).T), cum, 0, cum[-1], k)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)   # ... and nothing shows
    # sane estimate on the same data: no row needs the exact kernel
    monkeypatch.delenv("WCX_SCREEN_CUT_R")
    idx2, dist2 = nt.get_ref_for_rows(X, cum, k, 0, cum[-1], mode=2)
    assert _lib.default_context().topk_stats()["fallback_rows"] == 0
    assert np.array_equal(idx2, oi) and np.array_equal(dist2, od)


def test_many_unresolvable_rows_take_the_tiled_redo(nt):
    """Data the fp16 screen cannot resolve for a tenth of the rows (a heavy-tailed spread of row
    norms: for a high-variance target every candidate lies inside the screen's error margin, its
    shortlist cannot be cut below capacity): thousands of rows are flagged, grouped by chromosome
    into 64-row tiles on the device and redone by the blocked exact kernel -- bit-exact, and in a
    fraction of a second (one-row redo blocks took ~0.2 s per row)."""
    import time
    from wisecondorx_amd import _lib
    rng = np.random.default_rng(340)
    mb = rng.integers(5000, 12000, 8)
    cum = np.cumsum(mb).tolist()
    B, S, k = cum[-1], 140, 150
    X = np.asfortranarray(1.0 + 0.1 * rng.gamma(2.0, 0.5, B)[:, None] * rng.standard_normal((B, S)))
    nt.get_ref_for_rows(X, cum, k, 0, 4096, mode=2)               # warm-up (allocations)
    t0 = time.perf_counter()
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B, mode=2)
    dt = time.perf_counter() - t0
    flagged = _lib.default_context().topk_stats()["fallback_rows"]
    assert flagged > 1000, flagged
    assert dt < 2.0, dt
    Xs = np.ascontiguousarray(np.asarray(X).T)
    for t in rng.choice(B, 60, replace=False):
        oi, od = CO.get_reference_rows(Xs, cum, int(t), int(t) + 1, k)
        assert np.array_equal(idx[t], oi[0]) and np.array_equal(dist[t], od[0]), int(t)


@pytest.mark.parametrize("k", [1, 2, 3, 63, 64, 65, 91, 128, 300, 513])
def test_null_ratios_few_rows_any_refsize(nt, k):
    """The no-ranking path (rows x 4 <= bins) for odd and even reference sizes (one or two middle
    order statistics), sizes around the 64-lane boundaries, with ties, repeats and a NaN -- against
    the oracle and against the same rows taken from the whole-matrix (rank path) call."""
    rng = np.random.default_rng(k)
    B, S = 2400, 13
    X = np.asfortranarray(1.0 + 0.05 * rng.standard_normal((B, S)))
    X[100:140, 2] = 1.0625                       # ties in one sample (equal doubles)
    X[200:260, 4] = 1.0 + np.arange(60) * 2.0 ** -40   # equal HIGH key halves, different doubles
    X[300, 7] = np.nan
    idx = rng.integers(0, B, (B, k)).astype(np.int32)
    idx[3, :] = rng.integers(100, 140, k)        # many ties
    idx[4, :] = rng.integers(200, 260, k)        # shared high halves: settled on the full doubles
    idx[5, :] = 205                              # one bin repeated (more than 64 equal elements at k >= 65)
    idx[6, k // 2] = 300                         # a NaN among the references of sample 7
    ids = list(range(S))
    few = nt.get_null_ratios(X, idx[:500], 0, 500, ids)
    with np.errstate(all="ignore"):
        want = O.null_ratios(X, idx[:500], 0, 500, ids)
    np.testing.assert_allclose(few, want, rtol=1e-12, atol=1e-13, equal_nan=True)
    assert np.isnan(few[6, 7])
    full = nt.get_null_ratios(X, idx, 0, B, ids)
    assert np.array_equal(few, full[:500], equal_nan=True)


@pytest.mark.parametrize("S,k", [(100, 100), (250, 120)])
def test_one_directional_sweep_hub_count_thresholds(nt, S, k, monkeypatch):
    """Round 6: the one-directional sweep (row shards, gonosomal passes, K < 256) takes its thresholds from
    COUNTS over the low-norm rows (csrc/screen_hub1.h) instead of the sampled pre-pass.  Same bits as the
    C oracle on a whole matrix, a row shard and a few-block range (candidate segments); the counters say
    the estimator ran (one cut per row and segment -- the final one --, trials chosen); with a count nobody
    reaches (forced) every row starts without an estimate and sweeps as the streaming top-k: same bits;
    and switched off (the sampled pre-pass): same bits."""
    from wisecondorx_amd import _lib
    from wisecondorx_amd.synth import corrected_matrix
    monkeypatch.setenv("WCX_SCREEN_SYM", "0")             # (K = 256 would take the symmetric sweep for all rows)
    X, mbpc, cum = corrected_matrix([7000, 6500, 6000, 5500, 5000, 4500], S, seed=S + 3)      # 34 500 rows
    B = cum[-1]
    Xs = np.ascontiguousarray(np.asarray(X).T)
    oi, od = CO.get_reference_rows_threaded(Xs, cum, 0, B, k)
    ctx = _lib.default_context()
    for r0, r1 in ((0, B), (9000, 21000), (B - 1800, B)):
        idx, dist = nt.get_ref_for_rows(X, cum, k, r0, r1, mode=2)
        st = ctx.topk_stats()
        bad = np.flatnonzero((idx != oi[r0:r1]).any(axis=1) | (dist != od[r0:r1]).any(axis=1))
        assert bad.size == 0, "rows [{}, {}): {} differ (first {})".format(r0, r1, bad.size, bad[:5])
        assert st["rows"] == r1 - r0 and st["fallback_rows"] == 0
        assert st["hub_trial_sum"] > 0 and st["hub_rows_without_estimate"] < 64       # the estimator ran
        assert st["appends"] < 4 * k * (r1 - r0)           # (the streaming top-k admits k ln(B / k) ~ 6 k per row)
    monkeypatch.setenv("WCX_HUB_TEST_FAIL", "1")
    idx, dist = nt.get_ref_for_rows(X, cum, k, 9000, 21000, mode=2)
    st = ctx.topk_stats()
    assert st["hub_rows_without_estimate"] == 12000 and st["fallback_rows"] == 0
    assert np.array_equal(idx, oi[9000:21000]) and np.array_equal(dist, od[9000:21000])
    monkeypatch.delenv("WCX_HUB_TEST_FAIL")
    monkeypatch.setenv("WCX_SCREEN_HUB", "0")
    idx, dist = nt.get_ref_for_rows(X, cum, k, 9000, 21000, mode=2)
    assert ctx.topk_stats()["hub_trial_sum"] == 0
    assert np.array_equal(idx, oi[9000:21000]) and np.array_equal(dist, od[9000:21000])


def test_one_directional_hub_thresholds_on_data_without_hubs(nt, monkeypatch):
    """Prototype-structured rows (every row = one of a few hundred prototypes + small noise): a row's
    neighbours are its prototype's other copies, NOT the low-norm rows -- the hub estimates come out far
    too loose, the lists overflow into the in-sweep cuts (rigorous) and nothing changes in the results."""
    from wisecondorx_amd import _lib
    monkeypatch.setenv("WCX_SCREEN_SYM", "0")
    rng = np.random.default_rng(12)
    mb = [6000, 5600, 5200, 4800, 4400, 4000]
    cum = np.cumsum(mb).tolist()
    B, S, k = cum[-1], 96, 60
    proto = 1.0 + 0.3 * rng.standard_normal((300, S))
    X = np.asfortranarray(proto[rng.integers(0, 300, B)] * (1.0 + 0.01 * rng.standard_normal((B, S))))
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B, mode=2)
    st = _lib.default_context().topk_stats()
    oi, od = CO.get_reference_rows_threaded(np.ascontiguousarray(X.T), cum, 0, B, k)
    bad = np.flatnonzero((idx != oi).any(axis=1) | (dist != od).any(axis=1))
    assert bad.size == 0, "{} of {} rows differ (first {})".format(bad.size, B, bad[:5])
    assert st["hub_trial_sum"] > 0
