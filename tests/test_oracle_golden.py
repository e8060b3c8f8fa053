"""Pin the oracle (oracle/wcx_oracle.py) against outputs of the REFERENCE captured by
tests/golden/make_golden.py.  CPU only."""
import numpy as np
import pytest

from oracle import wcx_oracle as O
from conftest import ref_dict_from_golden, sample_from_counts


def _X(g, key="Xs", nb=None):
    Xs = g[key]
    X = Xs.T          # (B,S) Fortran-ordered view, like newref_tools.py:147
    return X if nb is None else np.asfortranarray(X[:nb])


@pytest.mark.parametrize("tag", ["A", "A1", "F", "M", "M2"])
def test_get_reference_bit_exact(g_search, tag):
    g = g_search
    mb = g[tag + "_mb"].tolist()
    cum = np.cumsum(mb).tolist()
    part, parts = g[tag + "_part"].tolist()
    X = _X(g, nb=cum[-1])
    idx, dist, nr = O.get_reference(X, mb, cum, 40, part, parts, g[tag + "_ids"].tolist())
    assert idx.dtype == np.int32
    assert np.array_equal(idx, g[tag + "_idx"])
    assert np.array_equal(dist, g[tag + "_dist"])
    assert np.array_equal(nr, g[tag + "_nr"], equal_nan=True)


def test_ties_stable_by_index(g_search):
    X = _X(g_search, "tie_Xs")
    chr_data = np.concatenate((X[:20], X[50:]))
    i, d = O.get_ref_for_bins(25, 20, 50, X, chr_data)
    assert np.array_equal(i, g_search["tie_idx"]) and np.array_equal(d, g_search["tie_dist"])


def test_fewer_than_k_padding(g_search):
    X = _X(g_search, "few_Xs")
    chr_data = np.concatenate((X[:10], X[18:]))
    i, d = O.get_ref_for_bins(40, 10, 18, X, chr_data)
    assert np.array_equal(i, g_search["few_idx"]) and np.array_equal(d, g_search["few_dist"])
    assert (i[:, 22:] == -1).all() and (d[:, 22:] == 1e10).all()


def test_nan_inf_never_admitted(g_search):
    X = _X(g_search, "nan_Xs")
    chr_data = np.concatenate((X[:20], X[30:]))
    i, d = O.get_ref_for_bins(45, 20, 30, X, chr_data)
    assert np.array_equal(i, g_search["nan_idx"]) and np.array_equal(d, g_search["nan_dist"])


@pytest.mark.parametrize("name", ["t0", "t1", "t2"])
def test_predict_normalize(g_pipe, name):
    g = g_pipe
    ref = ref_dict_from_golden(g)
    gender = str(g[name + "_gender"])
    sample = sample_from_counts(g[name + "_counts"], g["cohort_bpc"])
    if gender == "M":       # overall_tools.py:48-53
        sample["23"] = sample["23"] * 2
        sample["24"] = sample["24"] * 2
    for tag, rg in (("A", "A"), ("G", gender)):
        r, z, w, n, mlr, mz = O.normalize(sample, ref, rg, 5)
        for nm, v in zip(("r", "z", "w", "n", "mlr", "mz"), (r, z, w, n, mlr, mz)):
            np.testing.assert_allclose(np.asarray(v), g["{}_{}_{}".format(name, tag, nm)],
                                       rtol=1e-12, atol=1e-12, equal_nan=True, err_msg=nm)
    np.testing.assert_allclose(O.get_optimal_cutoff(ref["distances"], 5), g[name + "_cutoff"],
                               rtol=1e-14)


@pytest.mark.parametrize("name", ["t0", "t1", "t2"])
def test_post_processing_and_segment_z(g_pipe, name):
    g = g_pipe
    ref = ref_dict_from_golden(g)
    gender = str(g[name + "_gender"])
    ap = "." + gender
    A = {k: g["{}_A_{}".format(name, k)] for k in ("r", "z", "w", "n", "mlr", "mz")}
    G = {k: g["{}_G_{}".format(name, k)] for k in ("r", "z", "w", "n")}
    r, z, w, n = O.merge_autosomes_gonosomes(A["r"], A["z"], A["w"], A["n"], float(A["mz"]),
                                             G["r"], G["z"], G["w"], G["n"])
    mask, bpc = ref["mask" + ap], ref["bins_per_chr" + ap].tolist()
    nr_aut = ref["null_ratios"]
    nr_gon = ref["null_ratios" + ap][len(nr_aut):]
    nr = np.array([x.tolist() for x in nr_aut] + [x.tolist() for x in nr_gon], dtype=object)
    results = {"results_r": r, "results_z": z, "results_w": w, "results_nr": nr}
    for k in results:
        results[k] = O.get_post_processed_result(20, results[k], n, mask, bpc)
    O.log_trans(results, float(A["mlr"]))
    if name == "t1":
        O.apply_blacklist(results, [tuple(x) for x in g["t1_blacklist"]], int(ref["binsize"]))
    flat = lambda key: np.concatenate([np.asarray(c, dtype=float) for c in results[key]])
    for key, gk in (("results_r", "_post_r"), ("results_z", "_post_z"), ("results_w", "_post_w")):
        np.testing.assert_allclose(flat(key), g[name + gk], rtol=1e-13, atol=0, err_msg=key)
    segs = [[int(s[0]), int(s[1]), int(s[2]), float(s[3])] for s in g[name + "_segs"]]
    zs = O.get_z_score(segs, results["results_nr"], results["results_r"], results["results_w"])
    isstr = np.array([isinstance(v, str) for v in zs])
    assert np.array_equal(isstr, g[name + "_segz_isstr"])
    got = np.array([np.nan if isinstance(v, str) else float(v) for v in zs])
    np.testing.assert_allclose(got, g[name + "_segz"], rtol=1e-11, atol=1e-11, equal_nan=True)


def test_partition_helpers():
    # newref_tools.py:227-247 known answers (hand-evaluated)
    assert O.get_part(0, 3, 10) == (0, 3) and O.get_part(2, 3, 10) == (6, 10)
    assert O.split_by_chr(0, 10, [4, 4, 10]) == [[0, 0, 4], [1, 4, 4], [2, 4, 10]]  # empty chr -> empty region
    assert O.split_by_chr(2, 7, [4, 6, 10]) == [[0, 2, 4], [1, 4, 6], [2, 6, 7]]


# ---- the C restatement (oracle/wcx_oracle.c): pinned against the same reference outputs -------
@pytest.mark.parametrize("tag", ["A", "A1", "F", "M", "M2"])
def test_c_oracle_get_reference(g_search, tag):
    """oracle/wcx_oracle.c + the gonosomal dummy rule of c_oracle.get_reference_rows vs the
    reference's get_reference output (bit-exact indices and distances)."""
    from oracle import c_oracle as CO
    from oracle.wcx_oracle import get_part
    g = g_search
    mb = g[tag + "_mb"].tolist()
    cum = np.cumsum(mb).tolist()
    part, parts = g[tag + "_part"].tolist()
    Xs = np.ascontiguousarray(g["Xs"][:, :cum[-1]])
    s, e = get_part(part - 1, parts, cum[-1])
    idx, dist = CO.get_reference_rows(Xs, cum, s, e, 40)
    assert np.array_equal(idx, g[tag + "_idx"])
    assert np.array_equal(dist, g[tag + "_dist"])


@pytest.mark.parametrize("tag,cs,ce,k", [("tie", 20, 50, 25), ("few", 10, 18, 40),
                                         ("nan", 20, 30, 45)])
def test_c_oracle_edge_cases(g_search, tag, cs, ce, k):
    """ties (stable by index), fewer than k candidates, NaN/inf/>=1e10 never admitted."""
    from oracle import c_oracle as CO
    idx, dist = CO.topk_rows(g_search[tag + "_Xs"], cs, ce, cs, ce, k)
    assert np.array_equal(idx, g_search[tag + "_idx"])
    assert np.array_equal(dist, g_search[tag + "_dist"])


@pytest.mark.parametrize("tag", ["A", "A1", "F", "M", "M2"])
def test_tiled_c_oracle_get_reference(g_search, tag):
    """oracle/wcx_oracle_tiled.c (the cache-tiled, threaded verifier of the full-size GPU tests) vs the
    reference's own get_reference output: bit-exact indices and distances."""
    from oracle import c_oracle as CO
    from oracle.wcx_oracle import get_part
    g = g_search
    mb = g[tag + "_mb"].tolist()
    cum = np.cumsum(mb).tolist()
    part, parts = g[tag + "_part"].tolist()
    Xs = np.ascontiguousarray(g["Xs"][:, :cum[-1]])
    s, e = get_part(part - 1, parts, cum[-1])
    idx, dist = CO.get_reference_rows_threaded(Xs, cum, s, e, 40, threads=3, rows_per_task=17)
    assert np.array_equal(idx, g[tag + "_idx"])
    assert np.array_equal(dist, g[tag + "_dist"])


@pytest.mark.parametrize("tag,cs,ce,k", [("tie", 20, 50, 25), ("few", 10, 18, 40), ("nan", 20, 30, 45)])
def test_tiled_c_oracle_edge_cases_and_random(g_search, tag, cs, ce, k):
    """The tiled variant on the tie / short / NaN fixtures, and == the per-row C oracle on a seeded
    problem whose tiles, row blocks and chromosome gap do not line up with anything."""
    from oracle import c_oracle as CO
    Xs = np.ascontiguousarray(g_search[tag + "_Xs"])
    S, B = Xs.shape
    idx = np.empty((ce - cs, k), dtype=np.int32)
    dist = np.empty((ce - cs, k))
    assert CO.lib_tiled().wcxo_topk_rows_tiled(Xs.ctypes.data, B, S, cs, ce, cs, ce, k, idx.ctypes.data,
                                               dist.ctypes.data) == 0
    assert np.array_equal(idx, g_search[tag + "_idx"]) and np.array_equal(dist, g_search[tag + "_dist"])
    rng = np.random.default_rng(k)
    Xr = 1.0 + 0.05 * rng.standard_normal((37, 1500))
    Xr[:, rng.integers(0, 1500, 40)] = Xr[:, rng.integers(0, 1500, 40)]      # duplicate rows: ties
    cum = [401, 777, 1500]
    i0, d0 = CO.get_reference_rows(Xr, cum, 0, 1500, 60)
    i1, d1 = CO.get_reference_rows_threaded(Xr, cum, 0, 1500, 60, threads=4, rows_per_task=45)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)


# ---- BASELINE config 1: predict one sample at 1 Mb bins vs a 50-sample reference ------------
def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def config1_reference(g, search):
    """Reference dict of tests/golden/config1.npz with indexes/distances REBUILT by `search`
    (X, masked_bins_per_chr, cum, k) -> (idx, dist) from the reference's PCA-corrected matrices and
    proven identical to the reference's own tables through their SHA-256 digests."""
    ref = {k[5:]: g[k] for k in g.files if k.startswith("ref__")}
    gender = str(g["test_gender"])
    for tag, ap in (("A", ""), ("G", "." + gender)):
        X = g[tag + "_Xs"].T                                  # (B,S) Fortran-ordered view
        mb = ref["masked_bins_per_chr" + ap].tolist()
        idx, dist = search(X, mb, np.cumsum(mb).tolist(), 300)
        assert np.array_equal(idx[::97], g[tag + "_idx_rows"]), tag
        assert np.array_equal(dist[::97], g[tag + "_dist_rows"]), tag
        assert _sha(idx) == str(g[tag + "_idx_sha"]), tag + ": indexes differ from the reference"
        assert _sha(dist) == str(g[tag + "_dist_sha"]), tag + ": distances differ from the reference"
        ref["indexes" + ap], ref["distances" + ap] = idx, dist
    return ref, gender


def config1_sample(g, gender):
    sample = sample_from_counts(g["test_counts"], g["cohort_bpc"])
    if gender == "M":       # overall_tools.py:48-53
        sample["23"] = sample["23"] * 2
        sample["24"] = sample["24"] * 2
    return sample


def test_config1_predict_1mb_vs_50_sample_reference():
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "config1.npz"), allow_pickle=False)

    def search(X, mb, cum, k):
        i, d, _ = O.get_reference(X, mb, cum, k, 1, 1, [0])
        return i, d
    ref, gender = config1_reference(g, search)
    sample = config1_sample(g, gender)
    np.testing.assert_allclose(O.get_optimal_cutoff(ref["distances"], 5), g["cutoff"], rtol=1e-14)
    for tag, rg in (("A", "A"), ("G", gender)):
        res = O.normalize(sample, ref, rg, 5)
        for nm, v in zip(("r", "z", "w", "n", "mlr", "mz"), res):
            np.testing.assert_allclose(np.asarray(v), g["{}_{}".format(tag, nm)], rtol=1e-12,
                                       atol=1e-12, equal_nan=True, err_msg=tag + nm)


@pytest.mark.parametrize("tag,cs,ce,k", [("tie", 20, 50, 25), ("few", 10, 18, 40),
                                         ("nan", 20, 30, 45)])
def test_scan_restatement_matches_reference(g_search, tag, cs, ce, k):
    """O.topk_scan (the per-candidate Python loop bench.py times as cpu_baseline) on the
    reference's edge-case fixtures."""
    X = _X(g_search, tag + "_Xs")
    chr_data = np.concatenate((X[:cs], X[ce:]))
    for t in range(cs, ce):
        i, d = O.topk_scan(O.sq_distances(chr_data, X[t, :]), k)
        assert np.array_equal(i, g_search[tag + "_idx"][t - cs])
        assert np.array_equal(d, g_search[tag + "_dist"][t - cs])
