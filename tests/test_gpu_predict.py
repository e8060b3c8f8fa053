"""GPU parity tests of the predict hot path against the reference-generated fixtures
(tests/golden/pipeline.npz) and the oracle.  Tolerance: north_star states 1e-5 relative for
z-scores / ratios; the kernels compute in fp64 and differ from NumPy only by summation order,
so the tests assert 1e-9."""
import argparse

import numpy as np
import pytest

from conftest import ref_dict_from_golden, sample_from_counts
from oracle import wcx_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.fixture(scope="module")
def pt():
    from wisecondorx_amd import predict_tools
    return predict_tools


@pytest.mark.parametrize("name", ["t0", "t1", "t2"])
def test_normalize_golden(pt, g_pipe, name):
    g = g_pipe
    ref = ref_dict_from_golden(g)
    gender = str(g[name + "_gender"])
    sample = sample_from_counts(g[name + "_counts"], g["cohort_bpc"])
    if gender == "M":
        sample["23"] = sample["23"] * 2
        sample["24"] = sample["24"] * 2
    args = argparse.Namespace(maskrepeats=5)
    cache = {}
    for tag, rg in (("A", "A"), ("G", gender)):
        r, z, w, n, mlr, mz = pt.normalize(args, sample, ref, rg, cache)
        exp = {k: g["{}_{}_{}".format(name, tag, k)] for k in ("r", "z", "w", "n", "mlr", "mz")}
        assert np.array_equal(n, exp["n"])
        np.testing.assert_allclose(r, exp["r"], rtol=RTOL, equal_nan=True)
        np.testing.assert_allclose(z, exp["z"], rtol=RTOL, atol=1e-9, equal_nan=True)
        np.testing.assert_allclose(w, exp["w"], rtol=1e-12)
        np.testing.assert_allclose(mlr, exp["mlr"], rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(mz, exp["mz"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(pt.get_optimal_cutoff(ref, 5, cache), g[name + "_cutoff"],
                               rtol=1e-12)


def test_batch_equals_single(pt, g_pipe):
    g = g_pipe
    ref = ref_dict_from_golden(g)
    xs = np.stack([g["t0_A_x"], g["t1_A_x"], g["t2_A_x"]])
    cache = {}
    cutoff = pt.get_optimal_cutoff(ref, 5, cache)
    zb, rb, nb, mlrb, mzb = pt.normalize_repeat_batch(xs, ref, cutoff, 0, 0, "", cache)
    for i in range(3):
        z, r, n, mlr, mz = pt.normalize_repeat(xs[i], ref, cutoff, 0, 0, "", cache)
        assert np.array_equal(z, zb[i], equal_nan=True) and np.array_equal(r, rb[i], equal_nan=True)
        assert np.array_equal(n, nb[i]) and mlr == mlrb[i] and mz == mzb[i]


@pytest.mark.parametrize("ns", [40, 131])
def test_large_batches_equal_single(pt, g_pipe, ns):
    """Batches of 16 .. 128 samples take the lane-per-sample passes + the rank medians of the last pass
    (counts and ratios bit-identical to the one-sample path, z to rounding); beyond 128 -- the ranking's
    limit -- the tiled kernel does the last pass (everything bit-identical).  Masked stretches, a zero
    stretch and a NaN included."""
    g = g_pipe
    ref = ref_dict_from_golden(g)
    rng = np.random.default_rng(ns)
    base = np.stack([g["t0_A_x"], g["t1_A_x"], g["t2_A_x"]])
    xs = base[rng.integers(0, 3, ns)] * (1.0 + 0.03 * rng.standard_normal((ns, base.shape[1])))
    xs[1, 50:90] = 0.0
    xs[2, 200:260] *= 1.6
    xs[3, 7] = np.nan
    cache = {}
    cutoff = pt.get_optimal_cutoff(ref, 5, cache)
    zb, rb, nb, mlrb, mzb = pt.normalize_repeat_batch(xs, ref, cutoff, 0, 0, "", cache)
    for i in list(range(6)) + [ns - 1]:
        z, r, n, mlr, mz = pt.normalize_repeat(xs[i], ref, cutoff, 0, 0, "", cache)
        assert np.array_equal(r, rb[i], equal_nan=True) and np.array_equal(n, nb[i])
        assert mlr == mlrb[i] or (np.isnan(mlr) and np.isnan(mlrb[i]))
        if ns > 128:
            assert np.array_equal(z, zb[i], equal_nan=True) and (mz == mzb[i] or (np.isnan(mz) and np.isnan(mzb[i])))
        else:
            np.testing.assert_allclose(zb[i], z, rtol=1e-9, atol=1e-9, equal_nan=True)


def test_seeded_k300_vs_oracle(pt):
    """k=300 (8 values per lane), masked bins, degenerate rows."""
    rng = np.random.default_rng(4)
    mb = [300, 260, 220, 200, 180, 160, 150, 140, 130, 120, 110, 100, 90, 80, 70, 60, 50, 50,
          40, 40, 30, 30]
    cum = np.cumsum(mb)
    B, k = int(cum[-1]), 300
    idx = np.empty((B, k), dtype=np.int32)
    dist = np.sort(rng.gamma(4.0, 0.05, (B, k)), axis=1)
    for c in range(22):
        cs = cum[c - 1] if c else 0
        n_cd = B - mb[c]
        for i in range(cs, cum[c]):
            idx[i] = rng.choice(n_cd, k, replace=False)
    dist[7, :] = 1e10            # a row with no reference bin under the cut-off
    idx[7, :] = -1
    ref = {"indexes": idx, "distances": dist, "masked_bins_per_chr": np.array(mb),
           "masked_bins_per_chr_cum": cum}
    x = 1.0 + 0.03 * rng.standard_normal(B)
    x[500:560] *= 1.5            # a gain -> |z| > 2.33 -> masked in later passes
    x[1000:1010] = 0.0
    cache = {}
    cutoff = pt.get_optimal_cutoff(ref, 5, cache)
    np.testing.assert_allclose(cutoff, O.get_optimal_cutoff(dist, 5), rtol=1e-12)
    np.testing.assert_allclose(pt.get_weights(ref, "", cache), O.get_weights(dist), rtol=1e-12)
    z, r, n, mlr, mz = pt.normalize_repeat(x, ref, cutoff, 0, 0, "", cache)
    oz, orr, on, omlr, omz = O.normalize_repeat(x, mb, cum, idx, dist, cutoff, 0, 0)
    assert np.array_equal(n, on)
    np.testing.assert_allclose(r, orr, rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(z, oz, rtol=RTOL, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose([mlr, mz], [omlr, omz], rtol=RTOL, atol=1e-12)
    assert np.isnan(z[7]) and n[7] == 0


def test_row_sharded_primitives_equal_batched_path(pt):
    """The multi-GPU building blocks (rows-handle, moment sweeps, per-pass kernel, nanmedian2)
    driven by wisecondorx_amd.dist at world=1, and with the rows split in two handles, give the
    same numbers as the single-call path."""
    import torch
    from wisecondorx_amd import _lib, dist as wd
    rng = np.random.default_rng(8)
    mb = [200, 180, 160, 140, 120, 100, 90, 80, 70, 60, 50, 50, 40, 40, 30, 30, 30, 20, 20, 20, 20, 20]
    cum = np.cumsum(mb)
    B, k = int(cum[-1]), 100
    idx = np.empty((B, k), dtype=np.int32)
    dist = np.sort(rng.gamma(4.0, 0.05, (B, k)), axis=1)
    for c in range(22):
        cs = cum[c - 1] if c else 0
        for i in range(cs, cum[c]):
            idx[i] = rng.choice(B - mb[c], k, replace=False)
    ref = {"indexes": idx, "distances": dist, "masked_bins_per_chr": np.array(mb),
           "masked_bins_per_chr_cum": cum}
    x = 1.0 + 0.03 * rng.standard_normal(B)
    x[300:340] *= 1.4
    cache = {}
    cutoff = pt.get_optimal_cutoff(ref, 5, cache)
    z, r, n, mlr, mz = pt.normalize_repeat(x, ref, cutoff, 0, 0, "", cache)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)   # share torch's stream
    be = wd.GpuBackend(ctx)
    d_idx, d_dist = torch.from_numpy(idx).to(dev), torch.from_numpy(dist).to(dev)
    xt = torch.from_numpy(x).to(dev)
    torch.cuda.synchronize()
    h = be.wrap_rows(d_idx, d_dist, B, k, cum, 0, B)
    c2 = wd.cutoff_sharded(be, h, 5, 1)
    np.testing.assert_allclose(c2, cutoff, rtol=1e-13)
    z2, r2, n2, mlr2, mz2 = wd.normalize_sharded(be, h, xt, B, 0, cutoff, 0, 1)
    ctx.sync()
    assert np.array_equal(z2.cpu().numpy(), z, equal_nan=True)
    assert np.array_equal(r2.cpu().numpy(), r, equal_nan=True)
    assert np.array_equal(n2.cpu().numpy(), n) and mlr2 == mlr and mz2 == mz
    be.free_ref(h)
    # two half handles: moments add up, one pass over each half reproduces the full pass
    half = B // 2
    h0 = be.wrap_rows(d_idx[:half].contiguous(), d_dist[:half].contiguous(), B, k, cum, 0, half)
    h1 = be.wrap_rows(d_idx[half:].contiguous(), d_dist[half:].contiguous(), B, k, cum, half, B - half)
    s0, c0 = be.moments(h0, float("inf"), 0.0, 0)
    s1, c1 = be.moments(h1, float("inf"), 0.0, 0)
    assert c0 + c1 == B * k
    np.testing.assert_allclose(s0 + s1, dist.sum(), rtol=1e-12)
    zB = torch.zeros(B, dtype=torch.float64, device=dev)
    rB, nB, lB = torch.zeros_like(zB), torch.zeros_like(zB), torch.zeros_like(zB)
    cout = xt.clone()
    for hh in (h0, h1):
        be.predict_pass(hh, xt, xt, cout, cutoff, 0, True, True, zB, rB, nB, lB)   # last: r too
    ctx.sync()
    oz, orr, on = O.normalize_once(x, x.copy(), mb, cum, idx, dist, cutoff, 0, 0)
    np.testing.assert_allclose(zB.cpu().numpy(), oz, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(rB.cpu().numpy(), orr, rtol=1e-9)
    assert np.array_equal(nB.cpu().numpy(), on)
    be.free_ref(h0)
    be.free_ref(h1)


def test_post_process_dev_matches_host_post_processing():
    """wcx_post_process_dev + wcx_weights_dev against get_post_processed_result x3 + log_trans on
    the host (predict_control.py:49-63, predict_tools.py:180-193): zero / negative / inf / nan
    ratios, ratio == 1, bins below minrefbins, NaN z, and a NaN weight -- which makes ALL weights 1
    (main.py:252-256); a second round without it checks the plain w / nanmean(w)."""
    import argparse
    import torch
    from wisecondorx_amd import _lib, predict_tools as pt
    rng = np.random.default_rng(21)
    n_bins = 5000
    mask = rng.random(n_bins) > 0.15
    B = int(mask.sum())
    r = np.exp(rng.normal(0, 0.1, B))
    r[[0, 3, 5, 7, 9, 11]] = [0.0, -1.0, np.inf, np.nan, 1.0, 1e-300]
    z = rng.normal(0, 1, B)
    z[2] = np.nan
    w = rng.uniform(0.5, 2, B)
    w[4] = np.nan
    n = rng.integers(100, 300, B).astype(float)
    m_lr, m_z = 0.0123, -0.2
    args = argparse.Namespace(minrefbins=150)
    bpc = [1200, 800, 3000]
    rem = {"mask": mask, "bins_per_chr": bpc}
    w_scaled = w / np.nanmean(w)
    if np.isnan(w_scaled).any() or np.isinf(w_scaled).any():          # main.py:252-256
        w_scaled = np.ones(len(w_scaled))
    assert np.all(w_scaled == 1.0)                                    # (the NaN weight triggers it)
    want = {"results_r": r, "results_z": z - m_z, "results_w": w_scaled}
    for key in want:
        want[key] = pt.get_post_processed_result(args, want[key], n, rem)
    pt.log_trans(want, m_lr)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(a).to(dev) for a in (z, r, n, w)]
    med = torch.tensor([m_lr, m_z], dtype=torch.float64, device=dev)
    pos = torch.from_numpy(np.flatnonzero(mask).astype(np.int32)).to(dev)
    host = torch.empty((3, n_bins), dtype=torch.float64).pin_memory()
    _lib.check(ctx.lib.wcx_post_process_dev(ctx.h, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                            d[3].data_ptr(), B, med[0:].data_ptr(), med[1:].data_ptr(),
                                            150.0, pos.data_ptr(), n_bins, host[0].data_ptr(),
                                            host[1].data_ptr(), host[2].data_ptr()))
    got = host.numpy()
    for row, key in enumerate(("results_r", "results_z", "results_w")):
        np.testing.assert_allclose(got[row], np.concatenate(want[key]), rtol=1e-12, atol=1e-15,
                                   equal_nan=True, err_msg=key)
    w[4] = 1.3                                                        # all weights finite now
    want_w = pt.get_post_processed_result(args, w / np.nanmean(w), n, rem)
    wd = {"results_r": pt.get_post_processed_result(args, r, n, rem), "results_z": pt.get_post_processed_result(args, z - m_z, n, rem),
          "results_w": want_w}
    pt.log_trans(wd, m_lr)
    d[3] = torch.from_numpy(w).to(dev)
    _lib.check(ctx.lib.wcx_post_process_dev(ctx.h, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                            d[3].data_ptr(), B, med[0:].data_ptr(), med[1:].data_ptr(),
                                            150.0, pos.data_ptr(), n_bins, host[0].data_ptr(),
                                            host[1].data_ptr(), host[2].data_ptr()))
    np.testing.assert_allclose(host.numpy()[2], np.concatenate(wd["results_w"]), rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("name", ["t0", "t1", "t2"])
def test_device_resident_full_predict_golden(pt, g_pipe, name):
    """dist.predict_full_dev: both normalisation passes, the A + gonosome merge, minrefbins /
    inflation / log2 transform (wcx_post_process_merge_dev) with everything device-resident, against
    the vectors the REFERENCE produced after get_post_processed_result + log_trans (_post_r/z/w of
    tests/golden/pipeline.npz; t1 additionally applies a blacklist there, so only its non-blacklisted
    bins are compared), then segment z of the golden segments on the device-resident vectors."""
    import torch
    from wisecondorx_amd import _lib
    from wisecondorx_amd import dist as wd
    g = g_pipe
    ref = ref_dict_from_golden(g)
    gender = str(g[name + "_gender"])
    ap = "." + gender
    sample = sample_from_counts(g[name + "_counts"], g["cohort_bpc"])
    if gender == "M":
        sample["23"] = sample["23"] * 2
        sample["24"] = sample["24"] * 2
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    A = {"idx": t(ref["indexes"]), "dist": t(ref["distances"]), "nr": t(ref["null_ratios"]),
         "cum": ref["masked_bins_per_chr_cum"]}
    G = {"idx": t(ref["indexes" + ap]), "dist": t(ref["distances" + ap]), "nr": t(ref["null_ratios" + ap]),
         "cum": ref["masked_bins_per_chr_cum" + ap]}
    xA = pt.project_pc(pt.coverage_normalize_and_mask(sample, ref, ""), ref, "")
    xG = pt.project_pc(pt.coverage_normalize_and_mask(sample, ref, ap), ref, ap)
    args = argparse.Namespace(minrefbins=20, maskrepeats=5, alpha=1e-4, seed=3)
    rem = {"args": args, "mask": ref["mask" + ap], "bins_per_chr": ref["bins_per_chr" + ap],
           "binsize": int(ref["binsize"]), "ref_gender": gender}
    rows, host = wd.predict_full_dev(be, A, G, t(xA), t(xG), rem, pt, want_host=True)
    keep = np.ones(host.shape[1], dtype=bool)
    if name == "t1":                                     # the fixture's blacklist zeroes these afterwards
        keep = g[name + "_post_r"] != 0
    for row, key in enumerate(("_post_r", "_post_z", "_post_w")):
        np.testing.assert_allclose(host[row][keep], g[name + key][keep], rtol=1e-9, atol=1e-12,
                                   equal_nan=True, err_msg=key)
    assert len(rows) >= len(rem["bins_per_chr"]) - 1 and all(len(r) == 5 for r in rows)
    if name != "t1":
        out = be._predict_full_bufs["out"]
        seg = np.ascontiguousarray(g[name + "_segs"], dtype=np.float64)
        off, off_p = _lib.i64_array(np.concatenate(([0], np.cumsum(rem["bins_per_chr"]))))
        z, nn = np.empty(len(seg)), np.empty(len(seg))
        _lib.check(ctx.lib.wcx_segment_z_dev(ctx.h, out[0].data_ptr(), out[2].data_ptr(), off_p, len(off) - 1,
                                             _lib.ptr(seg), len(seg), _lib.ptr(z), _lib.ptr(nn)))
        assert np.array_equal(nn == 0, g[name + "_segz_isstr"])
        got = np.where(nn == 0, np.nan, z)
        np.testing.assert_allclose(got, g[name + "_segz"], rtol=1e-9, atol=1e-9, equal_nan=True)


def test_device_prep_and_batch_equals_single(pt, g_pipe):
    """wcx_predict_prep_dev (coverage normalisation + mask + PCA projection on the device) against
    coverage_normalize_and_mask + project_pc (pinned to the reference through the golden normalise
    test); predict_batch_dev of the three golden samples (female reference) == three single-sample
    calls, bit for bit (per-bin vectors and result rows)."""
    import torch
    from wisecondorx_amd import _lib
    from wisecondorx_amd import dist as wd
    g = g_pipe
    ref = ref_dict_from_golden(g)
    samples = [sample_from_counts(g[n + "_counts"], g["cohort_bpc"]) for n in ("t0", "t1", "t2")]
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xs = {}
    for ap in ("", ".F"):
        d_counts = t(pt.sample_counts_matrix(samples, ref, ap))
        x = pt.prepare_batch_dev(d_counts, ref, ap, ctx)
        host = np.stack([pt.project_pc(pt.coverage_normalize_and_mask(s, ref, ap), ref, ap) for s in samples])
        np.testing.assert_allclose(x.cpu().numpy(), host, rtol=1e-12)
        xs[ap] = x
    A = {"idx": t(ref["indexes"]), "dist": t(ref["distances"]), "nr": t(ref["null_ratios"]),
         "cum": ref["masked_bins_per_chr_cum"]}
    G = {"idx": t(ref["indexes.F"]), "dist": t(ref["distances.F"]), "nr": t(ref["null_ratios.F"]),
         "cum": ref["masked_bins_per_chr_cum.F"]}
    args = argparse.Namespace(minrefbins=20, maskrepeats=5, alpha=1e-3, seed=3)
    rem = {"args": args, "mask": ref["mask.F"], "bins_per_chr": ref["bins_per_chr.F"],
           "binsize": int(ref["binsize"]), "ref_gender": "F"}
    rows_b, host_b = wd.predict_batch_dev(be, A, G, xs[""], xs[".F"], rem, pt, want_host=True)
    for i in range(3):
        rows_1, host_1 = wd.predict_full_dev(be, A, G, xs[""][i], xs[".F"][i], rem, pt, want_host=True)
        assert rows_1 == rows_b[i]
        assert np.array_equal(host_1, host_b[:, i, :], equal_nan=True)
