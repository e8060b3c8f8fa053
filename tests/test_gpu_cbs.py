"""GPU tests of a14-a17: post-processing + segment z against the reference-generated fixtures,
and CBS.  CBS breakpoints are PARITY UNPINNED (DNAcopy is not part of the reference repo and R
is absent): the tests pin (i) the reference-owned CBS.R logic around the DNAcopy call against
the oracle restatement, (ii) recovery of planted change-points, (iii) determinism in the seed."""
import argparse

import numpy as np
import pytest

from conftest import ref_dict_from_golden
from oracle import wcx_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    from wisecondorx_amd import predict_tools
    return predict_tools


@pytest.mark.parametrize("name", ["t0", "t1", "t2"])
def test_post_processing_and_segment_z_golden(pt, g_pipe, name, tmp_path):
    g = g_pipe
    ref = ref_dict_from_golden(g)
    gender = str(g[name + "_gender"])
    ap = "." + gender
    resA = [g["{}_A_{}".format(name, k)] for k in ("r", "z", "w", "n", "mlr", "mz")]
    resG = [g["{}_G_{}".format(name, k)] for k in ("r", "z", "w", "n", "mlr", "mz")]
    r, z, w, n, weights_ok = pt.merge_autosomes_gonosomes(resA, resG)
    assert weights_ok
    args = argparse.Namespace(minrefbins=20)
    rem = {"args": args, "mask": ref["mask" + ap], "bins_per_chr": ref["bins_per_chr" + ap],
           "binsize": int(ref["binsize"])}
    nr_aut = ref["null_ratios"]
    nr_gon = ref["null_ratios" + ap][len(nr_aut):]
    m = max(nr_aut.shape[1], nr_gon.shape[1])
    nr = np.full((len(nr_aut) + len(nr_gon), m), np.nan)
    nr[:len(nr_aut), :nr_aut.shape[1]] = nr_aut
    nr[len(nr_aut):, :nr_gon.shape[1]] = nr_gon
    results = {"results_r": r, "results_z": z, "results_w": w, "results_nr": nr}
    for k in results:
        results[k] = pt.get_post_processed_result(args, results[k], n, rem)
    pt.log_trans(results, float(resA[4]))
    if name == "t1":
        bl = tmp_path / "bl.bed"
        bl.write_text("".join("\t".join(row) + "\n" for row in g["t1_blacklist"]))
        args.blacklist = str(bl)
        pt.apply_blacklist(rem, results)
    flat = lambda key: np.concatenate([np.asarray(c, dtype=float) for c in results[key]])
    for key, gk in (("results_r", "_post_r"), ("results_z", "_post_z"), ("results_w", "_post_w")):
        np.testing.assert_allclose(flat(key), g[name + gk], rtol=1e-12, atol=0, err_msg=key)
    segs = [[int(s[0]), int(s[1]), int(s[2]), float(s[3])] for s in g[name + "_segs"]]
    zs = pt.get_z_score(segs, results)
    isstr = np.array([isinstance(v, str) for v in zs])
    assert np.array_equal(isstr, g[name + "_segz_isstr"])
    got = np.array([np.nan if isinstance(v, str) else float(v) for v in zs])
    np.testing.assert_allclose(got, g[name + "_segz"], rtol=1e-9, atol=1e-9, equal_nan=True)


def _noise_results(rng, n_per_chr, sd=0.05):
    r = [rng.normal(0, sd, n) for n in n_per_chr]
    w = [rng.uniform(0.5, 2.0, n) for n in n_per_chr]
    return {"results_r": r, "results_w": w}


def test_cbs_planted_changepoints(pt):
    rng = np.random.default_rng(1)
    n_per_chr = [600] * 4 + [300] * 19
    res = _noise_results(rng, n_per_chr)
    res["results_r"][2][200:260] += 0.4            # interior gain
    res["results_r"][5][:80] -= 0.5                # loss at the start of a chromosome
    res["results_r"][0][100:104] = 0               # a few blacklisted bins (0 = missing)
    segs = pt.run_cbs(res, "F", 1e-4, 100000, 7)
    by_chr = {}
    for c, s, e, r in segs:
        by_chr.setdefault(c, []).append((s, e, r))
    assert len(by_chr) == 23
    c2 = sorted(by_chr[2])
    assert len(c2) == 3
    assert abs(c2[1][0] - 200) <= 2 and abs(c2[1][1] - 260) <= 2 and abs(c2[1][2] - 0.4) < 0.05
    c5 = sorted(by_chr[5])
    assert len(c5) == 2 and abs(c5[0][1] - 80) <= 2 and abs(c5[0][2] + 0.5) < 0.05
    # pure-noise chromosomes stay in one piece
    assert sum(len(v) for c, v in by_chr.items() if c not in (2, 5)) == 21
    # deterministic in the seed
    assert pt.run_cbs(res, "F", 1e-4, 100000, 7) == segs


def test_cbs_wrapper_logic_matches_oracle(pt):
    """No change-points (noise, strict alpha): the output is decided by the reference-owned
    CBS.R code only -- all-NA chromosome dropped, split over long NA runs (pieces start AT the
    last NA bin), >=2-bin rule, weighted re-mean, weight 0 -> 1."""
    rng = np.random.default_rng(3)
    n_per_chr = [400] * 24
    res = _noise_results(rng, n_per_chr, sd=0.02)
    res["results_r"][1][:] = 0                     # all-NA chromosome
    res["results_r"][3][100:160] = 0               # long NA run (60 > 20 at 100 kb) -> split
    res["results_r"][4][50:60] = 0                 # short NA run -> no split
    res["results_r"][6][0:30] = 0                  # leading NAs
    res["results_r"][6][395:] = 0                  # trailing NAs
    res["results_w"][7][10:20] = 0                 # zero weights -> 1
    segs = pt.run_cbs(res, "M", 1e-9, 100000, 1)

    def one_segment(c, y, w, alpha, state):
        ok = np.flatnonzero(~np.isnan(y))
        return [(int(ok[0]) + 1, int(ok[-1]) + 1)]

    exp = O.cbs_r_wrapper(res["results_r"], res["results_w"], "M", 1e-9, 100000, 1, one_segment)
    assert [s[:3] for s in segs] == [s[:3] for s in exp]
    np.testing.assert_allclose([s[3] for s in segs], [s[3] for s in exp], rtol=1e-12)
    assert not any(s[0] == 1 for s in segs)
    assert [s[:3] for s in segs if s[0] == 3] == [[3, 0, 100], [3, 159, 400]]  # piece starts AT the last NA bin


@pytest.mark.parametrize("n_big", [20000, 40000])
def test_cbs_long_chromosomes(pt, n_big):
    """Chromosomes beyond 16 384 bins (chr1/chr2 below ~14 kb bins) and beyond the LDS sort
    capacity of the permutation kernel (32 768 keys -> global-scratch variant): planted
    change-points are found where they are, noise stays whole (the old bucket scan
    covered only 1024 buckets; the old API refused > 32 768 bins)."""
    rng = np.random.default_rng(11)
    n_per_chr = [n_big, 3000, 250]
    res = _noise_results(rng, n_per_chr, sd=0.06)
    res["results_r"][0][7000:7400] += 0.3              # interior gain on the long chromosome
    res["results_r"][0][n_big - 900:] -= 0.35          # loss running to its end
    res["results_r"][1][rng.random(3000) < 0.03] = 0   # scattered missing bins
    segs = pt.run_cbs(res, "F", 1e-4, 5000, 3)
    c0 = sorted((s, e, r) for c, s, e, r in segs if c == 0)
    assert len(c0) == 4, c0
    assert abs(c0[1][0] - 7000) <= 3 and abs(c0[1][1] - 7400) <= 3 and abs(c0[1][2] - 0.3) < 0.03
    assert abs(c0[3][0] - (n_big - 900)) <= 3 and c0[3][1] == n_big and abs(c0[3][2] + 0.35) < 0.03
    assert len([1 for c, s, e, r in segs if c == 1]) == 1 and len([1 for c, s, e, r in segs if c == 2]) == 1
    assert pt.run_cbs(res, "F", 1e-4, 5000, 3) == segs  # deterministic in the seed


def test_cbs_short_arc_bound_is_an_exact_shortcut(pt):
    """Hybrid tests whose observed statistic exceeds what ANY permutation can reach with a short arc
    are decided without running their permutations (short_arc_bound in cbs_seg.hip).  Run with
    DNAcopy's t >= 7 rule disabled (debug flag 2), so that long aberrations reach the permutation
    stage at all: the segmentation must be identical with the bound disabled as well (flag 32), and
    the bound must actually fire on long aberrations while weak ones still go through the
    permutations."""
    from wisecondorx_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(11)
    n_per_chr = [3000, 2500, 2000, 1500] + [400] * 19
    cases = []
    for trial in range(4):
        res = _noise_results(rng, n_per_chr, sd=0.08)
        res["results_r"][0][500:1700] += 0.58            # long, strong: bound shortcut
        res["results_r"][1][300:330] += 0.25             # short, moderate: permutations decide
        res["results_r"][2][1000:1400] -= 0.1            # long, weak
        res["results_r"][3 + trial][50:60] += 0.35
        for c in range(23):
            res["results_r"][c][rng.random(n_per_chr[c]) < 0.03] = 0
        cases.append(res)
    default = [pt.run_cbs(res, "F", 1e-4, 100000, 3, ctx) for res in cases]
    before = ctx.cbs_stats()["bound_shortcuts"]
    ctx.lib.wcx_debug_flags(ctx.h, 2)
    try:
        with_shortcut = [pt.run_cbs(res, "F", 1e-4, 100000, 3, ctx) for res in cases]
        fired = ctx.cbs_stats()["bound_shortcuts"] - before
        ctx.lib.wcx_debug_flags(ctx.h, 2 | 32)
        without = [pt.run_cbs(res, "F", 1e-4, 100000, 3, ctx) for res in cases]
        assert ctx.cbs_stats()["bound_shortcuts"] - before == fired     # none while disabled
    finally:
        ctx.lib.wcx_debug_flags(ctx.h, 0)
    assert with_shortcut == without == default
    assert fired >= 4
    segs0 = sorted(s for s in with_shortcut[0] if s[0] == 0)
    assert len(segs0) == 3 and abs(segs0[1][1] - 500) <= 2 and abs(segs0[1][2] - 1700) <= 2


@pytest.mark.parametrize("with_inf", [False, True])
def test_cbs_batch_dev_equals_the_host_api(pt, with_inf):
    """wcx_cbs_batch_dev (per-bin vectors in HBM: series compacted on the device, x | w | positions
    exported beside the first round, CBS.R:84-129 restated on the compacted series) == wcx_cbs_batch
    on the same vectors from the host: the same segments and bit-identical means -- NA runs (zeros,
    NaN) short and long, leading / trailing NAs, an all-NA chromosome, zero weights, planted
    change-points.  with_inf: +-inf values are dropped from the series but are not NA to the
    post-processing; the device path then falls back to the host's r / w (same answer)."""
    import torch
    from wisecondorx_amd import _lib
    rng = np.random.default_rng(12)
    n_per_chr = [700, 500, 433, 301] + [260] * 19
    ns, n_chr = 5, 23
    off = np.concatenate(([0], np.cumsum(n_per_chr))).astype(np.int64)
    n_bins = int(off[-1]) + 37                               # (bins beyond the 23 chromosomes: ignored)
    r = rng.normal(0, 0.05, (ns, n_bins))
    w = rng.uniform(0.5, 2.0, (ns, n_bins))
    for s in range(ns):
        c = [1, 3, 0, 2, 5][s]
        a = int(off[c]) + 120
        r[s, a:a + 90] += 0.5                                # a change-point pair
        r[s, off[2] + 50:off[2] + 58] = 0                    # short NA run
        r[s, off[4] + 100:off[4] + 130] = np.nan             # long NA run (30 > 4 at 500 kb): split
        r[s, off[6]:off[6] + 11] = 0                         # leading NAs
        r[s, off[7] + 255:off[7 + 1]] = 0                    # trailing NAs
        w[s, off[8] + 10:off[8] + 20] = 0                    # zero weights -> 1
    r[2, off[9]:off[10]] = 0                                 # all-NA chromosome
    if with_inf:
        r[1, off[11] + 40] = np.inf
        r[3, off[4] + 110] = -np.inf                         # inside the NaN run: breaks it for CBS.R:84-113
    ctx = _lib.default_context()
    off_a, off_p = _lib.i64_array(off)
    cap = 512

    def host():
        seg = np.empty((ns, cap, 4)); cnt = np.zeros(ns, dtype=np.int32)
        _lib.check(ctx.lib.wcx_cbs_batch(ctx.h, _lib.ptr(np.ascontiguousarray(r)), _lib.ptr(np.ascontiguousarray(w)),
                                         ns, n_bins, off_p, n_chr, 1e-4, 500000, 3, _lib.ptr(seg), cap, _lib.ptr(cnt)))
        return [seg[i, :cnt[i]].copy() for i in range(ns)]

    def device():
        d_r = torch.from_numpy(r).cuda(); d_w = torch.from_numpy(w).cuda()
        torch.cuda.synchronize()
        seg = np.empty((ns, cap, 4)); cnt = np.zeros(ns, dtype=np.int32)
        _lib.check(ctx.lib.wcx_cbs_batch_dev(ctx.h, d_r.data_ptr(), d_w.data_ptr(), ns, n_bins, off_p, n_chr,
                                             1e-4, 500000, 3, _lib.ptr(seg), cap, _lib.ptr(cnt)))
        return [seg[i, :cnt[i]].copy() for i in range(ns)]

    # the device call FIRST, on a context of its own whose host staging area is poisoned (debug flag
    # 64): the host-side decisions (short-arc bound, edge statistics) read a series' x | w only where
    # ensure_resident exported them -- a missing export reads NaN here instead of the values an earlier
    # host call left in the same staging area
    fresh = _lib.Context(0)
    fresh.lib.wcx_debug_flags(fresh.h, 64)
    main_ctx, ctx = ctx, fresh
    try:
        got_fresh = device()
    finally:
        ctx = main_ctx
        fresh.close()
    want, got = host(), device()
    assert sum(len(x) for x in want) > ns * 22               # (change-points and NA splits were found)
    for a, b in zip(want, got_fresh):
        assert a.shape == b.shape
        assert np.array_equal(a, b, equal_nan=True)
    for a, b in zip(want, got):
        assert a.shape == b.shape
        assert np.array_equal(a, b, equal_nan=True)
    again = device()                                          # (staging buffers reused)
    for a, b in zip(want, again):
        assert np.array_equal(a, b, equal_nan=True)


def test_cbs_block_bound_search_equals_the_plain_search(pt):
    """The best arc of every segment comes from the block-bound search (k_cbs_blockstats / coarse /
    prune / pairmax: only block pairs whose bound reaches a lower bound of the maximum are evaluated)
    once a call holds more than 4e9 arcs, from the plain striped search (k_cbs_arcmax) below that.
    Six samples with 15 kb-sized chromosomes in ONE call (pruned) == the same samples one call each
    (plain): identical segments and means -- the permutation keys do not depend on the batch."""
    from wisecondorx_amd import _lib
    rng = np.random.default_rng(21)
    n_per_chr = [16600, 16100, 13200, 12700, 12100, 11400, 10600, 9700, 9200, 8900, 9000, 8900, 7600, 7100,
                 6800, 6000, 5500, 5400, 3900, 4300, 3100, 3400, 10400]
    ns = 6
    off = np.concatenate(([0], np.cumsum(n_per_chr))).astype(np.int64)
    n_bins = int(off[-1])
    r = rng.normal(0, 0.06, (ns, n_bins))
    w = rng.uniform(0.5, 2.0, (ns, n_bins))
    for s in range(ns):
        for _ in range(3):
            c = int(rng.integers(0, 23)); a = int(off[c] + rng.integers(0, n_per_chr[c] - 400))
            r[s, a:a + int(rng.integers(20, 400))] += rng.choice([-0.3, 0.25, 0.5])
        for _ in range(20):
            a = int(rng.integers(0, n_bins)); r[s, a:a + int(rng.integers(1, 40))] = 0
    ctx = _lib.default_context()
    off_a, off_p = _lib.i64_array(off)
    cap = 1024
    listed0 = ctx.cbs_stats()["arc_pairs_listed"]

    def run(rows):
        seg = np.empty((len(rows), cap, 4)); cnt = np.zeros(len(rows), dtype=np.int32)
        rr = np.ascontiguousarray(r[rows]); ww = np.ascontiguousarray(w[rows])
        _lib.check(ctx.lib.wcx_cbs_batch(ctx.h, _lib.ptr(rr), _lib.ptr(ww), len(rows), n_bins, off_p, 23, 1e-4,
                                         15000, 5, _lib.ptr(seg), cap, _lib.ptr(cnt)))
        return [seg[i, :cnt[i]].copy() for i in range(len(rows))]

    batch = run(list(range(ns)))
    listed1 = ctx.cbs_stats()["arc_pairs_listed"]
    assert listed1 > listed0                                      # (the block-bound search ran)
    assert sum(len(x) for x in batch) > ns * 23
    for s in range(ns):
        single = run([s])[0]
        assert single.shape == batch[s].shape
        assert np.array_equal(single, batch[s], equal_nan=True)
    assert ctx.cbs_stats()["arc_pairs_listed"] == listed1         # (... and not in the one-sample calls)
