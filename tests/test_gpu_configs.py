"""-m gpu parity tests at the sizes of BASELINE.json's configs that round 1 left untested:
  config 1  predict one sample at 1 Mb bins vs the 50-sample reference (reference-run fixture)
  config 4  newref 15 kb x 500 samples, refsize 300 (the problem the 8-GPU build shards), one device
  config 5  predict batch of 96 samples at 15 kb with on-GPU CBS
All through the C-ABI.  Bit-exact for indices / distances / bin counts, 1e-9 for z / ratios
(north_star: 1e-5)."""
import argparse
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import c_oracle as CO
from oracle import wcx_oracle as O

pytestmark = pytest.mark.gpu


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------------------------- config 1
def test_config1_predict_1mb_vs_50_sample_reference():
    """The reference's own CPU run (tests/golden/config1.npz, made by importing the reference):
    the GPU search on the reference's PCA-corrected matrices reproduces its indexes/distances
    (SHA-256 of the full tables), then pt.normalize reproduces its r / z / w / n / medians."""
    from test_oracle_golden import config1_reference, config1_sample
    from wisecondorx_amd import newref_tools as nt, predict_tools as pt
    g = np.load(os.path.join(GOLDEN, "config1.npz"), allow_pickle=False)

    def search(X, mb, cum, k):
        return nt.get_ref_for_rows(X, cum, k, 0, cum[-1])
    ref, gender = config1_reference(g, search)
    sample = config1_sample(g, gender)
    args = argparse.Namespace(maskrepeats=5)
    cache = {}
    np.testing.assert_allclose(pt.get_optimal_cutoff(ref, 5, cache), g["cutoff"], rtol=1e-12)
    for tag, rg in (("A", "A"), ("G", gender)):
        r, z, w, n, mlr, mz = pt.normalize(args, sample, ref, rg, cache)
        exp = {k: g["{}_{}".format(tag, k)] for k in ("r", "z", "w", "n", "mlr", "mz")}
        assert np.array_equal(n, exp["n"]), tag
        np.testing.assert_allclose(r, exp["r"], rtol=1e-9, equal_nan=True)
        np.testing.assert_allclose(z, exp["z"], rtol=1e-9, atol=1e-9, equal_nan=True)
        np.testing.assert_allclose(w, exp["w"], rtol=1e-12)
        np.testing.assert_allclose([mlr, mz], [exp["mlr"], exp["mz"]], rtol=1e-9, atol=1e-12)


# ------------------------------------------------------------------------------------- config 4
def test_config4_newref_15kb_500_samples():
    """15 kb bins (~182 k masked autosomal rows) x 500 samples, k = 300, MFMA screen path on one
    device (the 8-GPU build runs exactly this per row shard): no row may fall back to the exact
    kernel; ALL 182 k rows bit-exact (indices and distances) vs the C oracle; null ratios of
    >= 10 % of the rows vs the NumPy oracle."""
    from wisecondorx_amd import _lib, newref_tools as nt
    from wisecondorx_amd.synth import bins_per_chr, corrected_matrix
    bpc = [int(b * 0.95) for b in bins_per_chr(15000)[:22]]
    S, k = 500, 300
    X, mbpc, cum = corrected_matrix(bpc, S, seed=45)
    B = cum[-1]
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B, mode=2)
    st = _lib.default_context().topk_stats()
    assert st["rows"] == B and st["fallback_rows"] == 0
    assert (np.diff(dist, axis=1) >= 0).all() and (idx >= 0).all()
    own = np.repeat(np.array(mbpc), np.array(mbpc))
    assert (idx < (B - own)[:, None]).all()                        # chr-excluded index space
    Xs = np.ascontiguousarray(np.asarray(X).T)
    # EVERY row against the C oracle (cache-tiled, on the host's cores; pinned to the reference's
    # fixtures in tests/test_oracle_golden.py): indices and distances bit for bit
    oi, od = CO.get_reference_rows_threaded(Xs, cum, 0, B, k)
    bad = np.flatnonzero((idx != oi).any(axis=1) | (dist != od).any(axis=1))
    assert bad.size == 0, "{} of {} rows differ (first {}); sha256 idx {} dist {}".format(
        bad.size, B, bad[:5], _sha(idx), _sha(dist))
    print("config 4: {} / {} rows bit-exact; sha256 idx {} dist {}".format(B, B, _sha(idx)[:16], _sha(dist)[:16]))
    # null ratios: >= 10 % of the rows (every 8th block of 64 rows, all chromosome borders included)
    ids = list(range(0, S, 37))
    nr_all = nt.get_null_ratios(X, idx, 0, B, ids)
    starts = sorted(set(list(range(0, B - 64, 512)) + [int(c) - 32 for c in cum[:-1]]))
    n_checked = 0
    for r0 in starts:
        exp = O.null_ratios(X, idx[r0:r0 + 64], r0, r0 + 64, ids)
        np.testing.assert_allclose(nr_all[r0:r0 + 64], exp, rtol=1e-12, atol=1e-13)
        n_checked += 64
    assert n_checked >= 0.1 * B
    # a row part like one of 8 ranks would build it (candidate segments path): same rows
    s, e = nt._get_part(3, 8, B)
    pi, pd = nt.get_ref_for_rows(X, cum, k, s, e, mode=2)
    assert _lib.default_context().topk_stats()["fallback_rows"] == 0
    assert np.array_equal(pi, idx[s:e]) and np.array_equal(pd, dist[s:e])


# ------------------------------------------------------------------------------------- config 5
@pytest.fixture(scope="module")
def ref15():
    """A 15 kb reference (S = 100 synthetic corrected matrix; neighbour tables by the GPU search,
    whose parity is covered above and in test_gpu_newref.py)."""
    from wisecondorx_amd import newref_tools as nt
    from wisecondorx_amd.synth import bins_per_chr, corrected_matrix
    full = bins_per_chr(15000)[:22]
    bpc = [int(b * 0.95) for b in full]
    X, mbpc, cum = corrected_matrix(bpc, 100, seed=55)
    B, k = cum[-1], 300
    idx, dist = nt.get_ref_for_rows(X, cum, k, 0, B, mode=2)
    nr = nt.get_null_ratios(X, idx, 0, B, list(range(0, 100, 2)))
    mask = np.zeros(int(np.sum(full)), dtype=bool)
    off = np.concatenate(([0], np.cumsum(full)))
    for c in range(22):
        mask[off[c]:off[c] + bpc[c]] = True
    ref = {"indexes": idx, "distances": dist, "null_ratios": nr, "mask": mask,
           "bins_per_chr": np.array(full), "masked_bins_per_chr": np.array(mbpc),
           "masked_bins_per_chr_cum": np.array(cum)}
    return ref, X


def test_config5_predict_batch_of_96_at_15kb(ref15):
    """96 samples through normalize_repeat_batch + segment_batch (CBS + segment z on the device):
    batch == single bit for bit, one whole sample vs the NumPy oracle, every planted CNV is
    segmented and called, segment z vs the oracle."""
    from wisecondorx_amd import _lib, predict_tools as pt
    ref, X = ref15
    mb, cum = ref["masked_bins_per_chr"].tolist(), ref["masked_bins_per_chr_cum"].tolist()
    B = cum[-1]
    rng = np.random.default_rng(96)
    n_batch = 96
    xs = np.asarray(X)[:, 7][None, :] * (1.0 + 0.02 * rng.standard_normal((n_batch, B)))
    planted = []
    for i in range(n_batch):
        c = i % 22
        cs = cum[c - 1] if c else 0
        a = cs + 100 + (37 * i) % (mb[c] - 2200)      # the 2000-bin CNV stays inside chromosome c
        f = 1.5 if i % 2 == 0 else 0.5
        xs[i, a:a + 2000] *= f
        planted.append((c, a - cs, a - cs + 2000, f))
    xs[5, 1000:1040] = 0.0                      # zero-coverage stretch
    cache = {}
    cutoff = pt.get_optimal_cutoff(ref, 5, cache)
    z, r, n, mlr, mz = pt.normalize_repeat_batch(xs, ref, cutoff, 0, 0, "", cache)
    # batch == single: reference-bin counts, ratios (the medians) and their median bit for bit; z to 1e-9
    # (round 6: a batch's last pass takes mean / sd from incrementally updated sums about the
    # bin's own value, the one-sample kernel from wave reductions: rounding apart -- largest where the
    # bin's own value lies many sd off its reference bins, the planted CNVs and the zero stretch;
    # north_star asks 1e-5 relative, the oracle comparisons below 1e-9)
    for i in (0, 5, 17, 48, 95):
        z1, r1, n1, mlr1, mz1 = pt.normalize_repeat(xs[i], ref, cutoff, 0, 0, "", cache)
        assert np.array_equal(r1, r[i], equal_nan=True)
        assert np.array_equal(n1, n[i]) and mlr1 == mlr[i]
        np.testing.assert_allclose(z[i], z1, rtol=1e-9, atol=1e-9, equal_nan=True)
        np.testing.assert_allclose(mz[i], mz1, rtol=1e-9, atol=1e-12)
    # one whole sample against the NumPy oracle (three dependent passes over all 182 k bins)
    oz, orr, on, omlr, omz = O.normalize_repeat(xs[5], mb, cum, ref["indexes"], ref["distances"],
                                                cutoff, 0, 0)
    assert np.array_equal(n[5], on)
    np.testing.assert_allclose(r[5], orr, rtol=1e-9, equal_nan=True)
    np.testing.assert_allclose(z[5], oz, rtol=1e-9, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose([mlr[5], mz[5]], [omlr, omz], rtol=1e-9, atol=1e-12)
    # CBS + segment z of the whole batch, striped over 4 contexts / streams
    w = pt.get_weights(ref, "", cache)
    args = argparse.Namespace(minrefbins=150, alpha=1e-4, seed=7)
    rem = {"args": args, "mask": ref["mask"], "bins_per_chr": ref["bins_per_chr"],
           "binsize": 15000, "ref_gender": "F"}
    off = np.concatenate(([0], np.cumsum(ref["bins_per_chr"]))).astype(int)
    nr_full = pt.inflate_results(ref["null_ratios"], rem)
    nr_chr = [nr_full[off[c]:off[c + 1]] for c in range(len(off) - 1)]
    ctxs = [_lib.default_context(0)] + [_lib.Context(0) for _ in range(3)]
    for c in ctxs:
        pt.attach_null_matrix(nr_chr, c)
    kept = {}

    def post(i):
        res = {"results_r": r[i], "results_z": z[i] - mz[i], "results_w": w / np.nanmean(w)}
        for key in res:
            res[key] = pt.get_post_processed_result(args, res[key], n[i], rem)
        res["results_nr"] = pt.ATTACHED
        pt.log_trans(res, mlr[i])
        if i in (0, 33):
            kept[i] = res
        return res
    rows = pt.segment_batch(list(range(n_batch)), rem, ctxs, post=post)
    assert len(rows) == n_batch
    for i, (c, a, b, f) in enumerate(planted):
        hits = [s for s in rows[i] if s[0] == c and s[1] < b and s[2] > a
                and abs(s[4] - np.log2(f)) < 0.15]
        assert hits, (i, planted[i], rows[i][:6])
        lo, hi = min(s[1] for s in hits), max(s[2] for s in hits)
        assert abs(lo - a) <= 3 and abs(hi - b) <= 3, (i, lo, hi, a, b)
        assert all(abs(s[3]) > 5 for s in hits if not isinstance(s[3], str))
    # segment z against the oracle (overall_tools.py:88-119) on two samples of the batch
    for i, res in kept.items():
        res = dict(res)
        res["results_nr"] = nr_chr
        segs = [[s[0], s[1], s[2], s[4]] for s in rows[i]]
        oz = O.get_z_score(segs, res["results_nr"], res["results_r"], res["results_w"])
        for s, ozv in zip(rows[i], oz):
            if isinstance(ozv, str):
                assert isinstance(s[3], str)
            else:
                np.testing.assert_allclose(s[3], ozv, rtol=1e-9, atol=1e-9)
    # the SAME batch device-resident end to end (dist.predict_batch_dev: normalisation outputs stay in
    # HBM through merge / post-processing, ONE batched CBS and ONE batched segment-z call): the same
    # segments, z-scores and ratios as the striped host-orchestrated path above
    import torch
    from wisecondorx_amd import dist as wd
    dev = torch.device("cuda", 0)
    ctx_t = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx_t)
    tt = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    A = {"idx": tt(ref["indexes"]), "dist": tt(ref["distances"]), "nr": tt(ref["null_ratios"]), "cum": cum}
    rem_dev = dict(rem, args=argparse.Namespace(minrefbins=150, alpha=1e-4, seed=7, maskrepeats=5))
    rows_dev = wd.predict_batch_dev(be, A, None, tt(xs), None, rem_dev, pt)
    assert len(rows_dev) == n_batch
    for i in range(n_batch):
        assert [s[:3] for s in rows_dev[i]] == [s[:3] for s in rows[i]], i
        for s, t in zip(rows_dev[i], rows[i]):
            if isinstance(t[3], str):
                assert isinstance(s[3], str)
            else:
                np.testing.assert_allclose([s[3], s[4]], [t[3], t[4]], rtol=1e-9, atol=1e-9)
