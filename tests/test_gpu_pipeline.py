"""End-to-end on the GPU through the CLI: newref on the golden cohort -> reference .npz with the
reference's keys/dtypes; every sub-reference bit-identical to the oracle run on the same
PCA-corrected matrix; predict --bed -> tables in the reference's formats, planted CNV called."""
import argparse
import os
import random

import numpy as np
import pytest

from conftest import ref_dict_from_golden, sample_from_counts
from oracle import wcx_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def built(g_pipe, tmp_path_factory):
    from wisecondorx_amd import main, npz_io
    tmp = tmp_path_factory.mktemp("pipe")
    bpc = g_pipe["cohort_bpc"]
    infiles = []
    for i, counts in enumerate(g_pipe["cohort_counts"]):
        p = str(tmp / "s{}.npz".format(i))
        npz_io.save_sample(p, sample_from_counts(counts, bpc), 4000000)
        infiles.append(p)
    out = str(tmp / "ref.npz")
    random.seed(11)
    main.main(["newref"] + infiles + [out, "--binsize", "4000000", "--refsize", "60",
                                      "--yfrac", "0.004"])
    return tmp, out, infiles


def test_reference_file_format(built, g_pipe):
    _, out, _ = built
    mine = np.load(out, encoding="latin1", allow_pickle=True)
    gold = ref_dict_from_golden(g_pipe)
    assert set(mine.files) == set(gold.keys())
    for k in gold:
        assert mine[k].dtype.kind == gold[k].dtype.kind, k
        if k.split(".")[0] in ("bins_per_chr", "has_female", "has_male", "is_nipt", "binsize"):
            assert np.array_equal(mine[k], gold[k]), k
        if k.split(".")[0] in ("indexes", "distances", "null_ratios", "pca_components"):
            assert mine[k].shape[1:] == gold[k].shape[1:], k
    assert mine["indexes"].dtype == np.int32 and mine["distances"].dtype == np.float64


def test_sub_references_match_oracle(built, g_pipe):
    """Same X (the build's deterministic PCA on the device, as the CLI runs it) -> same neighbours
    as the oracle, A/F/M passes."""
    from wisecondorx_amd import _lib, prep
    from wisecondorx_amd.overall_tools import gender_correct
    _, out, _ = built
    mine = np.load(out, encoding="latin1", allow_pickle=True)
    bpc = g_pipe["cohort_bpc"]
    genders = [str(x) for x in g_pipe["cohort_genders"]]
    samples = np.array([gender_correct(sample_from_counts(c, bpc), g)
                        for c, g in zip(g_pipe["cohort_counts"], genders)])
    g = np.array(genders)
    total_mask, bins_per_chr = prep.get_mask(samples)
    total_mask = total_mask & prep.get_mask(samples[g == "F"])[0] & prep.get_mask(samples[g == "M"])[0]
    for gender, sub, ap in (("A", samples, ""), ("F", samples[g == "F"], ".F"),
                            ("M", samples[g == "M"], ".M")):
        p = prep.prepare(sub, gender, total_mask, bins_per_chr, ctx=_lib.default_context(0))
        assert np.array_equal(p["mask"], mine["mask" + ap])
        cum = p["masked_bins_per_chr_cum"].tolist()
        oi, od, _ = O.get_reference(p["X"], p["masked_bins_per_chr"].tolist(), cum, 60, 1, 1, [0])
        assert np.array_equal(mine["indexes" + ap], oi), gender
        assert np.array_equal(mine["distances" + ap], od), gender
        assert mine["null_ratios" + ap].shape == (cum[-1], min(len(sub), 100))


def test_predict_cli_tables(built, g_pipe):
    from wisecondorx_amd import main, npz_io
    tmp, out, _ = built
    from wisecondorx_amd.synth import Cohort
    co = Cohort(4000000, struct_seed=11, female_y=0.1)
    test = co.sample(9001, "M", reads=4e6, cnv=[(3, 10, 25, 1.5)])
    sp = str(tmp / "test.npz")
    npz_io.save_sample(sp, test, 4000000)
    outid = str(tmp / "ID")
    main.main(["predict", sp, out, outid, "--bed", "--minrefbins", "20", "--seed", "3",
               "--zscore", "4"])
    bins = open(outid + "_bins.bed").read().splitlines()
    assert bins[0] == "chr\tstart\tend\tid\tratio\tzscore"
    assert bins[1].split("\t")[:4] == ["1", "1", "4000000", "1:1-4000000"]
    assert len(bins) - 1 == int(np.sum(co.bpc))          # male: 24 chromosomes of bins
    segs = [l.split("\t") for l in open(outid + "_segments.bed").read().splitlines()]
    assert segs[0] == ["chr", "start", "end", "ratio", "zscore"]
    abr = [l.split("\t") for l in open(outid + "_aberrations.bed").read().splitlines()[1:]]
    hits = [a for a in abr if a[0] == "3" and a[5] == "gain"]
    assert hits, abr
    cov = np.zeros(60, dtype=bool)
    for a in hits:
        cov[int((int(a[1]) - 1) / 4e6):int(int(a[2]) / 4e6)] = True
        assert abs(float(a[3]) - np.log2(1.5)) < 0.2, hits
    assert cov[10:25].sum() >= 11 and cov[:8].sum() == 0 and cov[27:].sum() == 0, hits
    stats = open(outid + "_statistics.txt").read()
    assert stats.startswith("chr\tratio.mean\tratio.median\tzscore\n")
    assert "Gender based on --yfrac (or manually overridden by --gender): M" in stats
    assert "Copy number profile abnormality (CPA) score" in stats


@pytest.mark.parametrize("S", [1700, 2100])
def test_pca_stage_large_cohort(S):
    """cohorts beyond S = 1638 need more than the default 64 KB of dynamic LDS for the
    eigenvectors (k_pca_comps) and fewer Gram slices; same agreement with the host PCA."""
    from wisecondorx_amd import _lib, prep
    rng = np.random.default_rng(S)
    B = 900
    load = rng.normal(size=(S, 5)) * np.array([5.0, 4.0, 3.0, 2.0, 1.4])   # five separated factors
    fac = rng.normal(size=(5, B))
    data = 1.0 + 0.01 * (load @ fac) + 0.002 * rng.normal(size=(S, B))
    ctx = _lib.default_context(0)
    Xh, ph = prep.train_pca(np.ascontiguousarray(data.T))
    Xg, pg = prep.train_pca_gpu(data, ctx, sample_major=True)
    np.testing.assert_allclose(pg.mean_, ph.mean_, rtol=1e-12)
    np.testing.assert_allclose(Xg, Xh, rtol=1e-9)


def test_pca_stage_on_gpu_matches_host(g_pipe):
    """f2: wcx_pca_begin/finish (Gram, components, reconstruction, ratio, distance profile on the
    device) vs prep.train_pca (host NumPy, itself pinned against scikit-learn's full-SVD PCA in
    tests/test_host.py) -- 1e-11; deterministic (two runs, identical bits); prepare() takes the
    same filter decision through either path on the cohort where the filter fires."""
    from conftest import GOLDEN
    from wisecondorx_amd import _lib, prep
    from wisecondorx_amd.overall_tools import gender_correct
    from wisecondorx_amd.synth import Cohort
    ctx = _lib.default_context(0)
    co = Cohort(100000, struct_seed=5)
    samples, genders = co.cohort(60, seed0=900, reads=2e6)
    samples = np.array(samples)
    mask, bpc = prep.get_mask(samples)
    data = prep.normalize_and_mask(samples, range(1, 23), mask[:int(np.sum(bpc[:22]))])
    Xh, ph = prep.train_pca(data)
    Xg, pg, d2m = prep.train_pca_gpu(data, ctx, want_dist=True)
    np.testing.assert_allclose(pg.mean_, ph.mean_, rtol=1e-13)
    np.testing.assert_allclose(Xg, Xh, rtol=1e-11)
    # components 4-5 sit in a near-degenerate noise subspace: compare the projector, not the basis
    np.testing.assert_allclose(pg.components_[:3], ph.components_[:3], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(pg.components_.T @ pg.components_ @ np.ones(Xh.shape[0]),
                               ph.components_.T @ ph.components_ @ np.ones(Xh.shape[0]), rtol=1e-7,
                               atol=1e-9)
    med_prof = np.median(Xh, axis=0)
    np.testing.assert_allclose(d2m, np.sum((Xh - med_prof) ** 2, axis=1), rtol=1e-9)
    Xg2, pg2, d2m2 = prep.train_pca_gpu(data, ctx, want_dist=True)
    assert np.array_equal(Xg, Xg2) and np.array_equal(pg.components_, pg2.components_)
    assert np.array_equal(d2m, d2m2)
    # the cohort in which the PCA-distance filter fires (reference-run fixture): same decisions
    g = np.load(os.path.join(GOLDEN, "prep_filter.npz"), allow_pickle=False)
    gcs = np.array([str(x) for x in g["cohort_genders"]])
    smp = np.array([gender_correct(sample_from_counts(c, g["cohort_bpc"]), gd)
                    for c, gd in zip(g["cohort_counts"], gcs)])
    tm, bins = prep.get_mask(smp)
    tm = tm & prep.get_mask(smp[gcs == "F"])[0] & prep.get_mask(smp[gcs == "M"])[0]
    for gender, sub in (("A", smp), ("F", smp[gcs == "F"]), ("M", smp[gcs == "M"])):
        p = prep.prepare(sub, gender, tm, bins, ctx=ctx)
        assert np.array_equal(p["mask"], g[gender + "_mask"]), gender


def test_statistics_file_matches_reference(tmp_path):
    """f3 pin of ID_statistics.txt (predict_output.py:197-263): every text field identical, every
    number within 1e-9 of the reference's file (the per-chromosome z-scores come out of the
    segment-z kernel; MSV / CPA / read count / gender lines are byte-identical)."""
    from test_host import _tables_case
    from wisecondorx_amd import predict_output as po
    g, rem, results = _tables_case(tmp_path)
    po.generate_output_tables(rem, results)
    mine = open(rem["args"].outid + "_statistics.txt").read().splitlines()
    gold = str(g["file_statistics_txt"]).splitlines()
    assert len(mine) == len(gold) == 1 + 24 + 5
    assert mine[0] == gold[0] and mine[25:] == gold[25:]
    for a, b in zip(mine[1:25], gold[1:25]):
        fa, fb = a.split("\t"), b.split("\t")
        assert fa[0] == fb[0] and len(fa) == len(fb) == 4
        np.testing.assert_allclose([float(v) for v in fa[1:]], [float(v) for v in fb[1:]], rtol=1e-9)
    for suffix in ("_bins.bed", "_segments.bed", "_aberrations.bed", "_regions.bed"):
        assert open(rem["args"].outid + suffix).read() == str(g["file" + suffix.replace(".", "_")])


def test_predict_one_dev_equals_the_host_path():
    """dist.predict_one_dev (what bench.py and the replica predict of a multi-GPU build run: cut-off,
    weights, three passes, wcx_post_process_dev, CBS, segment z -- nothing but three result vectors
    leaves the device) against the step-by-step host mirror of predict_control.normalize +
    get_post_processed_result + log_trans + exec_cbs on the same reference."""
    import torch
    import bench
    from wisecondorx_amd import _lib, dist as wd, predict_tools as pt
    co, p, test = bench.make_workload(100000, 40)
    X = p["X"]
    cum = np.asarray(p["masked_bins_per_chr_cum"], dtype=np.int64)
    B, S, k = int(cum[-1]), X.shape[1], 100
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    Xrow = torch.from_numpy(np.ascontiguousarray(X)).to(dev)
    ids = np.arange(S, dtype=np.int32)
    idx, dist, nr, _ = wd.newref_sharded(Xrow, B, cum, k, ids, be, 0, 1)
    x = pt.project_pc(pt.coverage_normalize_and_mask(test, p, ""), p, "")
    d_x = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    args = argparse.Namespace(minrefbins=30, alpha=1e-4, seed=3, maskrepeats=5)
    rem = {"args": args, "mask": p["mask"], "bins_per_chr": p["bins_per_chr"], "binsize": 100000,
           "ref_gender": "F"}
    rows = wd.predict_one_dev(be, idx, dist, nr, d_x, B, k, cum, rem, pt)
    rows_again = wd.predict_one_dev(be, idx, dist, nr, d_x, B, k, cum, rem, pt)   # cached buffers
    assert rows == rows_again

    ref = dict(p)
    ref.update({"indexes": idx.cpu().numpy(), "distances": dist.cpu().numpy(),
                "null_ratios": nr.cpu().numpy()})
    cache = {}
    cutoff = pt.get_optimal_cutoff(ref, 5, cache)
    w = pt.get_weights(ref, "", cache)
    z, r, n, mlr, mz = pt.normalize_repeat(x, ref, cutoff, 0, 0, "", cache)
    res = {"results_r": r, "results_z": z - mz, "results_w": w / np.nanmean(w)}
    for key in res:
        res[key] = pt.get_post_processed_result(args, res[key], n, rem)
    pt.log_trans(res, mlr)
    off = np.concatenate(([0], np.cumsum(p["bins_per_chr"]))).astype(int)
    nr_full = pt.inflate_results(ref["null_ratios"], rem)
    res["results_nr"] = [nr_full[off[c]:off[c + 1]] for c in range(len(off) - 1)]
    want = pt.exec_cbs(rem, res, _lib.default_context(0))
    assert len(rows) == len(want) and len(rows) >= 23
    for a, b in zip(rows, want):
        assert a[:3] == b[:3]
        np.testing.assert_allclose([float(a[3]), a[4]], [float(b[3]), b[4]], rtol=1e-9, atol=1e-9)
    assert any(abs(s[4] - np.log2(1.5)) < 0.1 for s in rows)          # the planted gain is called


def test_mask_skew_default_is_upstream_and_aligned_masks_is_opt_in(tmp_path):
    """The cohort of tests/golden/mask_skew.npz makes the F / M passes' PCA-distance filter drop an
    autosomal bin the finished A reference still holds (newref_control.py:48-54).  DEFAULT newref
    reproduces upstream's three masks bit for bit; upstream's own predict raises IndexError on that
    reference (recorded in the fixture: predict_control.py:50 on results_nr) and ours refuses it too;
    --aligned-masks keeps the autosomal masks equal and predict works."""
    from wisecondorx_amd import main, npz_io
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_skew.npz"))
    assert str(g["predict_results_nr"]) == "IndexError" and str(g["predict_results_r"]) == "ok"
    bpc = g["cohort_bpc"]
    infiles = []
    for i, counts in enumerate(g["cohort_counts"]):
        p = str(tmp_path / "s{}.npz".format(i))
        npz_io.save_sample(p, sample_from_counts(counts, bpc), 4000000)
        infiles.append(p)
    sp = str(tmp_path / "test.npz")
    npz_io.save_sample(sp, sample_from_counts(g["test_counts"], bpc), 4000000)
    n_aut = int(np.sum(bpc[:22]))
    common = ["--binsize", "4000000", "--refsize", "40", "--yfrac", "0.004"]
    # default: upstream's masks; the skew makes the reference unusable, so newref stops with an error
    # and leaves no file (nor a temporary one) behind
    out0 = str(tmp_path / "ref_default.npz")
    random.seed(5)
    with pytest.raises(SystemExit) as e0:
        main.main(["newref"] + infiles + [out0] + common)
    assert e0.value.code == 1
    assert [f for f in os.listdir(str(tmp_path)) if f.startswith("ref_default")] == []
    # --reference-mask-skew (deprecated alias): upstream's file as it is, skew included
    out = str(tmp_path / "ref.npz")
    random.seed(5)
    main.main(["newref"] + infiles + [out, "--reference-mask-skew"] + common)
    mine = np.load(out, allow_pickle=True)
    assert np.array_equal(mine["mask"], g["mask"])
    assert np.array_equal(mine["mask.F"], g["mask_F"])
    assert np.array_equal(mine["mask.M"], g["mask_M"])
    assert mine["mask"][:n_aut].sum() > mine["mask.F"][:n_aut].sum()
    with pytest.raises(SystemExit) as ei:
        main.main(["predict", sp, out, str(tmp_path / "ID"), "--bed", "--minrefbins", "10", "--seed", "3"])
    assert ei.value.code == 1
    # --aligned-masks: the gonosomal passes keep the autosomal mask of the A pass; predict runs
    out2 = str(tmp_path / "ref_aligned.npz")
    random.seed(5)
    main.main(["newref"] + infiles + [out2, "--aligned-masks"] + common)
    fixed = np.load(out2, allow_pickle=True)
    assert np.array_equal(fixed["mask"], g["mask"])
    assert np.array_equal(fixed["mask.F"][:n_aut], fixed["mask"][:n_aut])
    assert np.array_equal(fixed["mask.M"][:n_aut], fixed["mask"][:n_aut])
    main.main(["predict", sp, out2, str(tmp_path / "ID2"), "--bed", "--minrefbins", "10", "--seed", "3"])
    assert os.path.exists(str(tmp_path / "ID2_bins.bed"))


def test_predict_full_dev_applies_the_blacklist_like_the_host_path(tmp_path):
    """dist.predict_full_dev with --blacklist: the blacklisted bins of r / z / w are zeroed on the device
    between the log2 transform and CBS (main.py:263-265, predict_tools.py:202-214) -- same per-bin
    vectors and same segments as the host mirror with apply_blacklist."""
    import torch
    import bench
    from wisecondorx_amd import _lib, dist as wd, predict_tools as pt
    co, p, test = bench.make_workload(100000, 40)
    X = p["X"]
    cum = np.asarray(p["masked_bins_per_chr_cum"], dtype=np.int64)
    B, S, k = int(cum[-1]), X.shape[1], 100
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    Xrow = torch.from_numpy(np.ascontiguousarray(X)).to(dev)
    ids = np.arange(S, dtype=np.int32)
    idx, dist, nr, _ = wd.newref_sharded(Xrow, B, cum, k, ids, be, 0, 1)
    x = pt.project_pc(pt.coverage_normalize_and_mask(test, p, ""), p, "")
    d_x = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    bl = str(tmp_path / "bl.bed")
    with open(bl, "w") as fh:
        fh.write("chr3\t1000000\t2600000\n7\t0\t450000\nchrX\t0\t100\n")
    args = argparse.Namespace(minrefbins=30, alpha=1e-4, seed=3, maskrepeats=5, blacklist=bl)
    rem = {"args": args, "mask": p["mask"], "bins_per_chr": p["bins_per_chr"], "binsize": 100000,
           "ref_gender": "F"}
    A = {"idx": idx, "dist": dist, "nr": nr, "cum": cum}
    rows, host = wd.predict_full_dev(be, A, None, d_x, None, rem, pt, want_host=True)

    ref = dict(p)
    ref.update({"indexes": idx.cpu().numpy(), "distances": dist.cpu().numpy(),
                "null_ratios": nr.cpu().numpy()})
    cache = {}
    cutoff = pt.get_optimal_cutoff(ref, 5, cache)
    w = pt.get_weights(ref, "", cache)
    z, r, n, mlr, mz = pt.normalize_repeat(x, ref, cutoff, 0, 0, "", cache)
    res = {"results_r": r, "results_z": z - mz, "results_w": w / np.nanmean(w)}
    for key in res:
        res[key] = pt.get_post_processed_result(args, res[key], n, rem)
    pt.log_trans(res, mlr)
    pt.apply_blacklist(rem, res)
    off = np.concatenate(([0], np.cumsum(p["bins_per_chr"]))).astype(int)
    for row, key in enumerate(("results_r", "results_z", "results_w")):
        want_vec = np.concatenate(res[key])
        np.testing.assert_allclose(host[row], want_vec, rtol=1e-9, atol=1e-12)
        assert np.all(host[row][off[2] + 10:off[2] + 27] == 0) and np.all(host[row][off[6]:off[6] + 5] == 0)
    nr_full = pt.inflate_results(ref["null_ratios"], rem)
    res["results_nr"] = [nr_full[off[c]:off[c + 1]] for c in range(len(off) - 1)]
    want = pt.exec_cbs(rem, res, _lib.default_context(0))
    assert len(rows) == len(want)
    for a, b in zip(rows, want):
        assert a[:3] == b[:3]
        np.testing.assert_allclose([float(a[3]), a[4]], [float(b[3]), b[4]], rtol=1e-9, atol=1e-9)


def test_predict_cli_batch_equals_single_runs(built, g_pipe, tmp_path, monkeypatch):
    """`predict --batch LIST` (device-resident batches, dist.predict_batch_dev; also striped over two
    worker processes sharing the device): the tables of every sample are the ones the one-sample
    CLI writes."""
    from wisecondorx_amd import main, npz_io
    from wisecondorx_amd.synth import Cohort
    tmp, ref, _ = built
    co = Cohort(4000000, struct_seed=11, female_y=0.1)
    specs = [(9001, "M", [(3, 10, 25, 1.5)]), (9002, "F", [(7, 5, 15, 0.5)]), (9003, "M", None),
             (9004, "F", [(12, 3, 12, 1.5)]), (9005, "F", None)]
    files = []
    for seed, g, cnv in specs:
        f = str(tmp_path / "t{}.npz".format(seed))
        npz_io.save_sample(f, co.sample(seed, g, reads=4e6, cnv=cnv), 4000000)
        files.append(f)
    common = ["--bed", "--minrefbins", "20", "--seed", "3", "--zscore", "4"]
    for i, f in enumerate(files):
        main.main(["predict", f, ref, str(tmp_path / "single{}".format(i))] + common)
    for gpus, tag in ((1, "b"), (2, "p")):
        lst = str(tmp_path / "list_{}.txt".format(tag))
        with open(lst, "w") as fh:
            for i, f in enumerate(files[1:], 1):
                fh.write("{}\t{}\n".format(f, tmp_path / "{}{}".format(tag, i)))
        monkeypatch.setenv("WCX_DIST_SHARE_DEVICE", "1")
        main.main(["predict", files[0], ref, str(tmp_path / "{}0".format(tag)), "--batch", lst,
                   "--gpus", str(gpus)] + common)
        for i in range(len(files)):
            for suffix in ("_bins.bed", "_segments.bed", "_aberrations.bed", "_statistics.txt"):
                a = open(str(tmp_path / "single{}{}".format(i, suffix))).read()
                b = open(str(tmp_path / "{}{}{}".format(tag, i, suffix))).read()
                if a != b:          # same rows; numbers to 1e-9 (batch and single sums differ in order)
                    la, lb = a.splitlines(), b.splitlines()
                    assert len(la) == len(lb), (tag, i, suffix)
                    for x, y in zip(la, lb):
                        fx, fy = x.split("\t"), y.split("\t")
                        assert len(fx) == len(fy)
                        for u, v in zip(fx, fy):
                            if u != v:
                                np.testing.assert_allclose(float(u), float(v), rtol=1e-7, atol=1e-9)


def test_newref_cli_constant_prefix_members(built, tmp_path, monkeypatch):
    """The gonosomal passes hand their tables on as npz_io.PrefixConst (the autosomal rows of indexes.F /
    distances.F / .M are the reference's 0 / 1 dummies, newref_tools.py:186-191, and never leave the
    device); from 8 MB of constants on, the writer stores them as pre-built deflate blocks.  With that
    limit lowered to the test cohort's size the file must hold the same arrays as the ordinary build for
    np.load (what the reference's predict uses) and for load_reference, and predict must give the same
    tables."""
    import zipfile
    from wisecondorx_amd import main, npz_io
    tmp, ref1, infiles = built
    monkeypatch.setattr(npz_io, "_BIG", 1 << 10)
    out = str(tmp_path / "ref_hybrid.npz")
    random.seed(11)
    main.main(["newref"] + infiles + [out, "--binsize", "4000000", "--refsize", "60", "--yfrac", "0.004"])
    with zipfile.ZipFile(out) as zf:
        for name in ("indexes.F.npy", "distances.F.npy", "indexes.M.npy", "distances.M.npy"):
            assert npz_io._hybrid_info(zf.getinfo(name).extra) is not None, name
        assert npz_io._hybrid_info(zf.getinfo("indexes.npy").extra) is None
        assert zf.testzip() is None
    a, b = np.load(ref1, allow_pickle=True), np.load(out, allow_pickle=True)
    fast = npz_io.load_reference(out)
    assert sorted(a.files) == sorted(b.files) == sorted(fast.keys())
    for key in a.files:
        for other in (b[key], fast[key]):
            assert a[key].dtype == np.asarray(other).dtype and np.array_equal(a[key], other, equal_nan=a[key].dtype.kind == "f"), key
    monkeypatch.undo()
    sp = infiles[3]
    outs = []
    for tag, ref in (("p1", ref1), ("p2", out)):
        oid = str(tmp_path / tag)
        main.main(["predict", sp, ref, oid, "--bed", "--minrefbins", "20", "--seed", "3"])
        outs.append([open(oid + sfx).read() for sfx in ("_bins.bed", "_segments.bed", "_statistics.txt")])
    assert outs[0] == outs[1]


def test_newref_cli_multi_process_equals_one_process(built, tmp_path, monkeypatch):
    """`newref --gpus 2`: one process per rank (here gloo, both ranks on the one device), every rank
    prepares the pass, keeps its row shard of the corrected matrix, ONE all-gather per pass
    (dist.newref_sharded / newref_gonosomal_sharded), rank 0 writes -- the same reference file as the
    one-process build: masks, PCA, indexes, distances and null ratios bit for bit."""
    from wisecondorx_amd import main
    tmp, ref1, infiles = built
    monkeypatch.setenv("WCX_DIST_BACKEND", "gloo")
    monkeypatch.setenv("WCX_DIST_SHARE_DEVICE", "1")
    out = str(tmp_path / "ref2.npz")
    random.seed(11)
    main.main(["newref"] + infiles + [out, "--binsize", "4000000", "--refsize", "60", "--yfrac", "0.004",
                                      "--gpus", "2"])
    a, b = np.load(ref1, allow_pickle=True), np.load(out, allow_pickle=True)
    assert sorted(a.files) == sorted(b.files)
    for key in a.files:
        if a[key].dtype.kind == "f":
            assert np.array_equal(a[key], b[key], equal_nan=True), key
        else:
            assert np.array_equal(a[key], b[key]), key
