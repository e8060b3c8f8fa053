"""CPU: host-side glue (bin rescaling, PCA, partition helpers, CLI surface, .npz formats)."""
import os

import numpy as np
import pytest

from oracle import wcx_oracle as O


def test_scale_sample_matches_reference_semantics():
    from wisecondorx_amd.overall_tools import scale_sample
    rng = np.random.default_rng(0)
    s = {"1": rng.integers(0, 50, 1003).astype(np.int32), "2": rng.integers(0, 50, 7).astype(np.int32)}
    out = scale_sample(s, 5000, 15000)
    for k, v in s.items():
        n = int(np.ceil(len(v) / 3.0))
        exp = np.array([v[int(i * 3.0):int(i * 3.0 + 3.0)].sum() for i in range(n)], dtype=np.int32)
        assert out[k].dtype == np.int32 and np.array_equal(out[k], exp)   # overall_tools.py:31-39
    assert scale_sample(s, 5000, 5000) is s


def test_train_pca_equals_full_svd_pca():
    from sklearn.decomposition import PCA
    from wisecondorx_amd import prep
    rng = np.random.default_rng(1)
    D = 1 + 0.1 * rng.standard_normal((400, 30))
    X, model = prep.train_pca(D)
    pca = PCA(n_components=5, svd_solver="full").fit(D.T)
    ref = (D.T / pca.inverse_transform(pca.transform(D.T))).T
    assert X.flags["F_CONTIGUOUS"]
    np.testing.assert_allclose(X, ref, rtol=1e-11)
    np.testing.assert_allclose(model.components_, pca.components_, atol=1e-11)


def test_partition_helpers_match_oracle():
    from wisecondorx_amd import newref_tools as nt
    cum = [0, 5, 9, 9, 20, 31]
    for parts in (1, 2, 3, 7):
        for p in range(parts):
            assert nt._get_part(p, parts, 31) == O.get_part(p, parts, 31)
            s, e = O.get_part(p, parts, 31)
            assert nt._split_by_chr(s, e, cum) == O.split_by_chr(s, e, cum)


def test_cli_surface_matches_reference():
    """Flags and defaults of main.py:314-488."""
    from wisecondorx_amd import main
    p = main.build_parser()
    a = p.parse_args(["predict", "in.npz", "ref.npz", "out"])
    assert (a.minrefbins, a.maskrepeats, a.alpha, a.zscore, a.beta, a.blacklist, a.gender,
            a.ylim, a.bed, a.plot, a.cairo, a.add_plot_title, a.seed, a.regions) == (
        150, 5, 1e-4, 5, None, None, None, "def", False, False, False, False, None, None)
    a = p.parse_args(["newref", "a.npz", "b.npz", "ref.npz"])
    assert (a.nipt, a.yfrac, a.plotyfrac, a.refsize, a.binsize, a.cpus) == (
        False, None, None, 300, 1e5, 1)
    assert a.infiles == ["a.npz", "b.npz"] and a.outfile == "ref.npz"
    a = p.parse_args(["convert", "x.bam", "x.npz"])
    assert (a.binsize, a.normdup, a.reference) == (5e3, False, None)
    a = p.parse_args(["gender", "x.npz", "ref.npz"])
    assert a.func is main.output_gender


def test_npz_roundtrip(tmp_path):
    from wisecondorx_amd import npz_io
    sample = {str(c): np.arange(c + 3, dtype=np.int32) for c in range(1, 25)}
    p = str(tmp_path / "s.npz")
    npz_io.save_sample(p, sample, 5000)
    s2, b = npz_io.load_sample(p)
    assert b == 5000 and all(np.array_equal(s2[k], sample[k]) for k in sample)
    assert all(s2[k].dtype == np.int32 for k in sample)
    # the direct reader (one inflate call per member) checks what np.load's zipfile would: a flipped
    # byte in the deflated sample, a truncated file; an uncompressed np.savez file loads as well
    import zipfile
    raw = bytearray(open(p, "rb").read())
    at = raw.find(b"sample.npy") + 200
    raw[at] ^= 0x5A
    bad = str(tmp_path / "bad.npz")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(zipfile.BadZipFile):
        npz_io.load_sample(bad)
    open(bad, "wb").write(bytes(raw[:len(raw) // 2]))
    with pytest.raises(zipfile.BadZipFile):
        npz_io.load_sample(bad)
    stored = str(tmp_path / "stored.npz")
    np.savez(stored, binsize=7000, sample=sample, quality={})
    s3, b3 = npz_io.load_sample(stored)
    assert b3 == 7000 and all(np.array_equal(s3[k], sample[k]) for k in sample)
    ref = {"indexes": np.zeros((3000, 300), np.int32), "distances": np.ones((3000, 300)),
           "mask": np.ones(10, bool), "binsize": 100000, "is_nipt": False, "trained_cutoff": 0.004}
    rp = npz_io.save_npz(str(tmp_path / "r.npz"), ref)
    back = np.load(rp, encoding="latin1", allow_pickle=True)     # how the reference reads it
    assert set(back.files) == set(ref)
    assert back["indexes"].dtype == np.int32 and int(back["binsize"]) == 100000
    assert not back["is_nipt"] and float(back["trained_cutoff"]) == 0.004


def test_reference_npz_fast_writer_and_reader(tmp_path):
    """save_npz writes its own ZIP (stored big members, CRCs from worker threads, positional
    writes): the archive must pass zipfile's CRC check, load with plain np.load exactly like the
    reference's files, keep Fortran order, and load_reference (direct reads of the stored members)
    must return the same arrays and dtypes."""
    import zipfile
    from wisecondorx_amd import npz_io
    rng = np.random.default_rng(3)
    ref = {"binsize": 15000, "indexes": rng.integers(0, 1000, (40000, 60)).astype(np.int32),
           "distances": rng.random((40000, 60)),
           "null_ratios.F": np.asfortranarray(rng.random((30000, 40))),
           "mask": rng.random(50000) > 0.1, "is_nipt": False, "trained_cutoff": 0.25,
           "pca_mean": rng.random(1000), "meta": np.array({"a": 1}, dtype=object)}
    p = npz_io.save_npz(str(tmp_path / "ref"), ref)
    assert p.endswith(".npz")
    with zipfile.ZipFile(p) as zf:
        assert zf.testzip() is None
        kinds = {i.filename: i.compress_type for i in zf.infolist()}
    assert kinds["distances.npy"] == zipfile.ZIP_STORED and kinds["mask.npy"] == zipfile.ZIP_DEFLATED
    plain = np.load(p, encoding="latin1", allow_pickle=True)
    fast = npz_io.load_reference(p)
    assert set(plain.files) == set(ref) == set(fast)
    for k, v in ref.items():
        if k == "meta":
            assert plain[k].item() == {"a": 1} and fast[k].item() == {"a": 1}
            continue
        for got in (plain[k], fast[k]):
            assert got.dtype == np.asarray(v).dtype and np.array_equal(got, np.asarray(v)), k
    assert fast["null_ratios.F"].flags.f_contiguous and plain["null_ratios.F"].flags.f_contiguous


def test_reference_npz_zip64_records(tmp_path, monkeypatch):
    """The ZIP64 branch of save_npz (members / offsets beyond 4 GB: 5 kb-bin references) with the
    threshold lowered so that a small archive takes it: ZIP64 extra fields in the local and
    central headers, ZIP64 end-of-central-directory record + locator -- readable by zipfile,
    np.load and load_reference."""
    import zipfile
    from wisecondorx_amd import npz_io
    monkeypatch.setattr(npz_io, "_Z64", 1 << 20)
    monkeypatch.setattr(npz_io, "_BIG", 1 << 18)
    rng = np.random.default_rng(4)
    ref = {"a": rng.random((300, 500)), "b": rng.integers(0, 9, (900, 400)).astype(np.int32),
           "c": rng.random(70000), "small": np.arange(5), "flag": True}
    p = npz_io.save_npz(str(tmp_path / "z64.npz"), ref)
    with zipfile.ZipFile(p) as zf:
        assert zf.testzip() is None
        assert {i.filename for i in zf.infolist()} == {k + ".npy" for k in ref}
    plain = np.load(p, allow_pickle=True)
    fast = npz_io.load_reference(p)
    for k, v in ref.items():
        assert np.array_equal(plain[k], np.asarray(v)) and np.array_equal(fast[k], np.asarray(v)), k


def test_convert_filters_match_reference_loop(monkeypatch):
    """convert (pysam absent here): the vectorised duplicate / MAPQ / proper-pair filters against
    a straight per-read loop with the reference's branching (convert_tools.py:75-104), on fake
    read columns including the carry-over of the previous read across contigs."""
    import types
    from wisecondorx_amd import convert_tools as ct
    rng = np.random.default_rng(0)

    class Read:
        def __init__(self, pos, mate, mq, paired, proper):
            self.pos, self.next_reference_start, self.mapping_quality = pos, mate, mq
            self.is_paired, self.is_proper_pair = paired, proper

    class FakeBam:
        references = ["chr1", "chrM", "2", "chrX"]
        lengths = [5000, 100, 3000, 2000]
        mapped, unmapped, nocoordinate = 1, 2, 3

        def __init__(self):
            self.reads = {}
            for c, L in zip(self.references, self.lengths):
                pos = np.sort(rng.integers(0, L, 400))
                pos[rng.random(400) < 0.3] = 0
                pos = np.sort(pos)
                self.reads[c] = [Read(int(p), int(rng.integers(0, 4)), int(rng.integers(0, 3)),
                                      bool(rng.random() < 0.6), bool(rng.random() < 0.8)) for p in pos]

        def fetch(self, c):
            return iter(self.reads[c])

    def loop(f, binsize, normdup):
        out, larp, larp2, dup, mq, pf, seen = {}, -1, -1, 0, 0, 0, 0
        for i, c in enumerate(f.references):
            name = c[3:] if c[:3].lower() == "chr" else c
            if name not in [str(x) for x in range(1, 25)] + ["X", "Y"]:
                continue
            counts = np.zeros(int(f.lengths[i] / float(binsize) + 1), dtype=np.int32)
            for r in f.fetch(c):
                if r.is_paired:
                    if not r.is_proper_pair:
                        pf += 1
                        continue
                    if not normdup and larp == r.pos and larp2 == r.next_reference_start:
                        dup += 1
                    elif r.mapping_quality >= 1:
                        counts[int(r.pos / binsize)] += 1
                    else:
                        mq += 1
                    larp2 = r.next_reference_start
                else:
                    if not normdup and larp == r.pos:
                        dup += 1
                    elif r.mapping_quality >= 1:
                        counts[int(r.pos / binsize)] += 1
                    else:
                        mq += 1
                seen += 1
                larp = r.pos
            out[{"X": "23", "Y": "24"}.get(name, name)] = counts
        return out, (dup, mq, pf, seen)

    f = FakeBam()
    monkeypatch.setattr(ct, "_open", lambda a: f)
    for normdup in (False, True):
        args = types.SimpleNamespace(infile="x.bam", reference=None, binsize=100.0, normdup=normdup)
        bins, q = ct.convert_reads(args)
        exp, (dup, mq, pf, seen) = loop(f, 100.0, normdup)
        assert bins["3"] is None and set(k for k, v in bins.items() if v is not None) == set(exp)
        for k in exp:
            assert bins[k].dtype == np.int32 and np.array_equal(bins[k], exp[k])
        assert (q["filter_rmdup"], q["filter_mapq"], q["pair_fail"], q["pre_retro"]) == (dup, mq, pf, seen)
        assert q["post_retro"] == sum(int(v.sum()) for v in exp.values())


def _golden_cohort(g):
    from conftest import sample_from_counts
    from wisecondorx_amd.overall_tools import gender_correct
    bpc = g["cohort_bpc"]
    genders = np.array([str(x) for x in g["cohort_genders"]])
    samples = np.array([gender_correct(sample_from_counts(c, bpc), gd)
                        for c, gd in zip(g["cohort_counts"], genders)])
    return samples, genders


def test_prepare_matches_reference_outputs(g_pipe):
    """a1/a3 pin: prep.get_mask + prep.prepare on the golden cohort reproduce what the REFERENCE's
    tool_newref wrote (mask, masked_bins_per_chr(_cum), bins_per_chr, pca_mean) for the A, F and
    M passes (tests/golden/pipeline.npz, keys ref__*).  pca_components are not compared: the
    reference's randomized SVD is unseeded."""
    from conftest import ref_dict_from_golden
    from wisecondorx_amd import prep
    ref = ref_dict_from_golden(g_pipe)
    samples, g = _golden_cohort(g_pipe)
    total_mask, bins_per_chr = prep.get_mask(samples)                 # main.py:82-88
    total_mask = total_mask & prep.get_mask(samples[g == "F"])[0] & prep.get_mask(samples[g == "M"])[0]
    for gender, sub, ap in (("A", samples, ""), ("F", samples[g == "F"], ".F"),
                            ("M", samples[g == "M"], ".M")):
        p = prep.prepare(sub, gender, total_mask, bins_per_chr)
        assert np.array_equal(p["mask"], ref["mask" + ap]), gender
        assert np.array_equal(p["bins_per_chr"], ref["bins_per_chr" + ap]), gender
        assert np.array_equal(p["masked_bins_per_chr"], ref["masked_bins_per_chr" + ap]), gender
        assert np.array_equal(p["masked_bins_per_chr_cum"], ref["masked_bins_per_chr_cum" + ap])
        np.testing.assert_allclose(p["pca_mean"], ref["pca_mean" + ap], rtol=1e-13, atol=0)
        assert p["pca_components"].shape == ref["pca_components" + ap].shape


def test_pca_distance_filter_fires_and_skews_masks():
    """a3 pin (tests/golden/prep_filter.npz, captured from the reference's tool_newref_prep): the
    PCA-distance filter removes nine bins in the A pass and ONE MORE autosomal bin in the F pass;
    because the shared mask is mutated in place (newref_control.py:51-54) after the A pass kept
    its copy, mask.F / mask.M end up with one autosomal bin fewer than mask -- reproduced, not
    fixed."""
    from conftest import GOLDEN, sample_from_counts
    from wisecondorx_amd import prep
    from wisecondorx_amd.overall_tools import gender_correct
    g = np.load(os.path.join(GOLDEN, "prep_filter.npz"), allow_pickle=False)
    bpc = g["cohort_bpc"]
    genders = np.array([str(x) for x in g["cohort_genders"]])
    samples = np.array([gender_correct(sample_from_counts(c, bpc), gd)
                        for c, gd in zip(g["cohort_counts"], genders)])
    total_mask, bins_per_chr = prep.get_mask(samples)
    total_mask = total_mask & prep.get_mask(samples[genders == "F"])[0] \
        & prep.get_mask(samples[genders == "M"])[0]
    assert np.array_equal(total_mask, g["total_mask_in"])
    n_aut = int(np.sum(bins_per_chr[:22]))
    for gender, sub in (("A", samples), ("F", samples[genders == "F"]),
                        ("M", samples[genders == "M"])):
        before = total_mask.copy()
        p = prep.prepare(sub, gender, total_mask, bins_per_chr)
        assert np.array_equal(np.where(before & ~total_mask)[0], g[gender + "_removed"]), gender
        for key in ("mask", "bins_per_chr", "masked_bins_per_chr", "masked_bins_per_chr_cum"):
            assert np.array_equal(p[key], g["{}_{}".format(gender, key)]), (gender, key)
        np.testing.assert_allclose(p["pca_mean"], g[gender + "_pca_mean"], rtol=1e-13, atol=0)
    assert len(g["A_removed"]) == 9 and len(g["F_removed"]) == 1
    assert g["A_mask"][:n_aut].sum() == g["F_mask"][:n_aut].sum() + 1      # the skew


def test_normalize_and_mask_sample_major_is_bit_identical():
    """normalize_and_mask_t (what prepare feeds the GPU PCA stage) == normalize_and_mask
    (newref_tools.py:110-129) transposed, bit for bit, incl. samples shorter than the longest."""
    from wisecondorx_amd import prep
    rng = np.random.default_rng(8)
    lens = {str(c): int(rng.integers(30, 90)) for c in range(1, 25)}
    samples = []
    for i in range(7):
        s = {k: rng.integers(0, 400, n).astype(np.int32) for k, n in lens.items()}
        if i == 2:
            s["5"] = s["5"][:-4]
            s["23"] = s["23"][:-1]
        samples.append(s)
    samples = np.array(samples)
    mask, bpc = prep.get_mask(samples)
    mask = mask & (rng.random(len(mask)) > 0.2)
    for last in (22, 23, 24):
        m = mask[:int(np.sum(bpc[:last]))]
        a = prep.normalize_and_mask(samples, range(1, last + 1), m)
        b = prep.normalize_and_mask_t(samples, range(1, last + 1), m)
        assert b.flags.c_contiguous and np.array_equal(a, b.T)


def test_frozen_autosomal_mask_in_gonosomal_passes():
    """The product default (main.build_sub_reference): on the same cohort the F / M passes keep the
    autosomal part of the mask of the finished A reference -- the one autosomal bin the reference's
    F pass drops stays, so predict can align autosomal and gonosomal results."""
    from conftest import GOLDEN, sample_from_counts
    from wisecondorx_amd import prep
    from wisecondorx_amd.overall_tools import gender_correct
    g = np.load(os.path.join(GOLDEN, "prep_filter.npz"), allow_pickle=False)
    bpc = g["cohort_bpc"]
    genders = np.array([str(x) for x in g["cohort_genders"]])
    samples = np.array([gender_correct(sample_from_counts(c, bpc), gd)
                        for c, gd in zip(g["cohort_counts"], genders)])
    total_mask, bins_per_chr = prep.get_mask(samples)
    total_mask = total_mask & prep.get_mask(samples[genders == "F"])[0] \
        & prep.get_mask(samples[genders == "M"])[0]
    n_aut = int(np.sum(bins_per_chr[:22]))
    pa = prep.prepare(samples, "A", total_mask, bins_per_chr)
    assert np.array_equal(pa["mask"], g["A_mask"])                     # the A pass is unchanged
    for gender in ("F", "M"):
        p = prep.prepare(samples[genders == gender], gender, total_mask, bins_per_chr, frozen=n_aut)
        assert np.array_equal(p["mask"][:n_aut], pa["mask"])
        assert np.array_equal(p["masked_bins_per_chr"][:22], pa["masked_bins_per_chr"])
        # whatever the reference's pass removed beyond the autosomes is still removed
        ref_removed = g[gender + "_removed"]
        assert not p["mask"][ref_removed[ref_removed >= n_aut]].any()


def _tables_case(tmp_path):
    """results / rem_input exactly as the reference's tool_test handed them to its
    generate_output_tables (tests/golden/tables.npz; exec_R stubbed, see make_golden.py)."""
    import argparse
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "tables.npz"), allow_pickle=False)
    bpc = g["bins_per_chr"]
    off = np.concatenate(([0], np.cumsum(bpc)))
    split = lambda a: [a[off[c]:off[c + 1]] for c in range(len(bpc))]
    segs = [[int(s[0]), int(s[1]), int(s[2]), "nan" if isstr else float(s[3]), float(s[4])]
            for s, isstr in zip(g["results_c_num"], g["results_c_isstr"])]
    results = {"results_r": split(g["results_r"]), "results_z": split(g["results_z"]),
               "results_w": split(g["results_w"]), "results_nr": split(g["results_nr"]),
               "results_c": segs}
    regions = tmp_path / "regions.bed"
    regions.write_text(str(g["regions_text"]))
    args = argparse.Namespace(outid=str(tmp_path / "ID"), zscore=float(g["zscore"]), beta=None,
                              regions=str(regions))
    rem = {"args": args, "binsize": int(g["binsize"]), "n_reads": int(g["n_reads"]),
           "ref_gender": str(g["ref_gender"]), "gender": str(g["gender"]), "bins_per_chr": bpc}
    return g, rem, results


def test_output_tables_byte_identical_to_reference(tmp_path):
    """f3 pin: ID_bins.bed, ID_segments.bed, ID_aberrations.bed, ID_regions.bed are BYTE-identical
    to the files the reference's predict_output.py:51-194 wrote for the same results."""
    from wisecondorx_amd import predict_output as po
    g, rem, results = _tables_case(tmp_path)
    po._generate_bins_bed(rem, results)
    po._generate_segments_and_aberrations_bed(rem, results)
    po._generate_regions_bed(rem, results)
    for suffix in ("_bins.bed", "_segments.bed", "_aberrations.bed", "_regions.bed"):
        mine = open(rem["args"].outid + suffix).read()
        assert mine == str(g["file" + suffix.replace(".", "_")]), suffix


def test_ref_qc_matches_reference(g_pipe):
    """f4: ref_qc.compute_metrics / qc_reference vs the reference's own ref_qc.py run on the golden
    reference and on doctored copies reaching the WARN and FAIL rules (tests/golden/ref_qc.npz)."""
    from conftest import GOLDEN, ref_dict_from_golden
    from wisecondorx_amd import ref_qc
    g = np.load(os.path.join(GOLDEN, "ref_qc.npz"), allow_pickle=False)
    plain = ref_dict_from_golden(g_pipe)
    wide = dict(plain)
    for suf in (".F", ".M"):
        wide["indexes" + suf] = np.tile(plain["indexes" + suf], (1, 3))
        wide["distances" + suf] = np.tile(plain["distances" + suf], (1, 3))
    spread = dict(wide)
    spread["distances.F"] = wide["distances.F"] * g["spread_scale_F"][:, None] * 4e1
    heavy = dict(wide)
    heavy["distances.M"] = wide["distances.M"] * 1e4
    heavy["distances.F"] = wide["distances.F"] * g["spread_scale_F"][:, None] * 4e3
    codes = {}
    for name, ref in (("plain", plain), ("wide", wide), ("spread", spread), ("heavy", heavy)):
        codes[name] = ref_qc.qc_reference(ref)
        assert codes[name] == int(g[name + "_code"]), name
        for suf in (".F", ".M"):
            m = ref_qc.compute_metrics(ref, suf)
            for key in ("n_bins", "n_valid", "n_mean_outlier", "n_low_refs"):
                assert m[key] == int(g["{}{}_{}".format(name, suf, key)]), (name, suf, key)
            for key in ("mean_of_means", "std_of_means", "outlier_pct"):
                np.testing.assert_allclose(m[key], float(g["{}{}_{}".format(name, suf, key)]), rtol=1e-12)
            if suf == ".M":
                for key, val in m["chrY"].items():
                    np.testing.assert_allclose(val, float(g["{}{}_chrY_{}".format(name, suf, key)]),
                                               rtol=1e-12)
    assert sorted(set(codes.values())) == [0, 1, 2]          # PASS, WARN and FAIL all reached
    assert ref_qc.qc_reference(os.path.join(GOLDEN, "no_such_file.npz")) == 2


def test_post_process_fused_equals_the_step_by_step_path():
    """post_process_fused == get_post_processed_result x3 + log_trans (the reference's sequence,
    predict_control.py:49-63 + predict_tools.py:180-193), incl. zero / negative / inf / nan ratios,
    ratio == 1 (log2 = 0: not shifted) and bins below minrefbins; its per-chromosome arrays are
    views that _flatten returns without a copy."""
    from types import SimpleNamespace
    from wisecondorx_amd import predict_tools as pt
    rng = np.random.default_rng(5)
    bpc = [40, 25, 31, 17]
    nb = sum(bpc)
    mask = rng.random(nb) > 0.2
    B = int(mask.sum())
    r = np.exp(rng.normal(0, 0.1, B))
    r[[0, 3, 5, 7, 9, 11]] = [0.0, -1.0, np.inf, np.nan, 1.0, 1e-300]
    z = rng.normal(0, 1, B)
    z[2] = np.nan
    w = rng.uniform(0.5, 2, B)
    n = rng.integers(100, 300, B).astype(float)
    args = SimpleNamespace(minrefbins=150)
    rem = {"mask": mask, "bins_per_chr": bpc}
    ref = {"results_r": r, "results_z": z, "results_w": w}
    for key in ref:
        ref[key] = pt.get_post_processed_result(args, ref[key], n, rem)
    pt.log_trans(ref, 0.0123)
    got = pt.post_process_fused(args, r, z, w, n, 0.0123, rem)
    for key in ref:
        assert len(got[key]) == len(bpc)
        for c in range(len(bpc)):
            np.testing.assert_array_equal(got[key][c], ref[key][c])
        flat = pt._flatten(got, key)
        assert flat.base is got[key][0].base or flat is got[key][0].base    # no copy
        np.testing.assert_array_equal(flat, np.concatenate(ref[key]))
        np.testing.assert_array_equal(pt._flatten(got, key, 2), np.concatenate(ref[key][:2]))
        np.testing.assert_array_equal(pt._flatten(ref, key), np.concatenate(ref[key]))


def test_reference_npz_reader_detects_corruption(tmp_path):
    """the direct reader of the big stored members verifies their CRC-32 (chunk CRCs from
    the reader threads, combined) and their sizes, like zipfile would; the writer replaces the target
    only after every write succeeded."""
    import zlib
    from wisecondorx_amd import npz_io
    a, b = os.urandom(1000), os.urandom(777)
    assert npz_io.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert npz_io.crc32_combine_py(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert npz_io.crc32_combine_py(zlib.crc32(a), zlib.crc32(b""), 0) == zlib.crc32(a)
    assert npz_io.crc32_combine(zlib.crc32(a), zlib.crc32(b""), 0) == zlib.crc32(a)
    rng = np.random.default_rng(0)
    arrs = {"indexes": rng.integers(0, 1000, (30000, 100)).astype(np.int32),
            "distances": rng.random((30000, 100)), "binsize": np.array(15000)}
    path = npz_io.save_npz(str(tmp_path / "ref.npz"), arrs)
    assert sorted(os.listdir(tmp_path)) == ["ref.npz"]                 # no .tmp left behind
    got = npz_io.load_reference(path)
    assert np.array_equal(got["indexes"], arrs["indexes"]) and np.array_equal(got["distances"], arrs["distances"])
    with np.load(path) as z:                                           # still an ordinary .npz
        assert np.array_equal(z["distances"], arrs["distances"])
    size = os.path.getsize(path)
    with open(path, "r+b") as fh:                                      # flip one bit of a big member
        fh.seek(size // 2)
        c = fh.read(1)
        fh.seek(size // 2)
        fh.write(bytes([c[0] ^ 4]))
    with pytest.raises(IOError, match="CRC-32 mismatch"):
        npz_io.load_reference(path)


def test_gender_model_cutoff_matches_reference():
    """f4: main.train_gender_model WITHOUT --yfrac against the reference's own run
    (tests/golden/gender.npz, made by importing newref_tools.train_gender_model): the mixture fit and
    the first local minimum of its density (newref_tools.py:36-62) give the same cut-off and the
    same genders."""
    import argparse
    from conftest import GOLDEN
    from wisecondorx_amd import main
    g = np.load(os.path.join(GOLDEN, "gender.npz"))
    samples = [{"1": np.array([o // 2, o - o // 2]), "24": np.array([yy])} for o, yy in zip(g["other"], g["y"])]
    np.random.seed(int(g["np_seed"]))
    genders, cut = main.train_gender_model(argparse.Namespace(plotyfrac=None, yfrac=None), samples)
    assert cut == float(g["cut_off"])
    assert genders == [str(x) for x in g["genders"]]
    assert [x == "M" for x in genders] == g["is_m"].tolist()
    # --yfrac given: no mixture, the value is used as is (newref_tools.py:55-56)
    g2, c2 = main.train_gender_model(argparse.Namespace(plotyfrac=None, yfrac=0.002), samples)
    assert c2 == 0.002 and g2 == genders


def test_newref_mask_skew_default_follows_upstream():
    """newref's default lets a gonosomal pass drop autosomal bins like upstream (frozen = 0); the fix is
    opt-in.  The fixture records what upstream's own predict does with such a reference: IndexError."""
    import os
    from wisecondorx_amd import main
    a = main.build_parser().parse_args(["newref", "a.npz", "out.npz"])
    assert a.aligned_masks is False
    assert a.reference_mask_skew is False
    assert main.build_parser().parse_args(["newref", "a.npz", "out.npz", "--aligned-masks"]).aligned_masks
    assert main.build_parser().parse_args(["newref", "a.npz", "out.npz",
                                           "--reference-mask-skew"]).reference_mask_skew    # deprecated alias
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_skew.npz"))
    n_aut = int(np.sum(g["cohort_bpc"][:22]))
    assert g["mask"][:n_aut].sum() == g["mask_F"][:n_aut].sum() + 1
    assert str(g["predict_results_nr"]) == "IndexError"
    assert g["len_merged"].tolist()[0] == g["len_merged"].tolist()[1] + 1


def test_npz_writer_streams_members_and_aborts_cleanly(tmp_path):
    """NpzWriter (the reference file is written while newref still computes): members added one by
    one with early write-back in between load back bit for bit; an aborted writer leaves no file."""
    from wisecondorx_amd import npz_io
    rng = np.random.default_rng(5)
    big = rng.standard_normal((3000, 300))
    idx = np.asfortranarray(rng.integers(0, 1 << 30, (2000, 300)).astype(np.int32))
    path = str(tmp_path / "ref.npz")
    w = npz_io.NpzWriter(path)
    w.add("distances", big)
    w.flush_async()
    w.add("indexes", idx)
    w.add("mask", np.arange(1000) % 3 == 0)
    w.flush_async()
    w.add("trained_cutoff", 0.25)
    w.add("is_nipt", False)
    assert w.close() == path
    with np.load(path) as z:
        assert set(z.files) == {"distances", "indexes", "mask", "trained_cutoff", "is_nipt"}
        assert np.array_equal(z["distances"], big) and np.array_equal(z["indexes"], idx)
        assert float(z["trained_cutoff"]) == 0.25 and not bool(z["is_nipt"])
    import zipfile
    assert zipfile.ZipFile(path).testzip() is None          # every CRC-32 right
    w = npz_io.NpzWriter(str(tmp_path / "gone.npz"))
    w.add("distances", big)
    w.flush_async()
    w.abort()
    assert sorted(os.listdir(tmp_path)) == ["ref.npz"]


def test_sample_counts_matrix_threaded_and_in_place():
    """The batch count matrix (predict_tools.py:36-44 layout: truncate / zero-pad every chromosome to the
    reference's bin count): threaded fill == the plain statement, also into a dirty preallocated buffer."""
    from wisecondorx_amd import predict_tools as pt
    rng = np.random.default_rng(0)
    bpc = [50, 40, 30, 7]
    ref = {"bins_per_chr": np.array(bpc)}
    samples = [{str(c + 1): rng.integers(0, 100, size=n + int(rng.integers(-3, 4))).astype(np.int32)
                for c, n in enumerate(bpc)} for _ in range(20)]
    starts = np.concatenate(([0], np.cumsum(bpc)))
    want = np.zeros((len(samples), int(starts[-1])), dtype=np.int32)
    for i, s in enumerate(samples):
        for c, n_ref in enumerate(bpc):
            n = min(n_ref, len(s[str(c + 1)]))
            want[i, starts[c]:starts[c] + n] = s[str(c + 1)][:n]
    assert np.array_equal(pt.sample_counts_matrix(samples, ref, ""), want)
    buf = np.full((25, int(starts[-1])), -7, dtype=np.int32)
    got = pt.sample_counts_matrix(samples[:5], ref, "", out=buf)
    assert got.shape == (5, int(starts[-1])) and np.array_equal(got, want[:5])


def test_native_float_format_is_pythons_repr():
    """wcx_format_floats (the native writer of ID_bins.bed) == repr(float) on special values, random
    doubles over 60 orders of magnitude and random bit patterns (host code: runs without a GPU)."""
    import struct
    from wisecondorx_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(0)
    vals = [0.1, 0.5, 1.0, -1.0, 1e-4, 1e-5, 9.999e-5, 1e15, 1e16, 1.5e16, 123456789.123, 1 / 3, -2 / 3, 5e-324,
            1.7976931348623157e308, 2.5, 100.0, 1e22, 1e23, 0.30000000000000004, float("nan"), float("inf"),
            float("-inf"), 1234567890123456.0, 12345678901234567.0, 0.001, 0.0001234, -0.0]
    vals += list(rng.normal(0, 1, 20000)) + list(np.exp(rng.normal(0, 30, 20000)))
    vals += [struct.unpack("d", struct.pack("Q", int(x)))[0] for x in rng.integers(0, 2 ** 63, 20000)]
    a = np.array(vals, dtype=np.float64)
    buf = np.empty(len(a) * 26, dtype=np.uint8)
    m = L.wcx_format_floats(_lib.ptr(a), len(a), b"\n", _lib.ptr(buf), buf.size)
    got = bytes(buf[:m]).decode().split("\n")[:-1]
    assert got == [repr(float(x)) for x in a.tolist()]
    assert L.wcx_format_floats(_lib.ptr(a), len(a), b"\n", _lib.ptr(buf), 10) == -1     # (capacity checked)


def test_top_eigh_equals_the_full_solve():
    """prep.top_eigh (LAPACK dsyevx for the 5 largest pairs) against numpy.linalg.eigh: eigenvalues to
    1e-14 relative, eigenvectors to 1e-12 up to sign, also for S <= k and a rank-deficient Gram."""
    from wisecondorx_amd import prep
    rng = np.random.default_rng(4)
    for S, B in ((3, 50), (12, 400), (60, 40), (200, 3000)):
        A = rng.normal(size=(B, S)) * np.linspace(1, 20, S)
        G = A.T @ A
        w, v = prep.top_eigh(G, 5)
        w0, v0 = np.linalg.eigh(G)
        k = min(5, S)
        w0, v0 = w0[::-1][:k], v0[:, ::-1][:, :k]
        assert w.shape == (k,) and v.shape == (S, k)
        np.testing.assert_allclose(w, w0, rtol=1e-13, atol=1e-13 * w0[0])
        np.testing.assert_allclose(np.abs(v), np.abs(v0), atol=1e-11)
        np.testing.assert_allclose(G @ v, v * w, atol=1e-11 * w0[0])


def test_reference_npz_deferred_members(tmp_path):
    """load_reference(defer=(".F", ".M")): the big members of the deferred suffixes are planned, not read
    (a predict needs ONE gonosomal set); ensure_loaded reads a set later -- same arrays, same CRC check."""
    from wisecondorx_amd import npz_io
    rng = np.random.default_rng(1)
    arrs = {"binsize": np.array(15000), "mask.F": rng.random(1000) < 0.5}
    for sfx in ("", ".F", ".M"):
        arrs["indexes" + sfx] = rng.integers(0, 1000, (30000, 100)).astype(np.int32)
        arrs["distances" + sfx] = rng.random((30000, 100))
    path = npz_io.save_npz(str(tmp_path / "ref.npz"), arrs)
    ref = npz_io.load_reference(path, defer=(".F", ".M"))
    assert sorted(ref.deferred) == ["distances.F", "distances.M", "indexes.F", "indexes.M"]
    assert "indexes.F" not in ref and np.array_equal(ref["indexes"], arrs["indexes"])
    assert np.array_equal(ref["mask.F"], arrs["mask.F"])                 # (small members are never deferred)
    npz_io.ensure_loaded(ref, ".F")
    assert sorted(ref.deferred) == ["distances.M", "indexes.M"]
    assert np.array_equal(ref["indexes.F"], arrs["indexes.F"]) and np.array_equal(ref["distances.F"], arrs["distances.F"])
    npz_io.ensure_loaded(ref, ".F")                                      # (idempotent)
    everything = npz_io.load_reference(path, defer=("",))
    assert len(everything.deferred) == 6 and int(everything["binsize"]) == 15000
    # a corrupted deferred member is caught when it is read
    with zipfile_member_offset(path, "distances.M.npy") as off:
        with open(path, "r+b") as fh:
            fh.seek(off + 4096)
            c = fh.read(1)
            fh.seek(off + 4096)
            fh.write(bytes([c[0] ^ 1]))
    ref2 = npz_io.load_reference(path, defer=(".M",))                    # (the other members are intact)
    with pytest.raises(IOError, match="CRC-32 mismatch"):
        npz_io.ensure_loaded(ref2, ".M")
    # ... and leaves nothing behind: no half-read table in the dict (the intact "indexes.M" of the same
    # call included), the plans still there, so a later access raises and a retry fails the same way
    assert "distances.M" not in ref2 and "indexes.M" not in ref2
    assert sorted(ref2.deferred) == ["distances.M", "indexes.M"]
    with pytest.raises(IOError, match="CRC-32 mismatch"):
        npz_io.ensure_loaded(ref2, ".M")


def test_reference_npz_constant_prefix_members(tmp_path):
    """npz_io.PrefixConst -- the tables of a gonosomal pass, whose autosomal rows are the reference's
    dummies (newref_tools.py:186-191): written as ONE deflated member (constant rows as pre-built deflate
    blocks, tail in stored blocks) that np.load inflates to the very array the reference's own writer
    would have stored, that zipfile's CRC check accepts, and that load_reference rebuilds from its
    description in the ZIP extra field (directly, deferred, and with a damaged tail detected)."""
    import zipfile
    from wisecondorx_amd import npz_io
    rng = np.random.default_rng(3)
    ct, g, k = 150000, 4111, 25
    tail_i = rng.integers(-1, 9000, (g, k)).astype(np.int32)
    tail_d = rng.random((g, k))
    full_i = np.concatenate([np.zeros((ct, k), np.int32), tail_i])
    full_d = np.concatenate([np.ones((ct, k)), tail_d])
    pc_i, pc_d = npz_io.PrefixConst(ct, 0, tail_i), npz_io.PrefixConst(ct, 1.0, tail_d)
    assert pc_d.shape == full_d.shape and len(pc_i) == ct + g and pc_d.nbytes == full_d.nbytes
    assert np.array_equal(np.asarray(pc_i), full_i) and np.array_equal(pc_d[ct - 1:ct + 2], full_d[ct - 1:ct + 2])
    arrs = {"binsize": np.array(15000), "indexes.F": pc_i, "distances.F": pc_d,
            "indexes.M": npz_io.PrefixConst(ct, 0, tail_i[:0]),                 # (no gonosomal row at all)
            "distances.M": npz_io.PrefixConst(7, 1.0, tail_d),                  # (short prefix: stored as ever)
            "null_ratios.F": rng.random((ct + g, 10))}
    path = npz_io.save_npz(str(tmp_path / "ref.npz"), arrs)
    assert os.path.getsize(path) < full_d.nbytes // 8 + arrs["null_ratios.F"].nbytes + arrs["distances.M"].nbytes
    with zipfile.ZipFile(path) as zf:
        assert zf.testzip() is None                                            # (every member's CRC-32)
        assert zf.getinfo("distances.F.npy").compress_type == zipfile.ZIP_DEFLATED
        assert npz_io._hybrid_info(zf.getinfo("distances.F.npy").extra)["prefix_bytes"] == ct * k * 8
        assert npz_io._hybrid_info(zf.getinfo("distances.M.npy").extra) is None     # (an ordinary member)
    with np.load(path) as z:                                                   # what the reference's predict does
        assert z["indexes.F"].dtype == np.int32 and np.array_equal(z["indexes.F"], full_i)
        assert z["distances.F"].dtype == np.float64 and np.array_equal(z["distances.F"], full_d)
        assert z["indexes.M"].shape == (ct, k) and not z["indexes.M"].any()
        assert np.array_equal(z["distances.M"], np.concatenate([np.ones((7, k)), tail_d]))
    ref = npz_io.load_reference(path)
    assert np.array_equal(ref["indexes.F"], full_i) and np.array_equal(ref["distances.F"], full_d)
    assert ref["indexes.M"].shape == (ct, k) and not ref["indexes.M"].any()
    assert np.array_equal(ref["null_ratios.F"], arrs["null_ratios.F"])
    ref = npz_io.load_reference(path, defer=(".F", ".M"))
    assert "distances.F" in ref.deferred and "distances.F" not in ref
    npz_io.ensure_loaded(ref, ".F")
    assert np.array_equal(ref["distances.F"], full_d) and np.array_equal(ref["indexes.F"], full_i)
    # one flipped bit in the tail's stored blocks (payload, then a block header)
    with zipfile.ZipFile(path) as zf:
        zi = zf.getinfo("distances.F.npy")
        with open(path, "rb") as fh:
            fh.seek(zi.header_offset + 26)
            nlen, elen = np.frombuffer(fh.read(4), dtype="<u2")
        end = zi.header_offset + 30 + int(nlen) + int(elen) + zi.compress_size
    for back, what in ((1000, "CRC-32 mismatch"), (tail_d.nbytes % 65535 + 4, "stored-block")):
        bad = str(tmp_path / "bad{}.npz".format(back))
        data = bytearray(open(path, "rb").read())
        data[end - back] ^= 0x10
        open(bad, "wb").write(bytes(data))
        with pytest.raises(IOError, match=what):
            npz_io.load_reference(bad)


class zipfile_member_offset:
    """Context manager: offset of the data of a stored member inside the archive."""

    def __init__(self, path, name):
        import struct
        import zipfile
        with zipfile.ZipFile(path) as zf, open(path, "rb") as fh:
            info = zf.getinfo(name)
            fh.seek(info.header_offset)
            lh = fh.read(30)
            nlen, elen = struct.unpack("<HH", lh[26:30])
            self.off = info.header_offset + 30 + nlen + elen

    def __enter__(self):
        return self.off

    def __exit__(self, *a):
        return False
