"""CPU: host-side glue (bin rescaling, PCA, partition helpers, CLI surface, .npz formats)."""
import numpy as np

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
    ref = {"indexes": np.zeros((3000, 300), np.int32), "distances": np.ones((3000, 300)),
           "mask": np.ones(10, bool), "binsize": 100000, "is_nipt": False, "trained_cutoff": 0.004}
    rp = npz_io.save_npz(str(tmp_path / "r.npz"), ref)
    back = np.load(rp, encoding="latin1", allow_pickle=True)     # how the reference reads it
    assert set(back.files) == set(ref)
    assert back["indexes"].dtype == np.int32 and int(back["binsize"]) == 100000
    assert not back["is_nipt"] and float(back["trained_cutoff"]) == 0.004
