#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE (/root/reference) in the build
container and running its own functions on small synthetic inputs.

Only data is committed: inputs and the reference's outputs.  No reference source is copied.
Run:  python tests/golden/make_golden.py     (needs /root/reference; never runs on the GPU box)

Shims:
  * pysam is absent -> a dummy module so `import wisecondorx.main` works (convert is not on
    the path).
  * predict_tools.project_pc raises on scikit-learn >= 1.5 -> patched with the <=1.4.2
    transform semantics the reference pins (setup.cfg:42).
  * CBS needs R/DNAcopy (absent) -> exec_R patched with hand-made segments.
"""
import argparse
import os
import random
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.modules.setdefault("pysam", types.ModuleType("pysam"))
warnings.filterwarnings("ignore")

import wisecondorx.main as ref_main                      # noqa: E402
import wisecondorx.newref_tools as ref_nt                # noqa: E402
import wisecondorx.predict_tools as ref_pt               # noqa: E402
import wisecondorx.predict_control as ref_pc             # noqa: E402
import wisecondorx.overall_tools as ref_ot               # noqa: E402

from wisecondorx_amd.synth import Cohort, corrected_matrix   # noqa: E402


def _project_pc_le14(sample_data, ref_file, ap):
    comp = ref_file["pca_components{}".format(ap)]
    mean = ref_file["pca_mean{}".format(ap)]
    t = np.dot(np.array([sample_data]) - mean, comp.T)
    rec = (np.dot(t, comp) + mean)[0]
    return sample_data / rec


ref_pt.project_pc = _project_pc_le14
ref_pc.project_pc = _project_pc_le14


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


# --------------------------------------------------------------------------- newref search
def golden_newref_search():
    out = {}
    # realistic values; 24 chromosomes so A / F-style / M-style passes all exist
    mb24 = [70, 66, 55, 52, 50, 47, 44, 40, 38, 37, 37, 36, 31, 29, 28, 25, 23, 22, 16, 18,
            13, 14, 43, 16]
    X, _, _ = corrected_matrix(mb24, 30, seed=7)
    assert X.flags["F_CONTIGUOUS"]
    out["Xs"] = np.ascontiguousarray(X.T)      # sample-major bytes of the F-ordered (B,S)
    for tag, nchr, part, parts in (("A", 22, 2, 3), ("A1", 22, 1, 1), ("F", 23, 1, 1),
                                   ("M", 24, 1, 1), ("M2", 24, 2, 2)):
        mb = mb24[:nchr]
        cum = np.cumsum(mb).tolist()
        Xp = X[:cum[-1], :]
        assert Xp.flags["F_CONTIGUOUS"] or True
        Xp = np.asfortranarray(Xp)
        random.seed(1000 + nchr + part)
        idx, dist, nr = ref_nt.get_reference(Xp, mb, cum, 40, part, parts)
        random.seed(1000 + nchr + part)
        ids = random.sample(range(Xp.shape[1]), min(Xp.shape[1], 100))
        out[tag + "_mb"] = np.array(mb)
        out[tag + "_part"] = np.array([part, parts])
        out[tag + "_idx"] = idx
        out[tag + "_dist"] = dist
        out[tag + "_nr"] = nr
        out[tag + "_ids"] = np.array(ids)
    # tie-heavy integer case (many exactly equal distances), direct get_ref_for_bins
    rng = np.random.default_rng(3)
    Xi = np.asfortranarray(rng.integers(0, 3, (120, 6)).astype(np.float64))
    chr_data = np.concatenate((Xi[:20, :], Xi[50:, :]))
    ti, td = ref_nt.get_ref_for_bins(25, 20, 50, Xi, chr_data)
    out["tie_Xs"] = np.ascontiguousarray(Xi.T)
    out["tie_idx"], out["tie_dist"] = ti, td
    # fewer than k candidates -> -1 / 1e10 padding
    Xf = np.asfortranarray(1.0 + 0.05 * rng.standard_normal((30, 8)))
    chr_data = np.concatenate((Xf[:10, :], Xf[18:, :]))
    fi, fd = ref_nt.get_ref_for_bins(40, 10, 18, Xf, chr_data)
    out["few_Xs"] = np.ascontiguousarray(Xf.T)
    out["few_idx"], out["few_dist"] = fi, fd
    # NaN / inf / huge candidates are never admitted
    Xn = np.asfortranarray(1.0 + 0.05 * rng.standard_normal((60, 8)))
    Xn[5, 3] = np.nan
    Xn[40, 0] = np.inf
    Xn[41, 2] = 3e5          # d ~ 9e10 >= 1e10 -> not admitted
    Xn[25, 1] = np.nan       # a NaN TARGET row: all distances NaN -> all padding
    chr_data = np.concatenate((Xn[:20, :], Xn[30:, :]))
    ni, nd = ref_nt.get_ref_for_bins(45, 20, 30, Xn, chr_data)
    out["nan_Xs"] = np.ascontiguousarray(Xn.T)
    out["nan_idx"], out["nan_dist"] = ni, nd
    save("newref_search.npz", **out)


# --------------------------------------------------------------------------- end-to-end
def write_sample(path, sample, binsize):
    np.savez_compressed(path, binsize=binsize, sample=sample, quality={})


def golden_pipeline():
    binsize = 4000000
    co = Cohort(binsize, struct_seed=11, female_y=0.1)
    samples, genders = co.cohort(24, seed0=500, reads=4e6)
    tmp = tempfile.mkdtemp(prefix="wcx_golden_")
    infiles = []
    for i, s in enumerate(samples):
        p = os.path.join(tmp, "s{}.npz".format(i))
        write_sample(p, s, binsize)
        infiles.append(p)
    args = argparse.Namespace(infiles=infiles, outfile=os.path.join(tmp, "ref.npz"),
                              nipt=False, yfrac=0.004, plotyfrac=None, refsize=60,
                              binsize=binsize, cpus=1)
    np.random.seed(5)
    random.seed(5)
    try:
        ref_main.tool_newref(args)
    except NameError as e:          # main.py:135 qc_reference never imported
        print("expected reference bug:", e)
    ref = np.load(args.outfile, encoding="latin1", allow_pickle=True)
    ref_dict = {k: ref[k] for k in ref.files}
    print({k: (v.shape, v.dtype) for k, v in ref_dict.items()})

    out = {"ref__" + k: v for k, v in ref_dict.items()}
    # cohort counts so the build's own newref can be run on identical inputs
    out["cohort_counts"] = np.stack([np.concatenate([s[str(c)] for c in range(1, 25)])
                                     for s in samples])
    out["cohort_genders"] = np.array(genders)
    out["cohort_bpc"] = np.array(co.bpc)

    pargs = argparse.Namespace(maskrepeats=5, minrefbins=20)
    tests = {
        "t0": co.sample(9001, "M", reads=4e6, cnv=[(3, 10, 25, 1.5)]),
        "t1": co.sample(9002, "F", reads=4e6, cnv=[(7, 5, 15, 0.5), (23, 10, 20, 1.5)]),
        "t2": co.sample(9003, "M", reads=4e6),
    }
    # t2: zeros + wild outliers to trigger the -1 masking and degenerate bins
    t2 = tests["t2"]
    t2["5"][3:9] = 0
    t2["9"][7] *= 40
    for name, sample0 in tests.items():
        gender = ref_pt.predict_gender(sample0, ref["trained_cutoff"])
        sample = {k: v.copy() for k, v in sample0.items()}
        sample = ref_ot.gender_correct(sample, gender)
        out[name + "_counts"] = np.concatenate([sample0[str(c)] for c in range(1, 25)])
        out[name + "_gender"] = np.array(gender)
        rA = ref_pc.normalize(pargs, sample, ref, "A")
        rG = ref_pc.normalize(pargs, sample, ref, gender)
        for tag, res in (("A", rA), ("G", rG)):
            for nm, v in zip(("r", "z", "w", "n", "mlr", "mz"), res):
                out["{}_{}_{}".format(name, tag, nm)] = np.asarray(v)
        # stage values feeding normalize_repeat
        ap = ""
        xA = ref_pt.coverage_normalize_and_mask(sample, ref, ap)
        out[name + "_A_cov"] = xA
        out[name + "_A_x"] = ref_pt.project_pc(xA, ref, ap)
        ap = "." + gender
        xG = ref_pt.coverage_normalize_and_mask(sample, ref, ap)
        out[name + "_G_x"] = ref_pt.project_pc(xG, ref, ap)
        out[name + "_cutoff"] = np.array(ref_pt.get_optimal_cutoff(ref, 5))

        # merge + post-processing exactly as main.py:216-275
        results_r, results_z, results_w, ref_sizes, m_lr, m_z = rA
        results_r_2, results_z_2, results_w_2, ref_sizes_2, _, _ = rG
        nr_aut = ref["null_ratios"]
        nr_gon = ref["null_ratios.{}".format(gender)][len(nr_aut):]
        rem_input = {
            "args": pargs, "binsize": int(ref["binsize"]), "ref_gender": gender,
            "mask": ref["mask.{}".format(gender)],
            "bins_per_chr": ref["bins_per_chr.{}".format(gender)],
        }
        results_r = np.append(results_r, results_r_2)
        results_z = np.append(results_z, results_z_2) - m_z
        results_w = np.append(results_w * np.nanmean(results_w_2),
                              results_w_2 * np.nanmean(results_w))
        results_w = results_w / np.nanmean(results_w)
        ref_sizes = np.append(ref_sizes, ref_sizes_2)
        null_ratios = np.array([x.tolist() for x in nr_aut] + [x.tolist() for x in nr_gon],
                               dtype=object)
        results = {"results_r": results_r, "results_z": results_z, "results_w": results_w,
                   "results_nr": null_ratios}
        for k in results.keys():
            results[k] = ref_pc.get_post_processed_result(pargs, results[k], ref_sizes,
                                                          rem_input)
        ref_pt.log_trans(results, m_lr)
        if name == "t1":
            bl = os.path.join(tmp, "bl.bed")
            with open(bl, "w") as fh:
                fh.write("chr2\t8000000\t19000000\nX\t0\t7900000\nchrY\t0\t100\n")
            pargs.blacklist = bl
            rem_input["args"] = pargs
            ref_pt.apply_blacklist(rem_input, results)
            out["t1_blacklist"] = np.array([["chr2", "8000000", "19000000"],
                                            ["X", "0", "7900000"], ["chrY", "0", "100"]])
        flat = lambda key: np.concatenate([np.asarray(c, dtype=float) for c in results[key]])
        out[name + "_post_r"] = flat("results_r")
        out[name + "_post_z"] = flat("results_z")
        out[name + "_post_w"] = flat("results_w")
        nchr = len(results["results_r"])
        # null ratios are ragged in the reference (autosomal rows have min(S,100) columns,
        # gonosomal rows min(S_gender,100)); store one matrix per group, zero rows where the
        # reference holds the int 0 placeholder (predict_tools.py:164).
        def nr_block(chrs, m):
            rows = []
            for c in chrs:
                for row in results["results_nr"][c]:
                    rows.append(np.asarray(row, dtype=float) if isinstance(row, (list, np.ndarray))
                                else np.zeros(m))
            return np.array(rows)
        out[name + "_post_nrA"] = nr_block(range(22), nr_aut.shape[1])
        out[name + "_post_nrG"] = nr_block(range(22, nchr), nr_gon.shape[1])
        # segment z on hand-made segments (CBS itself needs R/DNAcopy: unpinned)
        bpc = list(rem_input["bins_per_chr"])
        segs = []
        for c in range(nchr):
            n = bpc[c]
            cuts = [0, n // 3, n // 3 + 2, n] if c % 2 == 0 else [0, n]
            for a, b in zip(cuts[:-1], cuts[1:]):
                if b - a < 1:
                    continue
                rr = np.asarray(results["results_r"][c][a:b], dtype=float)
                ww = np.asarray(results["results_w"][c][a:b], dtype=float)
                keep = rr != 0
                segr = float(np.sum(rr[keep] * ww[keep]) / np.sum(ww[keep])) if keep.any() else 0.0
                segs.append([c, a, b, segr])
        segs.append([0, 0, 1, 50.0])      # clipping to +1000
        zs = ref_ot.get_z_score([list(s) for s in segs], results)
        out[name + "_segs"] = np.array(segs, dtype=float)
        out[name + "_segz"] = np.array([np.nan if isinstance(z, str) else float(z) for z in zs])
        out[name + "_segz_isstr"] = np.array([isinstance(z, str) for z in zs])
    save("pipeline.npz", **out)
    golden_ref_qc(args.outfile, ref_dict)


def golden_ref_qc(ref_path, ref_dict):
    """The reference's ref_qc on the golden reference file and on two doctored copies (a WARN and a
    FAIL case): per-sub-reference metrics (ref_qc._compute_metrics) and the overall return code."""
    import wisecondorx.ref_qc as ref_qc
    out = {}
    cases = {"plain": dict(ref_dict)}
    wide = dict(ref_dict)            # refsize 180 >= MINREFBINS, so that the other rules are reached
    for suf in (".F", ".M"):
        wide["indexes" + suf] = np.tile(ref_dict["indexes" + suf], (1, 3))
        wide["distances" + suf] = np.tile(ref_dict["distances" + suf], (1, 3))
    cases["wide"] = wide
    warn = dict(wide)
    warn["distances.F"] = np.array(wide["distances.F"]) * np.linspace(1, 60, len(wide["distances.F"]))[:, None] * 4e1
    cases["spread"] = warn
    fail = dict(wide)
    fail["distances.M"] = np.array(wide["distances.M"]) * 1e4
    fail["distances.F"] = np.array(wide["distances.F"]) * np.linspace(1, 60, len(wide["distances.F"]))[:, None] * 4e3
    cases["heavy"] = fail
    for name, d in cases.items():
        path = os.path.join(os.path.dirname(ref_path), "qc_{}.npz".format(name))
        np.savez(path, **d)
        out[name + "_code"] = np.array(ref_qc.qc_reference(path))
        npz = np.load(path, encoding="latin1", allow_pickle=True)
        for suf in (".F", ".M"):
            m = ref_qc._compute_metrics(npz, suf)
            for key in ("n_bins", "n_valid", "mean_of_means", "std_of_means", "n_mean_outlier",
                        "outlier_pct", "n_low_refs"):
                out["{}{}_{}".format(name, suf, key)] = np.array(m[key])
            if m.get("chrY"):
                for key, val in m["chrY"].items():
                    out["{}{}_chrY_{}".format(name, suf, key)] = np.array(val)
    out["spread_scale_F"] = np.linspace(1, 60, len(wide["distances.F"]))
    save("ref_qc.npz", **out)


# --------------------------------------------------------------------------- a1/a3 prep pins
def golden_prep_filter():
    """A cohort in which the PCA-distance bin filter (newref_control.py:38-54) fires in the A
    pass AND again in the F pass on an autosomal bin: the shared mask is mutated in place after
    the A pass saved its copy, so mask.F/.M hold one autosomal bin fewer than mask (SURVEY.md
    "latent A/F/M mask skew").  Captured: the reference's prep outputs of the three passes."""
    import wisecondorx.newref_control as ref_nc
    binsize = 4000000
    co = Cohort(binsize, struct_seed=21, female_y=0.1)
    samples, genders = co.cohort(28, seed0=700, reads=4e6)
    rng = np.random.default_rng(9)
    sd = 1.6
    spec = [(2, 10, "FM"), (5, 7, "F"), (6, 7, "F"), (7, 9, "F"), (8, 3, "F"), (9, 3, "M"),
            (10, 3, "M"), (11, 5, "M"), (12, 5, "M"), (3, 5, "F"), (4, 5, "F"), (13, 5, "F"),
            (24, 4, "M"), (23, 11, "F")]
    for c, b, who in spec:          # per-sample heavy multiplicative noise on single bins
        for i, s in enumerate(samples):
            if genders[i] in who:
                s[str(c)][b] = int(s[str(c)][b] * np.exp(rng.normal(0, sd)))
    out = {"cohort_counts": np.stack([np.concatenate([s[str(c)] for c in range(1, 25)])
                                      for s in samples]),
           "cohort_genders": np.array(genders), "cohort_bpc": np.array(co.bpc)}
    samples = np.array([ref_ot.gender_correct(s, g) for s, g in zip(samples, genders)])
    g = np.array(genders)
    total_mask, bpc = ref_nt.get_mask(samples)                       # main.py:82-88
    total_mask = total_mask & ref_nt.get_mask(samples[g == "F"])[0] \
        & ref_nt.get_mask(samples[g == "M"])[0]
    out["total_mask_in"] = total_mask.copy()
    tmp = tempfile.mkdtemp(prefix="wcx_golden_prep_")
    for gender, sub in (("A", samples), ("F", samples[g == "F"]), ("M", samples[g == "M"])):
        args = argparse.Namespace(prepdatafile=os.path.join(tmp, "d.npy"),
                                  prepfile=os.path.join(tmp, "p.npz"), binsize=binsize)
        np.random.seed(3)
        before = total_mask.copy()
        ref_nc.tool_newref_prep(args, sub, gender, total_mask, bpc)
        p = np.load(args.prepfile)
        for key in ("mask", "bins_per_chr", "masked_bins_per_chr", "masked_bins_per_chr_cum",
                    "pca_mean"):
            out["{}_{}".format(gender, key)] = p[key]
        out[gender + "_removed"] = np.where(before & ~total_mask)[0]
        print(gender, "filter removed bins", out[gender + "_removed"])
    n_aut = int(np.sum(bpc[:22]))
    assert len(out["A_removed"]) > 0 and len(out["F_removed"]) > 0
    assert out["F_removed"].max() < n_aut, "the F pass must drop an AUTOSOMAL bin"
    assert out["A_mask"][:n_aut].sum() != out["F_mask"][:n_aut].sum()
    save("prep_filter.npz", **out)


def golden_mask_skew():
    """The prep_filter cohort (the F / M passes' PCA-distance filter drops an AUTOSOMAL bin after the
    A pass saved its mask) through the reference's WHOLE newref, then its predict merge on one
    sample: what does upstream do with the skewed reference it has just built?  Recorded: the three
    masks, and the outcome of main.py:232-275 -- results_r/z/w post-process (every bin after the
    dropped one shifted), results_nr RAISES IndexError in get_post_processed_result
    (predict_control.py:50: the merged null-ratio table has sum(mask.F) rows, ref_sizes one entry per
    merged result).  Upstream cannot predict on such a reference."""
    binsize = 4000000
    co = Cohort(binsize, struct_seed=21, female_y=0.1)
    samples, genders = co.cohort(28, seed0=700, reads=4e6)
    rng = np.random.default_rng(9)
    sd = 1.6
    spec = [(2, 10, "FM"), (5, 7, "F"), (6, 7, "F"), (7, 9, "F"), (8, 3, "F"), (9, 3, "M"),
            (10, 3, "M"), (11, 5, "M"), (12, 5, "M"), (3, 5, "F"), (4, 5, "F"), (13, 5, "F"),
            (24, 4, "M"), (23, 11, "F")]
    for c, b, who in spec:          # (same perturbation as golden_prep_filter)
        for i, s in enumerate(samples):
            if genders[i] in who:
                s[str(c)][b] = int(s[str(c)][b] * np.exp(rng.normal(0, sd)))
    tmp = tempfile.mkdtemp(prefix="wcx_golden_skew_")
    infiles = []
    for i, s in enumerate(samples):
        p = os.path.join(tmp, "s{}.npz".format(i))
        write_sample(p, s, binsize)
        infiles.append(p)
    args = argparse.Namespace(infiles=infiles, outfile=os.path.join(tmp, "ref.npz"), nipt=False,
                              yfrac=0.004, plotyfrac=None, refsize=40, binsize=binsize, cpus=1)
    np.random.seed(5)
    random.seed(5)
    try:
        ref_main.tool_newref(args)
    except NameError as e:          # main.py:135 qc_reference never imported
        print("expected reference bug:", e)
    ref = np.load(args.outfile, encoding="latin1", allow_pickle=True)
    n_aut = int(np.sum(ref["bins_per_chr"]))
    out = {"cohort_counts": np.stack([np.concatenate([s[str(c)] for c in range(1, 25)])
                                      for s in samples]),
           "cohort_genders": np.array(genders), "cohort_bpc": np.array(co.bpc),
           "mask": ref["mask"], "mask_F": ref["mask.F"], "mask_M": ref["mask.M"]}
    assert int(ref["mask"][:n_aut].sum()) > int(ref["mask.F"][:n_aut].sum())
    pargs = argparse.Namespace(maskrepeats=5, minrefbins=10)
    sample0 = co.sample(9002, "F", reads=4e6, cnv=[(7, 5, 15, 0.5)])
    out["test_counts"] = np.concatenate([sample0[str(c)] for c in range(1, 25)])
    gender = ref_pt.predict_gender(sample0, ref["trained_cutoff"])
    sample = ref_ot.gender_correct({k: v.copy() for k, v in sample0.items()}, gender)
    rA = ref_pc.normalize(pargs, sample, ref, "A")
    rG = ref_pc.normalize(pargs, sample, ref, gender)
    nr_aut = ref["null_ratios"]
    nr_gon = ref["null_ratios.{}".format(gender)][len(nr_aut):]
    rem_input = {"args": pargs, "binsize": int(ref["binsize"]), "ref_gender": gender,
                 "mask": ref["mask.{}".format(gender)],
                 "bins_per_chr": ref["bins_per_chr.{}".format(gender)]}
    ref_sizes = np.append(rA[3], rG[3])
    null_ratios = np.array([x.tolist() for x in nr_aut] + [x.tolist() for x in nr_gon], dtype=object)
    outcome = {}
    for nm, v in (("results_r", np.append(rA[0], rG[0])), ("results_nr", null_ratios)):
        try:
            ref_pc.get_post_processed_result(pargs, v, ref_sizes, rem_input)
            outcome[nm] = "ok"
        except Exception as e:      # noqa: BLE001
            outcome[nm] = type(e).__name__
    print("upstream predict on its own skewed reference:", outcome)
    assert outcome == {"results_r": "ok", "results_nr": "IndexError"}
    out["predict_gender"] = np.array(gender)
    out["predict_results_r"] = np.array(outcome["results_r"])
    out["predict_results_nr"] = np.array(outcome["results_nr"])
    out["len_merged"] = np.array([len(ref_sizes), len(null_ratios), int(np.sum(rem_input["mask"]))])
    save("mask_skew.npz", **out)


# --------------------------------------------------------------------------- f3 output tables
def golden_tables():
    """Byte pins of the `predict --bed --regions` tables: the reference's own tool_test run end to
    end on the tiny cohort with only the R call stubbed (exec_R -> hand-made segments: DNAcopy is
    not available).  Captured: the arguments of generate_output_tables (what the formatting code
    sees) and the files it wrote."""
    import wisecondorx.predict_output as ref_po
    binsize = 4000000
    co = Cohort(binsize, struct_seed=11, female_y=0.1)
    samples, genders = co.cohort(24, seed0=500, reads=4e6)
    tmp = tempfile.mkdtemp(prefix="wcx_golden_tab_")
    infiles = []
    for i, smp in enumerate(samples):
        path = os.path.join(tmp, "s{}.npz".format(i))
        write_sample(path, smp, binsize)
        infiles.append(path)
    args = argparse.Namespace(infiles=infiles, outfile=os.path.join(tmp, "ref.npz"), nipt=False,
                              yfrac=0.004, plotyfrac=None, refsize=60, binsize=binsize, cpus=1)
    np.random.seed(5)
    random.seed(5)
    try:
        ref_main.tool_newref(args)
    except NameError as e:
        print("expected reference bug:", e)
    test = co.sample(9001, "M", reads=4e6, cnv=[(3, 10, 25, 1.5)])
    tpath = os.path.join(tmp, "t.npz")
    write_sample(tpath, test, binsize)
    regions = os.path.join(tmp, "regions.bed")
    with open(regions, "w") as fh:
        fh.write("chr3\t40000001\t100000000\tplanted\n2\t1\t12000000\tquiet\nchr1\t999000000\t999900000\tbeyond\n")

    def stub_exec_R(json_dict):          # stands in for Rscript CBS.R (DNAcopy absent)
        out = []
        for c, (rr, ww) in enumerate(zip(json_dict["results_r"], json_dict["results_w"])):
            n = len(rr)
            cuts = [0, 10, 25, n] if c == 2 else ([0, n // 2, n] if c % 3 == 0 else [0, n])
            for a, b in zip(cuts[:-1], cuts[1:]):
                r = np.asarray(rr[a:b], dtype=float)
                w = np.asarray(ww[a:b], dtype=float)
                keep = r != 0
                if b - a < 1 or not keep.any():
                    continue
                out.append({"chr": c + 1, "s": a, "e": b,
                            "r": float(np.sum(r[keep] * w[keep]) / np.sum(w[keep]))})
        return out
    ref_pt.exec_R = stub_exec_R
    captured = {}
    real_tables = ref_po.generate_output_tables

    def spy_tables(rem_input, results):
        captured["rem"], captured["res"] = rem_input, results
        real_tables(rem_input, results)
    ref_main.generate_output_tables = spy_tables
    outid = os.path.join(tmp, "ID")
    targs = argparse.Namespace(infile=tpath, reference=args.outfile, outid=outid, minrefbins=20,
                               maskrepeats=5, alpha=1e-4, zscore=4.0, beta=None, blacklist=None,
                               gender=None, ylim="def", bed=True, plot=False, cairo=False,
                               add_plot_title=False, seed=3, regions=regions)
    ref_main.tool_test(targs)
    ref_main.generate_output_tables = real_tables
    rem, res = captured["rem"], captured["res"]
    out = {"binsize": np.array(rem["binsize"]), "n_reads": np.array(rem["n_reads"]),
           "ref_gender": np.array(rem["ref_gender"]), "gender": np.array(rem["gender"]),
           "bins_per_chr": np.asarray(rem["bins_per_chr"]), "zscore": np.array(4.0),
           "regions_text": np.array(open(regions).read())}
    nchr = len(res["results_r"])
    for key in ("results_r", "results_z", "results_w"):
        out[key] = np.concatenate([np.asarray(c, dtype=float) for c in res[key]])
    m = max(len(row) for c in res["results_nr"] for row in c if np.ndim(row) > 0)
    nr = np.full((len(out["results_r"]), m), np.nan)
    i = 0
    for c in res["results_nr"]:
        for row in c:
            if np.ndim(row) > 0:
                nr[i, :len(row)] = row
            else:
                nr[i, :] = 0.0                 # the int 0 placeholder of masked bins
            i += 1
    out["results_nr"] = nr
    out["results_c_num"] = np.array([[s[0], s[1], s[2], np.nan if isinstance(s[3], str) else s[3], s[4]]
                                     for s in res["results_c"]], dtype=float)
    out["results_c_isstr"] = np.array([isinstance(s[3], str) for s in res["results_c"]])
    assert nchr == 24
    for suffix in ("_bins.bed", "_segments.bed", "_aberrations.bed", "_statistics.txt", "_regions.bed"):
        out["file" + suffix.replace(".", "_")] = np.array(open(outid + suffix).read())
    print(open(outid + "_aberrations.bed").read())
    print(open(outid + "_statistics.txt").read()[-400:])
    save("tables.npz", **out)


# --------------------------------------------------------------------------- BASELINE config 1
def golden_config1():
    """BASELINE.json configs[0]: predict one synthetic sample at 1 Mb bins (~3.1 k bins) against
    a 50-sample reference, refsize 300 -- the reference's own CPU path end to end
    (tool_newref, then predict_control.normalize for the autosomes and the gonosomes).

    To keep the fixture small, the 2 x (B x 300) indexes/distances are stored as SHA-256 digests
    next to the reference's PCA-corrected matrices (the bytes np.save wrote, newref_control.py:68):
    a parity test first rebuilds the neighbour tables from those matrices, proves them identical
    to the reference's through the digests, then runs predict on them and compares with the
    reference's outputs."""
    import hashlib
    import wisecondorx.newref_control as ref_nc
    binsize = 1000000
    co = Cohort(binsize, struct_seed=41, female_y=0.1)
    samples, genders = co.cohort(50, seed0=4100, reads=1e7)
    tmp = tempfile.mkdtemp(prefix="wcx_golden_c1_")
    infiles = []
    for i, s in enumerate(samples):
        p = os.path.join(tmp, "s{}.npz".format(i))
        write_sample(p, s, binsize)
        infiles.append(p)
    # capture the PCA-corrected matrix of every pass as the reference saves it
    captured = {}
    real_prep = ref_nc.tool_newref_prep

    def spy_prep(args, samples_, gender, mask, bins_per_chr):
        real_prep(args, samples_, gender, mask, bins_per_chr)
        captured[gender] = np.load(args.prepdatafile)
    ref_main.tool_newref_prep = spy_prep
    args = argparse.Namespace(infiles=infiles, outfile=os.path.join(tmp, "ref.npz"),
                              nipt=False, yfrac=0.004, plotyfrac=None, refsize=300,
                              binsize=binsize, cpus=1)
    np.random.seed(41)
    random.seed(41)
    try:
        ref_main.tool_newref(args)
    except NameError as e:          # main.py:135 qc_reference never imported
        print("expected reference bug:", e)
    finally:
        ref_main.tool_newref_prep = real_prep
    ref = np.load(args.outfile, encoding="latin1", allow_pickle=True)
    test = co.sample(4999, "M", reads=1e7, cnv=[(4, 40, 70, 1.5), (23, 20, 60, 0.5)])
    gender = ref_pt.predict_gender(test, ref["trained_cutoff"])
    assert gender == "M"
    sample = ref_ot.gender_correct({k: v.copy() for k, v in test.items()}, gender)
    pargs = argparse.Namespace(maskrepeats=5, minrefbins=150)
    out = {"test_counts": np.concatenate([test[str(c)] for c in range(1, 25)]),
           "test_gender": np.array(gender), "cohort_bpc": np.array(co.bpc)}
    sha = lambda a: np.array(hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest())
    for tag, ap, rg in (("A", "", "A"), ("G", "." + gender, gender)):
        X = captured[rg]
        assert X.flags["F_CONTIGUOUS"] and X.shape[0] == ref["indexes" + ap].shape[0]
        out[tag + "_Xs"] = np.ascontiguousarray(X.T)          # the bytes of the F-ordered (B,S)
        for key in ("mask", "bins_per_chr", "masked_bins_per_chr", "masked_bins_per_chr_cum",
                    "pca_components", "pca_mean"):
            out["ref__" + key + ap] = ref[key + ap]
        idx, dist = ref["indexes" + ap], ref["distances" + ap]
        out[tag + "_idx_sha"], out[tag + "_dist_sha"] = sha(idx), sha(dist)
        out[tag + "_idx_rows"] = idx[::97]                     # a few rows in clear, for debugging
        out[tag + "_dist_rows"] = dist[::97]
        res = ref_pc.normalize(pargs, sample, ref, rg)
        for nm, v in zip(("r", "z", "w", "n", "mlr", "mz"), res):
            out["{}_{}".format(tag, nm)] = np.asarray(v)
    out["ref__binsize"] = ref["binsize"]
    out["ref__trained_cutoff"] = ref["trained_cutoff"]
    out["cutoff"] = np.array(ref_pt.get_optimal_cutoff(ref, 5))
    save("config1.npz", **out)


def golden_gender():
    """f4: the reference's own gender model (newref_tools.train_gender_model, :21-68) on a cohort whose
    Y-read fractions have a real bimodal valley, WITHOUT --yfrac: the Gaussian-mixture fit and the
    argrelextrema cut-off search (:57-62) run.  np.random is seeded for the mixture's k-means
    initialisation."""
    import argparse
    from wisecondorx import newref_tools as ref_nt
    rng = np.random.default_rng(21)
    n = 46
    is_m = rng.random(n) < 0.5
    total = rng.integers(8_000_000, 25_000_000, n)
    yfrac = np.where(is_m, rng.normal(0.0042, 0.0006, n), rng.normal(0.00035, 0.00012, n)).clip(1e-5)
    y = np.round(total * yfrac).astype(np.int64)
    other = (total - y).astype(np.int64)
    samples = [{"1": np.array([o // 2, o - o // 2]), "24": np.array([yy])} for o, yy in zip(other, y)]
    args = argparse.Namespace(plotyfrac=None, yfrac=None)
    np.random.seed(7)
    genders, cut = ref_nt.train_gender_model(args, samples)
    assert set(genders) == {"M", "F"} and 0.0005 < cut < 0.004
    save("gender.npz", other=other, y=y, genders=np.array(genders), cut_off=np.array(cut), np_seed=7,
         is_m=is_m)


def golden_cbs_bdry():
    """NOT a reference output (DNAcopy is not in the reference repository): the sequential-boundary
    table of the CBS ORACLE (oracle/cbs_oracle.getbdry, scipy hypergeometric CDF, ~50 s) for
    alpha = 0.01, stored so that tests need not re-derive it.  tests/test_oracle_cbs.py re-derives
    its first blocks and checks the product's own derivation (wcx_cbs_getbdry) against all of it."""
    from oracle import cbs_oracle as CO
    tab = np.array(CO.getbdry(0.05, 10000, 101), dtype=np.int32)
    save("cbs_bdry.npz", eta=0.05, nperm=10000, max_ones=101, table=tab)


def golden_example_bed():
    """The one DNAcopy output the reference repository ships: docs/include/example.bed (a 100 kb NIPT
    trisomy-21 case) -- ID_bins.bed holds the per-bin log2 ratios CBS.R was given (NaN = blacklisted,
    i.e. ratio 0 on input), ID_segments.bed what DNAcopy + CBS.R's NA-gap split made of them.  Data
    only: ratios per chromosome (1..22, X), and the segments as (chr0, first bin, end bin exclusive,
    ratio to 4 decimals, z).  The weights of that run (1 / mean sqrt reference distance) are not
    shipped, so tests run the segmentation with unit weights and compare the BOUNDARIES."""
    d = "/root/reference/docs/include/example.bed"
    names = [str(i) for i in range(1, 23)] + ["X"]
    ratios = {n: [] for n in names}
    binsize = None
    for line in open(os.path.join(d, "ID_bins.bed")).read().splitlines()[1:]:
        c, a, b, _, r, _ = line.split("\t")
        binsize = binsize or int(b) - int(a) + 1
        assert (int(a) - 1) // binsize == len(ratios[c])
        ratios[c].append(float("nan") if r == "NaN" else float(r))
    segs = []
    for line in open(os.path.join(d, "ID_segments.bed")).read().splitlines()[1:]:
        c, a, b, r, z = line.split("\t")
        segs.append([names.index(c), (int(a) - 1) // binsize, int(b) // binsize, float(r), float(z)])
    flat = np.concatenate([np.array(ratios[n]) for n in names])
    save("example_bed.npz", binsize=binsize, bins_per_chr=np.array([len(ratios[n]) for n in names]),
         ratios=flat, segments=np.array(segs))


if __name__ == "__main__":
    if "example" in sys.argv[1:]:
        golden_example_bed()
        sys.exit(0)
    if "bdry" in sys.argv[1:]:
        golden_cbs_bdry()
    if "gender" in sys.argv[1:]:
        golden_gender()
    which = sys.argv[1:] or ["search", "pipeline", "prep_filter", "config1", "tables"]
    if "tables" in which:
        golden_tables()
    if "config1" in which:
        golden_config1()
    if "search" in which:
        golden_newref_search()
    if "pipeline" in which:
        golden_pipeline()
    if "prep_filter" in which:
        golden_prep_filter()
    if "mask_skew" in which:
        golden_mask_skew()
