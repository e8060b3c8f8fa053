"""Output tables of `predict --bed` in the reference's formats (predict_output.py:51-263):
ID_bins.bed, ID_segments.bed, ID_aberrations.bed, ID_statistics.txt, ID_regions.bed.
Plain text formatting on the host; the only numerical part (per-chromosome z-scores) goes
through the same segment-z kernel as the CBS segments."""
import re

import numpy as np

from .overall_tools import get_cpa, get_median_segment_variance
from .predict_tools import get_z_score


def _chr_name(c):
    name = str(c + 1)
    return {"23": "X", "24": "Y"}.get(name, name)


def generate_output_tables(rem_input, results, ctx=None):
    """ctx: the library context that holds the attached null matrix (default: the default context)."""
    _generate_bins_bed(rem_input, results)
    _generate_segments_and_aberrations_bed(rem_input, results)
    _generate_chr_statistics_file(rem_input, results, ctx)
    if rem_input["args"].regions is not None:
        _generate_regions_bed(rem_input, results)


def _fmt(v):
    """str() of a Python float, like the reference's str(x) on list elements."""
    return "nan" if v == 0 else str(float(v))


def _fmt_all(values):
    """_fmt of every element (repr of a Python float is its str)."""
    return ["nan" if v == 0 else repr(v) for v in np.asarray(values, dtype=np.float64).tolist()]


def _generate_bins_bed(rem_input, results):
    """ID_bins.bed (predict_output.py:51-75 of the reference): 206 k rows at 15 kb.  The rows of a
    chromosome are laid out natively (wcx_format_bins_bed: floats printed as Python's str(float), 0 as
    "nan") -- formatting them here cost 0.15 s of a 0.5 s predict."""
    from . import _lib
    lib = _lib.load()
    binsize = int(rem_input["binsize"])
    with open("{}_bins.bed".format(rem_input["args"].outid), "wb") as fh:
        fh.write(b"chr\tstart\tend\tid\tratio\tzscore\n")
        for c in range(len(results["results_r"])):
            name = _chr_name(c).encode()
            rs = np.ascontiguousarray(results["results_r"][c], dtype=np.float64)
            zs = np.ascontiguousarray(results["results_z"][c], dtype=np.float64)
            n = len(rs)
            buf = np.empty(n * (2 * len(name) + 140) + 16, dtype=np.uint8)
            m = lib.wcx_format_bins_bed(name, n, binsize, _lib.ptr(rs), _lib.ptr(zs), _lib.ptr(buf), buf.size)
            if m < 0:
                raise _lib.WcxError("wcx_format_bins_bed: bad arguments for chromosome {}".format(c + 1))
            fh.write(memoryview(buf)[:m])


def _aberration_cutoff(beta, ploidy):
    return np.log2((ploidy - (beta / 2)) / ploidy), np.log2((ploidy + (beta / 2)) / ploidy)


def _generate_segments_and_aberrations_bed(rem_input, results):
    args = rem_input["args"]
    with open("{}_segments.bed".format(args.outid), "w") as seg_f, \
            open("{}_aberrations.bed".format(args.outid), "w") as abr_f:
        seg_f.write("chr\tstart\tend\tratio\tzscore\n")
        abr_f.write("chr\tstart\tend\tratio\tzscore\ttype\n")
        for segment in results["results_c"]:
            name = _chr_name(segment[0])
            row = [name, int(segment[1] * rem_input["binsize"] + 1),
                   int(segment[2] * rem_input["binsize"]), segment[4], segment[3]]
            line = "\t".join(str(x) for x in row)
            seg_f.write(line + "\n")
            ploidy = 1 if (name in ("X", "Y") and rem_input["ref_gender"] == "M") else 2
            if args.beta is not None:
                lo, hi = _aberration_cutoff(args.beta, ploidy)
                if float(segment[4]) > hi:
                    abr_f.write(line + "\tgain\n")
                elif float(segment[4]) < lo:
                    abr_f.write(line + "\tloss\n")
            elif isinstance(segment[3], str):
                continue
            elif float(segment[3]) > args.zscore:
                abr_f.write(line + "\tgain\n")
            elif float(segment[3]) < -args.zscore:
                abr_f.write(line + "\tloss\n")


def _generate_chr_statistics_file(rem_input, results, ctx=None):
    n_chr = len(results["results_r"])
    with np.errstate(all="ignore"):
        means = [np.ma.average(np.asarray(results["results_r"][c], dtype=float),
                               weights=np.asarray(results["results_w"][c], dtype=float))
                 if np.sum(results["results_w"][c]) != 0 else float("nan") for c in range(n_chr)]
        medians = []
        for c in range(n_chr):
            v = np.asarray(results["results_r"][c], dtype=float)
            v = v[v != 0]
            medians.append(np.median(v) if v.size else float("nan"))
    results_c_chr = [[c, 0, rem_input["bins_per_chr"][c] - 1, means[c]] for c in range(n_chr)]
    msv = round(float(get_median_segment_variance(results["results_c"], results["results_r"])), 5)
    cpa = round(float(get_cpa(results["results_c"], rem_input["binsize"])), 5)
    chr_z = get_z_score(results_c_chr, results, ctx)
    with open("{}_statistics.txt".format(rem_input["args"].outid), "w") as fh:
        fh.write("chr\tratio.mean\tratio.median\tzscore\n")
        for c in range(n_chr):
            fh.write("\t".join(str(x) for x in [_chr_name(c), means[c], medians[c], chr_z[c]]) + "\n")
        fh.write("Gender based on --yfrac (or manually overridden by --gender): {}\n".format(
            rem_input["gender"]))
        fh.write("Number of reads: {}\n".format(rem_input["n_reads"]))
        fh.write("Standard deviation of the ratios per chromosome: {}\n".format(
            round(float(np.nanstd(np.asarray(means, dtype=float))), 5)))
        fh.write("Median segment variance per bin (doi: 10.1093/nar/gky1263): {}\n".format(msv))
        fh.write("Copy number profile abnormality (CPA) score (doi: 10.1186/s13073-020-00735-4): "
                 "{}\n".format(cpa))


def _generate_regions_bed(rem_input, results):
    binsize = rem_input["binsize"]
    with open("{}_regions.bed".format(rem_input["args"].outid), "w") as out, \
            open(rem_input["args"].regions) as fh:
        out.write("chr\tstart\tend\tname\tratio\tzscore\n")
        for line in fh:
            if not line.strip():
                continue
            region = line.strip().split("\t")
            assert len(region) >= 4, "Regions file must have at least 4 columns: chr, start, end, name"
            chr_name, start, end, name = region[:4]
            core = re.sub("chr", "", chr_name)
            c = {"X": 22, "Y": 23}.get(core)
            if c is None:
                c = int(core) - 1
            if c >= len(results["results_r"]):
                out.write("Skipping invalid region: {}\n".format("\t".join(region)))
                continue
            s_bin = int(start) // binsize
            e_bin = min(int(end) // binsize, rem_input["bins_per_chr"][c] - 1)
            if s_bin < 0 or e_bin < 0 or s_bin > e_bin:
                out.write("Skipping invalid region: {}\n".format("\t".join(region)))
                continue
            rr = np.asarray(results["results_r"][c][s_bin:e_bin + 1], dtype=float)
            ww = np.asarray(results["results_w"][c][s_bin:e_bin + 1], dtype=float)
            zz = np.asarray(results["results_z"][c][s_bin:e_bin + 1], dtype=float)
            if rr.size == 0:
                out.write("Skipping region with no bins: {}\n".format("\t".join(region)))
                continue
            with np.errstate(all="ignore"):
                rm = np.ma.average(rr, weights=ww) if ww.sum() != 0 else float("nan")
                zm = np.ma.average(zz, weights=ww) if ww.sum() != 0 else float("nan")
            out.write("\t".join(str(x) for x in [chr_name, start, end, name,
                                                 "nan" if rm == 0 else rm,
                                                 "nan" if zm == 0 else zm]) + "\n")
