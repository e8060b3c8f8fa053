"""Host helpers shared by newref and predict (mirror of the reference's overall_tools.py;
O(bins) NumPy glue -- the per-bin Python loops of the reference are vectorised)."""
import logging
import sys

import numpy as np


def scale_sample(sample, from_size, to_size):
    """Sum consecutive bins to go from `from_size` to `to_size` bp bins
    (overall_tools.py:19-40; same validation, same int32 result, reshape-sum instead of a
    Python loop per bin)."""
    if not to_size or from_size == to_size:
        return sample
    if to_size == 0 or from_size == 0 or to_size < from_size or to_size % from_size > 0:
        logging.critical("Impossible binsize scaling requested: {} to {}".format(
            int(from_size), int(to_size)))
        sys.exit()
    scale = int(to_size // from_size)
    out = dict()
    for chr_name in sample:
        chr_data = sample[chr_name]
        if chr_data is None:
            out[chr_name] = None
            continue
        chr_data = np.asarray(chr_data)
        new_len = int(np.ceil(len(chr_data) / float(scale)))
        padded = np.zeros(new_len * scale, dtype=np.int64)
        padded[:len(chr_data)] = chr_data
        out[chr_name] = padded.reshape(new_len, scale).sum(axis=1).astype(np.int32)
    return out


def gender_correct(sample, gender):
    """Double the gonosomal counts of males (overall_tools.py:48-53)."""
    if gender == "M":
        sample["23"] = sample["23"] * 2
        sample["24"] = sample["24"] * 2
    return sample


def get_median_segment_variance(results_c, results_r):
    """MSV: median over segments of the variance of their non-zero bin ratios
    (overall_tools.py:127-135)."""
    variances = []
    for segment in results_c:
        seg = np.asarray(results_r[segment[0]][int(segment[1]):int(segment[2])], dtype=float)
        seg = seg[seg != 0]
        if seg.size:
            variances.append(np.var(seg))
    return np.median(variances)


def get_cpa(results_c, binsize):
    """Copy-number profile abnormality score (overall_tools.py:143-148)."""
    x = 0
    for segment in results_c:
        z = segment[3]          # rows are [chr, start, end, z, ratio] (predict_tools.py:259-263)
        z = float("nan") if isinstance(z, str) else abs(z)
        x += (segment[2] - segment[1] + 1) * binsize * z
    return x / len(results_c) * (10 ** -8)


def predict_gender(sample, trained_cutoff):
    """Y-read fraction against the cut-off trained by newref (predict_tools.py:17-24)."""
    total = float(np.sum([np.sum(sample[x]) for x in sample.keys()]))
    return "M" if float(np.sum(sample["24"])) / total > trained_cutoff else "F"
