"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A NumPy restatement of the WisecondorX newref/predict hot path, each
function citing the reference file:line it follows (paths relative to
/root/reference/src/wisecondorx/).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package (wisecondorx_amd/) never does.

Pinning status:
  * a4-a7, a9-a15, a17 (newref search, null ratios, predict normalisation, post-processing,
    segment z): PINNED -- asserted against the reference itself, imported from
    /root/reference in the build container, on the fixtures in tests/golden/ (generator:
    tests/golden/make_golden.py; checks: tests/test_oracle_golden.py).
  * a16 CBS breakpoints (DNAcopy::segment, Bioconductor DNAcopy 1.76.0 -- conda.yml:14 --
    not vendored, R absent): pinned on the one DNAcopy run the reference ships
    (docs/include/example.bed, 50 segments reproduced bin for bin), otherwise PARITY UNPINNED
    against DNAcopy.  The segmentation algorithm
    is restated in oracle/cbs_oracle.py (cbs_segment() there plugs into cbs_r_wrapper()
    below as its segment_fn); the reference-owned code around the DNAcopy call
    (CBS.R:41-63,84-129) is restated exactly in cbs_r_wrapper().
"""
import math
import random

import numpy as np

Z_MASK = 2.3263478740408408  # scipy.stats.norm.ppf(0.99), predict_tools.py:104


# --------------------------------------------------------------------------- newref (a4-a7)

def get_part(partnum, outof, bincount):
    """newref_tools.py:244-247."""
    start_bin = int(bincount / float(outof) * partnum)
    end_bin = int(bincount / float(outof) * (partnum + 1))
    return start_bin, end_bin


def split_by_chr(start, end, chr_bin_sums):
    """newref_tools.py:227-241 (same outputs; regions [chr_idx, start, end])."""
    areas = []
    tmp = [0, start, 0]
    for i, val in enumerate(chr_bin_sums):
        tmp[0] = i
        if val >= end:
            break
        if start < val < end:
            tmp[2] = val
            areas.append(tmp)
            tmp = [i, val, 0]
        tmp[1] = val
    tmp[2] = end
    areas.append(tmp)
    return areas


def sq_distances(chr_data, x_row):
    """newref_tools.py:260.  With Fortran-ordered inputs (train_pca returns corrected.T,
    newref_tools.py:147) NumPy reduces axis 1 column by column: a strictly sequential
    left-to-right fp64 sum of separately rounded (c-t), (.)^2.  Restated as that loop so
    the value does not depend on array layout."""
    n, s = chr_data.shape
    acc = np.zeros(n, dtype=np.float64)
    for j in range(s):
        diff = chr_data[:, j] - x_row[j]
        acc = acc + diff * diff
    return acc


def topk_stable(d, ref_size):
    """newref_tools.py:261-275: bisect_right insertion with strict `binVal < cur_max`
    admission, initial sentinels 1e10/-1  ==  first ref_size of a stable sort by
    (distance, candidate index) restricted to d < 1e10 (NaN never admitted)."""
    ok = np.flatnonzero(d < 1e10)
    order = ok[np.argsort(d[ok], kind="stable")][:ref_size]
    idx = np.full(ref_size, -1, dtype=np.int32)
    dist = np.full(ref_size, 1e10, dtype=np.float64)
    idx[:len(order)] = order
    dist[:len(order)] = d[order]
    return idx, dist


def topk_scan(d, ref_size):
    """newref_tools.py:261-275 as the reference executes it: one pass over ALL candidates in
    Python, keeping an ascending list of the ref_size best (a candidate enters only if strictly
    below the current worst; it goes after equal values, the worst drops out).  Same result as
    topk_stable(); kept because THIS loop is most of the reference's run time, so
    bench.py's cpu_baseline times it."""
    from bisect import bisect_right
    best_d = [1e10] * ref_size
    best_i = [-1] * ref_size
    worst = 1e10
    for cand, val in enumerate(d):
        if val < worst:
            at = bisect_right(best_d, val)
            best_d[at:at] = [val]
            best_i[at:at] = [cand]
            del best_d[-1], best_i[-1]
            worst = best_d[-1]
    return np.array(best_i, dtype=np.int32), np.array(best_d, dtype=np.float64)


def get_ref_for_bins(ref_size, start, end, X, chr_data):
    """newref_tools.py:255-278."""
    ref_indexes = np.zeros((end - start, ref_size), dtype=np.int32)
    ref_distances = np.ones((end - start, ref_size))
    for this_bin in range(start, end):
        d = sq_distances(chr_data, X[this_bin, :])
        i, v = topk_stable(d, ref_size)
        ref_indexes[this_bin - start, :] = i
        ref_distances[this_bin - start, :] = v
    return ref_indexes, ref_distances


def null_ratios(X, index_array, start_num, end_num, sample_ids):
    """newref_tools.py:210-223.  NOTE the reference quirk: index rows (chromosome-excluded
    index space) are applied to the FULL sample vector without re-offsetting (:219-221)."""
    out = np.zeros((end_num - start_num, len(sample_ids)))
    samples = np.transpose(X)
    for null_i, case_i in enumerate(sample_ids):
        sample = samples[case_i]
        for bin_i in range(start_num, end_num):
            ref = sample[index_array[bin_i - start_num]]
            with np.errstate(all="ignore"):
                out[bin_i - start_num][null_i] = np.log2(sample[bin_i] / np.median(ref))
    return out


def pick_null_samples(n_samples, rng=random):
    """newref_tools.py:214-217: random.sample(range(S), min(S, 100))."""
    return rng.sample(range(n_samples), min(n_samples, 100))


def get_reference(X, masked_bins_per_chr, masked_bins_per_chr_cum, ref_size, part,
                  split_parts, sample_ids=None):
    """newref_tools.py:155-224."""
    big_indexes, big_distances = [], []
    bincount = masked_bins_per_chr_cum[-1]
    start_num, end_num = get_part(part - 1, split_parts, bincount)
    regions = split_by_chr(start_num, end_num, masked_bins_per_chr_cum)
    for region in regions:
        chrom, start, end = region
        if start_num > start:
            start = start_num
        if end_num < end:
            end = end_num
        if len(masked_bins_per_chr_cum) > 22 and chrom != 22 and chrom != 23:
            big_indexes.extend(np.zeros((end - start, ref_size), dtype=np.int32))
            big_distances.extend(np.ones((end - start, ref_size)))
            continue
        lo = masked_bins_per_chr_cum[chrom] - masked_bins_per_chr[chrom]
        hi = masked_bins_per_chr_cum[chrom]
        chr_data = np.concatenate((X[:lo, :], X[hi:, :]))
        pi, pd_ = get_ref_for_bins(ref_size, start, end, X, chr_data)
        big_indexes.extend(pi)
        big_distances.extend(pd_)
    index_array = np.array(big_indexes)
    distance_array = np.array(big_distances)
    if sample_ids is None:
        sample_ids = pick_null_samples(X.shape[1])
    nr = null_ratios(X, index_array, start_num, end_num, sample_ids)
    return index_array, distance_array, nr


# --------------------------------------------------------------------------- predict (a9-a13)

def coverage_normalize_and_mask(sample, bins_per_chr, mask):
    """predict_tools.py:32-48."""
    by_chr = []
    for c in range(1, len(bins_per_chr) + 1):
        this_chr = np.zeros(bins_per_chr[c - 1], dtype=float)
        min_len = min(bins_per_chr[c - 1], len(sample[str(c)]))
        this_chr[:min_len] = sample[str(c)][:min_len]
        by_chr.append(this_chr)
    all_data = np.concatenate(by_chr, axis=0)
    all_data = all_data / np.sum(all_data)
    return all_data[mask]


def project_pc(sample_data, components, mean):
    """predict_tools.py:56-65 with scikit-learn<=1.4.2 PCA.transform semantics
    (setup.cfg:42): t = (x - mean) . C^T ; rec = t . C + mean ; x / rec."""
    t = np.dot(np.array([sample_data]) - mean, components.T)
    reconstructed = (np.dot(t, components) + mean)[0]
    return sample_data / reconstructed


def get_optimal_cutoff(distances, repeats):
    """predict_tools.py:74-82 (always on the autosomal `distances`)."""
    cutoff = float("inf")
    for _ in range(repeats):
        mask = distances < cutoff
        average = np.average(distances[mask])
        stddev = np.std(distances[mask])
        cutoff = average + 3 * stddev
    return cutoff


def get_weights(distances):
    """predict_tools.py:152-155."""
    inverse_weights = [np.mean(np.sqrt(x)) for x in distances]
    return np.array([1 / x for x in inverse_weights])


def normalize_once(test_data, test_copy, masked_bins_per_chr, masked_bins_per_chr_cum,
                   indexes, distances, optimal_cutoff, ct, cp, row_range=None):
    """predict_tools.py:111-142.  row_range=(lo,hi) restricts the work to those bins (used by
    the row-sharded multi-GPU tests; results of other bins stay 0)."""
    n = masked_bins_per_chr_cum[-1]
    results_z = np.zeros(n)[ct:]
    results_r = np.zeros(n)[ct:]
    ref_sizes = np.zeros(n)[ct:]
    i, i2 = ct, 0
    for c in list(range(len(masked_bins_per_chr)))[cp:]:
        start = masked_bins_per_chr_cum[c] - masked_bins_per_chr[c]
        end = masked_bins_per_chr_cum[c]
        chr_data = np.concatenate((test_copy[:start], test_copy[end:]))
        for index in indexes[start:end]:
            if row_range is not None and not (row_range[0] <= i < row_range[1]):
                i += 1
                i2 += 1
                continue
            ref_data = chr_data[index[distances[i] < optimal_cutoff]]
            ref_data = ref_data[ref_data >= 0]
            with np.errstate(all="ignore"):
                ref_stdev = np.std(ref_data)
                results_z[i2] = (test_data[i] - np.mean(ref_data)) / ref_stdev
                results_r[i2] = test_data[i] / np.median(ref_data)
            ref_sizes[i2] = ref_data.shape[0]
            i += 1
            i2 += 1
    return results_z, results_r, ref_sizes


def normalize_repeat(test_data, masked_bins_per_chr, masked_bins_per_chr_cum, indexes,
                     distances, optimal_cutoff, ct, cp):
    """predict_tools.py:94-108."""
    results_z = results_r = ref_sizes = None
    test_copy = np.copy(test_data)
    for _ in range(3):
        results_z, results_r, ref_sizes = normalize_once(
            test_data, test_copy, masked_bins_per_chr, masked_bins_per_chr_cum,
            indexes, distances, optimal_cutoff, ct, cp)
        with np.errstate(all="ignore"):
            test_copy[ct:][np.abs(results_z) >= Z_MASK] = -1
    with np.errstate(all="ignore"):
        m_lr = np.nanmedian(np.log2(results_r))
        m_z = np.nanmedian(results_z)
    return results_z, results_r, ref_sizes, m_lr, m_z


def normalize(sample, ref, ref_gender, maskrepeats=5):
    """predict_control.py:21-39.  `ref` is a dict / NpzFile with the reference keys."""
    if ref_gender == "A":
        ap, cp, ct = "", 0, 0
    else:
        ap = ".{}".format(ref_gender)
        cp = 22
        ct = ref["masked_bins_per_chr_cum" + ap][cp - 1]
    x = coverage_normalize_and_mask(sample, ref["bins_per_chr" + ap], ref["mask" + ap])
    x = project_pc(x, ref["pca_components" + ap], ref["pca_mean" + ap])
    results_w = get_weights(ref["distances" + ap])[ct:]
    cutoff = get_optimal_cutoff(ref["distances"], maskrepeats)
    z, r, n, m_lr, m_z = normalize_repeat(
        x, ref["masked_bins_per_chr" + ap], ref["masked_bins_per_chr_cum" + ap],
        ref["indexes" + ap], ref["distances" + ap], cutoff, ct, cp)
    return r, z, results_w, n, m_lr, m_z


# --------------------------------------------------------------------------- merge / post (a14-a15)

def merge_autosomes_gonosomes(rA, zA, wA, nA, m_z, rG, zG, wG, nG):
    """main.py:242-257."""
    with np.errstate(all="ignore"):
        r = np.append(rA, rG)
        z = np.append(zA, zG) - m_z
        w = np.append(wA * np.nanmean(wG), wG * np.nanmean(wA))
        w = w / np.nanmean(w)
    if np.isnan(w).any() or np.isinf(w).any():
        w = np.ones(len(w))
    n = np.append(nA, nG)
    return r, z, w, n


def inflate_results(results, mask):
    """predict_tools.py:163-170."""
    temp = [0 for _ in mask]
    j = 0
    for i, val in enumerate(mask):
        if val:
            temp[i] = results[j]
            j += 1
    return temp


def get_post_processed_result(minrefbins, result, ref_sizes, mask, bins_per_chr):
    """predict_control.py:49-63."""
    result = result.copy()
    infinite_mask = ref_sizes < minrefbins
    result[infinite_mask] = 0
    inflated = inflate_results(result, mask)
    final = []
    for c in range(len(bins_per_chr)):
        final.append(inflated[sum(bins_per_chr[:c]):sum(bins_per_chr[:c + 1])])
    return final


def log_trans(results, log_r_median):
    """predict_tools.py:180-193 (in place on the dict of per-chr lists)."""
    with np.errstate(all="ignore"):
        for c in range(len(results["results_r"])):
            results["results_r"][c] = np.log2(results["results_r"][c])
    results["results_r"] = [x.tolist() for x in results["results_r"]]
    for c in range(len(results["results_r"])):
        for i, rR in enumerate(results["results_r"][c]):
            if not np.isfinite(rR):
                results["results_r"][c][i] = 0
                results["results_z"][c][i] = 0
                results["results_w"][c][i] = 0
            if results["results_r"][c][i] != 0:
                results["results_r"][c][i] = results["results_r"][c][i] - log_r_median


def apply_blacklist(results, blacklist, binsize):
    """predict_tools.py:202-233.  blacklist = list of (chr_name, s, e) BED rows."""
    bed = {}
    for chr_name, s, e in blacklist:
        if chr_name[:3] == "chr":
            chr_name = chr_name[3:]
        if chr_name == "X":
            chr_name = "23"
        if chr_name == "Y":
            chr_name = "24"
        c = int(chr_name) - 1
        bed.setdefault(c, []).append([int(int(s) / binsize), int(int(e) / binsize) + 1])
    for c in bed:
        for s_e in bed[c]:
            for pos in range(s_e[0], s_e[1]):
                if len(results["results_r"]) < 24 and c == 23:
                    continue
                if pos >= len(results["results_r"][c]) or pos < 0:
                    continue
                results["results_r"][c][pos] = 0
                results["results_z"][c][pos] = 0
                results["results_w"][c][pos] = 0


# --------------------------------------------------------------------------- segment z (a17)

def get_z_score(results_c, results_nr, results_r, results_w):
    """overall_tools.py:88-119.  Per segment: weighted (w) average of every null-ratio
    column over the segment's bins with r != 0, ignoring non-finite entries; z of the
    segment ratio against mean/sd of those averages, clipped to +-1000; "nan" string when
    undefined."""
    zs = []
    for segment in results_c:
        c, s, e = segment[0], segment[1], segment[2]
        rr = results_r[c][s:e]
        keep = [i for i in range(len(rr)) if rr[i] != 0]
        nr = np.array([np.asarray(results_nr[c][s:e][i], dtype=float) for i in keep],
                      dtype=float)
        w = np.array([results_w[c][s:e][i] for i in keep], dtype=float)
        with np.errstate(all="ignore"):
            if nr.size == 0:
                null_segments = np.array([])
            else:
                nr = np.where(np.isfinite(nr), nr, np.nan)
                null_segments = []
                for col in nr.T:
                    # np.ma.average: (a*w).filled(0).sum() / w[~mask].filled(0).sum()
                    ok = ~np.isnan(col)
                    if not ok.any():
                        null_segments.append(np.nan)
                        continue
                    num = np.sum(np.where(ok, col * w, 0.0))
                    den = np.sum(np.where(ok, w, 0.0))
                    null_segments.append(num / den)
                null_segments = np.array(null_segments, dtype=float)
            fin = null_segments[np.isfinite(null_segments)] if null_segments.size else null_segments
            if fin.size == 0:
                null_mean = null_sd = float("nan")
            else:
                null_mean = float(np.mean(fin))
                null_sd = float(np.std(fin))
            if math.isnan(null_mean) or math.isnan(null_sd):
                zs.append("nan")
                continue
            if null_sd == 0:
                diff = segment[3] - null_mean
                z = float("nan") if diff == 0 else math.copysign(float("inf"), diff)
            else:
                z = (segment[3] - null_mean) / null_sd
        if math.isnan(z):
            # min()/max() with NaN first operand return NaN in the reference
            # (overall_tools.py:114-115: min(z,1000) -> z when z is NaN)
            zs.append(z)
            continue
        z = min(z, 1000)
        z = max(z, -1000)
        zs.append(z)
    return zs


# --------------------------------------------------------------------------- CBS wrapper (a16)

def cbs_r_wrapper(results_r, results_w, ref_gender, alpha, binsize, seed, segment_fn):
    """CBS.R:21-132 around the DNAcopy call.  segment_fn(y, w, alpha, seed) -> list of
    (start_1based, end_1based_inclusive) over ONE chromosome's non-NA-inclusive vector
    (DNAcopy drops NA rows itself and reports loc.start/loc.end in the x coordinate,
    here 1-based bin index, CBS.R:49,70-73)."""
    n_chr = 24 if ref_gender == "M" else 23
    out = []
    na_limit = int((binsize / 2000000.0) ** -1)  # CBS.R:95 as.integer(...)
    state = {"seed": seed}
    for c in range(n_chr):
        y = np.array(results_r[c], dtype=float)
        w = np.array(results_w[c], dtype=float)
        y[y == 0] = np.nan                # CBS.R:41
        w[w == 0] = 1.0                   # CBS.R:42 (1^-99 == 1)
        if np.all(np.isnan(y)):           # CBS.R:56-63
            continue
        segs = segment_fn(c, y, w, alpha, state)
        for (s1, e1) in segs:             # 1-based inclusive, CBS.R:84-113
            seg = y[s1 - 1:e1]
            isna = np.isnan(seg).astype(int)
            diff_na = np.diff(isna)
            start_pos = np.flatnonzero(diff_na == 1) + 1 + s1 - 1
            end_pos = np.flatnonzero(diff_na == -1) + 1 + s1 - 1
            m = min(len(start_pos), len(end_pos))
            # R recycles on unequal lengths; DNAcopy segments start and end on non-NA
            # bins so NA runs are interior and the two vectors have equal length.
            start_pos, end_pos = start_pos[:m], end_pos[:m]
            sel = (end_pos - start_pos) > na_limit
            start_pos, end_pos = start_pos[sel], end_pos[sel]
            inv_s = np.concatenate(([s1], end_pos))
            inv_e = np.concatenate((start_pos, [e1]))
            sel2 = (inv_e - inv_s) > 0    # CBS.R:103
            for a, b in zip(inv_s[sel2], inv_e[sel2]):
                yy = y[a - 1:b]
                ww = w[a - 1:b]
                ok = ~np.isnan(yy)        # CBS.R:122-127 weighted.mean(na.rm=T)
                r = float(np.sum(yy[ok] * ww[ok]) / np.sum(ww[ok])) if ok.any() else float("nan")
                out.append([c, int(a) - 1, int(b), r])   # CBS.R:129 s-1; e exclusive
    return out
