"""Heuristic QC verdict on a finished reference (the reference's ref_qc.py:140-218, which its
`newref` calls but never imports -- main.py:135 raises NameError after the file is written; here
it is wired properly).  Same metrics, thresholds, log lines and return codes; the per-bin loops
are array reductions."""
import logging
import os

import numpy as np

MINREFBINS = 150
OUTLIER_N_SIGMA = 3
_LABEL = {"": "A", ".F": "F", ".M": "M"}


def gender_suffixes(ref):
    """Sub-references that get a verdict: the gonosomal ones when present, else the autosomal one
    (ref_qc.py:11-20)."""
    out = [suf for suf in (".F", ".M") if "bins_per_chr" + suf in ref]
    if not out and "bins_per_chr" in ref:
        out = [""]
    return out


def _block_metrics(mean_d, n_refs, cutoff):
    valid = np.isfinite(mean_d)
    m = mean_d[valid]
    return {"n_valid": int(valid.sum()), "mean_of_means": float(np.mean(m)),
            "std_of_means": float(np.std(m)), "n_mean_outlier": int(np.sum(m >= cutoff)),
            "n_low_refs": int(np.sum(n_refs < MINREFBINS))}


def compute_metrics(ref, suf):
    """Per-bin mean distance statistics of one sub-reference (ref_qc.py:69-105): mean / std of the
    per-bin mean distances, bins beyond mean + 3 sd, bins with fewer than 150 reference bins; for
    the male sub-reference the same over the chrY rows."""
    if "indexes" + suf not in ref or "distances" + suf not in ref:
        return None
    distances = ref["distances" + suf]
    const_rows = 0
    if hasattr(distances, "n_prefix") and distances.ndim == 2 and distances.shape[1]:
        # npz_io.PrefixConst (a gonosomal pass's table): the mean of a row of k equal values is that value,
        # exactly -- the constant rows are never materialised
        const_rows, const_mean = distances.n_prefix, float(distances.fill)
        distances = np.asarray(distances.tail, dtype=float)
    else:
        distances = np.asarray(distances, dtype=float)
    n_bins = len(distances) + const_rows
    if n_bins == 0:
        return {"n_bins": 0}
    with np.errstate(all="ignore"):
        mean_d = distances.mean(axis=1) if distances.ndim == 2 and distances.shape[1] else \
            np.full(n_bins, np.nan)
    if const_rows:
        mean_d = np.concatenate([np.full(const_rows, const_mean), mean_d])
    n_refs = np.full(n_bins, ref["indexes" + suf].shape[1] if distances.ndim == 2 else 0)
    if not np.isfinite(mean_d).any():
        return {"n_bins": n_bins, "n_valid": 0}
    ok = mean_d[np.isfinite(mean_d)]
    cutoff = float(np.mean(ok)) + OUTLIER_N_SIGMA * float(np.std(ok))
    out = {"n_bins": n_bins}
    out.update(_block_metrics(mean_d, n_refs, cutoff))
    out["outlier_pct"] = 100.0 * out["n_mean_outlier"] / out["n_valid"]
    out["chrY"] = None
    cum_key = "masked_bins_per_chr_cum" + suf
    if suf == ".M" and cum_key in ref and len(np.atleast_1d(ref[cum_key])) >= 24:
        cum = np.atleast_1d(ref[cum_key])
        y0, y1 = int(cum[22]), int(cum[23])
        if y0 >= y1:
            out["chrY"] = {"n_bins": 0}
        elif not np.isfinite(mean_d[y0:y1]).any():
            out["chrY"] = {"n_bins": y1 - y0, "n_valid": 0, "mean_of_means": float("nan")}
        else:
            cy = {"n_bins": y1 - y0}
            cy.update(_block_metrics(mean_d[y0:y1], n_refs[y0:y1], cutoff))
            out["chrY"] = cy
    return out


def verdict(m, male):
    """(PASS | WARN | FAIL, reason) -- ref_qc.py:108-137: the female/autosomal rule looks at the
    spread of the per-bin mean distances, the male rule at their level and at chrY."""
    if m is None or m.get("n_valid", 0) == 0:
        return "FAIL", "no data"
    if m["n_low_refs"] > 0:
        return "WARN", "n_refs<{} in {} bins".format(MINREFBINS, m["n_low_refs"])
    if male:
        if m["mean_of_means"] > 10:
            return "FAIL", "mean(per-bin mean dist) = {:.2f} (heavy tail)".format(m["mean_of_means"])
        if m["mean_of_means"] > 2:
            return "WARN", "mean(per-bin mean dist) = {:.2f}".format(m["mean_of_means"])
        cy = m.get("chrY")
        if cy and cy.get("n_valid", 0) > 0 and np.isfinite(cy.get("mean_of_means", np.nan)):
            if cy["mean_of_means"] > 100:
                return "FAIL", "chrY mean distance = {:.1f} (very poor chrY)".format(cy["mean_of_means"])
            if cy["mean_of_means"] > 5:
                return "WARN", "chrY mean distance = {:.1f}".format(cy["mean_of_means"])
    else:
        if m["std_of_means"] > 10:
            return "FAIL", "std(per-bin mean dist) = {:.2f} (high)".format(m["std_of_means"])
        if m["std_of_means"] > 2:
            return "WARN", "std(per-bin mean dist) = {:.2f}".format(m["std_of_means"])
    if m["outlier_pct"] > 1:
        return "WARN", "outlier bins = {:.2f}%".format(m["outlier_pct"])
    return "PASS", ""


def qc_reference(reference, precomputed=None):
    """QC of a reference .npz (path) or of the in-memory dict `newref` is about to write.
    Returns the worst severity: 0 (PASS), 1 (WARN), 2 (FAIL) -- ref_qc.py:140-218.
    precomputed: {suffix: compute_metrics(...)} of sub-references whose tables the caller has already
    let go of (newref takes a pass's metrics right after the pass)."""
    if isinstance(reference, dict):
        ref, where = reference, "(in memory)"
    else:
        where = os.path.realpath(str(reference))
        if not os.path.exists(where):
            logging.error("QC check skipped: file not found: {}".format(where))
            return 2
        with np.load(where, encoding="latin1", allow_pickle=True) as npz:
            ref = {k: npz[k] for k in npz.files}
    suffixes = gender_suffixes(ref)
    if not suffixes:
        logging.error("QC failed: no bins_per_chr / bins_per_chr.F / bins_per_chr.M in npz")
        return 2
    logging.info("Starting ref-QC for file: {}".format(where))
    try:
        logging.info("Reference binsize: {} bp".format(int(np.atleast_1d(ref["binsize"])[0])))
    except Exception:
        logging.info("Reference binsize: (unknown)")
    worst = 0
    loggers = {"PASS": logging.info, "WARN": logging.warning, "FAIL": logging.error}
    for suf in suffixes:
        label = _LABEL[suf]
        m = precomputed[suf] if precomputed and suf in precomputed else compute_metrics(ref, suf)
        if m is None:
            logging.warning("[{}] no indexes/distances — skip".format(label))
            continue
        if m.get("n_valid", 0) == 0:
            logging.error("[{}] n_bins={}, n_valid=0 — FAIL".format(label, m["n_bins"]))
            worst = 2
            continue
        v, msg = verdict(m, male=(label == "M"))
        worst = max(worst, {"PASS": 0, "WARN": 1, "FAIL": 2}[v])
        log = loggers[v]
        log("[{}] n_bins={}, mean(dist)={:.4f}, std(dist)={:.4f}, outliers={} ({:.2f}%), "
            "n_refs<{}={}".format(label, m["n_bins"], m["mean_of_means"], m["std_of_means"],
                                  m["n_mean_outlier"], m["outlier_pct"], MINREFBINS, m["n_low_refs"]))
        cy = m.get("chrY")
        if cy and cy.get("n_valid", 0) > 0:
            log("       chrY: n_bins={}, mean={:.4f}, std={:.4f}, outliers={}, n_refs<{}={}".format(
                cy["n_bins"], cy["mean_of_means"], cy["std_of_means"], cy["n_mean_outlier"],
                MINREFBINS, cy["n_low_refs"]))
        log("         -> {}".format(v) + (": {}".format(msg) if msg else ""))
    if worst == 0:
        logging.info("QC Overall Verdict: PASS")
    elif worst == 1:
        logging.warning("QC Overall Verdict: WARN (review metrics above)")
    else:
        logging.error("QC Overall Verdict: FAIL (ref may cause poor predictions; consider rebuilding "
                      "or more samples)")
    return worst
