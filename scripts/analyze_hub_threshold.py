#!/usr/bin/env python3
"""How tight is a RIGOROUS threshold taken from the low-norm rows only?

For a sample of target rows of the bench cohort's A pass: the k-th smallest distance among the
candidates of the hub set H (the fraction f of rows with the smallest centred norm, own chromosome
excluded) bounds the true k-th distance from above; how many candidates lie below it (= what a sweep
with that threshold admits), against the r-th-of-a-uniform-1/16-sample estimate of round 2-4?
CPU only (host prep + NumPy); usage: analyze_hub_threshold.py [n_samples] [n_targets]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workload(S, binsize=15000, seed=0):
    cache = "/tmp/wcx_X_%d_%d.npy" % (binsize, S)
    if os.path.exists(cache):
        d = np.load(cache + ".meta.npz")
        return np.load(cache, mmap_mode="r"), d["cum"]
    from wisecondorx_amd import prep
    from wisecondorx_amd.overall_tools import gender_correct
    from wisecondorx_amd.synth import Cohort
    co = Cohort(binsize, struct_seed=1234 + seed, female_y=0.1)
    samples, genders = co.cohort(S, seed0=100 + seed)
    samples = np.array([gender_correct(s_, g_) for s_, g_ in zip(samples, genders)])
    g = np.array(genders)
    total_mask, bpc = prep.get_mask(samples)
    total_mask = total_mask & prep.get_mask(samples[g == "F"])[0] & prep.get_mask(samples[g == "M"])[0]
    p = prep.prepare(samples, "A", total_mask, bpc, ctx=None)
    X = np.ascontiguousarray(p["X"])            # (B, S)
    np.save(cache, X)
    np.savez(cache + ".meta.npz", cum=p["masked_bins_per_chr_cum"])
    return X, p["masked_bins_per_chr_cum"]


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    k = 300
    t0 = time.time()
    X, cum = workload(S)
    B = X.shape[0]
    print("X", X.shape, "%.1f s" % (time.time() - t0))
    Xc = X - X.mean(axis=0)
    n2 = np.einsum("ij,ij->i", Xc, Xc)
    order = np.argsort(n2, kind="stable")
    rank = np.empty(B, dtype=np.int64)
    rank[order] = np.arange(B)
    chrom = np.searchsorted(cum, np.arange(B), side="right")
    rng = np.random.default_rng(1)
    targets = np.sort(rng.choice(B, nt, replace=False))
    D = n2[targets][:, None] + n2[None, :] - 2.0 * (Xc[targets] @ Xc.T)     # (nt, B)
    D[chrom[targets][:, None] == chrom[None, :]] = np.inf
    srt = np.sort(D, axis=1)
    dk = srt[:, k - 1]
    print("norm^2 quantiles (1,6.25,25,50,75,99 %%): %s" % np.percentile(n2, [1, 6.25, 25, 50, 75, 99]))
    # today's estimate: r-th smallest of the uniform sample b = 0 mod 16
    samp = np.arange(0, B, 16)
    r = 48
    est = np.sort(D[:, samp], axis=1)[:, r - 1]
    adm = (D <= est[:, None]).sum(axis=1)
    print("uniform 1/16, r = %d: admitted per row mean %.0f median %.0f max %d; estimate fails %d" %
          (r, adm.mean(), np.median(adm), adm.max(), int((est < dk).sum())))
    for f in (1 / 128, 1 / 64, 1 / 32, 1 / 16, 1 / 8):
        H = order[:int(B * f)]
        inH = np.zeros(B, bool)
        inH[H] = True
        dh = np.sort(D[:, H], axis=1)[:, k - 1]            # rigorous bound on the k-th distance
        adm = (D <= dh[:, None]).sum(axis=1)
        adm_out = ((D <= dh[:, None]) & ~inH[None, :]).sum(axis=1)
        # share of the true k nearest inside H
        nn = np.argsort(D, axis=1)[:, :k]
        share = inH[nn].mean()
        # 6 % slack on the distance stands in for the filter margins (E, Q)
        dh6 = dh * 1.06
        adm6 = (D <= dh6[:, None]).sum(axis=1)
        adm6_out = ((D <= dh6[:, None]) & ~inH[None, :]).sum(axis=1)
        print("hub f = 1/%-4d |H| = %6d: share of true kNN in H %.3f; admitted mean %.0f (outside H %.0f) "
              "p90 %.0f max %d | +6%%: %.0f (outside %.0f)" %
              (round(1 / f), len(H), share, adm.mean(), adm_out.mean(), np.percentile(adm, 90), adm.max(),
               adm6.mean(), adm6_out.mean()))
        # by norm decile of the target
        dec = np.minimum(9, rank[targets] * 10 // B)
        print("   admitted by target norm decile: " +
              " ".join("%d:%.0f" % (d_, adm[dec == d_].mean()) for d_ in range(10) if np.any(dec == d_)))


if __name__ == "__main__":
    main()
