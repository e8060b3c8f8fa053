"""Synthetic shallow-WGS read-depth generator.

Produces per-chromosome int32 bin-count dicts in the reference's sample format
(keys "1".."24", see /root/reference/src/wisecondorx/convert_tools.py:60-64,110-119)
on hg38 chromosome lengths, with shared per-bin structure (mappability + three
latent bias factors) so that a within-sample reference is learnable.
Used by tests/, bench.py and tests/golden/make_golden.py -- never by the product path.
"""
import numpy as np

HG38_LENGTHS = [
    248956422, 242193529, 198295559, 190214555, 181538259, 170805979,
    159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
    114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
    58617616, 64444167, 46709983, 50818468, 156040895, 57227415,
]


def bins_per_chr(binsize):
    # convert_tools.py:60-64: int(length / binsize + 1)
    return [int(length / float(binsize) + 1) for length in HG38_LENGTHS]


class Cohort:
    """Shared structure (base mappability, latent factors) for one synthetic cohort."""

    def __init__(self, binsize, struct_seed=1234, female_y=0.002, zero_frac=0.05):
        self.binsize = int(binsize)
        self.bpc = bins_per_chr(binsize)
        self.offsets = np.concatenate(([0], np.cumsum(self.bpc)))
        nb = int(self.offsets[-1])
        rng = np.random.default_rng(struct_seed)
        base = rng.gamma(20.0, 1.0 / 20.0, nb)
        base[rng.random(nb) < zero_frac] = 0.0
        self.base = base
        self.fac = rng.standard_normal((3, nb))
        self.female_y = female_y

    def sample(self, seed, gender="F", reads=2e7, cnv=None):
        """One sample dict. cnv = list of (chr_1based, start_bin, end_bin, factor)."""
        rng = np.random.default_rng(seed)
        w = rng.normal(0.0, 0.03, 3)
        lam = self.base * np.exp(w @ self.fac)
        xs, xe = self.offsets[22], self.offsets[23]
        ys, ye = self.offsets[23], self.offsets[24]
        if gender == "M":
            lam[xs:xe] *= 0.5
            lam[ys:ye] *= 0.5
        else:
            lam[ys:ye] *= self.female_y
        if cnv:
            for (c, s, e, f) in cnv:
                o = self.offsets[c - 1]
                lam[o + s:o + e] *= f
        lam = lam * (reads / lam.sum())
        counts = rng.poisson(lam).astype(np.int32)
        return {str(c + 1): counts[self.offsets[c]:self.offsets[c + 1]].copy()
                for c in range(24)}

    def cohort(self, n, seed0=100, reads=2e7):
        """n samples, alternating M/F."""
        genders = ["M" if i % 2 == 0 else "F" for i in range(n)]
        return [self.sample(seed0 + i, genders[i], reads) for i in range(n)], genders


def corrected_matrix(n_bins_per_chr, n_samples, seed=0, sigma=0.05):
    """Direct synthetic PCA-corrected matrix X (B x S, Fortran order like
    newref_tools.train_pca returns, newref_tools.py:147): values ~1 +- sigma with
    a per-bin noise scale and weak inter-bin structure so neighbours are non-trivial.
    Returns (X, masked_bins_per_chr, masked_bins_per_chr_cum)."""
    rng = np.random.default_rng(seed)
    mb = np.asarray(n_bins_per_chr, dtype=np.int64)
    B = int(mb.sum())
    n_proto = 64
    proto = rng.standard_normal((n_proto, n_samples))
    assign = rng.integers(0, n_proto, B)
    scale = sigma * rng.gamma(8.0, 1.0 / 8.0, B)
    noise = rng.standard_normal((n_samples, B))
    Xs = 1.0 + scale[None, :] * (0.6 * proto[assign].T + 0.8 * noise)
    X = Xs.T  # (B, S) Fortran-ordered view
    return X, mb.tolist(), np.cumsum(mb).tolist()
