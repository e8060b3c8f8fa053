"""Host-side mirror of the reference's predict numerical layer (predict_tools.py,
predict_control.py) -- same function names and argument meaning; the O(B*k) work runs in
libwcx_hip.so on the MI355X, the O(B) glue stays in NumPy.
"""
import numpy as np

from . import _lib


class DeviceReference:
    """indexes{ap}/distances{ap} of a reference .npz resident in HBM (uploaded once per
    batch; SURVEY.md §8b `wcx_ref_upload`)."""

    def __init__(self, ref_file, ap="", ctx=None):
        self.ctx = ctx or _lib.default_context()
        idx = np.ascontiguousarray(ref_file["indexes{}".format(ap)], dtype=np.int32)
        dist = np.ascontiguousarray(ref_file["distances{}".format(ap)], dtype=np.float64)
        cum, cum_p = _lib.i64_array(ref_file["masked_bins_per_chr_cum{}".format(ap)])
        self.B, self.k = idx.shape
        self.chr_cum = cum
        h = _lib.vp()
        _lib.check(self.ctx.lib.wcx_ref_upload(self.ctx.h, _lib.ptr(idx), _lib.ptr(dist), self.B,
                                               self.k, cum_p, len(cum), _lib.C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.wcx_ref_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _dev(ref_file, ap, cache):
    key = ("devref", ap)
    if cache is not None and key in cache:
        return cache[key]
    d = DeviceReference(ref_file, ap)
    if cache is not None:
        cache[key] = d
    return d


def coverage_normalize_and_mask(sample, ref_file, ap):
    """predict_tools.py:32-48 (O(B) host glue)."""
    bpc = ref_file["bins_per_chr{}".format(ap)]
    by_chr = []
    for c in range(1, len(bpc) + 1):
        this_chr = np.zeros(bpc[c - 1], dtype=float)
        min_len = min(bpc[c - 1], len(sample[str(c)]))
        this_chr[:min_len] = sample[str(c)][:min_len]
        by_chr.append(this_chr)
    all_data = np.concatenate(by_chr, axis=0)
    all_data = all_data / np.sum(all_data)
    return all_data[ref_file["mask{}".format(ap)]]


def project_pc(sample_data, ref_file, ap):
    """predict_tools.py:56-65 with the scikit-learn<=1.4.2 transform the reference pins
    (setup.cfg:42): x / (((x - mean) . C^T) . C + mean)."""
    comp = ref_file["pca_components{}".format(ap)]
    mean = ref_file["pca_mean{}".format(ap)]
    t = np.dot(np.array([sample_data]) - mean, comp.T)
    reconstructed = (np.dot(t, comp) + mean)[0]
    return sample_data / reconstructed


def get_optimal_cutoff(ref_file, repeats, cache=None):
    """predict_tools.py:74-82: always on the autosomal `distances`."""
    d = _dev(ref_file, "", cache)
    out = _lib.C.c_double()
    _lib.check(d.ctx.lib.wcx_cutoff(d.ctx.h, d.h, int(repeats), _lib.C.byref(out)))
    return out.value


def get_weights(ref_file, ap, cache=None):
    """predict_tools.py:152-155."""
    d = _dev(ref_file, ap, cache)
    out = np.empty(d.B, dtype=np.float64)
    _lib.check(d.ctx.lib.wcx_weights(d.ctx.h, d.h, _lib.ptr(out)))
    return out


def normalize_repeat(test_data, ref_file, optimal_cutoff, ct, cp, ap, cache=None):
    """predict_tools.py:94-108: returns (results_z, results_r, ref_sizes, m_lr, m_z)."""
    z, r, n, mlr, mz = normalize_repeat_batch(np.asarray(test_data)[None, :], ref_file,
                                              optimal_cutoff, ct, cp, ap, cache)
    return z[0], r[0], n[0], mlr[0], mz[0]


def normalize_repeat_batch(test_batch, ref_file, optimal_cutoff, ct, cp, ap, cache=None):
    """Batched normalize_repeat: test_batch float64[n_samples][B]."""
    d = _dev(ref_file, ap, cache)
    x = np.ascontiguousarray(test_batch, dtype=np.float64)
    ns, B = x.shape
    if B != d.B:
        raise ValueError("sample vector length {} != reference bins {}".format(B, d.B))
    Bp = B - int(ct)
    z = np.empty((ns, Bp))
    r = np.empty((ns, Bp))
    n = np.empty((ns, Bp))
    mlr = np.empty(ns)
    mz = np.empty(ns)
    _lib.check(d.ctx.lib.wcx_predict_normalize(
        d.ctx.h, d.h, _lib.ptr(x), ns, float(optimal_cutoff), int(ct), int(cp), _lib.ptr(z),
        _lib.ptr(r), _lib.ptr(n), _lib.ptr(mlr), _lib.ptr(mz)))
    return z, r, n, mlr, mz


def normalize(args, sample, ref_file, ref_gender, cache=None):
    """predict_control.py:21-39."""
    if ref_gender == "A":
        ap, cp, ct = "", 0, 0
    else:
        ap = ".{}".format(ref_gender)
        cp = 22
        ct = int(ref_file["masked_bins_per_chr_cum{}".format(ap)][cp - 1])
    sample = coverage_normalize_and_mask(sample, ref_file, ap)
    sample = project_pc(sample, ref_file, ap)
    results_w = get_weights(ref_file, ap, cache)[ct:]
    optimal_cutoff = get_optimal_cutoff(ref_file, args.maskrepeats, cache)
    results_z, results_r, ref_sizes, m_lr, m_z = normalize_repeat(
        sample, ref_file, optimal_cutoff, ct, cp, ap, cache)
    return results_r, results_z, results_w, ref_sizes, m_lr, m_z
