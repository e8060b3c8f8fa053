/* CPU ORACLE (C) -- TEST INFRASTRUCTURE ONLY, never linked into the product library.
 *
 * Plain-C restatement of the reference's per-bin search, used where the NumPy oracle
 * (oracle/wcx_oracle.py) is too slow, and as bench.py's `cpu_baseline` ("port").
 * Pinned against the reference through tests/golden/newref_search.npz
 * (tests/test_oracle_golden.py::test_c_oracle_*).
 *
 * Follows /root/reference/src/wisecondorx/newref_tools.py:255-278:
 *   :260      d[c] = sum_j (chr_data[c][j] - x[j])^2  -- on the Fortran-ordered matrix NumPy
 *             accumulates sample by sample, so every candidate sees the sequential fp64 sum
 *             ((d0^2 + d1^2) + d2^2) + ... ; compile with -ffp-contract=off.
 *   :261-275  sorted list of the ref_size smallest, bisect_right insertion, strict
 *             `binVal < cur_max` admission, sentinels -1 / 1e10.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Xs: sample-major double[S][B]; own chromosome rows [cs,ce) are excluded from the
 * candidates; stored index = position in the concatenation X[:cs] ++ X[ce:]. */
int wcxo_topk_rows(const double *Xs, int64_t B, int S, int64_t cs, int64_t ce,
                   int64_t row_begin, int64_t row_end, int k, int32_t *out_idx,
                   double *out_dist) {
  const int64_t nc = B - (ce - cs);
  double *d = (double *)malloc(sizeof(double) * (size_t)(nc > 0 ? nc : 1));
  if (!d) return 1;
  for (int64_t t = row_begin; t < row_end; ++t) {
    /* newref_tools.py:260, sample by sample */
    for (int64_t c = 0; c < nc; ++c) d[c] = 0.0;
    for (int j = 0; j < S; ++j) {
      const double *col = Xs + (int64_t)j * B;
      const double xt = col[t];
      for (int64_t c = 0; c < nc; ++c) {
        const int64_t g = c < cs ? c : c + (ce - cs);
        const double diff = col[g] - xt;
        const double sq = diff * diff;
        d[c] = d[c] + sq;
      }
    }
    /* newref_tools.py:261-275 */
    int32_t *idx = out_idx + (t - row_begin) * (int64_t)k;
    double *dist = out_dist + (t - row_begin) * (int64_t)k;
    for (int i = 0; i < k; ++i) { idx[i] = -1; dist[i] = 1e10; }
    double cur_max = 1e10;
    for (int64_t c = 0; c < nc; ++c) {
      const double v = d[c];
      if (v < cur_max) {
        /* bisect.bisect (== bisect_right) on the ascending list */
        int lo = 0, hi = k;
        while (lo < hi) {
          int mid = (lo + hi) / 2;
          if (v < dist[mid]) hi = mid; else lo = mid + 1;
        }
        /* pop(-1) then insert(pos) */
        memmove(dist + lo + 1, dist + lo, sizeof(double) * (size_t)(k - 1 - lo));
        memmove(idx + lo + 1, idx + lo, sizeof(int32_t) * (size_t)(k - 1 - lo));
        dist[lo] = v;
        idx[lo] = (int32_t)c;
        cur_max = dist[k - 1];
      }
    }
  }
  free(d);
  return 0;
}
