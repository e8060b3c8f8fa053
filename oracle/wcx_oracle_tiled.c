/* CPU ORACLE (C), cache-tiled variant -- TEST INFRASTRUCTURE ONLY, never linked into the product.
 *
 * The same arithmetic and the same selection as oracle/wcx_oracle.c (which follows
 * /root/reference/src/wisecondorx/newref_tools.py:255-278 row by row and is pinned against the
 * reference's fixtures), reorganised so that EVERY row of a 15 kb problem can be verified in
 * seconds on many host cores: a block of RB target rows meets the candidates tile by tile, so a
 * tile of the sample-major matrix (S x CT doubles) is fetched from memory once per RB rows instead
 * of once per row.  Per (target, candidate) the operations and their order are untouched:
 *   d = ((d0^2 + d1^2) + d2^2) + ...  over the samples j = 0 .. S-1, every sub / mul / add rounded
 *   separately (newref_tools.py:260 on the Fortran-ordered matrix; compile with -ffp-contract=off),
 * followed by the reference's bisect_right insertion with strict `v < cur_max` admission
 * (newref_tools.py:261-275).  tests/test_oracle_golden.py asserts this file == wcx_oracle.c bit for
 * bit on the reference fixtures and on seeded random problems.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RB 32
#define CT 128

static void accumulate(const double *Xs, int64_t B, int S, const int64_t *rows, int nr,
                       int64_t g0, int64_t g1, int64_t c0, double *d, int64_t nc) {
  /* global candidate rows [g0, g1) = stored candidate indices [c0, c0 + g1 - g0) */
  for (int64_t gt = g0; gt < g1; gt += CT) {
    const int64_t len = g1 - gt < CT ? g1 - gt : CT;
    const int64_t ct = c0 + (gt - g0);
    for (int r = 0; r < nr; ++r) {
      double *restrict dr = d + (int64_t)r * nc + ct;
      const int64_t t = rows[r];
      for (int j = 0; j < S; ++j) {
        const double *col = Xs + (int64_t)j * B;
        const double xt = col[t];
        const double *restrict cg = col + gt;
        for (int64_t c = 0; c < len; ++c) {
          const double diff = cg[c] - xt;
          const double sq = diff * diff;
          dr[c] = dr[c] + sq;
        }
      }
    }
  }
}

int wcxo_topk_rows_tiled(const double *Xs, int64_t B, int S, int64_t cs, int64_t ce,
                         int64_t row_begin, int64_t row_end, int k, int32_t *out_idx,
                         double *out_dist) {
  const int64_t nc = B - (ce - cs);
  double *d = (double *)malloc(sizeof(double) * (size_t)(nc > 0 ? nc : 1) * RB);
  if (!d) return 1;
  for (int64_t tb = row_begin; tb < row_end; tb += RB) {
    const int nr = (int)(row_end - tb < RB ? row_end - tb : RB);
    int64_t rows[RB];
    for (int r = 0; r < nr; ++r) rows[r] = tb + r;
    memset(d, 0, sizeof(double) * (size_t)nc * (size_t)nr);
    accumulate(Xs, B, S, rows, nr, 0, cs, 0, d, nc);          /* candidates before the own chromosome */
    accumulate(Xs, B, S, rows, nr, ce, B, cs, d, nc);         /* ... and after it */
    for (int r = 0; r < nr; ++r) {
      /* newref_tools.py:261-275 (same code as wcx_oracle.c) */
      const double *dv = d + (int64_t)r * nc;
      int32_t *idx = out_idx + (tb + r - row_begin) * (int64_t)k;
      double *dist = out_dist + (tb + r - row_begin) * (int64_t)k;
      for (int i = 0; i < k; ++i) { idx[i] = -1; dist[i] = 1e10; }
      double cur_max = 1e10;
      for (int64_t c = 0; c < nc; ++c) {
        const double v = dv[c];
        if (v < cur_max) {
          int lo = 0, hi = k;
          while (lo < hi) {
            int mid = (lo + hi) / 2;
            if (v < dist[mid]) hi = mid; else lo = mid + 1;
          }
          memmove(dist + lo + 1, dist + lo, sizeof(double) * (size_t)(k - 1 - lo));
          memmove(idx + lo + 1, idx + lo, sizeof(int32_t) * (size_t)(k - 1 - lo));
          dist[lo] = v;
          idx[lo] = (int32_t)c;
          cur_max = dist[k - 1];
        }
      }
    }
  }
  free(d);
  return 0;
}
