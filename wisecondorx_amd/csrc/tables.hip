// Output tables (replaces the per-row string formatting of the reference's
// predict_output.py:51-75 for ID_bins.bed -- 206 k rows at 15 kb, two floats each).  Host code only:
// the rows are laid out in one buffer, the floats printed exactly like Python's repr(float)
// (= str(float), what the reference's str(x) on list elements produces): shortest digits that round-
// trip, exponent form for values < 1e-4 or >= 1e16, ".0" appended to integers.
#include <charconv>
#include <cstring>
#include <thread>
#include <vector>

#include "wcx_common.h"

namespace {

// repr(float) of Python 3 (Python/pystrtod.c format_float_short, mode 'r'); returns the end pointer.
char *py_repr(double v, char *p) {
  if (v != v) { memcpy(p, "nan", 3); return p + 3; }
  if (v == HUGE_VAL) { memcpy(p, "inf", 3); return p + 3; }
  if (v == -HUGE_VAL) { memcpy(p, "-inf", 4); return p + 4; }
  char t[40];
  const std::to_chars_result rr = std::to_chars(t, t + sizeof(t), v, std::chars_format::scientific);
  // t = [-]d[.ddd]e[+-]XX : shortest round-trip digits
  const char *s = t, *end = rr.ptr;
  if (*s == '-') { *p++ = '-'; ++s; }
  char dig[24];
  int nd = 0;
  const char *e = s;
  while (e < end && *e != 'e') { if (*e != '.') dig[nd++] = *e; ++e; }
  int ex = 0;
  {
    const char *q = e + 1;
    const bool neg = *q == '-';
    if (*q == '-' || *q == '+') ++q;
    while (q < end) ex = ex * 10 + (*q++ - '0');
    if (neg) ex = -ex;
  }
  const int decpt = ex + 1;              // value = 0.d1 d2 ... x 10^decpt
  if (decpt <= -4 || decpt > 16) {       // exponent form: d[.ddd]e+XX (at least two exponent digits)
    *p++ = dig[0];
    if (nd > 1) { *p++ = '.'; memcpy(p, dig + 1, (size_t)(nd - 1)); p += nd - 1; }
    *p++ = 'e';
    int x = decpt - 1;
    *p++ = x < 0 ? '-' : '+';
    if (x < 0) x = -x;
    char xb[8];
    int nx = 0;
    do { xb[nx++] = (char)('0' + x % 10); x /= 10; } while (x);
    if (nx < 2) xb[nx++] = '0';
    while (nx) *p++ = xb[--nx];
    return p;
  }
  if (decpt <= 0) {                      // 0.000ddd
    *p++ = '0'; *p++ = '.';
    for (int i = 0; i < -decpt; ++i) *p++ = '0';
    memcpy(p, dig, (size_t)nd);
    return p + nd;
  }
  if (decpt >= nd) {                     // ddd000.0
    memcpy(p, dig, (size_t)nd); p += nd;
    for (int i = nd; i < decpt; ++i) *p++ = '0';
    *p++ = '.'; *p++ = '0';
    return p;
  }
  memcpy(p, dig, (size_t)decpt); p += decpt;
  *p++ = '.';
  memcpy(p, dig + decpt, (size_t)(nd - decpt));
  return p + (nd - decpt);
}

char *put_i64(int64_t v, char *p) {
  char b[24];
  int n = 0;
  do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = b[--n];
  return p;
}

}  // namespace

extern "C" {

int64_t wcx_format_floats(const double *v, int64_t n, char sep, char *out, int64_t cap) {
  if (!v || !out || n < 0) return -1;
  if (cap < n * 26) return -1;
  char *p = out;
  for (int64_t i = 0; i < n; ++i) { p = py_repr(v[i], p); *p++ = sep; }
  return p - out;
}

int64_t wcx_format_bins_bed(const char *chr_name, int64_t n, int64_t binsize, const double *r, const double *z,
                            char *out, int64_t cap) {
  if (!chr_name || !r || !z || !out || n < 0 || binsize <= 0) return -1;
  const size_t ln = strlen(chr_name);
  if (cap < n * (int64_t)(2 * ln + 4 * 20 + 2 * 26 + 8)) return -1;
  char *p = out;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t a = i * binsize + 1, b = (i + 1) * binsize;
    memcpy(p, chr_name, ln); p += ln; *p++ = '\t';
    p = put_i64(a, p); *p++ = '\t';
    p = put_i64(b, p); *p++ = '\t';
    memcpy(p, chr_name, ln); p += ln; *p++ = ':';
    p = put_i64(a, p); *p++ = '-';
    p = put_i64(b, p); *p++ = '\t';
    if (r[i] == 0.0) { memcpy(p, "nan", 3); p += 3; } else p = py_repr(r[i], p);
    *p++ = '\t';
    if (z[i] == 0.0) { memcpy(p, "nan", 3); p += 3; } else p = py_repr(z[i], p);
    *p++ = '\n';
  }
  return p - out;
}

// Bin counts of a batch of samples -> one int32 matrix over the reference's bin layout (replaces the loop of
// predict_tools.py:36-44 per sample: every chromosome truncated or zero-padded to the reference's
// bins_per_chr).  src[s * n_chr + c] = the int32 counts of chromosome c of sample s (len[...] of them);
// out int32 [n_samples][sum(bins_per_chr)].  79 MB at 15 kb x 96 samples: a host memcpy job, spread over
// n_threads threads by sample (one Python thread: 10 ms; 16 threads here: under 1 ms).
int wcx_layout_counts(const int32_t *const *src, const int64_t *len, int n_samples, int n_chr,
                      const int64_t *bins_per_chr, int32_t *out, int n_threads) {
  if (!src || !len || !bins_per_chr || !out || n_samples < 0 || n_chr <= 0) {
    wcx_set_error("bad argument: wcx_layout_counts");
    return WCX_ERR_ARG;
  }
  int64_t n_bins = 0;
  for (int c = 0; c < n_chr; ++c) n_bins += bins_per_chr[c];
  auto one = [&](int s) {
    int32_t *row = out + (int64_t)s * n_bins;
    for (int c = 0; c < n_chr; ++c) {
      const int64_t want = bins_per_chr[c];
      const int64_t have = len[(int64_t)s * n_chr + c] < want ? len[(int64_t)s * n_chr + c] : want;
      if (have > 0) memcpy(row, src[(int64_t)s * n_chr + c], (size_t)have * 4);
      if (have < want) memset(row + (have > 0 ? have : 0), 0, (size_t)(want - (have > 0 ? have : 0)) * 4);
      row += want;
    }
  };
  if (n_threads > n_samples) n_threads = n_samples;
  if (n_threads <= 1) {
    for (int s = 0; s < n_samples; ++s) one(s);
    return WCX_OK;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&, t] { for (int s = t; s < n_samples; s += n_threads) one(s); });
  for (auto &x : th) x.join();
  return WCX_OK;
}

}  // extern "C"
