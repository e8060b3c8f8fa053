// Circular binary segmentation on the GPU + segment z-scores.
//
// a16 replaces predict_tools.exec_cbs -> Rscript include/CBS.R -> DNAcopy::segment
// (predict_tools.py:242-257, CBS.R:21-132).  The reference-owned code around the DNAcopy call
// (NA masking, weight fix-up CBS.R:41-42, dropping all-NA chromosomes :56-63, splitting segments
// over long NA runs :84-113, weighted re-mean :122-127, 0-based starts :129) is reproduced
// exactly.  The segmentation itself lives in Bioconductor DNAcopy 1.76.0 (conda.yml:14), which is
// NOT part of the reference repository and cannot run here (no R): PARITY UNPINNED.  It is
// restated from the published algorithm with DNAcopy's defaults (Olshen et al. 2004;
// Venkatraman & Olshen 2007): weighted circular binary segmentation, max-arc statistic on
// weighted partial sums, "hybrid" p-value for n > nmin=200 (Siegmund tail approximation for arcs
// longer than kmax=25 + permutation reference distribution for the short arcs, nperm=10000,
// early stop as soon as the exceedance budget is spent), min.width=2, edge test of each of two
// change-points, undo.splits="none".  Differences that make breakpoint parity impossible even
// with R available: R's Mersenne-Twister permutation stream (here: counter-based hash), and
// the edge test (here: weighted two-sample Student t instead of a permutation t-test).
//
// GPU mapping: one workgroup per permutation -- random keys, bitonic sort in LDS (the
// permutation), weighted re-centring, prefix scan, max over short arcs; the observed all-arc
// maximum is an O(n^2) pairwise kernel.  Compute/latency bound, reported as wall-clock only.
//
// a17 replaces overall_tools.get_z_score (overall_tools.py:88-119): HBM-bound column sums.
#include <algorithm>
#include <cmath>

#include "wave_sort.h"
#include "wcx_common.h"

namespace {

constexpr int NTP = 1024;

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

struct ArcBest {
  unsigned long long packed;  // (float bits of B) << 32 | i << 16 | j   (n <= 32768: 16 bits each)
};

// observed statistic: max over arcs (i,j], minw <= j-i <= n-minw, of the between-sum-of-squares
__global__ __launch_bounds__(256) void k_cbs_arcmax(const double *__restrict__ S,
                                                    const double *__restrict__ Wp, int n, int minw,
                                                    ArcBest *__restrict__ best) {
  const double W = Wp[n];
  unsigned long long loc = 0;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const double si = S[i], wi = Wp[i];
    for (int j = i + minw + threadIdx.x; j <= n; j += 256) {
      const int a = j - i;
      if (n - a < minw) break;
      const double d = S[j] - si, wa = Wp[j] - wi;
      const float b = (float)(d * d / (wa * (W - wa) / W));
      if (b == b) {
        const unsigned long long p = ((unsigned long long)__float_as_uint(b) << 32) |
                                     ((unsigned long long)i << 16) | (unsigned long long)(j & 0xffff);
        loc = p > loc ? p : loc;
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long o = __shfl_xor(loc, m, 64);
    loc = o > loc ? o : loc;
  }
  if ((threadIdx.x & 63) == 0 && loc) atomicMax(&best->packed, loc);
}

// One workgroup per permutation.  y = centred residual * sqrt(w) (exchangeable under H0),
// rw = sqrt(w), Wp = prefix sums of w (float).  out[p] = permuted max statistic (t^2).
__global__ __launch_bounds__(NTP) void k_cbs_perm(const float *__restrict__ y,
                                                  const float *__restrict__ rw,
                                                  const float *__restrict__ Wp, int n, int npad,
                                                  int ibits, int minw, int kmax, int hybrid,
                                                  unsigned long long seed, int perm0,
                                                  float *__restrict__ out) {
  extern __shared__ unsigned int sk[];  // npad words: keys, then the permuted weighted series
  __shared__ float red[NTP / 64];
  const int tid = threadIdx.x;
  const int p = perm0 + blockIdx.x;
  const unsigned long long s0 = mix64(seed ^ ((unsigned long long)p * 0xd1342543de82ef95ull));
  // Random permutation = order of the hashed keys (unique: the index sits in the low bits).  The
  // keys are uniform, so a bucket sort on their leading bits is O(n): count, scan, scatter (keys
  // are re-hashed, no second array), then an insertion sort of the ~16 keys of each bucket --
  // the same permutation as a full sort of the keys at a tenth of the LDS traffic.
  unsigned int *bc = sk + npad;                       // [nbk] bucket counters / cursors
  const int nbk = npad >= 1024 ? npad / 16 : (npad >= 16 ? npad / 16 : 1);
  int lb = 0;
  while ((1 << lb) < nbk) ++lb;
  auto key_of = [&](int i) {
    return ((unsigned int)(mix64(s0 + (unsigned long long)i) >> (32 + ibits)) << ibits) | (unsigned int)i;
  };
  auto bucket_of = [&](unsigned int key) { return lb ? (int)(key >> (32 - lb)) : 0; };
  for (int b = tid; b < nbk; b += NTP) bc[b] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += NTP) atomicAdd(&bc[bucket_of(key_of(i))], 1u);
  __syncthreads();
  {   // exclusive scan of the bucket counts (nbk <= NTP)
    __shared__ unsigned int iscan[NTP];
    const unsigned int mine = tid < nbk ? bc[tid] : 0u;
    iscan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < NTP; off <<= 1) {
      const unsigned int v = tid >= off ? iscan[tid - off] : 0u;
      __syncthreads();
      iscan[tid] += v;
      __syncthreads();
    }
    if (tid < nbk) bc[tid] = iscan[tid] - mine;
  }
  __syncthreads();
  for (int i = tid; i < n; i += NTP) {
    const unsigned int key = key_of(i);
    sk[atomicAdd(&bc[bucket_of(key)], 1u)] = key;     // afterwards bc[b] = end of bucket b
  }
  __syncthreads();
  for (int b = tid; b < nbk; b += NTP) {
    const int lo = b ? (int)bc[b - 1] : 0, hi = (int)bc[b];
    for (int i = lo + 1; i < hi; ++i) {
      const unsigned int kx = sk[i];
      int j = i - 1;
      while (j >= lo && sk[j] > kx) { sk[j + 1] = sk[j]; --j; }
      sk[j + 1] = kx;
    }
  }
  __syncthreads();
  const unsigned int imask = (1u << ibits) - 1u;
  // weighted mean of the permuted series: sum_i w_i (y_pi(i) / rw_i) = sum_i rw_i y_pi(i)
  auto block_sum = [&](float v) {
    v = (float)wcx::wave_sum((double)v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < NTP / 64; ++w) t += red[w];
    return t;
  };
  float part = 0.f;
  for (int i = tid; i < n; i += NTP) part += rw[i] * y[sk[i] & imask];
  const float W = Wp[n];
  const float mean = block_sum(part) / W;
  float tssl = 0.f;
  for (int i = tid; i < npad; i += NTP) {
    float cx = 0.f;
    if (i < n) {
      const float r = rw[i];
      const float v = y[sk[i] & imask] / r - mean;
      cx = r * r * v;          // w_i v_i
      tssl += cx * v;          // w_i v_i^2
    }
    sk[i] = __float_as_uint(cx);
  }
  const float tss = block_sum(tssl);
  // inclusive prefix scan of sk (as floats): serial chunks + scan of chunk totals
  const int chunk = npad / NTP > 0 ? npad / NTP : 1;
  const int nth = npad / chunk;   // threads that own a chunk
  __shared__ float tot[NTP];
  float run = 0.f;
  if (tid < nth) {
    for (int c = 0; c < chunk; ++c) {
      run += __uint_as_float(sk[tid * chunk + c]);
      sk[tid * chunk + c] = __float_as_uint(run);
    }
  }
  tot[tid] = tid < nth ? run : 0.f;
  __syncthreads();
  for (int off = 1; off < NTP; off <<= 1) {
    const float v = tid >= off ? tot[tid - off] : 0.f;
    __syncthreads();
    tot[tid] += v;
    __syncthreads();
  }
  if (tid < nth && tid > 0) {
    const float base = tot[tid - 1];
    for (int c = 0; c < chunk; ++c)
      sk[tid * chunk + c] = __float_as_uint(__uint_as_float(sk[tid * chunk + c]) + base);
  }
  __syncthreads();
  auto Sx = [&](int i) { return i == 0 ? 0.f : __uint_as_float(sk[i - 1]); };   // S_0 = 0
  float bmax = 0.f;
  const int amax_all = n - minw;
  if (hybrid) {
    const int a_hi = kmax < amax_all ? kmax : amax_all;
    const int na = a_hi - minw + 1;
    if (na > 0)
      for (int q = tid; q < na * (n + 1); q += NTP) {
        const int a = minw + q % na, i = q / na;
        if (i + a > n) continue;
        const float d = Sx(i + a) - Sx(i), wa = Wp[i + a] - Wp[i];
        const float b = d * d / (wa * (W - wa) / W);
        bmax = b > bmax ? b : bmax;
      }
    const int a_lo = (n - kmax > a_hi + 1) ? n - kmax : a_hi + 1;   // complement is short
    for (int a = a_lo; a <= amax_all; ++a)
      for (int i = tid; i + a <= n; i += NTP) {
        const float d = Sx(i + a) - Sx(i), wa = Wp[i + a] - Wp[i];
        const float b = d * d / (wa * (W - wa) / W);
        bmax = b > bmax ? b : bmax;
      }
  } else {
    for (int i = 0; i < n; ++i)
      for (int j = i + minw + tid; j <= n && n - (j - i) >= minw; j += NTP) {
        const float d = Sx(j) - Sx(i), wa = Wp[j] - Wp[i];
        const float b = d * d / (wa * (W - wa) / W);
        bmax = b > bmax ? b : bmax;
      }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(bmax, m, 64); bmax = o > bmax ? o : bmax; }
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = bmax;
  __syncthreads();
  if (tid == 0) {
    float b = 0.f;
    for (int w = 0; w < NTP / 64; ++w) b = red[w] > b ? red[w] : b;
    out[blockIdx.x] = b / ((tss - b) / (float)(n - 2));
  }
}

// nu(x) series for a grid of x values: one workgroup per x, threads over the terms
//   ln nu = ln 2 - 2 ln x - 2 sum_{k>=1} Phi(-x sqrt(k)/2) / k      (terms vanish once x sqrt(k)/2 > 8.5)
__global__ __launch_bounds__(256) void k_nu_series(const double *__restrict__ xs,
                                                   double *__restrict__ out) {
  const double x = xs[blockIdx.x];
  __shared__ double red[4];
  double acc = 0.0;
  if (x > 0.01) {
    const double kmax_d = (17.0 / x) * (17.0 / x);
    const long long kmax = kmax_d < 4.0e7 ? (long long)kmax_d + 1 : 40000000ll;
    for (long long k = 1 + threadIdx.x; k <= kmax; k += 256) {
      const double dk = (double)k;
      acc += 0.5 * erfc(x * sqrt(dk) * 0.5 * 0.70710678118654752440) / dk;   // Phi(-x sqrt(k)/2)/k
    }
  }
  acc = wcx::wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = red[0] + red[1] + red[2] + red[3];
    out[blockIdx.x] = x > 0.01 ? exp(log(2.0) - 2.0 * log(x) - 2.0 * s) : exp(-0.583 * x);
  }
}

// ------------------------------------------------------------------ host-side statistics
double fpnorm(double x) { return 0.5 * erfc(-x / M_SQRT2); }

double it1tsq(double x, double a) {   // integral of 1/(t(1-t))^2 over [x, x+a]
  double y = x + a - 0.5;
  double r = 8.0 * y / (1.0 - 4.0 * y * y) + 2.0 * log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
  y = x - 0.5;
  r -= 8.0 * y / (1.0 - 4.0 * y * y) + 2.0 * log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
  return r;
}

// P(max over arcs with delta <= length/m <= 1-delta of the CBS statistic >= b), Gaussian null.
// The ngrid nu() evaluations (10^4..10^6 series terms each for long chromosomes) run on the GPU.
int tailp_gpu(wcx_ctx *ctx, double *d_x, double *d_nu, double b, double delta, int m, int ngrid,
              double *result) {
  std::vector<double> xs(ngrid), tls(ngrid), nus(ngrid);
  const double dincr = (0.5 - delta) / ngrid;
  const double bsqrtm = b / sqrt((double)m);
  double tl = 0.5 - dincr, t = 0.5 - 0.5 * dincr;
  for (int i = 0; i < ngrid; ++i) {
    xs[i] = bsqrtm / sqrt(t * (1.0 - t));
    tls[i] = tl;
    tl -= dincr;
    t -= dincr;
  }
  WCX_HIP(hipMemcpyAsync(d_x, xs.data(), ngrid * 8, hipMemcpyHostToDevice, ctx->stream));
  k_nu_series<<<ngrid, 256, 0, ctx->stream>>>(d_x, d_nu);
  WCX_HIP(hipGetLastError());
  WCX_HIP(hipMemcpyAsync(nus.data(), d_nu, ngrid * 8, hipMemcpyDeviceToHost, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  double acc = 0.0;
  for (int i = 0; i < ngrid; ++i) acc += nus[i] * nus[i] * it1tsq(tls[i], dincr);
  *result = 9.973557e-2 * b * b * b * exp(-b * b / 2.0) * acc;
  return WCX_OK;
}

// regularised incomplete beta (continued fraction) -> two-sided Student t p-value
double betacf(double a, double b, double x) {
  const double eps = 3e-16, fpmin = 1e-300;
  double qab = a + b, qap = a + 1, qam = a - 1, c = 1, d = 1 - qab * x / qap;
  if (fabs(d) < fpmin) d = fpmin;
  d = 1 / d;
  double h = d;
  for (int m = 1; m <= 300; ++m) {
    int m2 = 2 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1 + aa * d; if (fabs(d) < fpmin) d = fpmin;
    c = 1 + aa / c; if (fabs(c) < fpmin) c = fpmin;
    d = 1 / d; h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1 + aa * d; if (fabs(d) < fpmin) d = fpmin;
    c = 1 + aa / c; if (fabs(c) < fpmin) c = fpmin;
    d = 1 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1) < eps) break;
  }
  return h;
}
double betai(double a, double b, double x) {
  if (x <= 0) return 0;
  if (x >= 1) return 1;
  double bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log(1 - x));
  if (x < (a + 1) / (a + b + 2)) return bt * betacf(a, b, x) / a;
  return 1 - bt * betacf(b, a, 1 - x) / b;
}
double t_two_sided(double t, double df) { return betai(df / 2, 0.5, df / (df + t * t)); }

// weighted two-sample t-test p-value of x[a..b) vs x[b..c)
double edge_pvalue(const double *x, const double *w, int a, int b, int c) {
  double W1 = 0, W2 = 0, s1 = 0, s2 = 0;
  for (int i = a; i < b; ++i) { W1 += w[i]; s1 += w[i] * x[i]; }
  for (int i = b; i < c; ++i) { W2 += w[i]; s2 += w[i] * x[i]; }
  const double m1 = s1 / W1, m2 = s2 / W2;
  double ss = 0;
  for (int i = a; i < b; ++i) ss += w[i] * (x[i] - m1) * (x[i] - m1);
  for (int i = b; i < c; ++i) ss += w[i] * (x[i] - m2) * (x[i] - m2);
  const int df = (c - a) - 2;
  if (df < 1) return 1.0;
  const double se2 = ss / df * (1.0 / W1 + 1.0 / W2);
  if (!(se2 > 0)) return (m1 != m2) ? 0.0 : 1.0;
  return t_two_sided((m1 - m2) / sqrt(se2), (double)df);
}

struct CbsParams {
  double alpha;
  int nperm = 10000, kmax = 25, nmin = 200, minw = 2, ngrid = 100;
  double tol = 1e-6;
  unsigned long long seed;
};

struct CbsWork {  // device buffers reused across tests
  double *dS = nullptr, *dWp = nullptr;
  float *dy = nullptr, *drw = nullptr, *dWpf = nullptr, *dout = nullptr;
  ArcBest *dbest = nullptr;
  double *dtx = nullptr, *dtnu = nullptr;   // tail-probability grid
  int cap = 0;
};

// One change-point test on x[0..n) (host arrays).  Returns ncpt and icpt (positions within the
// segment: the segment splits AFTER element icpt).
int cbs_test(wcx_ctx *ctx, CbsWork &wk, const double *x, const double *w, int n,
             const CbsParams &P, unsigned long long test_id, int *ncpt, int icpt[2]) {
  *ncpt = 0;
  if (n < 2 * P.minw) return WCX_OK;
  std::vector<double> xc(n), S(n + 1), Wp(n + 1);
  double W = 0, sw = 0;
  for (int i = 0; i < n; ++i) { W += w[i]; sw += w[i] * x[i]; }
  const double mean = sw / W;
  double tss = 0;
  S[0] = 0; Wp[0] = 0;
  for (int i = 0; i < n; ++i) {
    xc[i] = x[i] - mean;
    tss += w[i] * xc[i] * xc[i];
    S[i + 1] = S[i] + w[i] * xc[i];
    Wp[i + 1] = Wp[i] + w[i];
  }
  if (!(tss > 0)) return WCX_OK;
  std::vector<float> y(n), rw(n), Wpf(n + 1);
  for (int i = 0; i < n; ++i) { rw[i] = (float)sqrt(w[i]); y[i] = (float)(xc[i] * sqrt(w[i])); }
  for (int i = 0; i <= n; ++i) Wpf[i] = (float)Wp[i];
  hipStream_t st = ctx->stream;
  WCX_HIP(hipMemcpyAsync(wk.dS, S.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(wk.dWp, Wp.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(wk.dy, y.data(), n * 4, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(wk.drw, rw.data(), n * 4, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(wk.dWpf, Wpf.data(), (n + 1) * 4, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemsetAsync(wk.dbest, 0, sizeof(ArcBest), st));
  const int gb = n < 2048 ? n : 2048;
  k_cbs_arcmax<<<gb, 256, 0, st>>>(wk.dS, wk.dWp, n, P.minw, wk.dbest);
  ArcBest hb;
  WCX_HIP(hipMemcpyAsync(&hb, wk.dbest, sizeof(hb), hipMemcpyDeviceToHost, st));
  WCX_HIP(hipStreamSynchronize(st));
  if (!hb.packed) return WCX_OK;
  const int bi = (int)((hb.packed >> 16) & 0xffff);
  int bj = (int)(hb.packed & 0xffff);
  if (bj <= bi) bj += 65536 * ((bi - bj) / 65536 + 1);   // j stored modulo 2^16 (n <= 32768: no-op)
  const double d = S[bj] - S[bi], wa = Wp[bj] - Wp[bi];
  const double bss = d * d / (wa * (W - wa) / W);
  const double ostat = bss / ((tss - bss) / (n - 2.0));     // t^2 of the best arc
  const bool hybrid = n > P.nmin;
  double pval2 = P.alpha;
  if (hybrid) {
    const double delta = (P.kmax + 1.0) / n;
    double pval1 = 0.0;
    int rct = tailp_gpu(ctx, wk.dtx, wk.dtnu, sqrt(ostat), delta, n, P.ngrid, &pval1);
    if (rct) return rct;
    if (pval1 > P.alpha) return WCX_OK;
    pval2 = P.alpha - pval1;
  }
  const int nrejc = (int)(pval2 * P.nperm);
  int npad = 64, ibits = 6;
  while (npad < n) { npad <<= 1; ++ibits; }
  int nrej = 0;
  bool significant = true;
  const int batch = 256;
  std::vector<float> hout(batch);
  WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cbs_perm),
                              hipFuncAttributeMaxDynamicSharedMemorySize, npad * 4 + (npad / 16 + 1) * 4));
  for (int p0 = 0; p0 < P.nperm && significant; p0 += batch) {
    const int nb = P.nperm - p0 < batch ? P.nperm - p0 : batch;
    k_cbs_perm<<<nb, NTP, (size_t)npad * 4 + (size_t)(npad / 16 + 1) * 4, st>>>(wk.dy, wk.drw, wk.dWpf, n, npad, ibits, P.minw,
                                                  P.kmax, hybrid ? 1 : 0,
                                                  P.seed ^ (test_id * 0x2545f4914f6cdd1dull), p0,
                                                  wk.dout);
    WCX_HIP(hipGetLastError());
    WCX_HIP(hipMemcpyAsync(hout.data(), wk.dout, nb * 4, hipMemcpyDeviceToHost, st));
    WCX_HIP(hipStreamSynchronize(st));
    for (int q = 0; q < nb; ++q)
      if (ostat <= (double)hout[q] && ++nrej > nrejc) { significant = false; break; }
  }
  if (!significant) return WCX_OK;
  if (bi == 0) { *ncpt = 1; icpt[0] = bj; }
  else if (bj == n) { *ncpt = 1; icpt[0] = bi; }
  else {
    // two change-points: keep each only if its own edge test is significant
    int k = 0;
    if (edge_pvalue(x, w, 0, bi, bj) <= P.alpha) icpt[k++] = bi;
    if (edge_pvalue(x, w, bi, bj, n) <= P.alpha) icpt[k++] = bj;
    *ncpt = k;
  }
  return WCX_OK;
}

// ------------------------------------------------------------------ a17 segment z
// Segment z in two steps so that long segments (a whole chromosome = 16 k bins at 15 kb) do not
// serialise on one workgroup: per-chunk partial weighted sums of every null column (chunks of
// <= SZ_CHUNK bins, accumulated in bin order), then one workgroup per segment adds its chunks in
// order -- deterministic, the same association for every launch.
constexpr int SZ_CHUNK = 256;

__global__ __launch_bounds__(128) void k_segz_partial(const double *__restrict__ r,
                                                      const double *__restrict__ w,
                                                      const double *__restrict__ nr, int m,
                                                      const int64_t *__restrict__ cb0,
                                                      const int64_t *__restrict__ cb1,
                                                      double *__restrict__ pnum,
                                                      double *__restrict__ pden,
                                                      int *__restrict__ pany) {
  const int c = blockIdx.x;
  const int j = threadIdx.x;
  if (j >= m) return;
  double num = 0.0, den = 0.0;
  int any = 0;
  for (int64_t b = cb0[c]; b < cb1[c]; ++b) {
    if (r[b] == 0.0) continue;                      // overall_tools.py:98-100
    const double v = nr[b * m + j];
    if (fabs(v) < HUGE_VAL) { num += v * w[b]; den += w[b]; any = 1; }   // :101-110
  }
  pnum[(int64_t)c * m + j] = num;
  pden[(int64_t)c * m + j] = den;
  pany[(int64_t)c * m + j] = any;
}

__global__ __launch_bounds__(128) void k_segment_z(const double *__restrict__ pnum,
                                                   const double *__restrict__ pden,
                                                   const int *__restrict__ pany, int m,
                                                   const int *__restrict__ chunk0,
                                                   const double *__restrict__ seg_r, int n_seg,
                                                   double *__restrict__ out_z,
                                                   double *__restrict__ out_nnull) {
  const int s = blockIdx.x;
  if (s >= n_seg) return;
  __shared__ double avg[128];
  const int j = threadIdx.x;
  double a = __builtin_nan("");
  if (j < m) {
    double num = 0.0, den = 0.0;
    bool any = false;
    for (int c = chunk0[s]; c < chunk0[s + 1]; ++c) {
      num += pnum[(int64_t)c * m + j];
      den += pden[(int64_t)c * m + j];
      any |= pany[(int64_t)c * m + j] != 0;
    }
    if (any) a = num / den;
  }
  avg[j] = a;
  __syncthreads();
  if (j == 0) {
    double sum = 0.0;
    int cnt = 0;
    for (int q = 0; q < m; ++q) if (fabs(avg[q]) < HUGE_VAL) { sum += avg[q]; ++cnt; }   // :111
    double z = __builtin_nan("");
    if (cnt > 0) {
      const double mean = sum / cnt;
      double ss = 0.0;
      for (int q = 0; q < m; ++q) if (fabs(avg[q]) < HUGE_VAL) { const double e = avg[q] - mean; ss += e * e; }
      const double sd = sqrt(ss / cnt);                // np.ma.std: population
      z = (seg_r[s] - mean) / sd;                      // :113
      if (z == z) { z = z < 1000.0 ? z : 1000.0; z = z > -1000.0 ? z : -1000.0; }   // :114-115
    }
    out_z[s] = z;
    if (out_nnull) out_nnull[s] = (double)cnt;
  }
}

}  // namespace

extern "C" {

int wcx_cbs(wcx_ctx *ctx, const double *r, const double *w, const int64_t *chr_off, int n_chr,
            double alpha, int64_t binsize, uint64_t seed, double *out_seg, int cap,
            int *out_count) {
  WCX_ARG(ctx && r && w && chr_off && out_seg && out_count, "NULL argument");
  WCX_ARG(n_chr > 0 && alpha > 0 && alpha <= 1 && binsize > 0 && cap >= 0, "bad parameters");
  WCX_HIP(hipSetDevice(ctx->device));
  int64_t maxn = 0;
  for (int c = 0; c < n_chr; ++c) maxn = std::max<int64_t>(maxn, chr_off[c + 1] - chr_off[c]);
  if (maxn > 32768) {
    wcx_set_error("wcx_cbs: %lld bins in one chromosome (max 32768 per chromosome)", (long long)maxn);
    return WCX_ERR_UNSUPPORTED;
  }
  CbsWork wk;
  const size_t nb = (size_t)maxn + 1;
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, nb * (8 + 8 + 4 + 4 + 4) + 4096 + 256 * 4 + 4096, &scr);
  if (rc) return rc;
  char *p = reinterpret_cast<char *>(scr);
  wk.dS = reinterpret_cast<double *>(p); p += nb * 8;
  wk.dWp = reinterpret_cast<double *>(p); p += nb * 8;
  wk.dy = reinterpret_cast<float *>(p); p += nb * 4;
  wk.drw = reinterpret_cast<float *>(p); p += nb * 4;
  wk.dWpf = reinterpret_cast<float *>(p); p += nb * 4;
  p = reinterpret_cast<char *>(((uintptr_t)p + 255) & ~(uintptr_t)255);
  wk.dbest = reinterpret_cast<ArcBest *>(p); p += 256;
  wk.dout = reinterpret_cast<float *>(p); p += 256 * 4;
  wk.dtx = reinterpret_cast<double *>(p); p += 1024;
  wk.dtnu = reinterpret_cast<double *>(p);
  CbsParams P;
  P.alpha = alpha;
  P.seed = seed;
  rc = wcx_timer_begin(ctx, "cbs");
  if (rc) return rc;
  const int na_limit = (int)(1.0 / ((double)binsize / 2000000.0));   // CBS.R:95 as.integer((binsize/2e6)^-1)
  int count = 0;
  unsigned long long test_id = 0;
  for (int c = 0; c < n_chr; ++c) {
    const int64_t o = chr_off[c];
    const int nall = (int)(chr_off[c + 1] - o);
    // CBS.R:41-42: ratio == 0 -> NA ; weight == 0 -> 1 ;  DNAcopy drops the NA rows itself
    std::vector<double> x, ww;
    std::vector<int> pos;   // 1-based bin index within the chromosome (CBS.R:49)
    for (int i = 0; i < nall; ++i) {
      const double v = r[o + i];
      if (v == 0.0 || v != v) continue;
      x.push_back(v);
      ww.push_back(w[o + i] == 0.0 ? 1.0 : w[o + i]);
      pos.push_back(i + 1);
    }
    const int n = (int)x.size();
    if (n == 0) continue;   // CBS.R:56-63 all-NA chromosome
    // recursive binary segmentation (DNAcopy changepoints(): stack of segment ends)
    std::vector<int> seg_end = {0, n}, change_loc;
    while (seg_end.size() > 1) {
      const int k = (int)seg_end.size();
      const int lo = seg_end[k - 2], hi = seg_end[k - 1];
      int ncpt = 0, icpt[2] = {0, 0};
      if (hi - lo >= 2 * P.minw) {
        rc = cbs_test(ctx, wk, x.data() + lo, ww.data() + lo, hi - lo, P, ++test_id, &ncpt, icpt);
        if (rc) return rc;
      }
      if (ncpt == 0) { change_loc.push_back(hi); seg_end.pop_back(); }
      else if (ncpt == 1) { seg_end.insert(seg_end.end() - 1, lo + icpt[0]); }
      else { seg_end.insert(seg_end.end() - 1, lo + icpt[0]); seg_end.insert(seg_end.end() - 1, lo + icpt[1]); }
    }
    std::sort(change_loc.begin(), change_loc.end());
    // segments in data index space -> 1-based loc.start / loc.end in bin coordinates
    int prev = 0;
    for (int e : change_loc) {
      const int s1 = pos[prev], e1 = pos[e - 1];   // inclusive, 1-based
      prev = e;
      // CBS.R:84-113 split over long NA runs; pieces start AT the last NA bin (reference quirk)
      std::vector<int> start_pos, end_pos;
      for (int b = s1; b < e1; ++b) {   // b, b+1 are 1-based bins inside the segment
        const bool na0 = (r[o + b - 1] == 0.0 || r[o + b - 1] != r[o + b - 1]);
        const bool na1 = (r[o + b] == 0.0 || r[o + b] != r[o + b]);
        if (!na0 && na1) start_pos.push_back(b);
        if (na0 && !na1) end_pos.push_back(b);
      }
      const size_t mm = std::min(start_pos.size(), end_pos.size());
      std::vector<int> inv_s = {s1}, inv_e;
      for (size_t q = 0; q < mm; ++q)
        if (end_pos[q] - start_pos[q] > na_limit) { inv_e.push_back(start_pos[q]); inv_s.push_back(end_pos[q]); }
      inv_e.push_back(e1);
      for (size_t q = 0; q < inv_s.size(); ++q) {
        const int a = inv_s[q], b = inv_e[q];
        if (!(b - a > 0)) continue;                 // CBS.R:103
        double num = 0, den = 0;                    // CBS.R:122-127 weighted.mean(na.rm=T)
        for (int t = a; t <= b; ++t) {
          const double v = r[o + t - 1];
          if (v == 0.0 || v != v) continue;
          const double wt = w[o + t - 1] == 0.0 ? 1.0 : w[o + t - 1];
          num += v * wt; den += wt;
        }
        if (count < cap) {
          out_seg[count * 4 + 0] = c;
          out_seg[count * 4 + 1] = a - 1;           // CBS.R:129
          out_seg[count * 4 + 2] = b;
          out_seg[count * 4 + 3] = den > 0 ? num / den : __builtin_nan("");
        }
        ++count;
      }
    }
  }
  rc = wcx_timer_end(ctx, "cbs");
  if (rc) return rc;
  *out_count = count;
  if (count > cap) {
    wcx_set_error("wcx_cbs: %d segments exceed the caller's capacity %d", count, cap);
    return WCX_ERR_ARG;
  }
  return WCX_OK;
}

int wcx_set_null_matrix(wcx_ctx *ctx, const double *nr, int64_t n_bins, int m) {
  WCX_ARG(ctx, "ctx is NULL");
  WCX_HIP(hipSetDevice(ctx->device));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->d_nullm) { WCX_HIP(hipFree(ctx->d_nullm)); ctx->d_nullm = nullptr; }
  ctx->nullm_bins = 0;
  ctx->nullm_m = 0;
  if (!nr) return WCX_OK;
  WCX_ARG(n_bins > 0 && m > 0 && m <= 128, "bad sizes (m <= 128)");
  const size_t bytes = (size_t)n_bins * m * 8;
  if (hipMalloc(reinterpret_cast<void **>(&ctx->d_nullm), bytes) != hipSuccess) {
    wcx_set_error("hipMalloc(%zu) for the null-ratio matrix failed", bytes);
    return WCX_ERR_NOMEM;
  }
  WCX_HIP(hipMemcpyAsync(ctx->d_nullm, nr, bytes, hipMemcpyHostToDevice, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  ctx->nullm_bins = n_bins;
  ctx->nullm_m = m;
  return WCX_OK;
}

int wcx_segment_z(wcx_ctx *ctx, const double *r, const double *w, const double *nr, int m,
                  const int64_t *chr_off, int n_chr, const double *seg, int n_seg, double *out_z,
                  double *out_nnull) {
  WCX_ARG(ctx && r && w && chr_off && seg && out_z, "NULL argument");
  WCX_ARG(n_chr > 0 && n_seg >= 0, "bad sizes");
  if (n_seg == 0) return WCX_OK;
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t nb = chr_off[n_chr];
  const bool attached = (nr == nullptr);
  if (attached) {
    WCX_ARG(ctx->d_nullm && ctx->nullm_bins == nb, "no attached null matrix of matching size");
    m = ctx->nullm_m;
  }
  WCX_ARG(m > 0 && m <= 128, "bad sizes (m <= 128)");
  std::vector<int64_t> b0(n_seg), b1(n_seg);
  std::vector<double> sr(n_seg);
  for (int s = 0; s < n_seg; ++s) {
    const int c = (int)seg[s * 4];
    WCX_ARG(c >= 0 && c < n_chr, "segment chromosome out of range");
    b0[s] = chr_off[c] + (int64_t)seg[s * 4 + 1];
    b1[s] = chr_off[c] + (int64_t)seg[s * 4 + 2];
    WCX_ARG(b0[s] >= chr_off[c] && b1[s] <= chr_off[c + 1] && b0[s] <= b1[s], "segment out of range");
    sr[s] = seg[s * 4 + 3];
  }
  // chunk table: segment s owns chunks [chunk0[s], chunk0[s+1])
  std::vector<int64_t> cb0, cb1;
  std::vector<int> chunk0(n_seg + 1);
  for (int s = 0; s < n_seg; ++s) {
    chunk0[s] = (int)cb0.size();
    for (int64_t b = b0[s]; b < b1[s]; b += SZ_CHUNK) {
      cb0.push_back(b);
      cb1.push_back(std::min<int64_t>(b + SZ_CHUNK, b1[s]));
    }
  }
  chunk0[n_seg] = (int)cb0.size();
  const size_t n_chunks = cb0.size();
  const size_t vb = (size_t)nb * 8, nrb = attached ? 0 : (size_t)nb * m * 8, sb = (size_t)n_seg * 8;
  const size_t cb = (n_chunks + 1) * 8, pb = (n_chunks + 1) * (size_t)m * 8;
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, 2 * vb + nrb + 4 * sb + 2 * cb + 3 * pb + 4096, &scr);
  if (rc) return rc;
  char *p = reinterpret_cast<char *>(scr);
  double *dr = (double *)p; p += vb;
  double *dw = (double *)p; p += vb;
  double *dnr = (double *)p; p += nrb;
  double *dsr = (double *)p; p += sb;
  double *dz = (double *)p; p += sb;
  double *dn = (double *)p; p += sb;
  int *dchunk0 = (int *)p; p += sb + 8;
  int64_t *dcb0 = (int64_t *)p; p += cb;
  int64_t *dcb1 = (int64_t *)p; p += cb;
  double *dpnum = (double *)p; p += pb;
  double *dpden = (double *)p; p += pb;
  int *dpany = (int *)p;
  hipStream_t st = ctx->stream;
  WCX_HIP(hipMemcpyAsync(dr, r, vb, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(dw, w, vb, hipMemcpyHostToDevice, st));
  if (!attached) WCX_HIP(hipMemcpyAsync(dnr, nr, nrb, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(dsr, sr.data(), sb, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(dchunk0, chunk0.data(), (size_t)(n_seg + 1) * 4, hipMemcpyHostToDevice, st));
  if (n_chunks) {
    WCX_HIP(hipMemcpyAsync(dcb0, cb0.data(), n_chunks * 8, hipMemcpyHostToDevice, st));
    WCX_HIP(hipMemcpyAsync(dcb1, cb1.data(), n_chunks * 8, hipMemcpyHostToDevice, st));
  }
  rc = wcx_timer_begin(ctx, "segment_z");
  if (rc) return rc;
  if (n_chunks)
    k_segz_partial<<<(unsigned)n_chunks, 128, 0, st>>>(dr, dw, attached ? ctx->d_nullm : dnr, m, dcb0,
                                                       dcb1, dpnum, dpden, dpany);
  k_segment_z<<<n_seg, 128, 0, st>>>(dpnum, dpden, dpany, m, dchunk0, dsr, n_seg, dz, dn);
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "segment_z");
  if (rc) return rc;
  WCX_HIP(hipMemcpyAsync(out_z, dz, sb, hipMemcpyDeviceToHost, st));
  if (out_nnull) WCX_HIP(hipMemcpyAsync(out_nnull, dn, sb, hipMemcpyDeviceToHost, st));
  WCX_HIP(hipStreamSynchronize(st));
  return WCX_OK;
}

}  // extern "C"
