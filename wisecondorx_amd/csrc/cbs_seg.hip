// Circular binary segmentation on the GPU.
//
// Replaces predict_tools.exec_cbs -> Rscript include/CBS.R -> DNAcopy::segment
// (predict_tools.py:242-257, CBS.R:21-132).  The reference-owned code around the DNAcopy call
// (NA masking, weight fix-up CBS.R:41-42, dropping all-NA chromosomes :56-63, splitting segments
// over long NA runs :84-113, weighted re-mean :122-127, 0-based starts :129) is reproduced
// exactly.  The segmentation itself lives in Bioconductor DNAcopy 1.76.0 (conda.yml:14), which is
// NOT part of the reference repository and cannot run here (no R): PARITY UNPINNED.  It is
// restated from the published algorithm with DNAcopy's defaults (Olshen et al. 2004;
// Venkatraman & Olshen 2007) and from the structure of DNAcopy's changepoints code as recalled:
//   * weighted max-arc statistic over all arcs with >= min.width = 2 points on each side;
//   * n > nmin = 200: "hybrid" p-value = Siegmund tail approximation for the arcs longer than
//     kmax = 25 + permutation reference distribution (nperm = 10 000) of the short-arc maximum;
//     otherwise the full permutation distribution; the observed statistic is compared as
//     0.99999 * ostat; the permutations stop as soon as the exceedance budget
//     floor(p2 * nperm) is spent (not significant);
//   * an interior arc yields two change-points, each kept only if its own two-sample permutation
//     test (weighted means of the shorter side, nperm permutations; skipped as "clearly
//     significant" when t^2 > 25 with >= 10 points) has p <= alpha; undo.splits = "none".
// Deviations that make breakpoint parity impossible even with R available: the permutation
// stream (counter-based hash here, R's Mersenne-Twister there) and DNAcopy's sequential stopping
// boundary for SIGNIFICANT tests (getbdry; at alpha = 1e-4 it can only save the last ~13 % of the
// permutations and never changes a decision by more than its error budget eta = 0.05 allows).
//
// GPU mapping -- LEVEL-SYNCHRONOUS and BATCHED: all chromosomes of all samples of a call advance
// together.  Per round: (1) one launch prepares every active segment (weighted centring, fp64
// prefix sums), (2) one launch finds every segment's best arc (fp64, striped over many
// workgroups), (3) one launch evaluates the tail probabilities, ONE device->host copy of the
// per-segment records, (4) one launch runs the permutations of every segment that needs them
// (one workgroup per permutation: a keyed Feistel bijection of [0, n) = the random permutation,
// weighted re-centring, prefix scan, short-arc maximum; a per-segment counter lets later
// workgroups exit as soon as the budget is spent), one copy of the counters, (5) the same kernel
// in "edge" mode for the two-sample tests.  The host only keeps the segment stack.
// Compute/latency bound; reported as wall-clock per sample.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <functional>
#include <vector>
#include <thread>

#include "wave_sort.h"
#include "wcx_common.h"

namespace {

constexpr int NTP = 1024;
constexpr int KMAXC = 25;   // DNAcopy's kmax (short-arc limit of the hybrid p-value), compile-time here
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// One active segment [lo, hi) of the concatenated (NA-free) data of all series of the call.
struct SegIn {
  int64_t lo;       // first element (index into the concatenated arrays)
  int32_t n;        // elements
  int32_t hybrid;   // n > nmin
};
// What the host needs back to decide.
struct SegOut {
  double tss, W, ostat, pval1;
  int32_t bi, bj;   // best arc (bi, bj], 0 <= bi < bj <= n
  int32_t valid, pad;
};
// A permutation job: the segmentation test of a segment, or one two-sample edge test.
struct PermJob {
  int64_t lo;        // first element of the (sub)series
  int32_t n;
  int32_t mode;      // 0 = max short-arc statistic (hybrid), 1 = max over all arcs, 2 = edge test
  int32_t m1, first; // edge test: size of the shorter side; 1 = it is the first m1 positions
  int32_t nrejc, pad;
  double ostat;      // threshold (already scaled by 0.99999)
  unsigned long long seed;
  int64_t qoff;      // mode 0: this job's block of the arc-weight table (k_cbs_arcweights)
};

// ---- (1) prepare: weighted centring + prefix sums -------------------------------------------
// S[lo + i] = sum_{t < i} w (x - mean), Wp likewise (both arrays have one slot per element + the
// segment's end slot lives at index hi of arrays sized N + 1: segments are disjoint and ordered,
// the end slot of one is the start slot of the next, rewritten by whoever runs later -- so every
// segment keeps its OWN end value in SegOut.W / a separate tail array).
__global__ __launch_bounds__(NTP) void k_cbs_prepare(const double *__restrict__ x,
                                                     const double *__restrict__ w,
                                                     const SegIn *__restrict__ segs,
                                                     double *__restrict__ S, double *__restrict__ Wp,
                                                     float *__restrict__ y, float *__restrict__ rw,
                                                     float *__restrict__ Wpf,
                                                     SegOut *__restrict__ out) {
  const SegIn sg = segs[blockIdx.x];
  const int n = sg.n, tid = threadIdx.x;
  const double *xs = x + sg.lo, *ws = w + sg.lo;
  __shared__ double red[2][NTP / 64];
  __shared__ double tot[2][NTP];
  auto block_sum2 = [&](double a, double b, double &ra, double &rb) {
    a = wcx::wave_sum(a);
    b = wcx::wave_sum(b);
    __syncthreads();
    if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = b; }
    __syncthreads();
    ra = 0.0; rb = 0.0;
    for (int q = 0; q < NTP / 64; ++q) { ra += red[0][q]; rb += red[1][q]; }
  };
  double sw = 0.0, sx = 0.0;
  for (int i = tid; i < n; i += NTP) { sw += ws[i]; sx += ws[i] * xs[i]; }
  double W, SX;
  block_sum2(sw, sx, W, SX);
  const double mean = SX / W;
  // chunked scan: thread t owns elements [t*chunk, (t+1)*chunk)
  const int chunk = (n + NTP - 1) / NTP;
  const int i0 = tid * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
  double ls = 0.0, lw = 0.0, lt = 0.0;
  for (int i = i0; i < i1; ++i) {
    const double c = xs[i] - mean;
    ls += ws[i] * c;
    lw += ws[i];
    lt += ws[i] * c * c;
  }
  tot[0][tid] = ls;
  tot[1][tid] = lw;
  __syncthreads();
  for (int off = 1; off < NTP; off <<= 1) {
    const double a = tid >= off ? tot[0][tid - off] : 0.0, b = tid >= off ? tot[1][tid - off] : 0.0;
    __syncthreads();
    tot[0][tid] += a;
    tot[1][tid] += b;
    __syncthreads();
  }
  double rs = tot[0][tid] - ls, rwt = tot[1][tid] - lw;     // exclusive prefix of this chunk
  for (int i = i0; i < i1; ++i) {
    S[sg.lo + i] = rs;
    Wp[sg.lo + i] = rwt;
    Wpf[sg.lo + i] = (float)rwt;
    const double c = xs[i] - mean, r = sqrt(ws[i]);
    y[sg.lo + i] = (float)(c * r);
    rw[sg.lo + i] = (float)r;
    rs += ws[i] * c;
    rwt += ws[i];
  }
  double tss, dummy;
  block_sum2(lt, 0.0, tss, dummy);
  if (tid == 0) {
    SegOut o;
    o.tss = tss; o.W = W; o.ostat = 0.0; o.pval1 = 0.0; o.bi = 0; o.bj = 0; o.valid = 0; o.pad = 0;
    out[blockIdx.x] = o;
  }
}

// ---- (2) observed statistic: best arc, fp64, striped ----------------------------------------
struct ArcItem { int32_t seg, i0, i1, pad; };
struct ArcBest { double b; int32_t i, j; };
constexpr int ARC_ROWS = 512;          // rows per stripe at most (LDS staging of the stripe's rows)

// prefix values at position p of segment (the end slot p = n is W / 0-sum by construction)
__device__ __forceinline__ double seg_S(const double *S, const SegIn &sg, int p) {
  return p < sg.n ? S[sg.lo + p] : 0.0;          // sum of w (x - mean) over the whole segment = 0
}
__device__ __forceinline__ double seg_W(const double *Wp, const SegIn &sg, double W, int p) {
  return p < sg.n ? Wp[sg.lo + p] : W;
}

__global__ __launch_bounds__(256) void k_cbs_arcmax(const double *__restrict__ S,
                                                    const double *__restrict__ Wp,
                                                    const SegIn *__restrict__ segs,
                                                    const SegOut *__restrict__ so,
                                                    const ArcItem *__restrict__ items, int minw,
                                                    ArcBest *__restrict__ best) {
  const ArcItem it = items[blockIdx.x];
  const SegIn sg = segs[it.seg];
  const double W = so[it.seg].W;
  const int n = sg.n;
  double bb = -1.0;
  float bf = -1.f;                      // fp32 image of the running maximum, a little below it
  int bi = 0, bj = 0;
  const float Wf = (float)W;
  // the stripe's rows sit in LDS (broadcast reads); every thread walks its own columns j and meets
  // all rows with S_j / W_j in registers: the prefix arrays are read once per stripe, not per row
  __shared__ double rs[ARC_ROWS], rwp[ARC_ROWS];
  const int nrow = it.i1 - it.i0;
  for (int q = threadIdx.x; q < nrow; q += 256) {
    rs[q] = seg_S(S, sg, it.i0 + q);
    rwp[q] = seg_W(Wp, sg, W, it.i0 + q);
  }
  __syncthreads();
  // columns that can pair with any row of the stripe: j in [i0 + minw, min(n, i1 - 1 + n - minw)]
  const int jlo = it.i0 + minw, jhi_all = it.i1 - 1 + n - minw < n ? it.i1 - 1 + n - minw : n;
  for (int j = jlo + threadIdx.x; j <= jhi_all; j += 256) {
    const double sj = seg_S(S, sg, j), wj = seg_W(Wp, sg, W, j);
    // rows i with minw <= j - i <= n - minw
    int qlo = j - (n - minw) - it.i0, qhi = j - minw - it.i0;
    qlo = qlo < 0 ? 0 : qlo;
    qhi = qhi > nrow - 1 ? nrow - 1 : qhi;
    for (int q = qlo; q <= qhi; ++q) {
      const double d = sj - rs[q], wa = wj - rwp[q];
      // cheap fp32 screen (relative error < 1e-6): only candidates within 4e-6 of the running
      // maximum get the two fp64 divisions
      const float df = (float)d, waf = (float)wa;
      const float b32 = df * df * Wf * __builtin_amdgcn_rcpf(waf * (Wf - waf));
      if (b32 >= bf) {
        const double b = d * d / (wa * (W - wa) / W);
        const int i = it.i0 + q;
        if (b > bb || (b == bb && (i < bi || (i == bi && j < bj)))) {
          bb = b; bi = i; bj = j; bf = (float)b * 0.999996f;
        }
      }
    }
  }
  // workgroup reduction, ties -> smallest (i, j): deterministic
  __shared__ double sb[256];
  __shared__ int si_[256], sj_[256];
  sb[threadIdx.x] = bb; si_[threadIdx.x] = bi; sj_[threadIdx.x] = bj;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const int o = threadIdx.x + off;
      const bool take = sb[o] > sb[threadIdx.x] ||
                        (sb[o] == sb[threadIdx.x] && (si_[o] < si_[threadIdx.x] ||
                                                      (si_[o] == si_[threadIdx.x] && sj_[o] < sj_[threadIdx.x])));
      if (take) { sb[threadIdx.x] = sb[o]; si_[threadIdx.x] = si_[o]; sj_[threadIdx.x] = sj_[o]; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { best[blockIdx.x].b = sb[0]; best[blockIdx.x].i = si_[0]; best[blockIdx.x].j = sj_[0]; }
}

// per segment: reduce its stripes (items [first[s], first[s+1])), t^2 of the best arc, and the
// grid of x values of the tail-probability integral
__global__ void k_cbs_arcfinish(const ArcBest *__restrict__ best, const int *__restrict__ first,
                                const SegIn *__restrict__ segs, SegOut *__restrict__ so, int kmax,
                                int ngrid, double *__restrict__ tx) {
  const int s = blockIdx.x;
  // best stripe of the segment; ties -> the first stripe (stripes ascend in i): deterministic
  __shared__ double rb[128];
  __shared__ int rq[128];
  {
    double bb = -1.0;
    int bq = 0x7fffffff;
    for (int q = first[s] + threadIdx.x; q < first[s + 1]; q += blockDim.x)
      if (best[q].b > bb) { bb = best[q].b; bq = q; }
    rb[threadIdx.x] = bb; rq[threadIdx.x] = bq;
    __syncthreads();
    for (int off = 64; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) {
        const int o = threadIdx.x + off;
        if (rb[o] > rb[threadIdx.x] || (rb[o] == rb[threadIdx.x] && rq[o] < rq[threadIdx.x])) {
          rb[threadIdx.x] = rb[o]; rq[threadIdx.x] = rq[o];
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    const double bb = rb[0];
    int bi = 0, bj = 0;
    if (bb > 0.0) { bi = best[rq[0]].i; bj = best[rq[0]].j; }
    SegOut o = so[s];
    const int n = segs[s].n;
    if (bb > 0.0 && o.tss > 0.0) {
      o.ostat = bb / ((o.tss - bb) / (n - 2.0));     // t^2 of the best arc
      o.bi = bi; o.bj = bj; o.valid = 1;
    }
    so[s] = o;
  }
  __syncthreads();
  // tail probability grid (Siegmund approximation; DNAcopy tailp): x_i = b / sqrt(m t (1 - t))
  const SegOut o = so[s];
  const int n = segs[s].n;
  if (!o.valid || !segs[s].hybrid) return;
  const double delta = (kmax + 1.0) / n;
  const double dincr = (0.5 - delta) / ngrid;
  const double bsqrtm = sqrt(o.ostat) / sqrt((double)n);
  for (int i = threadIdx.x; i < ngrid; i += blockDim.x) {
    const double t = 0.5 - 0.5 * dincr - i * dincr;
    tx[(int64_t)s * ngrid + i] = bsqrtm / sqrt(t * (1.0 - t));
  }
}

// nu(x) series: one workgroup per (grid point, segment), threads over the terms
//   ln nu = ln 2 - 2 ln x - 2 sum_{k>=1} Phi(-x sqrt(k)/2) / k      (terms vanish once x sqrt(k)/2 > 8.5)
__global__ __launch_bounds__(256) void k_nu_series(const double *__restrict__ xs,
                                                   const SegIn *__restrict__ segs,
                                                   const SegOut *__restrict__ so, int ngrid,
                                                   double *__restrict__ out) {
  const int s = blockIdx.y;
  if (!so[s].valid || !segs[s].hybrid) return;
  const double x = xs[(int64_t)s * ngrid + blockIdx.x];
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
    const double sum = red[0] + red[1] + red[2] + red[3];
    out[(int64_t)s * ngrid + blockIdx.x] = x > 0.01 ? exp(log(2.0) - 2.0 * log(x) - 2.0 * sum)
                                                     : exp(-0.583 * x);
  }
}

__device__ __forceinline__ double it1tsq(double x, double a) {   // integral of 1/(t(1-t))^2 over [x, x+a]
  double y = x + a - 0.5;
  double r = 8.0 * y / (1.0 - 4.0 * y * y) + 2.0 * log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
  y = x - 0.5;
  r -= 8.0 * y / (1.0 - 4.0 * y * y) + 2.0 * log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
  return r;
}

// P(max over arcs with delta <= length/m <= 1-delta of the CBS statistic >= b), Gaussian null
__global__ void k_cbs_tailp(const double *__restrict__ nu, const SegIn *__restrict__ segs,
                            SegOut *__restrict__ so, int kmax, int ngrid) {
  const int s = blockIdx.x;
  if (threadIdx.x != 0 || !so[s].valid || !segs[s].hybrid) return;
  const int n = segs[s].n;
  const double b = sqrt(so[s].ostat);
  const double delta = (kmax + 1.0) / n;
  const double dincr = (0.5 - delta) / ngrid;
  double acc = 0.0, tl = 0.5 - dincr;
  for (int i = 0; i < ngrid; ++i) {
    const double v = nu[(int64_t)s * ngrid + i];
    acc += v * v * it1tsq(tl, dincr);
    tl -= dincr;
  }
  so[s].pval1 = 9.973557e-2 * b * b * b * exp(-b * b / 2.0) * acc;
}

// Exclusive scan over the NTP threads of a workgroup: wave scan + one LDS hop (2 barriers).
__device__ __forceinline__ float block_excl_scan_f32(float v, float *ws) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float incl = v;     // wave inclusive scan by shuffles (fp32 adds in a fixed order: deterministic)
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  __syncthreads();
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  float base = 0.f;
  for (int q = 0; q < wave; ++q) base += ws[q];
  return base + incl - v;
}

// ---- (4) permutations ------------------------------------------------------------------------
// One workgroup per permutation of one job.  y = centred residual * sqrt(w) (exchangeable under
// H0), rw = sqrt(w), Wpf = prefix sums of w.  A permutation whose statistic reaches the job's
// threshold bumps nrej[job]; once nrej > nrejc the remaining workgroups of the job return at once.
// BIG: the sort buffer lives in global scratch (n > LDS capacity; slot = blockIdx.x, the grid is
// then limited and strides over the permutations).
// Arc weights of a hybrid permutation job: q[a - 2][i] = W / (w_a (W - w_a)) for the arc (i, i + a],
// a = 2 .. KMAXC, 0 where the arc does not exist.  The weights are not permuted (DNAcopy permutes
// the data under fixed weights), so this is shared by all permutations of the job.
__global__ void k_cbs_arcweights(const float *__restrict__ rw_all, const float *__restrict__ Wpf_all,
                                 const PermJob *__restrict__ jobs, int minw, float *__restrict__ qtab) {
  const PermJob jb = jobs[blockIdx.y];
  if (jb.mode != 0) return;
  const int n = jb.n;
  const int npad = (n + NTP - 1) / NTP * NTP;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  const float *rw = rw_all + jb.lo, *Wpf = Wpf_all + jb.lo;
  const float W = Wpf[n - 1] + rw[n - 1] * rw[n - 1];
  const int amax_all = n - minw;
  const int a_hi = KMAXC < amax_all ? KMAXC : amax_all;
  float *q = qtab + jb.qoff;
  const float w0 = i < n ? Wpf[i] : W;
  for (int a = 2; a <= KMAXC; ++a) {
    float v = 0.f;
    if (a >= minw && a <= a_hi && i + a <= n) {
      const float wa = (i + a < n ? Wpf[i + a] : W) - w0;
      v = W / (wa * (W - wa));
    }
    q[(size_t)(a - 2) * npad + i] = v;
  }
}

template <bool BIG>
__global__ __launch_bounds__(NTP) void k_cbs_perm(const float *__restrict__ y_all,
                                                  const float *__restrict__ rw_all,
                                                  const float *__restrict__ Wpf_all,
                                                  const PermJob *__restrict__ jobs, int nperm,
                                                  int npad_max, int minw, int kmax,
                                                  const float *__restrict__ qtab,
                                                  unsigned int *__restrict__ big_scr,
                                                  unsigned int *__restrict__ nrej) {
  extern __shared__ unsigned int lds[];
  __shared__ float red[NTP / 64];
  __shared__ float tot[NTP];
  const int tid = threadIdx.x;
  const PermJob jb = jobs[blockIdx.y];
  const int n = jb.n;
  const int npad = (n + NTP - 1) / NTP * NTP;
  const float *y = y_all + jb.lo, *rw = rw_all + jb.lo, *Wpf = Wpf_all + jb.lo;
  unsigned int *sk = BIG ? big_scr + (size_t)blockIdx.x * npad_max : lds;   // [npad]
  const float W = Wpf[n - 1] + rw[n - 1] * rw[n - 1];
  auto block_sum = [&](float v) {
    v = (float)wcx::wave_sum((double)v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int q = 0; q < NTP / 64; ++q) t += red[q];
    return t;
  };
  auto Wat = [&](int i) { return i < n ? Wpf[i] : W; };
  // Random permutation WITHOUT a sort: a keyed bijection of [0, n).  Four Feistel rounds on
  // Z_a x Z_a, a = ceil(sqrt(n)) (so the domain a*a exceeds n by < 2 sqrt(n) + 1 and the
  // cycle walk back into [0, n) almost never iterates); round function = murmur3's 32-bit
  // finaliser of (half + round key), mapped to [0, a) by a high multiply.  The null distribution
  // of the short-arc maximum under these permutations is indistinguishable from numpy's shuffles
  // (two-sample KS on 4000 + 4000 permutations, p = 0.36; 3 rounds already pass).
  if (tid < 32) sk[npad + tid] = 0u;     // the arc loop reads up to 32 slots past the scan
  unsigned int fa = (unsigned int)sqrtf((float)n);
  while ((unsigned long long)fa * fa < (unsigned long long)n) ++fa;
  while (fa > 1 && (unsigned long long)(fa - 1) * (fa - 1) >= (unsigned long long)n) --fa;
  const float inv_fa = 1.0f / (float)fa;
  __shared__ unsigned int s_spent;
  for (int p = blockIdx.x; p < nperm; p += gridDim.x) {
    __syncthreads();
    if (tid == 0) s_spent = atomicAdd(&nrej[blockIdx.y], 0u) > (unsigned int)jb.nrejc ? 1u : 0u;
    __syncthreads();
    if (s_spent) return;                                   // budget spent (uniform decision)
    const unsigned long long s0 = mix64(jb.seed ^ ((unsigned long long)p * 0xd1342543de82ef95ull));
    const unsigned long long s1 = mix64(s0 + 1ull);
    const unsigned int rk[4] = {(unsigned int)s0, (unsigned int)(s0 >> 32), (unsigned int)s1,
                                (unsigned int)(s1 >> 32)};
    auto perm_at = [&](int i) {
      unsigned int x = (unsigned int)i;
      do {
        unsigned int L = (unsigned int)((float)x * inv_fa);
        int R = (int)(x - L * fa);
        if (R < 0) { --L; R += (int)fa; }
        else if (R >= (int)fa) { ++L; R -= (int)fa; }
        unsigned int Rr = (unsigned int)R;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          unsigned int h = Rr + rk[r];
          h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
          unsigned int sN = L + __umulhi(h, fa);
          sN = sN >= fa ? sN - fa : sN;
          L = Rr; Rr = sN;
        }
        x = L * fa + Rr;
      } while (x >= (unsigned int)n);
      return (int)x;
    };
    bool exceed;
    if (jb.mode == 2) {
      // two-sample edge test: |weighted mean of the shorter side| of the permuted series (the
      // series is centred: the overall weighted mean is 0 up to the permutation's reweighting)
      // (y is centred on the mean of the segment the sub-series was cut from: re-centre on the
      // sub-series' own weighted mean c first, y' = y - c rw)
      float cw = 0.f, ws = 0.f;
      for (int i = tid; i < n; i += NTP) { cw += rw[i] * y[i]; ws += rw[i] * rw[i]; }
      const float Wsub = block_sum(ws);
      const float c = block_sum(cw) / Wsub;
      float part = 0.f, w1 = 0.f, all = 0.f;
      for (int i = tid; i < n; i += NTP) {
        const int src = perm_at(i);
        const float v = rw[i] * (y[src] - c * rw[src]);    // w_i * (permuted value at position i)
        all += v;
        const bool in1 = jb.first ? i < jb.m1 : i >= n - jb.m1;
        if (in1) { part += v; w1 += rw[i] * rw[i]; }
      }
      const float s_all = block_sum(all), s1 = block_sum(part), W1 = block_sum(w1);
      const float xbar = s_all / Wsub;
      exceed = (double)fabsf(s1 / W1 - xbar) >= jb.ostat;
    } else {
      // weighted mean of the permuted series: sum_i w_i (y_pi(i) / rw_i) = sum_i rw_i y_pi(i)
      float part = 0.f;
      for (int i = tid; i < n; i += NTP) { sk[i] = __float_as_uint(y[perm_at(i)]); part += rw[i] * __uint_as_float(sk[i]); }
      const float mean = block_sum(part) / W;
      float tssl = 0.f;
      for (int i = tid; i < npad; i += NTP) {
        float cx = 0.f;
        if (i < n) {
          const float r = rw[i];
          const float v = __uint_as_float(sk[i]) * __builtin_amdgcn_rcpf(r) - mean;
          cx = r * r * v;          // w_i v_i
          tssl += cx * v;          // w_i v_i^2
        }
        sk[i] = __float_as_uint(cx);
      }
      const float tss = block_sum(tssl);
      // inclusive prefix scan of sk (as floats): serial chunks + scan of chunk totals
      const int chunk = npad / NTP > 0 ? npad / NTP : 1;
      const int nth = npad / chunk;   // threads that own a chunk
      float run = 0.f;
      if (tid < nth) {
        for (int c = 0; c < chunk; ++c) {
          run += __uint_as_float(sk[tid * chunk + c]);
          sk[tid * chunk + c] = __float_as_uint(run);
        }
      }
      const float base = block_excl_scan_f32(tid < nth ? run : 0.f, tot);
      if (tid < nth && tid > 0) {
        for (int c = 0; c < chunk; ++c)
          sk[tid * chunk + c] = __float_as_uint(__uint_as_float(sk[tid * chunk + c]) + base);
      }
      __syncthreads();
      auto Sx = [&](int i) { return i == 0 ? 0.f : __uint_as_float(sk[i - 1]); };   // S_0 = 0; defined to npad + 32
      float bmax = 0.f;
      const int amax_all = n - minw;
      if (jb.mode == 0) {
        const int a_hi = kmax < amax_all ? kmax : amax_all;
        // short arcs (i, i + a], a = 2 .. 25: a thread owns 8 consecutive start positions, keeps
        // their 33 prefix sums in registers and multiplies d^2 by the job's precomputed
        // permutation-independent arc weight W / (w_a (W - w_a)) (0 for arcs that do not exist);
        // two arcs per packed fp32 instruction
        const float *q = qtab + jb.qoff;
        f32x2 bm = {0.f, 0.f};
        for (int i0 = tid * 8; i0 < n; i0 += 8 * NTP) {
          float sv[8 + KMAXC];
#pragma unroll
          for (int t = 0; t < 8 + KMAXC; ++t) sv[t] = Sx(i0 + t);
          const float *qp = q + i0;
          float4 q0 = *reinterpret_cast<const float4 *>(qp);
          float4 q1 = *reinterpret_cast<const float4 *>(qp + 4);
#pragma unroll
          for (int a = 2; a <= KMAXC; ++a) {
            // next arc length's weights are fetched while this one is evaluated (the barrier keeps
            // the compiler from hoisting all 48 loads to the top: that spilled)
            const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            if (a < KMAXC) {
              qp += npad;
              q0 = *reinterpret_cast<const float4 *>(qp);
              q1 = *reinterpret_cast<const float4 *>(qp + 4);
            }
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              // (scalar differences: a packed subtract would need a second, odd-aligned copy of
              // sv in registers for the odd arc lengths -- that spilled)
              float dx = sv[j + a] - sv[j], dy = sv[j + 1 + a] - sv[j + 1];
#if defined(__HIP_DEVICE_COMPILE__)
              asm volatile("" : "+v"(dx), "+v"(dy));
#endif
              f32x2 d = {dx, dy};
              const f32x2 qq = {qv[j], qv[j + 1]};
              d = d * d * qq;
              bm.x = __builtin_fmaxf(bm.x, d.x);
              bm.y = __builtin_fmaxf(bm.y, d.y);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        bmax = bm.x > bm.y ? bm.x : bm.y;
        const int a_lo = (n - kmax > a_hi + 1) ? n - kmax : a_hi + 1;   // complement is short
        for (int a = a_lo; a <= amax_all; ++a)
          for (int i = tid; i + a <= n; i += NTP) {
            const float d = Sx(i + a) - Sx(i), wa = Wat(i + a) - Wat(i);
            const float b = d * d / (wa * (W - wa) / W);
            bmax = b > bmax ? b : bmax;
          }
      } else {
        for (int i = 0; i < n; ++i)
          for (int j = i + minw + tid; j <= n && n - (j - i) >= minw; j += NTP) {
            const float d = Sx(j) - Sx(i), wa = Wat(j) - Wat(i);
            const float b = d * d / (wa * (W - wa) / W);
            bmax = b > bmax ? b : bmax;
          }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(bmax, m, 64); bmax = o > bmax ? o : bmax; }
      __syncthreads();
      if ((tid & 63) == 0) red[tid >> 6] = bmax;
      __syncthreads();
      float b = 0.f;
      for (int q = 0; q < NTP / 64; ++q) b = red[q] > b ? red[q] : b;
      const double pstat = (double)b / (((double)tss - (double)b) / (double)(n - 2));
      exceed = pstat >= jb.ostat;
    }
    if (tid == 0 && exceed) atomicAdd(&nrej[blockIdx.y], 1u);
    __syncthreads();
  }
}

// Upper bound of the permutation statistic of the hybrid test (the maximum over arcs of at most
// kmax points, or whose complement has at most kmax points) over ALL permutations of the series,
// from O(n) sums and the kmax largest |y|.  With y_j = r_j (x_j - mean), r_j = sqrt(w_j), a
// permutation pi puts y_pi(i) at position i (weights stay): part = sum r_i y_pi(i),
// tss' = sum y^2 - part^2 / W, arc sum d = sum_arc r_i y_pi(i) - (part / W) w_arc.
//   |part| <= sqrt(sum (r_i - rbar)^2 sum y^2) + rbar |sum y|                      (Cauchy-Schwarz)
//   |d|    <= r_max Y_a + |part| a w_max / W,   Y_a = the a largest |y|
//   w_arc (W - w_arc) >= min over the end points of [a w_min, a w_max]             (concave)
// A strongly significant segment (a long aberration) has an observed statistic far above this
// bound; its 10 000 permutations cannot produce a single exceedance and are not run.
static double short_arc_bound(const double *x, const double *w, int n, int minw, int kmax) {
  double W = 0, sx = 0;
  for (int i = 0; i < n; ++i) { W += w[i]; sx += w[i] * x[i]; }
  const double mean = sx / W;
  double sy = 0, syy = 0, sr = 0, wmin = w[0], wmax = w[0];
  std::vector<double> ay((size_t)n);
  for (int i = 0; i < n; ++i) {
    const double r = sqrt(w[i]), y = r * (x[i] - mean);
    sy += y; syy += y * y; sr += r;
    wmin = std::min(wmin, w[i]); wmax = std::max(wmax, w[i]);
    ay[(size_t)i] = fabs(y);
  }
  const double rbar = sr / n;
  double srr = 0;
  for (int i = 0; i < n; ++i) { const double e = sqrt(w[i]) - rbar; srr += e * e; }
  const double part = sqrt(srr * syy) + rbar * fabs(sy);
  const double tss_min = syy - part * part / W;
  const int a_hi = std::min(kmax, n - minw);
  if (a_hi < minw || !(tss_min > 0)) return HUGE_VAL;
  std::partial_sort(ay.begin(), ay.begin() + a_hi, ay.end(), std::greater<double>());
  const double rmax = sqrt(wmax);
  double Y = 0, bmax = 0;
  for (int a = 1; a <= a_hi; ++a) {
    Y += ay[(size_t)a - 1];
    if (a < minw) continue;
    if (a * wmax >= W) return HUGE_VAL;
    const double d = rmax * Y + part * a * wmax / W;
    const double den = std::min(a * wmin * (W - a * wmin), a * wmax * (W - a * wmax)) / W;
    bmax = std::max(bmax, d * d / den);
  }
  if (!(tss_min > bmax)) return HUGE_VAL;
  return bmax / ((tss_min - bmax) / (n - 2.0));
}

// ------------------------------------------------------------------ host side
struct CbsParams {
  double alpha;
  int nperm = 10000, kmax = 25, nmin = 200, minw = 2, ngrid = 100;
  unsigned long long seed;
};

// device arena of one call, grown on demand
struct Arena {
  wcx_ctx *ctx;
  char *base = nullptr;
  size_t off = 0, cap = 0;
  template <typename T> T *take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T *p = reinterpret_cast<T *>(base + off);
    off += count * sizeof(T);
    return p;
  }
};

constexpr int LDS_KEYS_MAX = 32768;      // n above this sorts in global scratch (k_cbs_perm<true>)
constexpr int BIG_GRID = 512;

}  // namespace

extern "C" {

int wcx_cbs_batch(wcx_ctx *ctx, const double *r, const double *w, int n_samples, int64_t n_bins,
                  const int64_t *chr_off, int n_chr, double alpha, int64_t binsize, uint64_t seed,
                  double *out_seg, int cap, int *out_count) {
  WCX_ARG(ctx && r && w && chr_off && out_seg && out_count, "NULL argument");
  WCX_ARG(n_samples > 0 && n_chr > 0 && alpha > 0 && alpha <= 1 && binsize > 0 && cap >= 0,
          "bad parameters");
  WCX_ARG(chr_off[n_chr] <= n_bins, "chr_off beyond n_bins");
  WCX_HIP(hipSetDevice(ctx->device));
  CbsParams P;
  P.alpha = alpha;
  P.seed = seed;
  hipStream_t st = ctx->stream;

  const auto T0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!(ctx->debug_flags & 8)) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count();
    fprintf(stderr, "[cbs] %-10s at %8.3f ms\n", what, ms);
  };
  // ---- NA-free series of every (sample, chromosome): CBS.R:41-42,56-63
  struct Series { int sample, chr; int64_t lo; int n; std::vector<int> seg_end, change_loc; };
  std::vector<Series> series;
  // two passes (count, then fill at known offsets) so that samples can be filled by host threads
  std::vector<int64_t> cnt_sc((size_t)n_samples * n_chr);
  auto is_na = [](double v) { return v == 0.0 || v != v; };   // ratio == 0 -> NA
  auto for_samples = [&](auto &&fn) {
    const int nt = std::min(n_samples, 8);
    if (nt <= 1) { for (int s = 0; s < n_samples; ++s) fn(s); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t] { for (int s = t; s < n_samples; s += nt) fn(s); });
    for (auto &x : th) x.join();
  };
  for_samples([&](int s) {
    for (int c = 0; c < n_chr; ++c) {
      const double *rr = r + (int64_t)s * n_bins + chr_off[c];
      const int nall = (int)(chr_off[c + 1] - chr_off[c]);
      int64_t m = 0;
      for (int i = 0; i < nall; ++i) m += is_na(rr[i]) ? 0 : 1;
      cnt_sc[(size_t)s * n_chr + c] = m;
    }
  });
  int64_t total = 0;
  std::vector<int64_t> lo_sc(cnt_sc.size());
  for (size_t q = 0; q < cnt_sc.size(); ++q) { lo_sc[q] = total; total += cnt_sc[q]; }
  // x | w | 1-based bin index within the chromosome (CBS.R:49), in the context's pinned staging area
  void *hstage = nullptr;
  {
    const int rcs = wcx_host_scratch(ctx, (size_t)total * 20 + 64, &hstage);
    if (rcs) return rcs;
  }
  double *hx = reinterpret_cast<double *>(hstage), *hw = hx + total;
  int *hpos = reinterpret_cast<int *>(hw + total);
  for_samples([&](int s) {
    for (int c = 0; c < n_chr; ++c) {
      const int64_t o = (int64_t)s * n_bins + chr_off[c];
      const int nall = (int)(chr_off[c + 1] - chr_off[c]);
      int64_t at = lo_sc[(size_t)s * n_chr + c];
      for (int i = 0; i < nall; ++i) {
        const double v = r[o + i];
        if (is_na(v)) continue;
        hx[at] = v;
        hw[at] = w[o + i] == 0.0 ? 1.0 : w[o + i];   // weight == 0 -> 1 (1^-99 == 1)
        hpos[at] = i + 1;
        ++at;
      }
    }
  });
  for (int s = 0; s < n_samples; ++s)
    for (int c = 0; c < n_chr; ++c) {
      const size_t q = (size_t)s * n_chr + c;
      if (cnt_sc[q] == 0) continue;                       // all-NA chromosome is dropped
      Series se;
      se.sample = s; se.chr = c; se.lo = lo_sc[q]; se.n = (int)cnt_sc[q];
      se.seg_end = {0, se.n};
      series.push_back(std::move(se));
    }
  const int64_t N = total;
  lap("series");
  int rc = wcx_timer_begin(ctx, "cbs");
  if (rc) return rc;
  if (N > 0) {
    int max_n = 0;
    for (const Series &se : series) max_n = std::max(max_n, se.n);
    const int max_segs = (int)series.size() + 8;          // active segments per round <= series
    // ---- device arena
    int npad_max = 64;
    while (npad_max < max_n) npad_max <<= 1;
    const bool any_big = max_n > LDS_KEYS_MAX;
    Arena A;
    A.ctx = ctx;
    const size_t max_items = (size_t)max_segs + (size_t)(N / 4) + 6144 + 64;
    size_t need = (size_t)N * (8 * 4 + 4 * 3) + (size_t)max_segs * (sizeof(SegIn) + sizeof(SegOut) + 8) +
                  (size_t)max_segs * P.ngrid * 16 + max_items * (sizeof(ArcItem) + sizeof(ArcBest)) +
                  (size_t)max_segs * 3 * (sizeof(PermJob) + 4) +
                  (any_big ? (size_t)BIG_GRID * (npad_max + 32) * 4 : 0) +
                  (size_t)(KMAXC - 1) * ((size_t)N + (size_t)NTP * max_segs) * 4 + (1 << 16);
    void *scr = nullptr;
    rc = wcx_scratch(ctx, need, &scr);
    if (rc) return rc;
    A.base = reinterpret_cast<char *>(scr);
    A.cap = need;
    double *dX = A.take<double>(N), *dW = A.take<double>(N), *dS = A.take<double>(N), *dWp = A.take<double>(N);
    float *dy = A.take<float>(N), *drw = A.take<float>(N), *dWpf = A.take<float>(N);
    SegIn *dseg = A.take<SegIn>(max_segs);
    SegOut *dso = A.take<SegOut>(max_segs);
    int *dfirst = A.take<int>(max_segs + 1);
    double *dtx = A.take<double>((size_t)max_segs * P.ngrid), *dnu = A.take<double>((size_t)max_segs * P.ngrid);
    ArcItem *ditems = A.take<ArcItem>(max_items);
    ArcBest *dbest = A.take<ArcBest>(max_items);
    PermJob *djobs = A.take<PermJob>((size_t)max_segs * 3);
    unsigned int *dnrej = A.take<unsigned int>((size_t)max_segs * 3);
    unsigned int *dbig = any_big ? A.take<unsigned int>((size_t)BIG_GRID * (npad_max + 32)) : nullptr;
    float *dq = A.take<float>((size_t)(KMAXC - 1) * ((size_t)N + (size_t)NTP * max_segs));
    WCX_HIP(hipMemcpyAsync(dX, hx, (size_t)N * 8, hipMemcpyHostToDevice, st));
    WCX_HIP(hipMemcpyAsync(dW, hw, (size_t)N * 8, hipMemcpyHostToDevice, st));
    const size_t lds_small = (size_t)(std::min((max_n + NTP - 1) / NTP * NTP, LDS_KEYS_MAX) + 32) * 4;
    WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cbs_perm<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small));

    // one batch of permutation jobs -> exceedance counts
    std::vector<unsigned int> hnrej;
    auto run_jobs = [&](std::vector<PermJob> &jobs) -> int {
      hnrej.assign(jobs.size(), 0u);
      if (jobs.empty()) return WCX_OK;
      // small jobs (LDS sort) and big jobs (global scratch) go to their own launches
      for (int big = 0; big < 2; ++big) {
        std::vector<PermJob> sel;
        std::vector<size_t> where;
        for (size_t q = 0; q < jobs.size(); ++q)
          if ((jobs[q].n > LDS_KEYS_MAX) == (big == 1)) { sel.push_back(jobs[q]); where.push_back(q); }
        if (sel.empty()) continue;
        size_t qoff = 0;
        int sel_max_n = 0;
        bool any_hybrid = false;
        for (PermJob &jb : sel) {
          jb.qoff = (int64_t)qoff;
          if (jb.mode == 0) {
            qoff += (size_t)(KMAXC - 1) * (size_t)((jb.n + NTP - 1) / NTP * NTP);
            any_hybrid = true;
          }
          sel_max_n = std::max(sel_max_n, jb.n);
        }
        WCX_HIP(hipMemcpyAsync(djobs, sel.data(), sel.size() * sizeof(PermJob), hipMemcpyHostToDevice, st));
        WCX_HIP(hipMemsetAsync(dnrej, 0, sel.size() * 4, st));
        if (any_hybrid) {
          const int npad_sel = (sel_max_n + NTP - 1) / NTP * NTP;
          k_cbs_arcweights<<<dim3((unsigned)(npad_sel / 256), (unsigned)sel.size()), 256, 0, st>>>(
              drw, dWpf, djobs, P.minw, dq);
          WCX_HIP(hipGetLastError());
        }
        if (big)
          k_cbs_perm<true><<<dim3(BIG_GRID, (unsigned)sel.size()), NTP, 64, st>>>(
              dy, drw, dWpf, djobs, P.nperm, npad_max + 32, P.minw, P.kmax, dq, dbig, dnrej);
        else
          k_cbs_perm<false><<<dim3((unsigned)P.nperm, (unsigned)sel.size()), NTP, lds_small, st>>>(
              dy, drw, dWpf, djobs, P.nperm, npad_max + 32, P.minw, P.kmax, dq, nullptr, dnrej);
        WCX_HIP(hipGetLastError());
        std::vector<unsigned int> got(sel.size());
        WCX_HIP(hipMemcpyAsync(got.data(), dnrej, sel.size() * 4, hipMemcpyDeviceToHost, st));
        WCX_HIP(hipStreamSynchronize(st));
        for (size_t q = 0; q < sel.size(); ++q) hnrej[where[q]] = got[q];
      }
      return WCX_OK;
    };

    // ---- level-synchronous recursion (DNAcopy changepoints(): stack of segment ends per series)
    struct Active { int series, lo, hi; };
    unsigned long long test_id = 0;
    for (;;) {
      // top-of-stack segment of every series that is not finished
      std::vector<Active> act;
      for (size_t q = 0; q < series.size(); ++q) {
        Series &se = series[q];
        while (se.seg_end.size() > 1) {
          const int k = (int)se.seg_end.size();
          const int lo = se.seg_end[k - 2], hi = se.seg_end[k - 1];
          if (hi - lo >= 2 * P.minw) { act.push_back({(int)q, lo, hi}); break; }
          se.change_loc.push_back(hi);            // too short to test: final
          se.seg_end.pop_back();
        }
      }
      if (act.empty()) break;
      const int ns = (int)act.size();
      std::vector<SegIn> hseg(ns);
      double arc_total = 0;
      for (int a = 0; a < ns; ++a) { const double n = act[a].hi - act[a].lo; arc_total += 0.5 * n * n; }
      const double arc_budget = std::max(arc_total / 6144.0, 16384.0);
      std::vector<ArcItem> items;
      std::vector<int> first(ns + 1);
      for (int a = 0; a < ns; ++a) {
        const Series &se = series[act[a].series];
        hseg[a].lo = se.lo + act[a].lo;
        hseg[a].n = act[a].hi - act[a].lo;
        hseg[a].hybrid = hseg[a].n > P.nmin ? 1 : 0;
        first[a] = (int)items.size();
        // stripes of rows with about `arc_budget` arcs each (row i meets ~n - i columns): ~6000
        // equal work items per round however few or short the active segments are
        const int n = hseg[a].n;
        for (int i0 = 0; i0 < n;) {
          const int rows = (int)std::min<double>(ARC_ROWS, std::max<double>(4.0, arc_budget / (double)(n - i0)));
          items.push_back({a, i0, std::min(i0 + rows, n), 0});
          i0 += rows;
        }
      }
      first[ns] = (int)items.size();
      WCX_ARG(items.size() <= max_items, "internal: arc stripe table overflow");
      WCX_HIP(hipMemcpyAsync(dseg, hseg.data(), (size_t)ns * sizeof(SegIn), hipMemcpyHostToDevice, st));
      WCX_HIP(hipMemcpyAsync(ditems, items.data(), items.size() * sizeof(ArcItem), hipMemcpyHostToDevice, st));
      WCX_HIP(hipMemcpyAsync(dfirst, first.data(), (size_t)(ns + 1) * 4, hipMemcpyHostToDevice, st));
      k_cbs_prepare<<<ns, NTP, 0, st>>>(dX, dW, dseg, dS, dWp, dy, drw, dWpf, dso);
      k_cbs_arcmax<<<(unsigned)items.size(), 256, 0, st>>>(dS, dWp, dseg, dso, ditems, P.minw, dbest);
      k_cbs_arcfinish<<<ns, 128, 0, st>>>(dbest, dfirst, dseg, dso, P.kmax, P.ngrid, dtx);
      k_nu_series<<<dim3(P.ngrid, ns), 256, 0, st>>>(dtx, dseg, dso, P.ngrid, dnu);
      k_cbs_tailp<<<ns, 64, 0, st>>>(dnu, dseg, dso, P.kmax, P.ngrid);
      WCX_HIP(hipGetLastError());
      std::vector<SegOut> hso(ns);
      WCX_HIP(hipMemcpyAsync(hso.data(), dso, (size_t)ns * sizeof(SegOut), hipMemcpyDeviceToHost, st));
      WCX_HIP(hipStreamSynchronize(st));

      // segmentation tests that need permutations
      std::vector<PermJob> jobs;
      std::vector<int> job_of(ns, -1);
      std::vector<int> verdict(ns, 0);     // 0 = no change, 1 = significant
      int n_shortcut = 0;
      for (int a = 0; a < ns; ++a) {
        if (!hso[a].valid) continue;
        double pval2 = P.alpha;
        if (hseg[a].hybrid) {
          if (hso[a].pval1 > P.alpha) continue;
          pval2 = P.alpha - hso[a].pval1;
        }
        PermJob jb;
        jb.lo = hseg[a].lo; jb.n = hseg[a].n; jb.mode = hseg[a].hybrid ? 0 : 1;
        jb.m1 = 0; jb.first = 0; jb.pad = 0;
        jb.nrejc = (int)(pval2 * P.nperm);
        jb.ostat = 0.99999 * hso[a].ostat;
        if (jb.mode == 0 && !(ctx->debug_flags & 32) &&
            short_arc_bound(hx + jb.lo, hw + jb.lo, jb.n, P.minw, P.kmax) * 1.05 < jb.ostat) {
          // no permutation of this series can reach the observed statistic with an arc of <= kmax
          // points: the exceedance count is 0 without running the nperm permutations
          verdict[a] = 1;
          ++n_shortcut;
          continue;
        }
        jb.seed = P.seed ^ ((++test_id) * 0x2545f4914f6cdd1dull);
        job_of[a] = (int)jobs.size();
        jobs.push_back(jb);
      }
      rc = run_jobs(jobs);
      if (rc) return rc;
      for (int a = 0; a < ns; ++a)
        if (job_of[a] >= 0) verdict[a] = hnrej[job_of[a]] <= (unsigned int)jobs[job_of[a]].nrejc ? 1 : 0;
      ctx->cbs_shortcuts += n_shortcut;

      // interior arcs: each of the two change-points needs its own two-sample test
      std::vector<PermJob> ejobs;
      struct EdgeRef { int a, which, shortcut; };
      std::vector<EdgeRef> eref;
      for (int a = 0; a < ns; ++a) {
        if (!verdict[a]) continue;
        const int n = hseg[a].n, bi = hso[a].bi, bj = hso[a].bj;
        if (bi == 0 || bj == n) continue;
        const double *x = hx + hseg[a].lo, *ww = hw + hseg[a].lo;
        // test 1: [0, bi) vs [bi, bj) ; test 2: [bi, bj) vs [bj, n)
        for (int which = 0; which < 2; ++which) {
          const int l = which == 0 ? 0 : bi, n12 = which == 0 ? bj : n - bi;
          const int n1 = which == 0 ? bi : bj - bi, n2 = n12 - n1;
          EdgeRef er{a, which, -1};
          if (n1 == 1 || n2 == 1) { er.shortcut = 0; eref.push_back(er); continue; }   // p = 1: not kept
          double w1 = 0, w2 = 0, s1 = 0, s2 = 0;
          for (int i = 0; i < n1; ++i) { w1 += ww[l + i]; s1 += ww[l + i] * x[l + i]; }
          for (int i = n1; i < n12; ++i) { w2 += ww[l + i]; s2 += ww[l + i] * x[l + i]; }
          const double xbar = (s1 + s2) / (w1 + w2);
          double tss = 0;
          for (int i = 0; i < n12; ++i) tss += ww[l + i] * (x[l + i] - xbar) * (x[l + i] - xbar);
          const bool first_short = n1 <= n2;
          const int m1 = first_short ? n1 : n2;
          const double wm = first_short ? w1 : w2, wo = first_short ? w2 : w1;
          const double dm = (first_short ? s1 / w1 : s2 / w2) - xbar;
          double tstat = dm * dm * wm * (wm + wo) / wo;
          tstat = tstat / ((tss - tstat) / (n12 - 2.0));
          if (tstat > 25.0 && m1 >= 10) { er.shortcut = 1; eref.push_back(er); continue; }   // p = 0
          // permutation test on the centred sub-series: needs its own prepared y / rw / Wpf, which
          // the segment's buffers hold for the WHOLE segment (centred on the segment mean): the
          // statistic below re-centres per permutation, so the segment's y serves as is
          PermJob jb;
          jb.lo = hseg[a].lo + l; jb.n = n12; jb.mode = 2; jb.m1 = m1; jb.first = first_short ? 1 : 0;
          jb.pad = 0;
          jb.nrejc = (int)(P.alpha * P.nperm);          // p <= alpha  <=>  nrej <= alpha * nperm
          jb.ostat = 0.99999 * fabs(dm);
          jb.seed = P.seed ^ ((++test_id) * 0x2545f4914f6cdd1dull);
          er.shortcut = -1 - (int)ejobs.size();          // (-1 - job index)
          ejobs.push_back(jb);
          eref.push_back(er);
        }
      }
      rc = run_jobs(ejobs);
      if (rc) return rc;
      std::vector<int> keep(ns * 2, 0);
      for (const EdgeRef &er : eref) {
        bool ok;
        if (er.shortcut >= 0) ok = er.shortcut == 1;
        else {
          const int q = -1 - er.shortcut;
          ok = hnrej[q] <= (unsigned int)ejobs[q].nrejc;
        }
        keep[er.a * 2 + er.which] = ok ? 1 : 0;
      }
      // ---- update the stacks
      for (int a = 0; a < ns; ++a) {
        Series &se = series[act[a].series];
        const int lo = act[a].lo, hi = act[a].hi, n = hseg[a].n;
        int ncpt = 0, icpt[2] = {0, 0};
        if (verdict[a]) {
          const int bi = hso[a].bi, bj = hso[a].bj;
          if (bi == 0) { ncpt = 1; icpt[0] = bj; }
          else if (bj == n) { ncpt = 1; icpt[0] = bi; }
          else {
            if (keep[a * 2]) icpt[ncpt++] = bi;
            if (keep[a * 2 + 1]) icpt[ncpt++] = bj;
          }
        }
        if (ncpt == 0) { se.change_loc.push_back(hi); se.seg_end.pop_back(); }
        else if (ncpt == 1) se.seg_end.insert(se.seg_end.end() - 1, lo + icpt[0]);
        else {
          se.seg_end.insert(se.seg_end.end() - 1, lo + icpt[0]);
          se.seg_end.insert(se.seg_end.end() - 1, lo + icpt[1]);
        }
      }
    }
  }
  rc = wcx_timer_end(ctx, "cbs");
  if (rc) return rc;
  lap("rounds");

  // ---- CBS.R:84-129 on the host: NA-run splitting, >= 2-bin rule, weighted re-mean, 0-based start
  const int na_limit = (int)(1.0 / ((double)binsize / 2000000.0));   // as.integer((binsize/2e6)^-1)
  std::vector<int> count(n_samples, 0);
  for_samples([&](int my_sample) {
  for (Series &se : series) {
    if (se.sample != my_sample) continue;
    std::sort(se.change_loc.begin(), se.change_loc.end());
    const int s = se.sample, c = se.chr;
    const int64_t o = (int64_t)s * n_bins + chr_off[c];
    const int *pos = hpos + se.lo;
    int prev = 0;
    for (int e : se.change_loc) {
      const int s1 = pos[prev], e1 = pos[e - 1];   // inclusive, 1-based
      prev = e;
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
        if (count[s] < cap) {
          double *dst = out_seg + ((size_t)s * cap + count[s]) * 4;
          dst[0] = c;
          dst[1] = a - 1;                           // CBS.R:129
          dst[2] = b;
          dst[3] = den > 0 ? num / den : __builtin_nan("");
        }
        ++count[s];
      }
    }
  }
  });
  lap("wrap-up");
  int over = 0;
  for (int s = 0; s < n_samples; ++s) { out_count[s] = count[s]; over = std::max(over, count[s]); }
  if (over > cap) {
    wcx_set_error("wcx_cbs: %d segments exceed the caller's capacity %d", over, cap);
    return WCX_ERR_ARG;
  }
  return WCX_OK;
}

int wcx_cbs_stats(wcx_ctx *ctx, int64_t out[4]) {
  WCX_ARG(ctx && out, "NULL argument");
  out[0] = ctx->cbs_shortcuts; out[1] = out[2] = out[3] = 0;
  return WCX_OK;
}

int wcx_cbs(wcx_ctx *ctx, const double *r, const double *w, const int64_t *chr_off, int n_chr,
            double alpha, int64_t binsize, uint64_t seed, double *out_seg, int cap,
            int *out_count) {
  WCX_ARG(chr_off != nullptr && n_chr > 0, "bad parameters");
  return wcx_cbs_batch(ctx, r, w, 1, chr_off[n_chr], chr_off, n_chr, alpha, binsize, seed, out_seg, cap,
                       out_count);
}

}  // extern "C"
