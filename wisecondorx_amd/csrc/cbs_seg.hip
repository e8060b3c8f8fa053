// Circular binary segmentation on the GPU.
//
// Replaces predict_tools.exec_cbs -> Rscript include/CBS.R -> DNAcopy::segment
// (predict_tools.py:242-257, CBS.R:21-132).  The reference-owned code around the DNAcopy call
// (NA masking, weight fix-up CBS.R:41-42, dropping all-NA chromosomes :56-63, splitting segments
// over long NA runs :84-113, weighted re-mean :122-127, 0-based starts :129) is reproduced
// exactly.  The segmentation itself lives in Bioconductor DNAcopy 1.76.0 (conda.yml:14), which is
// NOT part of the reference repository and cannot run here (no R): PARITY UNPINNED against DNAcopy.
// What it IS checked against: the test suite's NumPy CBS oracle, an independent statement of the same
// algorithm (Olshen et al. 2004; Venkatraman & Olshen 2007; the decision flow of DNAcopy's
// changepoints() / wfindcpt / wtpermp / tailp / getbdry as recalled) -- identical change-points,
// exceedance counts and stopping points on a fuzz set (tests/test_gpu_cbs_oracle.py).
//
// Decision flow of one test of a segment with n points (weights w, series centred on its weighted
// mean; "[D]" = DNAcopy behaviour as recalled, unverifiable offline):
//   * n < 2 min.width (= 4), or max - min <= 1.49e-8 [D: all.equal(diff(range), 0)]: no change;
//   * observed statistic: max over arcs (i, j], min.width <= j - i <= n - min.width, of
//     bss = (S_j - S_i)^2 / (w_a (W - w_a) / W); t^2 = bss / ((tss - bss) / (n - 2)), with
//     tss <= bss + 1e-4 replaced by bss + 1 [D]; all in fp64;
//   * t <= 0.1: no change [D]; t >= 7 with >= 10 points on the shorter side of the arc: change
//     without a p-value [D];
//   * n > nmin = 200 ("hybrid"): p1 = Siegmund tail approximation for the arcs holding a weight
//     fraction in [delta, 1 - delta], delta = min weight of any arc of kmax + 1 points / W [D:
//     getmncwt]; p1 > alpha: no change; else nperm = 10 000 permutations of the maximum over arcs
//     of <= kmax = 25 points (or whose complement has <= kmax points) against 0.99999 t^2 with the
//     exceedance budget nrejc = int((alpha - p1) nperm); n <= nmin: all arcs, budget int(alpha nperm);
//   * the permutations are consumed IN ORDER under the sequential boundary of Venkatraman & Olshen
//     (getbdry(eta = 0.05, nperm, floor(nperm alpha) + 1) [D]): not significant as soon as the
//     exceedances pass the budget, significant as soon as permutation number np reaches
//     boundary[exceedances so far];
//   * an interior arc gives two change-points, each kept only if its two-sample permutation test
//     (weighted mean of the shorter side; t^2 > 25 with >= 10 points: kept without permutations [D])
//     has p <= alpha; undo.splits = "none".
// Permuted series [D: wxperm]: y = sqrt(w) (x - mean) is exchangeable under H0; position i receives
// y[pi(i)] / sqrt(w_i), i.e. the weighted value v_i = sqrt(w_i) y[pi(i)].  Their total T is not 0
// (unlike the unweighted case), and an arc and its complement only carry the same statistic for a
// centred series: the permuted series is re-centred on its weighted mean (v_i - (T / W) w_i) and
// its total sum of squares is tss - T^2 / W (sum v^2 / w = sum y^2 is permutation invariant).
// Whether DNAcopy re-centres is not recalled with certainty; without it the arcs whose complement
// is short pick up T^2 ~ n var(sqrt w) and swamp the statistic.  pi = keyed Feistel bijection
// (specification: DESIGN.md section 6; key = f(seed, chromosome, segment, kind of test) --
// independent of batch position and scheduling).
//
// GPU mapping -- LEVEL-SYNCHRONOUS and BATCHED: all chromosomes of all samples of a call advance
// together.  Per round: (1) one launch prepares every active segment (weighted centring, fp64
// prefix sums), (2) one launch finds every segment's best arc (fp64, striped over many
// workgroups), (3) one launch evaluates the tail probabilities, ONE device->host copy of the
// per-segment records, (4) the permutation tests of all segments: one workgroup per permutation.
// The hybrid statistic is SCREENED in fp32 -- arc sums are local prefix sums of <= 33 raw values in
// registers, so their error is bounded by a per-job constant; a permutation whose statistic lies
// within that bound of the threshold is flagged and re-evaluated in fp64 by k_cbs_perm_exact --
// every exceedance is recorded as a BIT (job, permutation), the host walks the bits in order under
// the sequential rule.  A workgroup skips its permutation when the exceedances recorded in EARLIER
// 256-permutation chunks already pass the budget (the sequential rule can then never reach it).
// Permutations run in two stages, [0, 512) then the rest for the undecided tests.  (5) the edge
// tests (fp64, one wave per permutation).  The host only keeps the segment stack.
// Compute/latency bound; reported as wall-clock per sample.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "wave_sort.h"
#include "wcx_common.h"

namespace {

constexpr int NTP = 1024;
constexpr int KMAXC = 25;    // DNAcopy's kmax (short-arc limit of the hybrid p-value), compile-time here
constexpr int QC_W = 32;     // complement-arc weight table: [KMAXC + 1][QC_W]
constexpr int PCHUNK = 256;  // permutations per skip-counter chunk
constexpr int NCH = 64;      // chunk counters per job (nperm <= 16384)
constexpr int STAGE_A = 512; // permutations of the first stage
typedef float f32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// Random permutation WITHOUT a sort: a keyed bijection of [0, n).  Four Feistel rounds on
// Z_a x Z_a, a = ceil(sqrt(n)) (so the domain a*a exceeds n by < 2 sqrt(n) + 1 and the cycle walk
// back into [0, n) almost never iterates); round function = murmur3's 32-bit finaliser of
// (half + round key), mapped to [0, a) by a high multiply.
struct Feistel {
  unsigned int fa, n, rk[4];
  float inv_fa;
  __device__ __forceinline__ void init(int n_, unsigned long long key, int p) {
    n = (unsigned int)n_;
    fa = (unsigned int)sqrtf((float)n_);
    while ((unsigned long long)fa * fa < (unsigned long long)n) ++fa;
    while (fa > 1 && (unsigned long long)(fa - 1) * (fa - 1) >= (unsigned long long)n) --fa;
    inv_fa = 1.0f / (float)fa;
    const unsigned long long s0 = mix64(key ^ ((unsigned long long)p * 0xd1342543de82ef95ull));
    const unsigned long long s1 = mix64(s0 + 1ull);
    rk[0] = (unsigned int)s0; rk[1] = (unsigned int)(s0 >> 32);
    rk[2] = (unsigned int)s1; rk[3] = (unsigned int)(s1 >> 32);
  }
  __device__ __forceinline__ int at(int i) const {
    unsigned int x = (unsigned int)i;
    do {
      unsigned int L;
      int R;
      if (n < (1u << 22)) {          // x < fa^2 < 2^23: (float)x is exact, the quotient is off by <= 1
        L = (unsigned int)((float)x * inv_fa);
        R = (int)(x - L * fa);
        if (R < 0) { --L; R += (int)fa; }
        else if (R >= (int)fa) { ++L; R -= (int)fa; }
      } else {
        L = x / fa;
        R = (int)(x - L * fa);
      }
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
    } while (x >= n);
    return (int)x;
  }
};

// One active segment [lo, hi) of the concatenated (NA-free) data of all series of the call.
struct SegIn {
  int64_t lo;       // first element (index into the concatenated arrays)
  int32_t n;        // elements
  int32_t hybrid;   // n > nmin
};
// What the host needs back to decide.
struct SegOut {
  double tss, W, mean, ostat, pval1, delta;
  double pval_lo;             // cheap proven lower bound of the tail probability (k_cbs_arcfinish)
  double range, wmin, wmax, ymax;   // max - min of x; smallest / largest weight; largest |sqrt(w) (x - mean)|
  int32_t bi, bj;             // best arc (bi, bj], 0 <= bi < bj <= n
  int32_t valid, pad;
};
// A permutation test: the segmentation test of a segment (mode 0 hybrid, 1 all arcs) or one
// two-sample edge test (mode 2).
struct PermJob {
  int64_t lo;            // first element of the (sub)series
  int32_t n, mode;
  int32_t m1, budget;    // edge: size of the shorter side; exceedances the test tolerates
  int32_t slot, pad;     // row of the bit / counter tables
  double thr;            // 0.99999 t^2                       (edge: 0.99999 |mean_short - xbar|)
  double tss;            // observed tss of the segment       (edge: xbar)
  double W;              // total weight                      (edge: weight of the shorter side)
  // fp32 screen of the hybrid statistic (sqrt(bss) units): exceedance <=> bss >= cthr * tss', tss' =
  // tss - T^2 / W (T = total of the permuted weighted series); D = bound of the fp32 error of
  // sqrt(bss), eT |T| = bound of the error of tss' from T's rounding.  A permutation within these
  // bounds of the threshold, or of the tss' <= bss + 1e-4 rule, is flagged for the fp64 kernel.
  double cthr, D, eT;
  unsigned long long key;
  int64_t qoff;          // mode 0: this job's block of the arc-weight table (k_cbs_arcweights)
};

// ---- block reductions (NT threads, NT / 64 waves; every thread gets the result) -------------
template <int NT, typename Op>
__device__ __forceinline__ double block_reduce_d(double v, double *red, Op op, double ident) {
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;     // v is already wave-reduced
  __syncthreads();
  double r = ident;
  for (int q = 0; q < NT / 64; ++q) r = op(r, red[q]);
  return r;
}
template <int NT>
__device__ __forceinline__ double block_sum_d(double v, double *red) {
  return block_reduce_d<NT>(wcx::wave_sum(v), red, [](double a, double b) { return a + b; }, 0.0);
}
template <int NT>
__device__ __forceinline__ double block_max_d(double v, double *red) {
  return block_reduce_d<NT>(wcx::wave_max_f64(v), red, [](double a, double b) { return a > b ? a : b; },
                            -HUGE_VAL);
}
template <int NT>
__device__ __forceinline__ double block_min_d(double v, double *red) {
  return block_reduce_d<NT>(wcx::wave_min_f64(v), red, [](double a, double b) { return a < b ? a : b; },
                            HUGE_VAL);
}

// ---- (1) prepare: weighted centring + prefix sums -------------------------------------------
// S[lo + i] = sum_{t < i} w (x - mean), Wp likewise (one slot per element; a segment's end values
// are 0 / W by construction and are not stored: segments are disjoint and ordered, the end slot of
// one is the start slot of the next).  yd = sqrt(w) (x - mean) (fp64) and its fp32 image y, rw =
// sqrt(w) (fp32) feed the permutation kernels.
__global__ __launch_bounds__(NTP) void k_cbs_prepare(const double *__restrict__ x,
                                                     const double *__restrict__ w,
                                                     const SegIn *__restrict__ segs,
                                                     double *__restrict__ S, double *__restrict__ Wp,
                                                     double *__restrict__ yd, float *__restrict__ y,
                                                     float *__restrict__ rw,
                                                     SegOut *__restrict__ out) {
  const SegIn sg = segs[blockIdx.x];
  const int n = sg.n, tid = threadIdx.x;
  const double *xs = x + sg.lo, *ws = w + sg.lo;
  __shared__ double red[NTP / 64];
  __shared__ double tot[2][NTP];
  double sw = 0.0, sx = 0.0, xmin = HUGE_VAL, xmax = -HUGE_VAL, wmin = HUGE_VAL, wmax = 0.0;
  for (int i = tid; i < n; i += NTP) {
    sw += ws[i]; sx += ws[i] * xs[i];
    xmin = xs[i] < xmin ? xs[i] : xmin;
    xmax = xs[i] > xmax ? xs[i] : xmax;
    wmin = ws[i] < wmin ? ws[i] : wmin;
    wmax = ws[i] > wmax ? ws[i] : wmax;
  }
  const double W = block_sum_d<NTP>(sw, red), SX = block_sum_d<NTP>(sx, red);
  xmin = block_min_d<NTP>(xmin, red);
  xmax = block_max_d<NTP>(xmax, red);
  wmin = block_min_d<NTP>(wmin, red);
  wmax = block_max_d<NTP>(wmax, red);
  const double mean = SX / W;
  // chunked scan: thread t owns elements [t*chunk, (t+1)*chunk)
  const int chunk = (n + NTP - 1) / NTP;
  const int i0 = tid * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
  double ls = 0.0, lw = 0.0, lt = 0.0, ymax = 0.0;
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
    const double c = xs[i] - mean, r = sqrt(ws[i]);
    const double yy = c * r;
    yd[sg.lo + i] = yy;
    y[sg.lo + i] = (float)yy;
    rw[sg.lo + i] = (float)r;
    ymax = fabs(yy) > ymax ? fabs(yy) : ymax;
    rs += ws[i] * c;
    rwt += ws[i];
  }
  const double tss = block_sum_d<NTP>(lt, red);
  ymax = block_max_d<NTP>(ymax, red);
  if (tid == 0) {
    SegOut o;
    o.tss = tss; o.W = W; o.mean = mean; o.ostat = 0.0; o.pval1 = 0.0; o.delta = 0.0; o.pval_lo = 0.0;
    o.range = xmax - xmin; o.wmin = wmin; o.wmax = wmax; o.ymax = ymax;
    o.bi = 0; o.bj = 0; o.valid = 0; o.pad = 0;
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

// ---- (2b) the same maximum with BLOCK BOUNDS: exact, but only block pairs that can hold it ------
// Positions 0 .. n of a segment are cut into blocks of PBS; per block the extremes of S (with their
// positions) and the range of the prefix weight.  For a pair of blocks (I <= J) every arc (i, j],
// i in I, j in J, has |S_j - S_i| <= D = max(max S_J - min S_I, max S_I - min S_J) and an arc weight
// in [wlo, whi] = [W_first(J) - W_last(I), W_last(J) - W_first(I)]; w (W - w) / W is concave, so it
// is at least the smaller end-point value: bss <= D^2 / that.  A lower bound L of the maximum comes
// from the arcs between the blocks' extreme positions.  Only pairs whose bound reaches L are
// evaluated arc by arc (the same fp32 screen + fp64 formula as k_cbs_arcmax); on noise that is the
// few diagonals next to the main one, ~2-3 % of the pairs.  Ties -> smallest (i, j) as before.
constexpr int PBS = 64;
struct BlkStat { double smin, smax, wlo, whi; int amin, amax; };
struct WorkItem { int seg, I, J; };

__device__ __forceinline__ double arc_bss(double si, double wi, double sj, double wj, double W) {
  const double d = sj - si, wa = wj - wi;
  return d * d / (wa * (W - wa) / W);
}

__global__ __launch_bounds__(256) void k_cbs_blockstats(const double *__restrict__ S,
                                                        const double *__restrict__ Wp,
                                                        const SegIn *__restrict__ segs,
                                                        const SegOut *__restrict__ so,
                                                        const int *__restrict__ boff,
                                                        const int *__restrict__ bseg,
                                                        int total_blocks, BlkStat *__restrict__ bs) {
  const int blk = (int)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blk >= total_blocks) return;
  const int s = bseg[blk];
  const SegIn sg = segs[s];
  const double W = so[s].W;
  const int p0 = (blk - boff[s]) * PBS, p = p0 + lane;
  const bool in = p <= sg.n;
  const double sv = in ? seg_S(S, sg, p) : 0.0;
  const double vmin = wcx::wave_min_f64(in ? sv : HUGE_VAL), vmax = wcx::wave_max_f64(in ? sv : -HUGE_VAL);
  const unsigned long long bmin = __ballot(in && sv == vmin), bmax = __ballot(in && sv == vmax);
  if (lane == 0) {
    const int plast = p0 + PBS - 1 < sg.n ? p0 + PBS - 1 : sg.n;
    BlkStat b;
    b.smin = vmin; b.smax = vmax;
    b.amin = p0 + __ffsll((long long)bmin) - 1; b.amax = p0 + __ffsll((long long)bmax) - 1;
    b.wlo = seg_W(Wp, sg, W, p0); b.whi = seg_W(Wp, sg, W, plast);
    bs[blk] = b;
  }
}

// lower bound L[s] of the maximum: arcs between the extreme positions of every block pair.
// ONE WAVE per block row I (a 256-thread workgroup per row left most of its threads idle -- a series
// has <= ~260 blocks -- and paid a dependent-load chain per workgroup: 273 k workgroups of a 96-sample
// call cost 1 ms here and 3 ms in k_cbs_prune); bseg[blk] = the row's segment (host table).
__global__ __launch_bounds__(256) void k_cbs_coarse(const double *__restrict__ S,
                                                    const double *__restrict__ Wp,
                                                    const SegIn *__restrict__ segs,
                                                    const SegOut *__restrict__ so,
                                                    const int *__restrict__ boff,
                                                    const int *__restrict__ bseg, int total_blocks,
                                                    const BlkStat *__restrict__ bs, int minw,
                                                    unsigned int *__restrict__ L) {
  const int blk = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blk >= total_blocks) return;
  const int s = bseg[blk];
  const SegIn sg = segs[s];
  const double W = so[s].W;
  const int n = sg.n, I = blk - boff[s], nb = boff[s + 1] - boff[s];
  const BlkStat bi = bs[blk];
  float best = 0.f;
  for (int J = I + lane; J < nb; J += 64) {
    const BlkStat bj = bs[boff[s] + J];
    const int pa[2] = {bi.amin, bi.amax}, pb[2] = {bj.amin, bj.amax};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int i = pa[q & 1], j = pb[q >> 1];
      if (i > j) { const int t = i; i = j; j = t; }
      const int a = j - i;
      if (a < minw || n - a < minw) continue;
      const double b = arc_bss(seg_S(S, sg, i), seg_W(Wp, sg, W, i), seg_S(S, sg, j), seg_W(Wp, sg, W, j), W);
      const float bf = __double2float_rd(b);
      best = bf > best ? bf : best;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(best, m, 64); best = o > best ? o : best; }
  if (lane == 0 && best > 0.f) atomicMax(&L[s], __float_as_uint(best));
}

// block pairs whose bound reaches L -> work list.  A wave walks PRW consecutive block rows and stages
// the kept pairs in LDS; the list's tail counter -- ONE address for the whole call -- is bumped once
// per flush (~1 per 16 rows) instead of once per kept pair or row chunk: the returning same-address
// atomics (3e5 .. 6e5 of them for 96 samples) were this kernel's 3 ms, not its arithmetic.
constexpr int PRW = 16;          // rows per wave
constexpr int PSTAGE = 256;      // staged pairs per wave
__global__ __launch_bounds__(256) void k_cbs_prune(const SegOut *__restrict__ so,
                                                   const int *__restrict__ boff,
                                                   const int *__restrict__ bseg, int total_blocks,
                                                   const BlkStat *__restrict__ bs,
                                                   const unsigned int *__restrict__ L,
                                                   WorkItem *__restrict__ work, unsigned int cap,
                                                   unsigned int *__restrict__ count) {
  __shared__ WorkItem stage[4][PSTAGE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  WorkItem *mine = stage[wave];
  int staged = 0;                                            // wave-uniform
  auto flush = [&]() {
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(count, (unsigned int)staged);
    base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
    for (int e = lane; e < staged; e += 64)
      if (base + (unsigned int)e < cap) work[base + (unsigned int)e] = mine[e];
    __builtin_amdgcn_wave_barrier();
    staged = 0;
  };
  const int row0 = ((int)blockIdx.x * 4 + wave) * PRW;
  for (int blk = row0; blk < row0 + PRW && blk < total_blocks; ++blk) {
    const int s = bseg[blk];
    const double W = so[s].W;
    const int I = blk - boff[s], nb = boff[s + 1] - boff[s];
    const BlkStat bi = bs[blk];
    const double Ls = (double)__uint_as_float(L[s]) * (1.0 - 1e-9);
    for (int J0 = I; J0 < nb; J0 += 64) {
      const int J = J0 + lane;
      bool keep = false;
      if (J < nb) {
        const BlkStat bj = bs[boff[s] + J];
        const double d1 = bj.smax - bi.smin, d2 = bi.smax - bj.smin;
        const double D = d1 > d2 ? d1 : d2;
        const double wlo = bj.wlo - bi.whi, whi = bj.whi - bi.wlo;
        const double g1 = wlo * (W - wlo) / W, g2 = whi * (W - whi) / W;
        const double g = g1 < g2 ? g1 : g2;
        keep = !(g > 0.0) || D * D / g >= Ls;            // (g <= 0: same / touching blocks, or the whole series)
      }
      const unsigned long long km = __ballot(keep);
      if (km) {
        if (staged + 64 > PSTAGE) flush();
        if (keep) {
          WorkItem wi;
          wi.seg = s; wi.I = I; wi.J = J;
          mine[staged + __popcll(km & ((1ull << lane) - 1ull))] = wi;
        }
        staged += __popcll(km);
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (staged) flush();
}

// the arcs of the listed block pairs, exactly (fp32 screen, fp64 formula, ties -> smallest (i, j)).
// ONE WAVE per pair, no LDS, no barrier: lane q holds position i0 + q of block I and position j0 + q
// of block J; the lane walks its own i against the 64 j (broadcast by readlane), then the wave's best
// is a butterfly.  (A 256-thread workgroup per pair spent its time in the dependent loads of the
// item and three barriers: 2.7 ms for the 6e5 pairs of a 96-sample call.)
__global__ __launch_bounds__(256) void k_cbs_pairmax(const double *__restrict__ S,
                                                     const double *__restrict__ Wp,
                                                     const SegIn *__restrict__ segs,
                                                     const SegOut *__restrict__ so,
                                                     const WorkItem *__restrict__ work, unsigned int cap,
                                                     const unsigned int *__restrict__ count, int minw,
                                                     ArcBest *__restrict__ res,
                                                     unsigned long long *__restrict__ bbits,
                                                     const unsigned int *__restrict__ Lseg) {
  const unsigned int nitem = *count < cap ? *count : cap;
  const int lane = threadIdx.x & 63;
  const unsigned int w0 = (blockIdx.x * 256u + threadIdx.x) >> 6, nwv = (gridDim.x * 256u) >> 6;
  for (unsigned int it = w0; it < nitem; it += nwv) {
    const WorkItem wk = work[it];
    const SegIn sg = segs[wk.seg];
    const double W = so[wk.seg].W;
    const float Wf = (float)W;
    const int n = sg.n, i0 = wk.I * PBS, j0 = wk.J * PBS;
    const int i = i0 + lane, pj = j0 + lane;
    const double si = i <= n ? seg_S(S, sg, i) : 0.0, wi = i <= n ? seg_W(Wp, sg, W, i) : 0.0;
    const double sJ = pj <= n ? seg_S(S, sg, pj) : 0.0, wJ = pj <= n ? seg_W(Wp, sg, W, pj) : 0.0;
    double bb = -1.0;
    // (round 6) the fp32 screen starts at the segment's LOWER BOUND of the maximum (k_cbs_coarse: arcs between
    // the blocks' extreme positions, rounded down) instead of at nothing: an arc below it cannot be the
    // segment's maximum, so a pair that does not hold it never pays an fp64 division (a pair's own running
    // maximum took ~ln 64 record-breaking arcs per lane to get there); the maximum itself, and every tie of
    // it, passes as before (b >= L, screen margin 4e-6).  A pair without such an arc reports b = -1.
    float bf = __uint_as_float(Lseg[wk.seg]) * 0.999996f;
    if (!(bf > 0.f)) bf = -1.f;
    int bi = 0, bj = 0;
    const int jmax = n - j0 < PBS - 1 ? n - j0 : PBS - 1;      // last position of block J inside the segment
    for (int qj = 0; qj <= jmax; ++qj) {
      const double sj = wcx::readlane_f64(sJ, qj), wj = wcx::readlane_f64(wJ, qj);
      const int j = j0 + qj, a = j - i;
      if (i > n || a < minw || n - a < minw) continue;
      const double d = sj - si, wa = wj - wi;
      const float df = (float)d, waf = (float)wa;
      const float b32 = df * df * Wf * __builtin_amdgcn_rcpf(waf * (Wf - waf));
      if (b32 >= bf) {
        const double b = d * d / (wa * (W - wa) / W);
        if (b > bb) { bb = b; bi = i; bj = j; bf = (float)b * 0.999996f; }   // (j ascending: first j wins ties)
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double ob = wcx::shfl_xor_f64(bb, m);
      const int oi = __shfl_xor(bi, m, 64), oj = __shfl_xor(bj, m, 64);
      const bool take = ob > bb || (ob == bb && (oi < bi || (oi == bi && oj < bj)));
      if (take) { bb = ob; bi = oi; bj = oj; }
    }
    if (lane == 0) {
      res[it].b = bb; res[it].i = bi; res[it].j = bj;
      if (bb > 0.0) atomicMax(&bbits[wk.seg], (unsigned long long)__double_as_longlong(bb));
    }
  }
}
// among the pairs that reached the segment's maximum: the smallest (i, j)
__global__ __launch_bounds__(256) void k_cbs_pairtie(const WorkItem *__restrict__ work, unsigned int cap,
                                                     const unsigned int *__restrict__ count,
                                                     const ArcBest *__restrict__ res,
                                                     const unsigned long long *__restrict__ bbits,
                                                     unsigned long long *__restrict__ bij) {
  const unsigned int nitem = *count < cap ? *count : cap;
  for (unsigned int it = blockIdx.x * 256 + threadIdx.x; it < nitem; it += gridDim.x * 256) {
    const ArcBest r = res[it];
    const int s = work[it].seg;
    if (r.b > 0.0 && (unsigned long long)__double_as_longlong(r.b) == bbits[s])
      atomicMin(&bij[s], ((unsigned long long)(unsigned int)r.i << 32) | (unsigned int)r.j);
  }
}
__global__ __launch_bounds__(256) void k_cbs_pairfinal(int ns, const unsigned long long *__restrict__ bbits,
                                                       const unsigned long long *__restrict__ bij,
                                                       ArcBest *__restrict__ best, int *__restrict__ first) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s > ns) return;
  first[s] = s;
  if (s == ns) return;
  ArcBest b;
  b.b = bbits[s] ? __longlong_as_double((long long)bbits[s]) : -1.0;
  b.i = (int)(bij[s] >> 32); b.j = (int)(bij[s] & 0xffffffffull);
  best[s] = b;
}

__device__ __forceinline__ double it1tsq(double x, double a) {   // integral of 1/(t(1-t))^2 over [x, x+a]
  double y = x + a - 0.5;
  double r = 8.0 * y / (1.0 - 4.0 * y * y) + 2.0 * log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
  y = x - 0.5;
  r -= 8.0 * y / (1.0 - 4.0 * y * y) + 2.0 * log((1.0 + 2.0 * y) / (1.0 - 2.0 * y));
  return r;
}

// per segment: reduce its stripes (items [first[s], first[s+1])), t^2 of the best arc, the weighted
// short-arc fraction delta, and the grid of x values of the tail-probability integral
__global__ __launch_bounds__(128) void k_cbs_arcfinish(const ArcBest *__restrict__ best,
                                                       const int *__restrict__ first,
                                                       const SegIn *__restrict__ segs,
                                                       const double *__restrict__ Wp,
                                                       SegOut *__restrict__ so, int kmax, int ngrid,
                                                       double alpha, double *__restrict__ tx) {
  const int s = blockIdx.x;
  // best stripe of the segment; ties -> the first stripe (stripes ascend in i): deterministic
  __shared__ double rb[128];
  __shared__ int rq[128];
  __shared__ double red[2];
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
  const SegIn sg = segs[s];
  const int n = sg.n;
  // weighted delta (DNAcopy getmncwt, recalled): smallest weight of an arc of kmax + 1 points,
  // wrap-around arcs (= complements of arcs of n - kmax - 1 points) included, over W
  double dmin = HUGE_VAL;
  const double W = so[s].W;
  if (sg.hybrid) {
    const int j = kmax + 1, nmj = n - j;
    for (int i = threadIdx.x; i + j <= n; i += blockDim.x) {
      const double wa = seg_W(Wp, sg, W, i + j) - seg_W(Wp, sg, W, i);
      dmin = wa < dmin ? wa : dmin;
    }
    if (nmj >= 1)
      for (int i = threadIdx.x; i + nmj <= n; i += blockDim.x) {
        const double wa = W - (seg_W(Wp, sg, W, i + nmj) - seg_W(Wp, sg, W, i));
        dmin = wa < dmin ? wa : dmin;
      }
    dmin = block_min_d<128>(dmin, red);
  }
  if (threadIdx.x == 0) {
    const double bb = rb[0];
    int bi = 0, bj = 0;
    if (bb > 0.0) { bi = best[rq[0]].i; bj = best[rq[0]].j; }
    SegOut o = so[s];
    if (bb > 0.0) {
      double t = o.tss;
      if (t <= bb + 1e-4) t = bb + 1.0;            // DNAcopy (recalled)
      o.ostat = bb / ((t - bb) / (n - 2.0));       // t^2 of the best arc
      o.bi = bi; o.bj = bj; o.valid = 1;
    }
    o.delta = sg.hybrid ? dmin / W : 0.0;
    so[s] = o;
  }
  __syncthreads();
  // tail probability grid (Siegmund approximation; DNAcopy tailp): x_i = b / sqrt(m t (1 - t))
  const SegOut o = so[s];
  if (!o.valid || !sg.hybrid) return;
  const double delta = o.delta;
  const double dincr = (0.5 - delta) / ngrid;
  const double bsqrtm = sqrt(o.ostat) / sqrt((double)n);
  // Most tested segments are noise: their tail probability is ~0.1 - 1, four orders of magnitude
  // above alpha, and all the decision needs is "p1 > alpha".  nu(x) >= exp(-0.583 x) / 2 for every
  // x > 0 (the small-x expansion halved; tests/test_oracle_cbs.py checks it against the series on a
  // dense grid), so the same quadrature with that cheap bound is a LOWER bound of p1: when it
  // already exceeds alpha the series (up to 1.3e5 erfc terms per grid point) is not evaluated.
  double acc = 0.0;
  for (int i = threadIdx.x; i < ngrid; i += blockDim.x) {
    const double t = 0.5 - 0.5 * dincr - i * dincr;
    const double x = bsqrtm / sqrt(t * (1.0 - t));
    tx[(int64_t)s * ngrid + i] = x;
    const double vlo = 0.5 * exp(-0.583 * x);
    acc += vlo * vlo * it1tsq(0.5 - dincr - i * dincr, dincr);
  }
  acc = block_sum_d<128>(acc, red);
  if (threadIdx.x == 0) {
    const double b = sqrt(o.ostat);
    so[s].pval_lo = 9.973557e-2 * b * b * b * exp(-b * b / 2.0) * acc;
    (void)alpha;
  }
}

// nu(x) series: one workgroup per (grid point, segment), threads over the terms
//   ln nu = ln 2 - 2 ln x - 2 sum_{k>=1} Phi(-x sqrt(k)/2) / k      (terms vanish once x sqrt(k)/2 > 8.5)
__global__ __launch_bounds__(256) void k_nu_series(const double *__restrict__ xs,
                                                   const SegIn *__restrict__ segs,
                                                   const SegOut *__restrict__ so, int ngrid,
                                                   double alpha, double *__restrict__ out) {
  const int s = blockIdx.y;
  if (!so[s].valid || !segs[s].hybrid || so[s].pval_lo > alpha) return;   // decided by the bound
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

// P(max over arcs with delta <= weight fraction <= 1-delta of the CBS statistic >= b), Gaussian null
__global__ void k_cbs_tailp(const double *__restrict__ nu, const SegIn *__restrict__ segs,
                            SegOut *__restrict__ so, int ngrid, double alpha) {
  const int s = blockIdx.x;
  if (threadIdx.x != 0 || !so[s].valid || !segs[s].hybrid) return;
  if (so[s].pval_lo > alpha) { so[s].pval1 = -so[s].pval_lo; return; }   // negative = "at least this"
  const double b = sqrt(so[s].ostat);
  const double delta = so[s].delta;
  const double dincr = (0.5 - delta) / ngrid;
  double acc = 0.0, tl = 0.5 - dincr;
  for (int i = 0; i < ngrid; ++i) {
    const double v = nu[(int64_t)s * ngrid + i];
    acc += v * v * it1tsq(tl, dincr);
    tl -= dincr;
  }
  so[s].pval1 = 9.973557e-2 * b * b * b * exp(-b * b / 2.0) * acc;
}

// ---- (4) permutations ------------------------------------------------------------------------
// Arc weights of a hybrid job, from the fp64 prefix weights, rounded once to fp32:
//   q[a - 2][i] = W / (w_a (W - w_a)) for the arc (i, i + a], a = 2 .. KMAXC (0 where the arc does
//   not exist), then qc[c][i] likewise for the arcs (i, i + n - c] whose COMPLEMENT has c = 2 ..
//   KMAXC points (i = 0 .. c; 0 where such an arc is also a short arc or does not exist).
// The weights are not permuted (DNAcopy permutes the data under fixed weights), so the table is
// shared by all permutations of the job.
__global__ void k_cbs_arcweights(const double *__restrict__ Wp_all, const PermJob *__restrict__ jobs,
                                 int minw, float *__restrict__ qtab) {
  const PermJob jb = jobs[blockIdx.y];
  if (jb.mode != 0) return;
  const int n = jb.n;
  const int npad = (n + NTP - 1) / NTP * NTP;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  const double *Wp = Wp_all + jb.lo;
  const double W = jb.W;
  const int amax_all = n - minw;
  const int a_hi = KMAXC < amax_all ? KMAXC : amax_all;
  float *q = qtab + jb.qoff;
  auto Wat = [&](int p) { return p < n ? Wp[p] : W; };
  const double w0 = Wat(i < n ? i : n);
  for (int a = 2; a <= KMAXC; ++a) {
    float v = 0.f;
    if (a >= minw && a <= a_hi && i + a <= n) {
      const double wa = Wat(i + a) - w0;
      v = (float)(W / (wa * (W - wa)));
    }
    q[(size_t)(a - 2) * npad + i] = v;
  }
  if (i < (KMAXC + 1) * QC_W) {
    const int c = i / QC_W, s = i % QC_W, a = n - c;
    float v = 0.f;
    if (c >= minw && c >= 2 && s <= c && a > a_hi && a >= minw) {
      const double wa = Wat(s + a) - Wat(s);
      v = (float)(W / (wa * (W - wa)));
    }
    q[(size_t)(KMAXC - 1) * npad + i] = v;
  }
}

// exceedances recorded in the 256-permutation chunks BEFORE permutation p's chunk (all = every
// chunk): once they pass the budget the sequential rule has stopped before p
__device__ __forceinline__ bool budget_spent(const unsigned int *cnt, int slot, int p, int budget,
                                             bool all, int lane) {
  const int lim = all ? NCH : p / PCHUNK;
  int c = 0;
  if (lane < lim)
    c = (int)__hip_atomic_load(&cnt[(size_t)slot * NCH + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return wcx::wave_sum_i(c) > budget;
}
__device__ __forceinline__ void record_exceed(unsigned int *bits, unsigned int *cnt, int nw, int slot,
                                              int p) {
  atomicOr(&bits[(size_t)slot * nw + (p >> 5)], 1u << (p & 31));
  atomicAdd(&cnt[(size_t)slot * NCH + p / PCHUNK], 1u);
}

// Hybrid statistic, fp32 screen.  One workgroup per permutation of one job.  y = centred residual *
// sqrt(w), rw = sqrt(w).  BIG: the permuted series lives in global scratch (n > LDS capacity; slot =
// blockIdx.x, the grid is then limited and strides over the permutations).
template <bool BIG>
__global__ __launch_bounds__(NTP) void k_cbs_perm_hyb(const float *__restrict__ y_all,
                                                      const float *__restrict__ rw_all,
                                                      const PermJob *__restrict__ jobs, int p0, int p1,
                                                      int slot_stride, const float *__restrict__ qtab,
                                                      float *__restrict__ big_scr,
                                                      unsigned int *__restrict__ bits,
                                                      unsigned int *__restrict__ cnt, int nw,
                                                      uint2 *__restrict__ flags,
                                                      unsigned int *__restrict__ nflag) {
  extern __shared__ float ldsf[];
  __shared__ double redd[NTP / 64];
  __shared__ int s_skip;
  const int tid = threadIdx.x;
  const PermJob jb = jobs[blockIdx.y];
  const int n = jb.n;
  const int npad = (n + NTP - 1) / NTP * NTP;
  const float *y = y_all + jb.lo, *rw = rw_all + jb.lo;
  float *sk = BIG ? big_scr + (size_t)blockIdx.x * slot_stride : ldsf;   // [npad + 32]
  const float *q = qtab + jb.qoff;
  const float *qc = q + (size_t)(KMAXC - 1) * npad;
  for (int p = p0 + (int)blockIdx.x; p < p1; p += gridDim.x) {
    __syncthreads();
    if (tid < 64) {
      const bool sp = budget_spent(cnt, jb.slot, p, jb.budget, false, tid);
      if (tid == 0) s_skip = sp ? 1 : 0;
    }
    __syncthreads();
    if (s_skip) return;                        // uniform; later permutations of this workgroup too
    Feistel f;
    f.init(n, jb.key, p);
    // permuted, weighted series (raw values; no prefix scan: arc sums are local, see below)
    double tsum = 0.0;
    for (int i = tid; i < npad + 32; i += NTP) {
      float v = 0.f;
      if (i < n) v = rw[i] * y[f.at(i)];
      sk[i] = v;
      tsum += (double)v;
    }
    const double T = block_sum_d<NTP>(tsum, redd);       // (barrier: sk is complete)
    // re-centre on the permuted series' weighted mean T / W: v_i - (T / W) w_i (then an arc and its
    // complement carry the same statistic and the wrap-around arcs are plain short sums)
    {
      const float mf = (float)(T / jb.W);
      for (int i = tid; i < n; i += NTP) { const float r = rw[i]; sk[i] -= mf * (r * r); }
    }
    __syncthreads();
    // short arcs (i, i + a], a = 2 .. 25: a thread owns 8 consecutive start positions, builds the
    // 33 LOCAL prefix sums of the 32 values that follow in registers (error bounded by the window,
    // not by the series) and multiplies d^2 by the job's arc weight (0 for arcs that do not exist);
    // two arcs per packed fp32 instruction
    f32x2 bm = {0.f, 0.f};
    for (int i0 = tid * 8; i0 < n; i0 += 8 * NTP) {
      float sv[8 + KMAXC];
      sv[0] = 0.f;
#pragma unroll
      for (int t4 = 0; t4 < 8; ++t4) {
        const float4 r = *reinterpret_cast<const float4 *>(sk + i0 + 4 * t4);
        sv[4 * t4 + 1] = sv[4 * t4] + r.x;
        sv[4 * t4 + 2] = sv[4 * t4 + 1] + r.y;
        sv[4 * t4 + 3] = sv[4 * t4 + 2] + r.z;
        sv[4 * t4 + 4] = sv[4 * t4 + 3] + r.w;
      }
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
          float dx = sv[j + a] - sv[j], dy = sv[j + 1 + a] - sv[j + 1];
          asm volatile("" : "+v"(dx), "+v"(dy));
          f32x2 d = {dx, dy};
          const f32x2 qq = {qv[j], qv[j + 1]};
          d = d * d * qq;
          bm.x = __builtin_fmaxf(bm.x, d.x);
          bm.y = __builtin_fmaxf(bm.y, d.y);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float bmax = bm.x > bm.y ? bm.x : bm.y;
    // arcs whose complement is short: (minus) the wrap-around sum of the first s and last c - s values
    if (tid < (KMAXC + 1) * QC_W) {
      const float qv = qc[tid];
      if (qv > 0.f) {
        const int c = tid / QC_W, s = tid % QC_W;
        float h = 0.f, tl = 0.f;
        for (int u = 0; u < s; ++u) h += sk[u];
        for (int u = n - (c - s); u < n; ++u) tl += sk[u];
        const float d = h + tl;
        const float b = d * d * qv;
        bmax = b > bmax ? b : bmax;
      }
    }
    const double b = block_max_d<NTP>((double)bmax, redd);
    if (tid == 0) {
      const double sb = sqrt(b);
      const double tssp = jb.tss - T * T / jb.W, dts = jb.eT * fabs(T);
      const double bthr = jb.cthr * tssp;
      const double sq = bthr > 0.0 ? sqrt(bthr) : 0.0;
      const double esq = jb.D + 4.8e-7 * sb + (bthr > 0.0 ? jb.cthr * dts / (2.0 * sq) : HUGE_VAL);
      const double gq = tssp - 1e-4 - dts;
      const double sqg = gq > 0.0 ? sqrt(gq) : 0.0;
      if (sb + esq >= sqg || (sb + esq >= sq && sb - esq < sq)) {
        const unsigned int at = atomicAdd(nflag, 1u);
        flags[at] = make_uint2(blockIdx.y, (unsigned int)p);
      } else if (sb - esq >= sq) {
        record_exceed(bits, cnt, nw, jb.slot, p);
      }
    }
  }
}

// bss -> t^2 with DNAcopy's guard (recalled): tss <= bss + 1e-4 is replaced by bss + 1
__device__ __forceinline__ double stat_from_bss(double bss, double tss, int n) {
  const double t = tss <= bss + 1e-4 ? bss + 1.0 : tss;
  return bss / ((t - bss) / (n - 2.0));
}

// Flagged permutations of hybrid jobs again, everything in fp64 (the permuted series in a global
// scratch slot per workgroup).  Rare: speed is irrelevant, the arithmetic is the plain statement.
__global__ __launch_bounds__(256) void k_cbs_perm_exact(const double *__restrict__ yd_all,
                                                        const double *__restrict__ w_all,
                                                        const double *__restrict__ Wp_all,
                                                        const PermJob *__restrict__ jobs,
                                                        const uint2 *__restrict__ flags,
                                                        const unsigned int *__restrict__ nflag,
                                                        double *__restrict__ scr, int slot_stride,
                                                        int minw, unsigned int *__restrict__ bits,
                                                        unsigned int *__restrict__ cnt, int nw) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const unsigned int nf = *nflag;
  double *v = scr + (size_t)blockIdx.x * slot_stride;
  for (unsigned int fi = blockIdx.x; fi < nf; fi += gridDim.x) {
    const uint2 fl = flags[fi];
    const PermJob jb = jobs[fl.x];
    const int n = jb.n, p = (int)fl.y;
    const double *yd = yd_all + jb.lo, *w = w_all + jb.lo, *Wp = Wp_all + jb.lo;
    const double W = jb.W;
    auto Wat = [&](int q) { return q < n ? Wp[q] : W; };
    Feistel f;
    f.init(n, jb.key, p);
    __syncthreads();
    double tsum = 0.0;
    for (int i = tid; i < n; i += 256) {
      const double t = sqrt(w[i]) * yd[f.at(i)];
      v[i] = t;
      tsum += t;
    }
    const double T = block_sum_d<256>(tsum, red);
    {
      const double m = T / W;
      for (int i = tid; i < n; i += 256) v[i] -= m * w[i];     // (a thread's own elements)
    }
    __threadfence_block();
    __syncthreads();
    const int amax_all = n - minw;
    const int a_hi = KMAXC < amax_all ? KMAXC : amax_all;
    double bmax = 0.0;
    for (int i = tid; i < n; i += 256) {
      double s = 0.0;
      for (int a = 1; a <= a_hi && i + a <= n; ++a) {
        s += v[i + a - 1];
        if (a >= minw) {
          const double wa = Wat(i + a) - Wat(i);
          const double b = s * s / (wa * (W - wa) / W);
          bmax = b > bmax ? b : bmax;
        }
      }
    }
    for (int t = tid; t < (KMAXC + 1) * QC_W; t += 256) {
      const int c = t / QC_W, s = t % QC_W, a = n - c;
      if (c >= minw && c >= 2 && s <= c && a > a_hi && a >= minw) {
        double h = 0.0, tl = 0.0;
        for (int u = 0; u < s; ++u) h += v[u];
        for (int u = n - (c - s); u < n; ++u) tl += v[u];
        const double d = h + tl;
        const double wa = Wat(s + a) - Wat(s);
        const double b = d * d / (wa * (W - wa) / W);
        bmax = b > bmax ? b : bmax;
      }
    }
    const double b = block_max_d<256>(bmax, red);
    if (tid == 0 && jb.thr <= stat_from_bss(b, jb.tss - T * T / W, n)) record_exceed(bits, cnt, nw, jb.slot, p);
  }
}

// n <= nmin: maximum over ALL arcs of the permuted series, fp64, one workgroup per permutation.
__global__ __launch_bounds__(256) void k_cbs_perm_full(const double *__restrict__ yd_all,
                                                       const double *__restrict__ w_all,
                                                       const double *__restrict__ Wp_all,
                                                       const PermJob *__restrict__ jobs, int p0, int p1,
                                                       int minw, unsigned int *__restrict__ bits,
                                                       unsigned int *__restrict__ cnt, int nw) {
  __shared__ double red[4];
  __shared__ double v[256], S[257], Wl[257];
  __shared__ int s_skip;
  const int tid = threadIdx.x;
  const PermJob jb = jobs[blockIdx.y];
  const int n = jb.n, p = p0 + (int)blockIdx.x;
  if (p >= p1) return;
  if (tid < 64) {
    const bool sp = budget_spent(cnt, jb.slot, p, jb.budget, false, tid);
    if (tid == 0) s_skip = sp ? 1 : 0;
  }
  __syncthreads();
  if (s_skip) return;
  const double *yd = yd_all + jb.lo, *w = w_all + jb.lo, *Wp = Wp_all + jb.lo;
  const double W = jb.W;
  Feistel f;
  f.init(n, jb.key, p);
  double vt = 0.0;
  if (tid < n) vt = sqrt(w[tid]) * yd[f.at(tid)];
  const double T = block_sum_d<256>(vt, red);
  if (tid < n) v[tid] = vt - T / W * w[tid];                 // re-centred on the permuted weighted mean
  if (tid <= n) Wl[tid] = tid < n ? Wp[tid] : W;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    S[0] = 0.0;
    for (int i = 0; i < n; ++i) { s += v[i]; S[i + 1] = s; }
  }
  __syncthreads();
  double bmax = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = i + minw + tid; j <= n && n - (j - i) >= minw; j += 256) {
      const double d = S[j] - S[i], wa = Wl[j] - Wl[i];
      const double b = d * d / (wa * (W - wa) / W);
      bmax = b > bmax ? b : bmax;
    }
  const double b = block_max_d<256>(bmax, red);
  if (tid == 0 && jb.thr <= stat_from_bss(b, jb.tss - T * T / W, n)) record_exceed(bits, cnt, nw, jb.slot, p);
}

// Two-sample edge test (DNAcopy wtpermp, recalled): the sub-series of n points (centred on the
// SEGMENT mean), statistic |sum over the LAST m1 positions of sqrt(w_i) y[pi(i)] / rm1 - xbar|.
// fp64, one wave per permutation; the total count decides, so any chunk's exceedances may stop it.
__global__ __launch_bounds__(256) void k_cbs_perm_edge(const double *__restrict__ yd_all,
                                                       const double *__restrict__ w_all,
                                                       const PermJob *__restrict__ jobs, int p0, int p1,
                                                       unsigned int *__restrict__ bits,
                                                       unsigned int *__restrict__ cnt, int nw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const PermJob jb = jobs[blockIdx.y];
  const int p = p0 + (int)blockIdx.x * 4 + wave;
  if (p >= p1) return;
  if (budget_spent(cnt, jb.slot, p, jb.budget, true, lane)) return;
  const int n = jb.n, m1 = jb.m1;
  const double *yd = yd_all + jb.lo, *w = w_all + jb.lo;
  Feistel f;
  f.init(n, jb.key, p);
  double acc = 0.0;
  for (int t = lane; t < m1; t += 64) {
    const int i = n - m1 + t;
    acc += sqrt(w[i]) * yd[f.at(i)];
  }
  const double xsum = wcx::wave_sum(acc);
  if (lane == 0 && jb.thr <= fabs(xsum / jb.W - jb.tss)) record_exceed(bits, cnt, nw, jb.slot, p);
}

// Upper bound of the hybrid permutation statistic over ALL permutations of the series, from O(n)
// sums and the kmax largest |y|.  With y_j = r_j (x_j - mean), r_j = sqrt(w_j), a permutation pi
// puts v_i = r_i y_pi(i) at position i; T = sum v_i, |T| <= sqrt(sum (r_i - rbar)^2 sum y^2) +
// rbar |sum y| (Cauchy-Schwarz); the re-centred arc sum is sum_arc v_i - (T / W) w_arc, so for an
// arc of a points (or the wrap-around arc of a points whose complement it is)
//   |d| <= r_max Y_a + |T| a w_max / W,   Y_a = the a largest |y|;   tss' = sum y^2 - T^2 / W;
// the arc weight factor W / (w_a (W - w_a)) is at most its value at the end points of
// [a w_min, a w_max] (convex).  A strongly significant segment has an observed statistic far above
// this bound; its permutations cannot produce a single exceedance and are not run.
static double short_arc_bound(const double *x, const double *w, int n, int minw, int kmax, double tss) {
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
  const double tmax = sqrt(srr * syy) + rbar * fabs(sy);
  const int a_hi = std::min(kmax, n - minw);
  if (a_hi < minw) return HUGE_VAL;
  std::partial_sort(ay.begin(), ay.begin() + a_hi, ay.end(), std::greater<double>());
  const double rmax = sqrt(wmax);
  double Y = 0, bmax = 0;
  for (int a = 1; a <= a_hi; ++a) {
    Y += ay[(size_t)a - 1];
    if (a < minw) continue;
    if (a * wmax >= W) return HUGE_VAL;
    const double d = rmax * Y + tmax * a * wmax / W;
    const double den = std::min(a * wmin * (W - a * wmin), a * wmax * (W - a * wmax)) / W;
    bmax = std::max(bmax, d * d / den);
  }
  const double tss_min = std::min(tss, syy) - tmax * tmax / W;
  if (!(tss_min > bmax + 1e-4)) return HUGE_VAL;
  return bmax / ((tss_min - bmax) / (n - 2.0));
}

// ---- sequential boundary (Venkatraman & Olshen 2007, section 2.2; DNAcopy getbdry, recalled) ----
// (long double: getbdry() divides by differences of pexceed values ~1e-5 apart; with double lgamma
// the last digits of lgamma(10001) ~ 8e4 decide single stopping points)
static long double lchoose(long double n, long double k) {
  if (k < 0 || k > n) return -HUGE_VALL;
  return lgammal(n + 1.0L) - lgammal(k + 1.0L) - lgammal(n - k + 1.0L);
}
// ib[r] = smallest number of permutations at which "at most r exceedances so far" has probability
// <= eta0 when n1s exceedances are scattered uniformly over nperm permutations.  The hypergeometric
// probabilities advance draw by draw (pmf recurrence, all terms positive).
static void etabdry(int nperm, double eta0, int n1s, int *ib) {
  std::vector<double> pm((size_t)n1s + 1, 0.0);
  pm[0] = 1.0;
  int k = 0;
  for (int i = 1; i <= nperm && k < n1s; ++i) {
    const double rem = (double)(nperm - (i - 1));
    const int xmax = std::min(i, n1s);
    for (int x = xmax; x >= 1; --x)
      pm[x] = pm[x] * (1.0 - (n1s - x) / rem) + pm[x - 1] * ((n1s - x + 1) / rem);
    pm[0] = pm[0] * (1.0 - n1s / rem);
    double cdf = 0.0;
    for (int x = 0; x <= k; ++x) cdf += pm[x];
    if (cdf <= eta0) ib[k++] = i;
  }
  while (k < n1s) ib[k++] = nperm;
}
static double pexceed(int nperm, int n1s, const int *b) {
  const long double lc = lchoose(nperm, n1s);
  long double p = expl(lchoose(nperm - b[0], n1s) - lc);
  if (n1s >= 2) p += expl(logl((long double)b[0]) + lchoose(nperm - b[1], n1s - 1) - lc);
  if (n1s >= 3) {
    const long double t = lchoose(nperm - b[2], n1s - 2) - lc;
    p += expl(logl((long double)b[0]) + logl(b[0] - 1.0L) - logl(2.0L) + t);
    p += expl(logl((long double)b[0]) + logl((long double)(b[1] - b[0])) + t);
  }
  for (int i = 4; i <= n1s; ++i) {
    const long double n1 = b[i - 4], n2 = b[i - 3], n3 = b[i - 2];
    const long double t = lchoose(nperm - b[i - 1], n1s - i + 1) - lc;
    p += expl(lchoose(n1, i - 1) + t);
    p += expl(lchoose(n1, i - 2) + logl(n3 - n1) + t);
    p += expl(lchoose(n1, i - 3) + logl(n2 - n1) + logl(n3 - n2) + t);
    if (n2 - n1 > 1.0L) p += expl(lchoose(n1, i - 3) + logl(n2 - n1) - logl(2.0L) + logl(n2 - n1 - 1.0L) + t);
  }
  return (double)p;
}
static void getbdry(double eta, int nperm, int max_ones, std::vector<int> &bdry) {
  bdry.assign((size_t)max_ones * (max_ones + 1) / 2, 0);
  bdry[0] = nperm - (int)(nperm * eta);
  double eta0 = eta;
  size_t l = 1;
  for (int j = 2; j <= max_ones; ++j) {
    int *b = bdry.data() + l;
    double etahi = eta0 * 1.1;
    etabdry(nperm, etahi, j, b);
    double phi = pexceed(nperm, j, b);
    double etalo = eta0 * 0.25;
    etabdry(nperm, etalo, j, b);
    double plo = pexceed(nperm, j, b);
    for (int it = 0; (etahi - etalo) / etalo > 1e-2 && it < 200 && phi != plo; ++it) {
      eta0 = etalo + (etahi - etalo) * (eta - plo) / (phi - plo);
      etabdry(nperm, eta0, j, b);
      const double pexcd = pexceed(nperm, j, b);
      if (pexcd > eta) { etahi = eta0; phi = pexcd; }
      else { etalo = eta0; plo = pexcd; }
    }
    l += (size_t)j;
  }
}
constexpr int MAX_ONES_CAP = 128;   // beyond: no early "significant" stop (see cbs_boundary)
static std::mutex g_bdry_mu;
static std::map<std::pair<int, int>, std::vector<int>> g_bdry;
// the nrejc + 1 stopping points of a test with budget nrejc
static std::vector<int> cbs_boundary(double alpha, int nperm, int nrejc) {
  const int max_ones = (int)floor(nperm * alpha) + 1;
  if (max_ones > MAX_ONES_CAP || nrejc + 1 > max_ones) {
    // alpha > ~1.3 %: the table costs O(max_ones^2 nperm) to build; such tests run all their
    // permutations instead (the boundary only ever saves work and errs with probability <= eta)
    static bool warned = false;
    if (!warned) {
      warned = true;
      fprintf(stderr, "[wcx_cbs] alpha = %g: sequential boundary not tabulated beyond %d exceedances; "
                      "significant tests run all %d permutations\n", alpha, MAX_ONES_CAP - 1, nperm);
    }
    return std::vector<int>((size_t)nrejc + 1, nperm);
  }
  std::lock_guard<std::mutex> g(g_bdry_mu);
  auto key = std::make_pair(nperm, max_ones);
  auto it = g_bdry.find(key);
  if (it == g_bdry.end()) {
    std::vector<int> t;
    getbdry(0.05, nperm, max_ones, t);
    it = g_bdry.emplace(key, std::move(t)).first;
  }
  const size_t o = (size_t)nrejc * (nrejc + 1) / 2;
  return std::vector<int>(it->second.begin() + o, it->second.begin() + o + nrejc + 1);
}

// ------------------------------------------------------------------ host side
struct CbsParams {
  double alpha;
  int nperm = 10000, kmax = 25, nmin = 200, minw = 2, ngrid = 100;
  unsigned long long seed;
};

static unsigned long long test_key(unsigned long long seed, int chrom, int lo, int hi, int kind) {
  unsigned long long k = mix64(seed);
  k = mix64(k ^ (unsigned long long)chrom);
  k = mix64(k ^ (((unsigned long long)lo << 32) | (unsigned long long)(unsigned int)hi));
  return mix64(k ^ (unsigned long long)kind);
}

// device arena of one call, grown on demand
struct Arena {
  char *base = nullptr;
  size_t off = 0;
  template <typename T> T *take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T *p = reinterpret_cast<T *>(base + off);
    off += count * sizeof(T);
    return p;
  }
};

constexpr int LDS_KEYS_MAX = 32768;      // n above this keeps the permuted series in global scratch
constexpr int BIG_GRID = 512;
constexpr int EXACT_GRID = 256;
constexpr size_t FLAG_CAP = (size_t)1 << 21;   // flagged (job, permutation) pairs per launch group
constexpr int TRACE_W = 20;

struct JobResult { int decided = 0, significant = 0, nrej = 0, np = 0; };

// walk the exceedance bits of permutations [0, upto) under the sequential rule
static void eval_sequential(const unsigned int *bits, int upto, int nperm, int nrejc,
                            const std::vector<int> &block, JobResult &r) {
  int nrej = 0;
  for (int np = 1; np <= upto; ++np) {
    if ((bits[(np - 1) >> 5] >> ((np - 1) & 31)) & 1u) ++nrej;
    if (nrej > nrejc) { r.decided = 1; r.significant = 0; r.nrej = nrej; r.np = np; return; }
    if (np >= block[(size_t)nrej]) { r.decided = 1; r.significant = 1; r.nrej = nrej; r.np = np; return; }
  }
  r.nrej = nrej; r.np = upto;
  if (upto >= nperm) { r.decided = 1; r.significant = 1; }
}

// ---- series assembly on the device (wcx_cbs_batch_dev): the NA-free compaction of CBS.R:41-42 /
// DNAcopy's is.finite() of every (sample, chromosome), straight from the per-bin vectors in HBM
struct ChrOff { int64_t off[32]; };
__device__ __forceinline__ bool cbs_is_drop(double v) { return v == 0.0 || !(fabs(v) < HUGE_VAL); }

// cnt[sample * n_chr + chr] = number of kept bins
// cnt[n_samples * n_chr] = number of +-inf values anywhere (dropped from the series but NOT "NA" to
// the post-processing of CBS.R:84-129: with any of them the host falls back to the full r / w)
__global__ __launch_bounds__(256) void k_cbs_count(const double *__restrict__ r, int64_t n_bins, ChrOff co,
                                                   int n_chr, int n_samples, int *__restrict__ cnt) {
  __shared__ int red[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const double *rr = r + (int64_t)s * n_bins + co.off[c];
  const int nall = (int)(co.off[c + 1] - co.off[c]);
  int m = 0, ninf = 0;
  for (int i = threadIdx.x; i < nall; i += 256) {
    const double v = rr[i];
    m += cbs_is_drop(v) ? 0 : 1;
    ninf += (v == HUGE_VAL || v == -HUGE_VAL) ? 1 : 0;
  }
  m = wcx::wave_sum_i(m);
  if (__any(ninf != 0)) { ninf = wcx::wave_sum_i(ninf); if ((threadIdx.x & 63) == 0) atomicAdd(&cnt[n_samples * n_chr], ninf); }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) cnt[s * n_chr + c] = red[0] + red[1] + red[2] + red[3];
}

// device -> PINNED HOST copy by a small grid: the runtime's own device-to-host copy is a blit kernel
// that fills every CU while it moves data at PCIe speed -- a kernel launched beside it waits (measured:
// k_cbs_prepare 0.2 -> 8.8 ms beside 600 MB of such copies).  64 workgroups keep the link busy and
// leave the chip to the round's kernels.
__global__ __launch_bounds__(256) void k_cbs_export(const uint4 *__restrict__ src, uint4 *__restrict__ dst_host,
                                                    int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
    dst_host[i] = src[i];
}

// x[lo + j] / w[lo + j] / pos[lo + j] = the j-th kept bin of the (sample, chromosome), in bin order; weight 0 -> 1
__global__ __launch_bounds__(256) void k_cbs_fill(const double *__restrict__ r, const double *__restrict__ w,
                                                  int64_t n_bins, ChrOff co, int n_chr,
                                                  const int64_t *__restrict__ lo, double *__restrict__ x,
                                                  double *__restrict__ xw, int *__restrict__ pos) {
  __shared__ int wtot[4];
  const int c = blockIdx.x, s = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t o = (int64_t)s * n_bins + co.off[c];
  const int nall = (int)(co.off[c + 1] - co.off[c]);
  int64_t at = lo[s * n_chr + c];
  for (int i0 = 0; i0 < nall; i0 += 256) {
    const int i = i0 + (int)threadIdx.x;
    const double v = i < nall ? r[o + i] : 0.0;
    const bool keep = i < nall && !cbs_is_drop(v);
    const unsigned long long m = __ballot(keep);
    __syncthreads();                                   // (wtot of the previous chunk fully read)
    if (lane == 0) wtot[wave] = __popcll(m);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int t = wtot[q]; before += q < wave ? t : 0; all += t; }
    if (keep) {
      const int64_t dst = at + before + __popcll(m & ((1ull << lane) - 1ull));
      const double wt = w[o + i];
      x[dst] = v;
      xw[dst] = wt == 0.0 ? 1.0 : wt;
      pos[dst] = i + 1;                                // 1-based bin within the chromosome (CBS.R:49)
    }
    at += all;
  }
}

// x | w of the listed ranges -> the SAME offsets of the pinned host arrays (the host's decisions read a
// series only once one of its segments is significant: ~5 % of them; exporting everything kept the
// link busy for 5 ms beside the first round and slowed that round's kernels)
struct RangeItem { int64_t lo; int32_t n, pad; };
__global__ __launch_bounds__(256) void k_cbs_export_ranges(const double *__restrict__ x, const double *__restrict__ w,
                                                           const RangeItem *__restrict__ items,
                                                           double *__restrict__ hx, double *__restrict__ hw) {
  const RangeItem it = items[blockIdx.x];
  for (int i = threadIdx.x; i < it.n; i += 256) { hx[it.lo + i] = x[it.lo + i]; hw[it.lo + i] = w[it.lo + i]; }
}

// CBS.R:122-127 weighted.mean of the final intervals: num = sum x w, den = sum w over the interval's
// kept bins IN ORDER (products and sums separately rounded, like the host loop it replaces: the file
// is compiled with -ffp-contract=off).  One WAVE per interval: 64 elements are loaded coalesced, the
// next 64 are in flight while the current ones are added one after the other out of the lanes
// (readlane; every lane carries the same running sums).  (One THREAD per interval walked its 16 k
// elements through 16 k dependent cache misses: 7 ms.)
__global__ __launch_bounds__(256) void k_cbs_interval_means(const double *__restrict__ x, const double *__restrict__ w,
                                                            const RangeItem *__restrict__ items, int n_items,
                                                            double *__restrict__ out) {
  // One wave per range; the sums run strictly in index order (the association the oracle uses).  The
  // products of 64 bins go through a wave-private LDS slice and every lane adds them from there
  // (broadcast reads, two dependent fp64 chains: ~10 cycles per bin; lane-to-lane broadcasts through
  // SGPRs cost ~8 x that, 0.77 ms for a whole-chromosome segment at 15 kb).
  __shared__ __attribute__((aligned(16))) double2 s_pw[4][2][64];   // [wave][buffer][bin] = (x w, w)
  const int q = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (q >= n_items) return;
  const RangeItem it = items[q];
  double num = 0.0, den = 0.0;
  double xv = lane < it.n ? x[it.lo + lane] : 0.0, wv = lane < it.n ? w[it.lo + lane] : 0.0;
  int buf = 0;
  for (int base = 0; base < it.n; base += 64, buf ^= 1) {
    const int nxt = base + 64 + lane;
    const double xn = nxt < it.n ? x[it.lo + nxt] : 0.0, wn = nxt < it.n ? w[it.lo + nxt] : 0.0;
    s_pw[wave][buf][lane] = make_double2(xv * wv, wv);
    __builtin_amdgcn_wave_barrier();
    const int cnt = it.n - base < 64 ? it.n - base : 64;
    const double2 *pw = s_pw[wave][buf];
    if (cnt == 64) {
#pragma unroll
      for (int j0 = 0; j0 < 64; j0 += 16) {
        double2 t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = pw[j0 + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) { num = num + t[u].x; den = den + t[u].y; }
      }
    } else {
      for (int j = 0; j < cnt; ++j) { const double2 t = pw[j]; num = num + t.x; den = den + t.y; }
    }
    xv = xn; wv = wn;
  }
  if (lane == 0) { out[2 * q] = num; out[2 * q + 1] = den; }
}

}  // namespace

extern "C" {

int wcx_cbs_getbdry(double eta, int nperm, int max_ones, int32_t *out) {
  WCX_ARG(out && eta > 0 && eta < 1 && nperm > 0 && nperm <= NCH * PCHUNK && max_ones >= 1 &&
          max_ones <= 1024, "bad parameters");
  std::vector<int> t;
  getbdry(eta, nperm, max_ones, t);
  for (size_t i = 0; i < t.size(); ++i) out[i] = t[i];
  return WCX_OK;
}

}  // extern "C"

// r / w: the per-bin vectors on the host.  d_r / d_w != NULL (wcx_cbs_batch_dev): the vectors are in
// HBM and r / w are PINNED host buffers, filled only if the data hold +-inf.  The series are then
// compacted on the device (no upload) and stay there; the host fetches a series' x | w when a decision
// first needs them (ensure_resident), the bin positions come down on the stream `aux` beside the
// rounds' kernels, and the wrap-up (CBS.R:84-129) takes its weighted means from a device kernel.
static int cbs_batch_impl(wcx_ctx *ctx, const double *r, const double *w, const double *d_r, const double *d_w,
                          hipStream_t aux, int n_samples, int64_t n_bins,
                          const int64_t *chr_off, int n_chr, double alpha, int64_t binsize, uint64_t seed,
                          double *out_seg, int cap, int *out_count) {
  WCX_ARG(ctx && r && w && chr_off && out_seg && out_count, "NULL argument");
  WCX_ARG(n_chr <= 31, "more than 31 chromosomes");
  WCX_ARG(n_samples > 0 && n_chr > 0 && alpha > 0 && alpha <= 1 && binsize > 0 && cap >= 0,
          "bad parameters");
  WCX_ARG(chr_off[n_chr] <= n_bins, "chr_off beyond n_bins");
  WCX_HIP(hipSetDevice(ctx->device));
  CbsParams P;
  P.alpha = alpha;
  P.seed = seed;
  hipStream_t st = ctx->stream;
  const bool strict = (ctx->debug_flags & 2) != 0;     // no t >= 7 / t^2 > 25 shortcuts (diagnostics)
  const bool tracing = (ctx->debug_flags & 128) != 0;
  ctx->cbs_trace.clear();
  const int nw = (P.nperm + 31) / 32;

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
  auto is_na = [](double v) { return v == 0.0 || v != v; };               // CBS.R:41 ratio == 0 -> NA; is.na()
  auto is_drop = [](double v) { return v == 0.0 || !std::isfinite(v); };  // DNAcopy segment(): is.finite()
  auto for_samples = [&](auto &&fn) {
    // (host threads for the per-sample assembly / wrap-up loops: up to 32, within what the machine has)
    static const int hw_threads = [] {
      const unsigned h = std::thread::hardware_concurrency();
      return (int)std::max(1u, std::min(32u, h ? h : 8u));
    }();
    const int nt = std::min(n_samples, hw_threads);
    if (nt <= 1) { for (int s = 0; s < n_samples; ++s) fn(s); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t] { for (int s = t; s < n_samples; s += nt) fn(s); });
    for (auto &x : th) x.join();
  };
  ChrOff co;
  for (int c = 0; c <= n_chr; ++c) co.off[c] = chr_off[c];
  int64_t *d_lo = nullptr;
  int n_inf = 0;            // device path: +-inf values in r (then the host needs r / w themselves)
  bool copies_queued = false;
  if (d_r) {
    void *scr2 = nullptr;
    const size_t nsc = cnt_sc.size();
    const int rc2 = wcx_scratch2(ctx, nsc * 16 + 64, &scr2);
    if (rc2) return rc2;
    d_lo = reinterpret_cast<int64_t *>(scr2);
    int *d_cnt = reinterpret_cast<int *>(d_lo + nsc);
    WCX_HIP(hipMemsetAsync(d_cnt + nsc, 0, 4, st));
    k_cbs_count<<<dim3((unsigned)n_chr, (unsigned)n_samples), 256, 0, st>>>(d_r, n_bins, co, n_chr, n_samples,
                                                                             d_cnt);
    WCX_HIP(hipGetLastError());
    std::vector<int> hc(nsc + 1);
    WCX_HIP(hipMemcpyAsync(hc.data(), d_cnt, (nsc + 1) * 4, hipMemcpyDeviceToHost, st));
    WCX_HIP(hipStreamSynchronize(st));
    for (size_t q = 0; q < nsc; ++q) cnt_sc[q] = hc[q];
    n_inf = hc[nsc];
  } else {
    for_samples([&](int s) {
      for (int c = 0; c < n_chr; ++c) {
        const double *rr = r + (int64_t)s * n_bins + chr_off[c];
        const int nall = (int)(chr_off[c + 1] - chr_off[c]);
        int64_t m = 0;
        for (int i = 0; i < nall; ++i) m += is_drop(rr[i]) ? 0 : 1;
        cnt_sc[(size_t)s * n_chr + c] = m;
      }
    });
  }
  int64_t total = 0;
  std::vector<int64_t> lo_sc(cnt_sc.size());
  for (size_t q = 0; q < cnt_sc.size(); ++q) { lo_sc[q] = total; total += cnt_sc[q]; }
  // x | w | 1-based bin index within the chromosome (CBS.R:49), in the context's pinned staging area
  // (each part padded to whole 16-byte words: the device path fills them with 16-byte stores)
  const int64_t totp = (total + 3) & ~(int64_t)3;
  void *hstage = nullptr;
  {
    const int rcs = wcx_host_scratch(ctx, (size_t)totp * 20 + 64, &hstage);
    if (rcs) return rcs;
  }
  double *hx = reinterpret_cast<double *>(hstage), *hw = hx + totp;
  int *hpos = reinterpret_cast<int *>(hw + totp);
  if (d_r && (ctx->debug_flags & 64)) {
    // (tests: on the device path the host copies exist only where ensure_resident put them; poisoned,
    //  a host read of a series nobody exported cannot go unnoticed -- the staging area is reused
    //  between calls and would otherwise still hold an earlier call's values)
    const double poison = __builtin_nan("");
    for (int64_t i = 0; i < 2 * totp; ++i) hx[i] = poison;
  }
  if (!d_r)
    for_samples([&](int s) {
      for (int c = 0; c < n_chr; ++c) {
        const int64_t o = (int64_t)s * n_bins + chr_off[c];
        const int nall = (int)(chr_off[c + 1] - chr_off[c]);
        int64_t at = lo_sc[(size_t)s * n_chr + c];
        for (int i = 0; i < nall; ++i) {
          const double v = r[o + i];
          if (is_drop(v)) continue;
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
  const double *dXs_all = nullptr, *dWs_all = nullptr;    // the compacted series in HBM (for the wrap-up)
  lap("series");
  int rc = wcx_timer_begin(ctx, "cbs");
  if (rc) return rc;
  if (N > 0) {
    int max_n = 0;
    for (const Series &se : series) max_n = std::max(max_n, se.n);
    const int max_segs = (int)series.size() + 8;          // active segments per round <= series
    const int max_jobs = max_segs * 2;                    // segmentation tests, then 2 edge tests each
    // ---- device arena
    const int npad_max = (max_n + NTP - 1) / NTP * NTP;
    const bool any_big = max_n > LDS_KEYS_MAX;
    const size_t max_items = (size_t)max_segs + (size_t)(N / 4) + 6144 + 64;
    const size_t max_blocks = (size_t)(N / PBS) + 2 * (size_t)max_segs + 64;
    const unsigned int work_cap = 1u << 22;                  // block pairs evaluated per round at most
    const size_t qtab_floats = (size_t)(KMAXC - 1) * ((size_t)N + (size_t)NTP * max_segs) +
                               (size_t)max_segs * (KMAXC + 1) * QC_W;
    const size_t need = (size_t)N * (8 * 5 + 4 * 3) + 4096 + (series.size() + 8) * sizeof(RangeItem) + (size_t)max_segs * (sizeof(SegIn) + sizeof(SegOut) + 8) +
                        (size_t)max_segs * P.ngrid * 16 + max_items * (sizeof(ArcItem) + sizeof(ArcBest)) +
                        (size_t)max_jobs * (2 * sizeof(PermJob) + (size_t)(nw + NCH) * 4) +
                        (any_big ? (size_t)BIG_GRID * (npad_max + 64) * 4 : 0) +
                        (size_t)EXACT_GRID * npad_max * 8 + FLAG_CAP * sizeof(uint2) + qtab_floats * 4 +
                        max_blocks * (sizeof(BlkStat) + 4) + (size_t)work_cap * (sizeof(WorkItem) + sizeof(ArcBest)) +
                        (size_t)max_segs * 32 + (1 << 16);
    void *scr = nullptr;
    rc = wcx_scratch(ctx, need, &scr);
    if (rc) return rc;
    Arena A;
    A.base = reinterpret_cast<char *>(scr);
    double *dX = A.take<double>(N), *dW = A.take<double>(N), *dS = A.take<double>(N), *dWp = A.take<double>(N);
    double *dYd = A.take<double>(N);
    dXs_all = dX; dWs_all = dW;
    int *dpos = A.take<int>((size_t)N + 4);
    RangeItem *drng = A.take<RangeItem>(series.size() + 8);
    float *dy = A.take<float>(N), *drw = A.take<float>(N);
    SegIn *dseg = A.take<SegIn>(max_segs);
    SegOut *dso = A.take<SegOut>(max_segs);
    int *dfirst = A.take<int>(max_segs + 1);
    double *dtx = A.take<double>((size_t)max_segs * P.ngrid), *dnu = A.take<double>((size_t)max_segs * P.ngrid);
    ArcItem *ditems = A.take<ArcItem>(max_items);
    ArcBest *dbest = A.take<ArcBest>(max_items);
    PermJob *djobs = A.take<PermJob>((size_t)max_jobs * 2);
    unsigned int *dbits = A.take<unsigned int>((size_t)max_jobs * nw);
    unsigned int *dcnt = A.take<unsigned int>((size_t)max_jobs * NCH + 64);
    unsigned int *dnflag = dcnt + (size_t)max_jobs * NCH;
    float *dbig = any_big ? A.take<float>((size_t)BIG_GRID * (npad_max + 64)) : nullptr;
    double *dexact = A.take<double>((size_t)EXACT_GRID * npad_max);
    uint2 *dflags = A.take<uint2>(FLAG_CAP);
    float *dq = A.take<float>(qtab_floats);
    BlkStat *dbs = A.take<BlkStat>(max_blocks);
    WorkItem *dwork = A.take<WorkItem>(work_cap);
    ArcBest *dres = A.take<ArcBest>(work_cap);
    int *dboff = A.take<int>(max_segs + 1);
    int *dbseg = A.take<int>(max_blocks);
    unsigned int *dL = A.take<unsigned int>(max_segs + 2);     // [ns] lower bounds | work count
    unsigned long long *dbbits = A.take<unsigned long long>(max_segs);
    unsigned long long *dbij = A.take<unsigned long long>(max_segs);
    static const bool no_prune = getenv("WCX_CBS_NOPRUNE") && atoi(getenv("WCX_CBS_NOPRUNE"));
    if (d_r) {
      WCX_HIP(hipMemcpyAsync(d_lo, lo_sc.data(), lo_sc.size() * 8, hipMemcpyHostToDevice, st));
      k_cbs_fill<<<dim3((unsigned)n_chr, (unsigned)n_samples), 256, 0, st>>>(d_r, d_w, n_bins, co, n_chr, d_lo,
                                                                            dX, dW, dpos);
      WCX_HIP(hipGetLastError());
      // the bin positions (wrap-up) come down on the copy stream beside the rounds' kernels
      WCX_HIP(hipEventRecord(ctx->ev_cbs_fill, st));
      WCX_HIP(hipStreamWaitEvent(aux, ctx->ev_cbs_fill, 0));
      // (x | w follow per series, on demand: ensure_resident below; arena slots and the staging area
      //  are padded to whole 16-byte words)
      k_cbs_export<<<16, 256, 0, aux>>>(reinterpret_cast<const uint4 *>(dpos), reinterpret_cast<uint4 *>(hpos),
                                        ((int64_t)N * 4 + 15) / 16);
      WCX_HIP(hipGetLastError());
      if (n_inf > 0) {
        const size_t rw_bytes = (size_t)n_samples * n_bins * 8;
        WCX_HIP(hipMemcpyAsync(const_cast<double *>(r), d_r, rw_bytes, hipMemcpyDeviceToHost, aux));
        WCX_HIP(hipMemcpyAsync(const_cast<double *>(w), d_w, rw_bytes, hipMemcpyDeviceToHost, aux));
      }
      copies_queued = true;
    } else {
      WCX_HIP(hipMemcpyAsync(dX, hx, (size_t)N * 8, hipMemcpyHostToDevice, st));
      WCX_HIP(hipMemcpyAsync(dW, hw, (size_t)N * 8, hipMemcpyHostToDevice, st));
    }
    const size_t lds_small = (size_t)(std::min(npad_max, LDS_KEYS_MAX) + 32) * 4;
    WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cbs_perm_hyb<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small));

    // One batch of permutation tests -> results.  seq[q] = the job's stopping points (modes 0 / 1).
    std::vector<JobResult> jres;
    std::vector<unsigned int> hbits;
    auto run_jobs = [&](std::vector<PermJob> &jobs, const std::vector<std::vector<int>> &seq) -> int {
      jres.assign(jobs.size(), JobResult());
      if (jobs.empty()) return WCX_OK;
      WCX_ARG(jobs.size() <= (size_t)max_jobs, "internal: permutation job table overflow");
      size_t qoff = 0;
      for (size_t q = 0; q < jobs.size(); ++q) {
        jobs[q].slot = (int)q;
        jobs[q].qoff = (int64_t)qoff;
        if (jobs[q].mode == 0)
          qoff += (size_t)(KMAXC - 1) * (size_t)((jobs[q].n + NTP - 1) / NTP * NTP) + (size_t)(KMAXC + 1) * QC_W;
      }
      WCX_ARG(qoff <= qtab_floats, "internal: arc-weight table overflow");
      WCX_HIP(hipMemcpyAsync(djobs, jobs.data(), jobs.size() * sizeof(PermJob), hipMemcpyHostToDevice, st));
      WCX_HIP(hipMemsetAsync(dbits, 0, jobs.size() * (size_t)nw * 4, st));
      WCX_HIP(hipMemsetAsync(dcnt, 0, jobs.size() * (size_t)NCH * 4, st));
      {
        int hyb_max_n = 0;
        for (const PermJob &jb : jobs) if (jb.mode == 0) hyb_max_n = std::max(hyb_max_n, jb.n);
        if (hyb_max_n > 0) {
          const int npad_sel = (hyb_max_n + NTP - 1) / NTP * NTP;
          k_cbs_arcweights<<<dim3((unsigned)(npad_sel / 256), (unsigned)jobs.size()), 256, 0, st>>>(
              dWp, djobs, P.minw, dq);
          WCX_HIP(hipGetLastError());
        }
      }
      hbits.resize(jobs.size() * (size_t)nw);
      std::vector<int> pending(jobs.size());
      for (size_t q = 0; q < jobs.size(); ++q) pending[q] = (int)q;
      const int stages[3] = {0, std::min(STAGE_A, P.nperm), P.nperm};
      for (int sg = 0; sg < 2 && !pending.empty(); ++sg) {
        const int p0 = stages[sg], p1 = stages[sg + 1];
        if (p1 <= p0) continue;
        // kernels take contiguous job ranges: group the pending jobs by kind into launches over
        // [first, last] runs of the (unsorted) job table via a compacted copy
        std::vector<PermJob> sel[4];     // 0 hybrid small, 1 hybrid big, 2 full, 3 edge
        for (int q : pending) {
          const PermJob &jb = jobs[(size_t)q];
          const int kind = jb.mode == 0 ? (jb.n > LDS_KEYS_MAX ? 1 : 0) : jb.mode == 1 ? 2 : 3;
          sel[kind].push_back(jb);
        }
        size_t joff = 0;
        PermJob *dsel = djobs + jobs.size();             // second half of the job table
        WCX_ARG(pending.size() + jobs.size() <= (size_t)max_jobs * 2, "internal: job table overflow");
        for (int kind = 0; kind < 4; ++kind) {
          std::vector<PermJob> &sj = sel[kind];
          if (sj.empty()) continue;
          // hybrid launches are cut so that their flag list cannot overflow
          const size_t per_launch = kind <= 1 ? std::max<size_t>(1, FLAG_CAP / (size_t)(p1 - p0)) : sj.size();
          for (size_t j0 = 0; j0 < sj.size(); j0 += per_launch) {
            const size_t nj = std::min(per_launch, sj.size() - j0);
            PermJob *dj = dsel + joff;
            WCX_HIP(hipMemcpyAsync(dj, sj.data() + j0, nj * sizeof(PermJob), hipMemcpyHostToDevice, st));
            joff += nj;
            if (kind <= 1) {
              WCX_HIP(hipMemsetAsync(dnflag, 0, 4, st));
              if (kind == 1)
                k_cbs_perm_hyb<true><<<dim3((unsigned)std::min(BIG_GRID, p1 - p0), (unsigned)nj), NTP, 0, st>>>(
                    dy, drw, dj, p0, p1, npad_max + 64, dq, dbig, dbits, dcnt, nw, dflags, dnflag);
              else
                k_cbs_perm_hyb<false><<<dim3((unsigned)(p1 - p0), (unsigned)nj), NTP, lds_small, st>>>(
                    dy, drw, dj, p0, p1, 0, dq, nullptr, dbits, dcnt, nw, dflags, dnflag);
              WCX_HIP(hipGetLastError());
              k_cbs_perm_exact<<<EXACT_GRID, 256, 0, st>>>(dYd, dW, dWp, dj, dflags, dnflag, dexact, npad_max,
                                                           P.minw, dbits, dcnt, nw);
            } else if (kind == 2) {
              k_cbs_perm_full<<<dim3((unsigned)(p1 - p0), (unsigned)nj), 256, 0, st>>>(dYd, dW, dWp, dj, p0, p1,
                                                                                      P.minw, dbits, dcnt, nw);
            } else {
              k_cbs_perm_edge<<<dim3((unsigned)((p1 - p0 + 3) / 4), (unsigned)nj), 256, 0, st>>>(
                  dYd, dW, dj, p0, p1, dbits, dcnt, nw);
            }
            WCX_HIP(hipGetLastError());
          }
        }
        WCX_HIP(hipMemcpyAsync(hbits.data(), dbits, jobs.size() * (size_t)nw * 4, hipMemcpyDeviceToHost, st));
        WCX_HIP(hipStreamSynchronize(st));
        std::vector<int> still;
        for (int q : pending) {
          const PermJob &jb = jobs[(size_t)q];
          const unsigned int *b = hbits.data() + (size_t)q * nw;
          JobResult &res = jres[(size_t)q];
          if (jb.mode == 2) {
            int nrej = 0;
            for (int t = 0; t < nw; ++t) nrej += __builtin_popcount(b[t]);
            res.nrej = nrej; res.np = p1;
            if (nrej > jb.budget) { res.decided = 1; res.significant = 0; }
            else if (p1 >= P.nperm) { res.decided = 1; res.significant = 1; }
          } else {
            eval_sequential(b, p1, P.nperm, jb.budget, seq[(size_t)q], res);
          }
          if (!res.decided) still.push_back(q);
        }
        pending.swap(still);
      }
      return WCX_OK;
    };

    // device path: the host copy of a series (hx / hw at its own offsets) is made when a decision first
    // needs it -- a segment that passed the p-value filters (short-arc bound) or is significant (edge
    // statistics); the stream is idle at those points, the export costs one small launch + sync
    std::vector<char> resident(series.size(), d_r ? 0 : 1);
    auto ensure_resident = [&](const std::vector<int> &want) -> int {
      std::vector<RangeItem> items;
      for (int q : want)
        if (!resident[(size_t)q]) {
          resident[(size_t)q] = 1;
          items.push_back({series[(size_t)q].lo, series[(size_t)q].n, 0});
        }
      if (items.empty()) return WCX_OK;
      WCX_HIP(hipMemcpyAsync(drng, items.data(), items.size() * sizeof(RangeItem), hipMemcpyHostToDevice, st));
      k_cbs_export_ranges<<<(unsigned)items.size(), 256, 0, st>>>(dX, dW, drng, hx, hw);
      WCX_HIP(hipGetLastError());
      WCX_HIP(hipStreamSynchronize(st));
      return WCX_OK;
    };

    // ---- level-synchronous recursion (DNAcopy changepoints(): stack of segment ends per series)
    struct Active { int series, lo, hi; };
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
      k_cbs_prepare<<<ns, NTP, 0, st>>>(dX, dW, dseg, dS, dWp, dYd, dy, drw, dso);
      // best arc of every segment: block-bound pruning (k_cbs_blockstats .. k_cbs_pairfinal); the
      // striped full search (k_cbs_arcmax) is the fallback when the work list overflows
      std::vector<int> boff(ns + 1);
      boff[0] = 0;
      for (int a = 0; a < ns; ++a) boff[a + 1] = boff[a] + (hseg[a].n + 1 + PBS - 1) / PBS;
      const int total_blocks = boff[ns];
      // (a single sample's ~1e9 arcs take the striped search 0.1 ms per round: the six launches and
      // the count read-back of the pruned search only pay from a few samples on)
      bool pruned = !no_prune && (size_t)total_blocks <= max_blocks && arc_total > 4e9;
      unsigned int *dcount = dL + ns;
      if (pruned) {
        WCX_HIP(hipMemcpyAsync(dboff, boff.data(), (size_t)(ns + 1) * 4, hipMemcpyHostToDevice, st));
        WCX_HIP(hipMemsetAsync(dL, 0, (size_t)(ns + 1) * 4, st));
        WCX_HIP(hipMemsetAsync(dbbits, 0, (size_t)ns * 8, st));
        WCX_HIP(hipMemsetAsync(dbij, 0xff, (size_t)ns * 8, st));
        std::vector<int> bseg((size_t)total_blocks);
        for (int a = 0; a < ns; ++a) std::fill(bseg.begin() + boff[a], bseg.begin() + boff[a + 1], a);
        WCX_HIP(hipMemcpyAsync(dbseg, bseg.data(), (size_t)total_blocks * 4, hipMemcpyHostToDevice, st));
        const unsigned rows4 = (unsigned)((total_blocks + 3) / 4);
        k_cbs_blockstats<<<rows4, 256, 0, st>>>(dS, dWp, dseg, dso, dboff, dbseg, total_blocks, dbs);
        k_cbs_coarse<<<rows4, 256, 0, st>>>(dS, dWp, dseg, dso, dboff, dbseg, total_blocks, dbs, P.minw, dL);
        k_cbs_prune<<<(unsigned)((total_blocks + 4 * PRW - 1) / (4 * PRW)), 256, 0, st>>>(
            dso, dboff, dbseg, total_blocks, dbs, dL, dwork, work_cap, dcount);
        k_cbs_pairmax<<<8192, 256, 0, st>>>(dS, dWp, dseg, dso, dwork, work_cap, dcount, P.minw, dres, dbbits, dL);
        k_cbs_pairtie<<<2048, 256, 0, st>>>(dwork, work_cap, dcount, dres, dbbits, dbij);
        k_cbs_pairfinal<<<(unsigned)((ns + 256) / 256), 256, 0, st>>>(ns, dbbits, dbij, dbest, dfirst);
        WCX_HIP(hipGetLastError());
        unsigned int hcount = 0;
        WCX_HIP(hipMemcpyAsync(&hcount, dcount, 4, hipMemcpyDeviceToHost, st));
        WCX_HIP(hipStreamSynchronize(st));
        ctx->cbs_pairs_listed += hcount;
        ctx->cbs_pairs_total += [&] { long long t = 0; for (int a = 0; a < ns; ++a) { const long long nb = boff[a + 1] - boff[a]; t += nb * (nb + 1) / 2; } return t; }();
        if (hcount > work_cap) pruned = false;           // (never on real data; keeps the result exact)
      }
      if (!pruned) {
        WCX_HIP(hipMemcpyAsync(dfirst, first.data(), (size_t)(ns + 1) * 4, hipMemcpyHostToDevice, st));
        k_cbs_arcmax<<<(unsigned)items.size(), 256, 0, st>>>(dS, dWp, dseg, dso, ditems, P.minw, dbest);
      }
      // (the bound must clear alpha with room: nu_lo^2 is 4x below the expansion it halves)
      const double alpha_skip = (ctx->debug_flags & 2) ? HUGE_VAL : 4.0 * P.alpha;
      k_cbs_arcfinish<<<ns, 128, 0, st>>>(dbest, dfirst, dseg, dWp, dso, P.kmax, P.ngrid, alpha_skip, dtx);
      k_nu_series<<<dim3(P.ngrid, ns), 256, 0, st>>>(dtx, dseg, dso, P.ngrid, alpha_skip, dnu);
      k_cbs_tailp<<<ns, 64, 0, st>>>(dnu, dseg, dso, P.ngrid, alpha_skip);
      WCX_HIP(hipGetLastError());
      std::vector<SegOut> hso(ns);
      WCX_HIP(hipMemcpyAsync(hso.data(), dso, (size_t)ns * sizeof(SegOut), hipMemcpyDeviceToHost, st));
      WCX_HIP(hipStreamSynchronize(st));
      lap("  stats");
      if (d_r) {      // (device path: the series whose short-arc bound the loop below will ask for)
        std::vector<int> want;
        for (int a = 0; a < ns; ++a) {
          const SegOut &o = hso[a];
          if (!hseg[a].hybrid || !o.valid || o.range <= 1.4901161193847656e-08) continue;
          const double ostat1 = sqrt(o.ostat);
          const int arc = std::min(o.bj - o.bi, hseg[a].n - o.bj + o.bi);
          if (ostat1 <= 0.1 || (!strict && ostat1 >= 7.0 && arc >= 10)) continue;
          if (o.pval1 < 0.0 || o.pval1 > P.alpha) continue;
          want.push_back(act[a].series);
        }
        rc = ensure_resident(want);
        if (rc) return rc;
      }
      lap("  host copy");

      // ---- decisions (DNAcopy wfindcpt, recalled) and the tests that need permutations
      std::vector<PermJob> jobs;
      std::vector<std::vector<int>> seq;
      std::vector<int> job_of(ns, -1);
      std::vector<int> verdict(ns, 0);     // 0 = no change, 1 = significant
      std::vector<int> why(ns, 0);         // 1 constant/invalid, 2 t<=0.1, 3 t>=7, 4 tailp, 5 perm, 6 bound
      int n_shortcut = 0;
      for (int a = 0; a < ns; ++a) {
        const SegOut &o = hso[a];
        const int n = hseg[a].n;
        if (!o.valid || o.range <= 1.4901161193847656e-08) { why[a] = 1; continue; }
        const double ostat1 = sqrt(o.ostat);
        if (ostat1 <= 0.1) { why[a] = 2; continue; }
        const int arc = std::min(o.bj - o.bi, n - o.bj + o.bi);
        if (!strict && ostat1 >= 7.0 && arc >= 10) { why[a] = 3; verdict[a] = 1; continue; }
        double pval2 = P.alpha;
        if (hseg[a].hybrid) {
          if (o.pval1 < 0.0 || o.pval1 > P.alpha) { why[a] = 4; continue; }   // (negative: a lower bound > alpha)
          pval2 = P.alpha - o.pval1;
        }
        PermJob jb;
        memset(&jb, 0, sizeof(jb));
        jb.lo = hseg[a].lo; jb.n = n; jb.mode = hseg[a].hybrid ? 0 : 1;
        jb.budget = (int)(pval2 * (double)P.nperm);
        jb.thr = 0.99999 * o.ostat;
        jb.tss = o.tss;
        jb.W = o.W;
        if (jb.mode == 0 && !(ctx->debug_flags & 32) &&
            short_arc_bound(hx + jb.lo, hw + jb.lo, jb.n, P.minw, P.kmax, o.tss) * 1.0001 < jb.thr) {
          // no permutation of this series can reach the observed statistic with a short arc: zero
          // exceedances whatever the order -- significant under the sequential rule too
          why[a] = 6; verdict[a] = 1; ++n_shortcut;
          continue;
        }
        if (jb.mode == 0) {
          // fp32 screen: arc sums are <= 33 local adds of v_i - m w_i, v_i a product of two rounded
          // factors: |d32 - d| <= u vmax (2700 + 75 n wmax / W) (the second term: the error of
          // m = T / W on an arc of <= 25 points), vmax = max sqrt(w) * max |y|, u = 2^-24; the arc
          // weights are rounded once (relative 2 u, in the 4.8e-7 sb term of the kernel)
          const double u = 5.9604644775390625e-08;
          const double vmax = sqrt(o.wmax) * o.ymax;
          const double wlo = P.minw * o.wmin;
          const double qmax = o.W / (wlo * (o.W - wlo));
          jb.cthr = jb.thr / ((n - 2.0) + jb.thr);
          jb.D = u * vmax * (2700.0 + 75.0 * n * o.wmax / o.W) * sqrt(qmax) * 1.001;
          jb.eT = 6.0 * u * n * vmax / o.W * 1.001;
          const double bthr = jb.cthr * o.tss;
          if (!(wlo > 0) || !(o.W > 2 * wlo) || !(bthr > 1e-30) || !(bthr < 1e30) || !std::isfinite(jb.D))
            jb.D = HUGE_VAL;                                       // everything goes to fp64
        }
        const Series &se = series[act[a].series];
        jb.key = test_key(P.seed, se.chr, act[a].lo, act[a].hi, 0);
        why[a] = 5;
        job_of[a] = (int)jobs.size();
        seq.push_back(cbs_boundary(P.alpha, P.nperm, jb.budget));
        jobs.push_back(jb);
      }
      rc = run_jobs(jobs, seq);
      if (rc) return rc;
      lap("  perms");
      std::vector<JobResult> seg_res(ns);
      for (int a = 0; a < ns; ++a)
        if (job_of[a] >= 0) { seg_res[a] = jres[(size_t)job_of[a]]; verdict[a] = seg_res[a].significant; }
      ctx->cbs_shortcuts += n_shortcut;

      // interior arcs: each of the two change-points needs its own two-sample test
      std::vector<PermJob> ejobs;
      struct EdgeRef { int a, which, shortcut; };
      std::vector<EdgeRef> eref;
      if (d_r) {      // (device path: the series of the significant segments with an interior arc)
        std::vector<int> want;
        for (int a = 0; a < ns; ++a)
          if (verdict[a] && hso[a].bi != 0 && hso[a].bj != hseg[a].n) want.push_back(act[a].series);
        rc = ensure_resident(want);
        if (rc) return rc;
      }
      for (int a = 0; a < ns; ++a) {
        if (!verdict[a]) continue;
        const int n = hseg[a].n, bi = hso[a].bi, bj = hso[a].bj;
        if (bi == 0 || bj == n) continue;
        const double mean = hso[a].mean;
        const double *x = hx + hseg[a].lo, *ww = hw + hseg[a].lo;
        // test 1: [0, bi) vs [bi, bj) ; test 2: [bi, bj) vs [bj, n)
        for (int which = 0; which < 2; ++which) {
          const int l = which == 0 ? 0 : bi, n12 = which == 0 ? bj : n - bi;
          const int n1 = which == 0 ? bi : bj - bi, n2 = n12 - n1;
          EdgeRef er{a, which, -1};
          if (n1 == 1 || n2 == 1) { er.shortcut = 0; eref.push_back(er); continue; }   // p = 1: not kept
          double w1 = 0, w2 = 0, s1 = 0, s2 = 0, ssq = 0;
          for (int i = 0; i < n1; ++i) { const double c = x[l + i] - mean; w1 += ww[l + i]; s1 += ww[l + i] * c; ssq += ww[l + i] * c * c; }
          for (int i = n1; i < n12; ++i) { const double c = x[l + i] - mean; w2 += ww[l + i]; s2 += ww[l + i] * c; ssq += ww[l + i] * c * c; }
          const double rn = w1 + w2;
          const double xbar = (s1 + s2) / rn;
          const double tss = ssq - rn * xbar * xbar;
          const bool first_short = n1 <= n2;
          const int m1 = first_short ? n1 : n2;
          const double wm = first_short ? w1 : w2, wo = first_short ? w2 : w1;
          const double dm = fabs((first_short ? s1 / w1 : s2 / w2) - xbar);
          double tstat = dm * dm * wm * rn / wo;
          tstat = tstat / ((tss - tstat) / (n12 - 2.0));
          if (!strict && tstat > 25.0 && m1 >= 10) { er.shortcut = 1; eref.push_back(er); continue; }   // kept
          PermJob jb;
          memset(&jb, 0, sizeof(jb));
          jb.lo = hseg[a].lo + l; jb.n = n12; jb.mode = 2; jb.m1 = m1;
          // kept <=> nrej / nperm <= alpha
          int budget = (int)floor(P.alpha * P.nperm);
          while ((double)(budget + 1) / (double)P.nperm <= P.alpha) ++budget;
          while (budget >= 0 && (double)budget / (double)P.nperm > P.alpha) --budget;
          jb.budget = budget;
          jb.thr = 0.99999 * dm;
          jb.tss = xbar;
          jb.W = wm;
          const Series &se = series[act[a].series];
          jb.key = test_key(P.seed, se.chr, act[a].lo, act[a].hi, 1 + which);
          er.shortcut = -1 - (int)ejobs.size();          // (-1 - job index)
          ejobs.push_back(jb);
          eref.push_back(er);
        }
      }
      rc = run_jobs(ejobs, {});
      if (rc) return rc;
      lap("  edges");
      std::vector<int> keep(ns * 2, 0), enrej(ns * 2, -2);
      for (const EdgeRef &er : eref) {
        bool ok;
        if (er.shortcut >= 0) { ok = er.shortcut == 1; enrej[er.a * 2 + er.which] = -1; }
        else {
          const int q = -1 - er.shortcut;
          ok = jres[(size_t)q].significant != 0;
          enrej[er.a * 2 + er.which] = jres[(size_t)q].nrej;
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
          if (bj == n) { ncpt = 1; icpt[0] = bi; }
          else if (bi == 0) { ncpt = 1; icpt[0] = bj; }
          else {
            if (keep[a * 2]) icpt[ncpt++] = bi;
            if (keep[a * 2 + 1]) icpt[ncpt++] = bj;
          }
        }
        if (tracing) {
          const double rec[TRACE_W] = {
              (double)se.sample, (double)se.chr, (double)lo, (double)hi, (double)n, (double)hso[a].bi,
              (double)hso[a].bj, hso[a].ostat, hseg[a].hybrid ? hso[a].pval1 : __builtin_nan(""),
              hso[a].delta, (double)why[a], job_of[a] >= 0 ? (double)jobs[(size_t)job_of[a]].budget : -1.0,
              job_of[a] >= 0 ? (double)seg_res[a].nrej : -1.0, job_of[a] >= 0 ? (double)seg_res[a].np : -1.0,
              (double)verdict[a], (double)ncpt, (double)keep[a * 2], (double)enrej[a * 2],
              (double)keep[a * 2 + 1], (double)enrej[a * 2 + 1]};
          ctx->cbs_trace.insert(ctx->cbs_trace.end(), rec, rec + TRACE_W);
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

  if (copies_queued) WCX_HIP(hipStreamSynchronize(aux));
  const bool compact = d_r != nullptr && n_inf == 0;   // device path: r / w stayed in HBM
  // ---- CBS.R:84-129 on the host: NA-run splitting, >= 2-bin rule, weighted re-mean, 0-based start
  const int na_limit = (int)(1.0 / ((double)binsize / 2000000.0));   // as.integer((binsize/2e6)^-1)
  std::vector<int> count(n_samples, 0);
  auto emit_seg = [&](int s, int c, int a, int b, double num, double den) {
    if (count[s] < cap) {
      double *dst = out_seg + ((size_t)s * cap + count[s]) * 4;
      dst[0] = c;
      dst[1] = a - 1;                               // CBS.R:129
      dst[2] = b;
      dst[3] = den > 0 ? num / den : __builtin_nan("");
    }
    ++count[s];
  };
  if (compact) {
    // The same statement on the compacted series (no +-inf in r: dropped == NA): the NA runs of a
    // segment are the gaps between consecutive kept bins, a run of bins p_t + 1 .. p_u - 1 has
    // start_pos = p_t, end_pos = p_u - 1; the intervals come from the positions alone (host threads),
    // their weighted means -- sums over the interval's kept bins in order -- from x | w where they
    // are: in HBM (k_cbs_interval_means), so that x and w never cross the link as a whole.
    struct Iv { int c, a, b, n; int64_t lo; };
    std::vector<std::vector<Iv>> ivs((size_t)n_samples);
    for_samples([&](int my_sample) {
      for (Series &se : series) {
        if (se.sample != my_sample) continue;
        std::sort(se.change_loc.begin(), se.change_loc.end());
        const int *pos = hpos + se.lo;
        int prev = 0;
        for (int e : se.change_loc) {
          const int e1 = pos[e - 1];
          int a = pos[prev], ia = prev;             // current interval: starts at bin a, first kept element ia
          auto close = [&](int b, int ie) {         // interval [a, b], kept elements [ia, ie)
            if (!(b - a > 0)) return;               // CBS.R:103
            ivs[(size_t)my_sample].push_back({se.chr, a, b, ie - ia, se.lo + ia});
          };
          for (int t = prev; t + 1 < e; ++t) {
            const int sp = pos[t], ep = pos[t + 1] - 1;
            if (ep - sp > na_limit) { close(sp, t + 1); a = ep; ia = t + 1; }
          }
          close(e1, e);
          prev = e;
        }
      }
    });
    std::vector<RangeItem> items;
    for (const auto &v : ivs) for (const Iv &iv : v) items.push_back({iv.lo, iv.n, 0});
    std::vector<double> sums(items.size() * 2);
    if (!items.empty()) {
      void *scr2 = nullptr;
      rc = wcx_scratch2(ctx, items.size() * (sizeof(RangeItem) + 16) + 64, &scr2);
      if (rc) return rc;
      RangeItem *d_items = reinterpret_cast<RangeItem *>(scr2);
      double *d_sums = reinterpret_cast<double *>(d_items + items.size());
      WCX_HIP(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(RangeItem), hipMemcpyHostToDevice, st));
      k_cbs_interval_means<<<(unsigned)((items.size() + 3) / 4), 256, 0, st>>>(dXs_all, dWs_all, d_items,
                                                                              (int)items.size(), d_sums);
      WCX_HIP(hipGetLastError());
      WCX_HIP(hipMemcpyAsync(sums.data(), d_sums, sums.size() * 8, hipMemcpyDeviceToHost, st));
      WCX_HIP(hipStreamSynchronize(st));
    }
    size_t q = 0;
    for (int s = 0; s < n_samples; ++s)
      for (const Iv &iv : ivs[(size_t)s]) { emit_seg(s, iv.c, iv.a, iv.b, sums[2 * q], sums[2 * q + 1]); ++q; }
  } else {
  for_samples([&](int my_sample) {
  for (Series &se : series) {
    if (se.sample != my_sample) continue;
    std::sort(se.change_loc.begin(), se.change_loc.end());
    const int s = se.sample, c = se.chr;
    const int64_t o = (int64_t)s * n_bins + chr_off[c];
    const int *pos = hpos + se.lo;
    int prev = 0;
    auto emit = [&](int a, int b, double num, double den) { emit_seg(s, c, a, b, num, den); };
    for (int e : se.change_loc) {
      const int s1 = pos[prev], e1 = pos[e - 1];   // inclusive, 1-based
      prev = e;
      std::vector<int> start_pos, end_pos;
      for (int b = s1; b < e1; ++b) {   // b, b+1 are 1-based bins inside the segment
        const bool na0 = is_na(r[o + b - 1]);
        const bool na1 = is_na(r[o + b]);
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
          if (is_na(v)) continue;
          const double wt = w[o + t - 1] == 0.0 ? 1.0 : w[o + t - 1];
          num += v * wt; den += wt;
        }
        emit(a, b, num, den);
      }
    }
  }
  });
  }
  lap("wrap-up");
  int over = 0;
  for (int s = 0; s < n_samples; ++s) { out_count[s] = count[s]; over = std::max(over, count[s]); }
  if (over > cap) {
    wcx_set_error("wcx_cbs: %d segments exceed the caller's capacity %d", over, cap);
    return WCX_ERR_ARG;
  }
  return WCX_OK;
}

extern "C" {

int wcx_cbs_batch(wcx_ctx *ctx, const double *r, const double *w, int n_samples, int64_t n_bins,
                  const int64_t *chr_off, int n_chr, double alpha, int64_t binsize, uint64_t seed,
                  double *out_seg, int cap, int *out_count) {
  return cbs_batch_impl(ctx, r, w, nullptr, nullptr, nullptr, n_samples, n_bins, chr_off, n_chr, alpha, binsize,
                        seed, out_seg, cap, out_count);
}

int wcx_cbs_batch_dev(wcx_ctx *ctx, const double *d_r, const double *d_w, int n_samples,
                      int64_t n_bins, const int64_t *chr_off, int n_chr, double alpha, int64_t binsize,
                      uint64_t seed, double *out_seg, int cap, int *out_count) {
  WCX_ARG(ctx && d_r && d_w && n_samples > 0 && n_bins > 0, "bad parameters");
  WCX_HIP(hipSetDevice(ctx->device));
  // (pinned destination of r | w for the rare +-inf fallback; see cbs_batch_impl)
  const size_t bytes = (size_t)n_samples * n_bins * 8;
  if (ctx->host_scratch2_bytes < 2 * bytes) {
    if (ctx->host_scratch2) { WCX_HIP(hipStreamSynchronize(ctx->stream)); WCX_HIP(hipHostFree(ctx->host_scratch2)); }
    ctx->host_scratch2 = nullptr; ctx->host_scratch2_bytes = 0;
    hipError_t e = hipHostMalloc(&ctx->host_scratch2, 2 * bytes + bytes / 2, hipHostMallocDefault);
    if (e != hipSuccess) {
      wcx_set_error("hipHostMalloc(%zu bytes) failed: %s", 2 * bytes + bytes / 2, hipGetErrorString(e));
      return WCX_ERR_NOMEM;
    }
    ctx->host_scratch2_bytes = 2 * bytes + bytes / 2;
  }
  double *hr = reinterpret_cast<double *>(ctx->host_scratch2), *hw = hr + (size_t)n_samples * n_bins;
  if (!ctx->copy_stream) {
    WCX_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    WCX_HIP(hipEventCreateWithFlags(&ctx->ev_cbs_fill, hipEventDisableTiming));
    WCX_HIP(hipEventCreateWithFlags(&ctx->ev_cbs_xw, hipEventDisableTiming));
  }
  hipStream_t aux = ctx->copy_stream;
  const int rc = cbs_batch_impl(ctx, hr, hw, d_r, d_w, aux, n_samples, n_bins, chr_off, n_chr, alpha, binsize,
                                seed, out_seg, cap, out_count);
  // (an early error return may leave the copies in flight: the pinned buffer outlives them, but the
  // next call must not start before they are done)
  if (rc) hipStreamSynchronize(aux);
  return rc;
}

int wcx_cbs_stats(wcx_ctx *ctx, int64_t out[4]) {
  WCX_ARG(ctx && out, "NULL argument");
  out[0] = ctx->cbs_shortcuts; out[1] = ctx->cbs_pairs_listed; out[2] = ctx->cbs_pairs_total; out[3] = 0;
  return WCX_OK;
}

int wcx_cbs_trace(wcx_ctx *ctx, double *out, int cap_records, int *count) {
  WCX_ARG(ctx && count && cap_records >= 0 && (out || cap_records == 0), "bad parameters");
  const int have = (int)(ctx->cbs_trace.size() / TRACE_W);
  *count = have;
  const int ncopy = std::min(have, cap_records);
  if (ncopy > 0) memcpy(out, ctx->cbs_trace.data(), (size_t)ncopy * TRACE_W * sizeof(double));
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
