// predict hot path: distance cut-off, weights, and the three
// masked within-sample normalisation passes.  All kernels are gather/scan work over the
// reference's indexes/distances held in HBM: HBM-bandwidth bound (B*k*4 bytes of indices per
// pass, B*k*8 bytes of distances per cut-off sweep).
#include "wave_sort.h"
#include "wcx_common.h"

namespace {

constexpr int NT = 256;
constexpr double Z_MASK = 2.3263478740408408;  // scipy.stats.norm.ppf(0.99), predict_tools.py:104

// ------------------------------------------------------------------------------------------
// a11 get_optimal_cutoff (predict_tools.py:74-82): repeats x { mean, std of dist < cutoff }.
// state[0]=cutoff state[1]=mean state[2]=count
__global__ __launch_bounds__(NT) void k_cut_partial(const double *__restrict__ d, int64_t n,
                                                    const double *__restrict__ state, int phase,
                                                    double *__restrict__ part_sum,
                                                    double *__restrict__ part_cnt) {
  const double cutoff = state[0];
  const double mean = state[1];
  double s = 0.0, c = 0.0;
  const int64_t stride = (int64_t)gridDim.x * NT * 2;
  for (int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * 2; i < n; i += stride) {
    double2 v;
    if (i + 1 < n) v = *reinterpret_cast<const double2 *>(d + i);
    else { v.x = d[i]; v.y = __builtin_nan(""); }
    if (v.x < cutoff) { if (phase == 0) { s += v.x; c += 1.0; } else { double e = v.x - mean; s += e * e; } }
    if (v.y < cutoff) { if (phase == 0) { s += v.y; c += 1.0; } else { double e = v.y - mean; s += e * e; } }
  }
  __shared__ double sh_s[NT / 64], sh_c[NT / 64];
  s = wcx::wave_sum(s);
  c = wcx::wave_sum(c);
  if ((threadIdx.x & 63) == 0) { sh_s[threadIdx.x >> 6] = s; sh_c[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tc = 0.0;
    for (int w = 0; w < NT / 64; ++w) { ts += sh_s[w]; tc += sh_c[w]; }
    part_sum[blockIdx.x] = ts;
    part_cnt[blockIdx.x] = tc;
  }
}

__global__ __launch_bounds__(NT) void k_cut_final(double *__restrict__ state, int phase,
                                                  const double *__restrict__ part_sum,
                                                  const double *__restrict__ part_cnt, int nparts) {
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nparts; i += NT) { s += part_sum[i]; c += part_cnt[i]; }
  __shared__ double sh_s[NT / 64], sh_c[NT / 64];
  s = wcx::wave_sum(s);
  c = wcx::wave_sum(c);
  if ((threadIdx.x & 63) == 0) { sh_s[threadIdx.x >> 6] = s; sh_c[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tc = 0.0;
    for (int w = 0; w < NT / 64; ++w) { ts += sh_s[w]; tc += sh_c[w]; }
    if (phase == 0) {
      state[2] = tc;
      state[1] = ts / tc;                       // np.average
    } else {
      const double sd = sqrt(ts / state[2]);    // np.std (population)
      state[0] = state[1] + 3.0 * sd;           // predict_tools.py:81
    }
  }
}

// (Round 6, built, measured, removed: partial + final as ONE launch -- every workgroup takes a ticket after
//  its partial and the last one adds them up -- and the same for the wide nanmedian's histogram + decide
//  pairs: 10 + 18 launches instead of 20 + 36 per predict.  The device-scope release every workgroup needs
//  before its ticket (__threadfence: on this multi-XCD part an L2 write-back + invalidate) costs more than
//  the launches it saves: cut-off 0.73 -> 1.30 ms at 15 kb, the 100 kb step 6.4 -> 7.0 ms.  A kernel
//  boundary is the cheap device-wide fence here.)
// out[0] = sum of part_sum, out[1] = sum of part_cnt  (row-sharded cut-off: the host all-reduces)
__global__ __launch_bounds__(NT) void k_sum_parts(const double *__restrict__ part_sum,
                                                  const double *__restrict__ part_cnt, int nparts,
                                                  double *__restrict__ out) {
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nparts; i += NT) { s += part_sum[i]; c += part_cnt[i]; }
  __shared__ double sh_s[NT / 64], sh_c[NT / 64];
  s = wcx::wave_sum(s);
  c = wcx::wave_sum(c);
  if ((threadIdx.x & 63) == 0) { sh_s[threadIdx.x >> 6] = s; sh_c[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tc = 0.0;
    for (int w = 0; w < NT / 64; ++w) { ts += sh_s[w]; tc += sh_c[w]; }
    out[0] = ts;
    out[1] = tc;
  }
}

// ------------------------------------------------------------------------------------------
// a12 get_weights (predict_tools.py:152-155): w_i = 1 / mean_k sqrt(dist[i][k]); wave per row.
__global__ __launch_bounds__(NT) void k_weights(const double *__restrict__ dist, int64_t B, int k,
                                                double *__restrict__ out) {
  const int lane = wcx::lane_id();
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t i = w0; i < B; i += nw) {
    double s = 0.0;
    for (int t = lane; t < k; t += 64) s += sqrt(dist[i * (int64_t)k + t]);
    s = wcx::wave_sum(s);
    if (lane == 0) out[i] = 1.0 / (s / (double)k);
  }
}

// ------------------------------------------------------------------------------------------
// a13: selection mask  sel[i][q] = ballot( dist[i][q*64+lane] < cutoff )   (predict_tools.py:133)
// dist / sel hold the rows [row0, ...) of the reference (row0 = 0 unless row-sharded);
// rows [lo, hi) are processed.
__global__ __launch_bounds__(NT) void k_select_mask(const double *__restrict__ dist, int k, int ipl,
                                                    double cutoff, int64_t lo, int64_t hi,
                                                    int64_t row0,
                                                    unsigned long long *__restrict__ sel) {
  const int lane = wcx::lane_id();
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t i = lo + w0; i < hi; i += nw) {
    const int64_t li = i - row0;
    for (int q = 0; q < ipl; ++q) {
      const int t = q * 64 + lane;
      const bool s = (t < k) && (dist[li * (int64_t)k + t] < cutoff);
      const unsigned long long m = __ballot(s);
      if (lane == 0) sel[li * ipl + q] = m;
    }
  }
}

struct ChrTable {
  int n_chr;
  int64_t cum[32];
};

// One masked pass of _normalize_once (predict_tools.py:111-142) for a batch of samples.
// grid.y = sample.  copy_in/copy_out: ping-pong of test_copy (predict_tools.py:98,104).
template <int IPL>
__global__ __launch_bounds__(NT) void k_normalize_pass(
    const double *__restrict__ x, const double *__restrict__ copy_in,
    double *__restrict__ copy_out, const int32_t *__restrict__ idx,
    const unsigned long long *__restrict__ sel, int64_t B, int k, int64_t ct, int64_t lo,
    int64_t hi, int64_t row0, ChrTable chr, double *__restrict__ out_z,
    double *__restrict__ out_r, double *__restrict__ out_n, double *__restrict__ out_lr,
    int last_pass) {
  const int lane = wcx::lane_id();
  __shared__ int s_hist[NT / 64][64];
  __shared__ double s_slots[NT / 64][64];
  const int s = blockIdx.y;
  const double *xs = x + (int64_t)s * B;
  const double *cin = copy_in + (int64_t)s * B;
  double *cout = copy_out + (int64_t)s * B;
  const int64_t Bp = B - ct;
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t i = lo + w0; i < hi; i += nw) {
    const int64_t li = i - row0;   // row of idx / sel (the reference may hold a row shard)
    // own chromosome [cs,ce) of row i
    int64_t cs = 0, ce = chr.cum[0];
    for (int c = 1; c < chr.n_chr && i >= ce; ++c) { cs = ce; ce = chr.cum[c]; }
    const int64_t own = ce - cs;
    const int64_t len_cd = B - own;  // len(chr_data), predict_tools.py:125-130
    double v[IPL];
    int nloc = 0;
    unsigned int keepmask = 0;
    double sloc = 0.0;
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
      const int t = q * 64 + lane;
      const bool selq = (t < k) && ((sel[li * IPL + q] >> lane) & 1ull);
      double val = HUGE_VAL;
      bool keep = false;
      if (selq) {
        int64_t c = idx[li * (int64_t)k + t];
        if (c < 0) c += len_cd;                       // NumPy negative index
        const int64_t g = c < cs ? c : c + own;       // chr_data index -> row
        const double cv = cin[g];
        keep = cv >= 0.0;                             // predict_tools.py:134
        if (keep) val = cv;
      }
      v[q] = val;
      if (keep) { nloc += 1; sloc += val; keepmask |= 1u << q; }
    }
    const int n = wcx::wave_sum_i(nloc);
    const double mean = wcx::wave_sum(sloc) / (double)n;
    double ssl = 0.0;
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
      if ((keepmask >> q) & 1u) { const double e = v[q] - mean; ssl += e * e; }
    }
    const double sd = sqrt(wcx::wave_sum(ssl) / (double)n);
    // the ratio (median) of a pass is only ever read after the LAST one (predict_tools.py:99-108:
    // the earlier passes feed nothing but the z-mask), so only that pass pays for the selection
    double med = 1.0;
    if (last_pass)
      med = wcx::wave_median_bucket<IPL>(v, keepmask, n, s_hist[threadIdx.x >> 6],
                                         s_slots[threadIdx.x >> 6], mean, sd);
    if (lane == 0) {
      const double xi = xs[i];
      const double z = (xi - mean) / sd;              // predict_tools.py:136
      const double r = xi / med;                      // predict_tools.py:137
      const int64_t o = (int64_t)s * Bp + (i - ct);
      out_z[o] = z;
      out_r[o] = r;
      out_n[o] = (double)n;
      if (last_pass) out_lr[o] = log2(r);
      cout[i] = (fabs(z) >= Z_MASK) ? -1.0 : cin[i];  // predict_tools.py:104
    }
  }
}

// Batched form: a wave owns one bin and a TILE of T samples.  test_copy is kept SAMPLE-MINOR
// (copyT[bin][NS]: the T values of a reference bin are one 8 T-byte piece), so the index row and
// the selection words of a bin are read ONCE per tile and every gather brings T samples; the
// statistics of the T samples follow one after the other in the same wave.
template <int IPL, int T>
__global__ __launch_bounds__(NT) void k_normalize_pass_tile(
    const double *__restrict__ x, const double *__restrict__ copy_in,
    double *__restrict__ copy_out, const int32_t *__restrict__ idx,
    const unsigned long long *__restrict__ sel, int64_t B, int k, int NS, int n_samples, int64_t ct,
    int64_t lo, int64_t hi, ChrTable chr, double *__restrict__ out_z, double *__restrict__ out_r,
    double *__restrict__ out_n, double *__restrict__ out_lr, int last_pass) {
  const int lane = wcx::lane_id();
  __shared__ int s_hist[NT / 64][64];
  __shared__ double s_slots[NT / 64][64];
  const int s0 = blockIdx.y * T;                 // first sample of this tile
  const int64_t Bp = B - ct;
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t i = lo + w0; i < hi; i += nw) {
    int64_t cs = 0, ce = chr.cum[0];
    for (int c = 1; c < chr.n_chr && i >= ce; ++c) { cs = ce; ce = chr.cum[c]; }
    const int64_t own = ce - cs;
    const int64_t len_cd = B - own;  // len(chr_data), predict_tools.py:125-130
    double v[T][IPL];
    unsigned int selmask = 0;
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
      const int t = q * 64 + lane;
      const bool selq = (t < k) && ((sel[i * IPL + q] >> lane) & 1ull);
      if (selq) {
        int64_t c = idx[i * (int64_t)k + t];
        if (c < 0) c += len_cd;                       // NumPy negative index
        const int64_t g = c < cs ? c : c + own;       // chr_data index -> row
        const double *src = copy_in + g * NS + s0;
#pragma unroll
        for (int s = 0; s < T; ++s) v[s][q] = src[s];
        selmask |= 1u << q;
      } else {
#pragma unroll
        for (int s = 0; s < T; ++s) v[s][q] = -1.0;   // fails the >= 0 test below
      }
    }
    // per sample: count, mean, sum of squares (wave reductions), median (last pass only); the
    // scalar tail (sqrt, two divisions, log2, stores) of the T samples then runs ONCE, sample s in
    // lane s, instead of T times in lane 0
    double st_n = 0.0, st_mean = 0.0, st_q = 0.0, st_med = 1.0;
#pragma unroll
    for (int s = 0; s < T; ++s) {
      if (s0 + s >= n_samples) break;                 // padding samples of the last tile
      int nloc = 0;
      unsigned int keepmask = 0;
      double sloc = 0.0;
      double vs[IPL];
#pragma unroll
      for (int q = 0; q < IPL; ++q) {
        const bool keep = ((selmask >> q) & 1u) && v[s][q] >= 0.0;   // predict_tools.py:134
        vs[q] = keep ? v[s][q] : HUGE_VAL;
        if (keep) { nloc += 1; sloc += v[s][q]; keepmask |= 1u << q; }
      }
      const int n = wcx::wave_sum_i(nloc);
      const double mean = wcx::wave_sum(sloc) / (double)n;
      double ssl = 0.0;
#pragma unroll
      for (int q = 0; q < IPL; ++q)
        if ((keepmask >> q) & 1u) { const double e = vs[q] - mean; ssl += e * e; }
      const double qsum = wcx::wave_sum(ssl);
      double med = 1.0;                                 // (only the last pass's ratio is ever read)
      if (last_pass)
        med = wcx::wave_median_bucket<IPL>(vs, keepmask, n, s_hist[threadIdx.x >> 6],
                                           s_slots[threadIdx.x >> 6], mean,
                                           (double)__builtin_sqrtf((float)qsum / (float)n));
      if (lane == s) { st_n = (double)n; st_mean = mean; st_q = qsum; st_med = med; }
    }
    if (lane < T && s0 + lane < n_samples) {
      const double sd = sqrt(st_q / st_n);
      const double xi = x[(int64_t)(s0 + lane) * B + i];
      const double z = (xi - st_mean) / sd;             // predict_tools.py:136
      const double r = xi / st_med;                     // predict_tools.py:137
      const int64_t o = (int64_t)(s0 + lane) * Bp + (i - ct);
      out_z[o] = z;
      out_r[o] = r;
      out_n[o] = st_n;
      if (last_pass) out_lr[o] = log2(r);
      copy_out[i * NS + s0 + lane] = (fabs(z) >= Z_MASK) ? -1.0 : copy_in[i * NS + s0 + lane];   // :104
    }
  }
}

// Passes 0 and 1 of a BATCH (predict_tools.py:99-104): they feed nothing but the z-mask of the next
// pass, so only count, mean and standard deviation are needed -- no median.  LANE = SAMPLE: a wave owns
// one bin and 64 samples; the bin's selected reference rows are walked once, every step one coalesced
// 512-byte load of copyT[row][64 samples], and each lane keeps its own sample's running sums in
// registers -- no cross-lane reduction at all (the tiled kernel below spends its time in three
// 64-lane reductions per sample and bin).  Sums are taken about c = the sample's own value of the
// bin: mean = c + S1 / n, var = (S2 - S1^2 / n) / n (c is close to the mean: no cancellation); a set
// of identical values ends like np.std's (see below).  (Measured, 96 samples at 15 kb: the three
// passes 42.8 -> 34.8 ms; a variant with two bins x 32 samples per wave -- no idle lanes for 96
// samples -- was slower, 36.4 ms: the kernel is bound by its per-element instructions, not by bytes.)
// st_S1 != nullptr (pass 0 of the incremental scheme, k_normalize_mask_incr): the sums, the count and the
// word of newly masked samples of every (bin, 64-sample tile) are kept; the masked copy is not written.
__global__ __launch_bounds__(NT) void k_normalize_mask_lanes(
    const double *__restrict__ xT, const double *__restrict__ copy_in, double *__restrict__ copy_out,
    const int32_t *__restrict__ idx, const unsigned long long *__restrict__ sel, int64_t B, int k,
    int ipl, int NS, int64_t lo, int64_t hi, ChrTable chr, double *__restrict__ st_S1 = nullptr,
    double *__restrict__ st_S2 = nullptr, int *__restrict__ st_n = nullptr,
    unsigned long long *__restrict__ st_mask = nullptr) {
  const int lane = wcx::lane_id();
  const int s = blockIdx.y * 64 + lane;             // my sample (NS is a multiple of 64)
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t i = lo + w0; i < hi; i += nw) {
    int64_t cs = 0, ce = chr.cum[0];
    for (int c = 1; c < chr.n_chr && i >= ce; ++c) { cs = ce; ce = chr.cum[c]; }
    const int64_t own = ce - cs;
    const int64_t len_cd = B - own;  // len(chr_data), predict_tools.py:125-130
    const double c0 = xT[i * NS + s];
    double S1a = 0.0, S2a = 0.0, S1b = 0.0, S2b = 0.0;
    int n = 0;
    for (int q = 0; q < ipl; ++q) {
      const int t = q * 64 + lane;
      int gv = 0;
      if (t < k) {
        int64_t c = idx[i * (int64_t)k + t];
        if (c < 0) c += len_cd;                       // NumPy negative index
        gv = (int)(c < cs ? c : c + own);             // chr_data index -> row
      }
      // (the selection word is the same for every lane: in SGPRs the walk over its bits -- ctz, clear,
      //  readlane index -- is scalar work; left in VGPRs it cost 11 of the loop's 24 vector instructions
      //  per reference row)
      const unsigned long long wv = sel[i * ipl + q];
      unsigned long long w = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(wv >> 32)) << 32) |
                             (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)wv);
      if (q == ipl - 1 && (k & 63)) w &= (1ull << (k & 63)) - 1ull;
      // the selected rows of this word, four at a time (their loads in flight together)
      while (w) {
        int g[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          on[u] = w != 0;
          const int b = on[u] ? __builtin_ctzll(w) : 0;
          w &= w - 1ull;                              // (0 & -1 = 0: stays empty)
          g[u] = __builtin_amdgcn_readlane(gv, b);
        }
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = copy_in[(int64_t)g[u] * NS + s];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool keep = on[u] && v[u] >= 0.0;     // predict_tools.py:134
          const double d = keep ? v[u] - c0 : 0.0;
          const double dd = d * d;
          if (u & 1) { S1b = S1b + d; S2b = S2b + dd; } else { S1a = S1a + d; S2a = S2a + dd; }
          n += keep ? 1 : 0;
        }
      }
    }
    // (a set of identical values v: np.std gives 0 or rounding dust, z = nan / 0 if the bin's own value
    //  equals v -- kept --, else inf or huge -- masked; the sums about c0 end the same way: S1 = 0 = S2
    //  for c0 = v, and var = dust or <= 0 otherwise)
    const double S1 = S1a + S1b, S2 = S2a + S2b, dn = (double)n;
    const double mean = c0 + S1 / dn;
    const double var = (S2 - S1 * (S1 / dn)) / dn;
    const double sd = sqrt(var > 0.0 ? var : 0.0);
    const double z = (c0 - mean) / sd;                // predict_tools.py:136
    if (st_S1) {
      st_S1[i * NS + s] = S1;
      st_S2[i * NS + s] = S2;
      st_n[i * NS + s] = n;
      const unsigned long long m = __ballot(fabs(z) >= Z_MASK && copy_in[i * NS + s] >= 0.0);
      if (lane == 0) st_mask[i * (NS >> 6) + blockIdx.y] = m;
    } else {
      copy_out[i * NS + s] = (fabs(z) >= Z_MASK) ? -1.0 : copy_in[i * NS + s];   // :104
    }
  }
}

// Pass 1 of a BATCH without a second sweep of the gathers.  Pass 0 (k_normalize_mask_lanes) left every
// (bin, sample)'s sums S1, S2 about c0 and count n over its selected, kept reference bins, and per (bin,
// 64-sample tile) the WORD of samples it masked (|z| >= 3).  Pass 1's set is pass 0's minus the reference
// bins masked meanwhile -- 0.3 % of them per sample: one in six (reference bin, tile) words is non-zero --
// so the sums are UPDATED: a lane (= reference slot) fetches its reference bin's mask word, the wave walks
// the non-zero ones, and only their samples' values are fetched and subtracted.  8 bytes per reference bin
// and tile + a sixth of the 512-byte rows instead of all of them.  Writes the masked copy pass 2 reads
// (cumulative: masked at pass 0 or now).  (The updated sums differ from freshly accumulated ones by
// rounding, ~1e-16 relative: a |z| within that of 3 could flip a mask bit against the one-sample path --
// the same caveat as for the lane-per-sample sums themselves)
// st_mask_new != nullptr (rank-median scheme): the updated sums and counts are written back and the word of
// samples masked NOW goes to st_mask_new -- the same kernel then runs once more as the statistics of the
// LAST pass (zT != nullptr: its set = this pass's minus the bins in st_mask; z and n, sample-minor, are the
// outputs; nothing else is written).
__global__ __launch_bounds__(NT) void k_normalize_mask_incr(
    const double *__restrict__ xT, double *__restrict__ copy_out, const int32_t *__restrict__ idx,
    const unsigned long long *__restrict__ sel, int64_t B, int k, int ipl, int NS, int64_t lo, int64_t hi,
    ChrTable chr, double *__restrict__ st_S1, double *__restrict__ st_S2,
    int *__restrict__ st_n, const unsigned long long *__restrict__ st_mask,
    const unsigned long long *__restrict__ st_was = nullptr, unsigned long long *__restrict__ st_mask_new = nullptr,
    double *__restrict__ zT = nullptr, double *__restrict__ nT = nullptr) {
  const int lane = wcx::lane_id();
  const int s = blockIdx.y * 64 + lane;             // my sample (NS is a multiple of 64)
  const int n_tiles = NS >> 6;
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t i = lo + w0; i < hi; i += nw) {
    int64_t cs = 0, ce = chr.cum[0];
    for (int c = 1; c < chr.n_chr && i >= ce; ++c) { cs = ce; ce = chr.cum[c]; }
    const int64_t own = ce - cs;
    const int64_t len_cd = B - own;
    const double c0 = xT[i * NS + s];
    if (zT) {
      // Statistics of the last pass for a bin with FEW selected reference bins (its distances mostly beyond
      // the cut-off: one or two references are common among them): the updated sums carry the rounding of
      // everything added and taken away before -- harmless against a variance of hundreds of values, but a
      // set of one value has variance 0 EXACTLY (z = +-inf, as np.std gives it), not dust.  Such a bin is
      // small for every sample alike (the selection is per bin): its few rows are walked again, mean and
      // squared deviations taken directly (copy_out = the masked copy pass 1 wrote).
      int nsel = 0;
      for (int q = 0; q < ipl; ++q) {
        unsigned long long w = sel[i * ipl + q];
        if (q == ipl - 1 && (k & 63)) w &= (1ull << (k & 63)) - 1ull;
        nsel += __popcll(w);
      }
      nsel = __builtin_amdgcn_readfirstlane(nsel);
      if (nsel <= 32) {
        double sum = 0.0, ss = 0.0;
        int ne = 0;
        for (int walk = 0; walk < 2; ++walk) {
          const double mean_e = walk ? sum / (double)ne : 0.0;
          for (int q = 0; q < ipl; ++q) {
            const int t = q * 64 + lane;
            int gv = 0;
            if (t < k) {
              int64_t c = idx[i * (int64_t)k + t];
              if (c < 0) c += len_cd;
              gv = (int)(c < cs ? c : c + own);
            }
            const unsigned long long wv = sel[i * ipl + q];
            unsigned long long w = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(wv >> 32)) << 32) |
                                   (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)wv);
            if (q == ipl - 1 && (k & 63)) w &= (1ull << (k & 63)) - 1ull;
            while (w) {
              const int b = __builtin_ctzll(w);
              w &= w - 1ull;
              const int g = __builtin_amdgcn_readlane(gv, b);
              const double v = copy_out[(int64_t)g * NS + s];
              if (v >= 0.0) {                         // predict_tools.py:134
                if (walk == 0) { sum += v; ne += 1; }
                else { const double e = v - mean_e; ss += e * e; }
              }
            }
          }
        }
        const double dne = (double)ne;
        const double mean_e = sum / dne;
        zT[i * NS + s] = (c0 - mean_e) / sqrt(ss / dne);
        nT[i * NS + s] = dne;
        continue;
      }
    }
    double S1 = st_S1[i * NS + s], S2 = st_S2[i * NS + s];
    int n = st_n[i * NS + s];
    for (int q = 0; q < ipl; ++q) {
      const int t = q * 64 + lane;
      int gv = 0;
      bool selq = false;
      if (t < k) {
        int64_t c = idx[i * (int64_t)k + t];
        if (c < 0) c += len_cd;
        gv = (int)(c < cs ? c : c + own);
        selq = (sel[i * ipl + q] >> lane) & 1ull;
      }
      // my reference bin's word of the samples pass 0 masked (0 for nearly all)
      const unsigned long long mw = selq ? st_mask[(int64_t)gv * n_tiles + blockIdx.y] : 0ull;
      unsigned long long todo = __ballot(mw != 0ull);
      // four reference bins at a time: their (sparse) loads in flight together -- one after the other the
      // walk is a chain of L2 round trips and costs as much as the full sweep it replaces
      while (todo) {
        int g[4];
        bool hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool on = todo != 0ull;
          const int j = on ? __builtin_ctzll(todo) : 0;
          todo &= todo - 1ull;                        // (0 & -1 = 0: stays empty)
          g[u] = __builtin_amdgcn_readlane(gv, j);
          const unsigned long long wj =
              ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(mw >> 32), j) << 32) |
              (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)mw, j);
          hit[u] = on && ((wj >> lane) & 1ull);
        }
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = hit[u] ? xT[(int64_t)g[u] * NS + s] : c0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double d = v[u] - c0;                 // (masked by pass 0 => it was kept: x >= 0; no hit: 0)
          S1 = S1 - d;
          S2 = S2 - d * d;
          n -= hit[u] ? 1 : 0;
        }
      }
    }
    const double dn = (double)n;
    const double mean = c0 + S1 / dn;
    const double var = (S2 - S1 * (S1 / dn)) / dn;
    const double sd = sqrt(var > 0.0 ? var : 0.0);
    const double z = (c0 - mean) / sd;                // predict_tools.py:136
    if (zT) {                                         // statistics of the last pass
      zT[i * NS + s] = z;
      nT[i * NS + s] = dn;
      continue;
    }
    const bool was = ((st_was ? st_was : st_mask)[i * n_tiles + blockIdx.y] >> lane) & 1ull;
    copy_out[i * NS + s] = (was || fabs(z) >= Z_MASK) ? -1.0 : c0;   // :104 (c0 < 0 stays as it is)
    if (st_mask_new) {
      st_S1[i * NS + s] = S1;
      st_S2[i * NS + s] = S2;
      st_n[i * NS + s] = n;
      const unsigned long long m = __ballot(!was && fabs(z) >= Z_MASK && c0 >= 0.0);
      if (lane == 0) st_mask_new[i * n_tiles + blockIdx.y] = m;
    }
  }
}

// sample-minor [B - ct rows from ct][NS] -> [n_samples][Bp] (the layout of the outputs)
__global__ __launch_bounds__(256) void k_from_sample_minor(const double *__restrict__ aT, int64_t ct, int64_t Bp,
                                                           int NS, int n_samples, double *__restrict__ out) {
  __shared__ double tile[32][33];
  const int64_t b0 = (int64_t)blockIdx.x * 32;
  const int s0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int64_t bb = b0 + r;
    const int sx = s0 + tx;
    tile[r][tx] = (bb < Bp && sx < NS) ? aT[(ct + bb) * NS + sx] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int sx = s0 + r;
    const int64_t bb = b0 + tx;
    if (sx < n_samples && bb < Bp) out[(int64_t)sx * Bp + bb] = tile[tx][r];
  }
}

// x [n_samples][B] -> sample-minor copies a[B][NS], b[B][NS] (padding samples = 0)
__global__ __launch_bounds__(256) void k_to_sample_minor(const double *__restrict__ x, int64_t B,
                                                         int n_samples, int NS, double *__restrict__ a,
                                                         double *__restrict__ b) {
  __shared__ double tile[32][33];
  const int64_t b0 = (int64_t)blockIdx.x * 32;
  const int s0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int s = s0 + r;
    const int64_t bb = b0 + tx;
    tile[r][tx] = (s < n_samples && bb < B) ? x[(int64_t)s * B + bb] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t bb = b0 + r;
    const int s = s0 + tx;
    if (bb < B && s < NS) { const double v = tile[tx][r]; a[bb * NS + s] = v; b[bb * NS + s] = v; }
  }
}

// ------------------------------------------------------------------------------------------
// np.nanmedian of a double array by 8-bit MSD radix select (one workgroup per array).
__device__ __forceinline__ unsigned long long f64_key(double x) {
  unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long kx) {
  unsigned long long b = (kx >> 63) ? (kx & 0x7fffffffffffffffull) : ~kx;
  return __longlong_as_double((long long)b);
}

constexpr int NTM = 1024;

__global__ __launch_bounds__(NTM) void k_nanmedian(const double *__restrict__ a0,
                                                   const double *__restrict__ a1, int64_t n,
                                                   int64_t array_stride, double *__restrict__ out0,
                                                   double *__restrict__ out1) {
  const double *a = (blockIdx.y ? a1 : a0) + (int64_t)blockIdx.x * array_stride;
  double *out = blockIdx.y ? out1 : out0;
  __shared__ unsigned int hist[2][256];
  __shared__ unsigned long long prefix[2];
  __shared__ long long rank[2];
  __shared__ long long nvalid;
  const int tid = threadIdx.x;
  if (tid == 0) nvalid = 0;
  __syncthreads();
  // (every walk over the array keeps NMU loads per thread in flight: with one, a workgroup's 16 waves
  //  moved 4 GB/s -- nine walks of 182 k doubles took 1.6 ms per call in a 96-sample batch)
  constexpr int NMU = 8;
  long long loc = 0;
  for (int64_t i0 = 0; i0 < n; i0 += (int64_t)NTM * NMU) {
    double xs[NMU];
#pragma unroll
    for (int u = 0; u < NMU; ++u) {
      const int64_t i = i0 + (int64_t)u * NTM + tid;
      xs[u] = i < n ? a[i] : __builtin_nan("");
    }
#pragma unroll
    for (int u = 0; u < NMU; ++u) loc += (xs[u] == xs[u]);
    if (i0 + NTM >= n) break;                      // (short rows -- the PCA stage's 500-value medians)
  }
  loc = (long long)wcx::wave_sum_i((int)loc);
  if ((tid & 63) == 0) atomicAdd((unsigned long long *)&nvalid, (unsigned long long)loc);
  __syncthreads();
  const long long nv = nvalid;
  if (nv == 0) { if (tid == 0) out[blockIdx.x] = __builtin_nan(""); return; }
  if (tid == 0) { rank[0] = (nv - 1) / 2; rank[1] = nv / 2; prefix[0] = 0; prefix[1] = 0; }
  unsigned long long himask = 0;
  for (int shift = 56; shift >= 0; shift -= 8) {
    if (tid < 256) { hist[0][tid] = 0; hist[1][tid] = 0; }
    __syncthreads();
    const unsigned long long p0 = prefix[0], p1 = prefix[1];
    const bool two = p1 != p0;                     // the two middle ranks usually share a prefix
    // wave-aggregated counting: in the leading digits (sign, exponent) whole waves agree, and 64
    // lanes adding to one LDS counter would serialise
    auto count = [&](bool on, unsigned int dg, unsigned int *h) {
      const unsigned long long m = __ballot(on);
      if (m == 0ull) return;
      const int first = __ffsll((long long)m) - 1;
      const unsigned int dgf = (unsigned int)__builtin_amdgcn_readlane((int)dg, first);
      if (__ballot(on && dg == dgf) == m) {
        if ((tid & 63) == first) atomicAdd(&h[dgf], (unsigned int)__popcll(m));
      } else if (on) {
        atomicAdd(&h[dg], 1u);
      }
    };
    for (int64_t i0 = 0; i0 < n; i0 += (int64_t)NTM * NMU) {
      double xs[NMU];
#pragma unroll
      for (int u = 0; u < NMU; ++u) {
        const int64_t i = i0 + (int64_t)u * NTM + tid;
        xs[u] = i < n ? a[i] : __builtin_nan("");
      }
#pragma unroll
      for (int u = 0; u < NMU; ++u) {
        if (i0 + (int64_t)u * NTM >= n) break;       // wave-uniform: nothing left in this walk
        const double x = xs[u];
        const bool ok = x == x;
        const unsigned long long kx = f64_key(x);
        const unsigned int dg = (unsigned int)((kx >> shift) & 255ull);
        count(ok && (kx & himask) == p0, dg, hist[0]);
        if (two) count(ok && (kx & himask) == p1, dg, hist[1]);
      }
    }
    __syncthreads();
    if (!two && tid < 256) hist[1][tid] = hist[0][tid];
    __syncthreads();
    if (tid < 2) {
      long long rk = rank[tid];
      int dg = 0;
      for (; dg < 256; ++dg) {
        const long long h = hist[tid][dg];
        if (rk < h) break;
        rk -= h;
      }
      rank[tid] = rk;
      prefix[tid] |= ((unsigned long long)dg) << shift;
    }
    himask |= 255ull << shift;
    __syncthreads();
  }
  if (tid == 0) {
    const double lo = key_f64(prefix[0]), hi = key_f64(prefix[1]);
    out[blockIdx.x] = (nv & 1) ? lo : (lo + hi) / 2.0;
  }
}

// ---- nanmedian of ONE long vector pair with the whole chip ---------------------------------------
// The single-workgroup k_nanmedian above is fine for a batch (one workgroup per sample) but for a
// single sample it leaves one CU walking 2 x 182 k doubles nine times (0.5 ms).  Here every pass is
// spread over many workgroups: LDS-private digit histograms -> global histogram -> a one-wave
// "decide" kernel that advances the two rank prefixes.  state[a] = {prefix0, prefix1, rank0, rank1,
// himask, nvalid} (as 64-bit words), hist[a][2][256].
struct NmState { unsigned long long prefix[2]; long long rank[2]; unsigned long long himask; long long nvalid; };

__global__ __launch_bounds__(256) void k_nm_count(const double *__restrict__ a0,
                                                  const double *__restrict__ a1, int64_t n,
                                                  NmState *__restrict__ st) {
  const double *a = blockIdx.y ? a1 : a0;
  long long loc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double x = a[i];
    loc += (x == x);
  }
  loc = (long long)wcx::wave_sum_i((int)loc);
  if ((threadIdx.x & 63) == 0 && loc) atomicAdd((unsigned long long *)&st[blockIdx.y].nvalid, (unsigned long long)loc);
}

__global__ void k_nm_init(NmState *__restrict__ st, unsigned int *__restrict__ hist) {
  const int a = blockIdx.x;
  if (threadIdx.x == 0) {
    const long long nv = st[a].nvalid;
    st[a].rank[0] = nv > 0 ? (nv - 1) / 2 : 0;
    st[a].rank[1] = nv / 2;
    st[a].prefix[0] = st[a].prefix[1] = 0;
    st[a].himask = 0;
  }
  for (int i = threadIdx.x; i < 512; i += blockDim.x) hist[a * 512 + i] = 0;
}

__global__ __launch_bounds__(256) void k_nm_hist(const double *__restrict__ a0,
                                                 const double *__restrict__ a1, int64_t n, int shift,
                                                 const NmState *__restrict__ st,
                                                 unsigned int *__restrict__ hist) {
  const int arr = blockIdx.y;
  const double *a = arr ? a1 : a0;
  __shared__ unsigned int lh[2][256];
  lh[0][threadIdx.x] = 0;
  lh[1][threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long p0 = st[arr].prefix[0], p1 = st[arr].prefix[1], himask = st[arr].himask;
  const bool two = p1 != p0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double x = a[i];
    if (x == x) {
      const unsigned long long kx = f64_key(x);
      const unsigned int dg = (unsigned int)((kx >> shift) & 255ull);
      if ((kx & himask) == p0) atomicAdd(&lh[0][dg], 1u);
      if (two && (kx & himask) == p1) atomicAdd(&lh[1][dg], 1u);
    }
  }
  __syncthreads();
  if (lh[0][threadIdx.x]) atomicAdd(&hist[arr * 512 + threadIdx.x], lh[0][threadIdx.x]);
  if (two && lh[1][threadIdx.x]) atomicAdd(&hist[arr * 512 + 256 + threadIdx.x], lh[1][threadIdx.x]);
}

__global__ __launch_bounds__(512) void k_nm_decide(NmState *__restrict__ st,
                                                    unsigned int *__restrict__ hist, int shift,
                                                    double *__restrict__ out0,
                                                    double *__restrict__ out1) {
  // 512 threads: half t = 0/1 scans the histogram of rank t's prefix (a shared one if equal)
  const int a = blockIdx.x;
  const int t = threadIdx.x >> 8, d = threadIdx.x & 255;
  const bool two = st[a].prefix[1] != st[a].prefix[0];
  __shared__ long long cum[2][256];
  const long long rk = st[a].rank[t];
  const long long mine = hist[a * 512 + ((t == 1 && two) ? 256 : 0) + d];
  cum[t][d] = mine;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const long long v = d >= off ? cum[t][d - off] : 0;
    __syncthreads();
    cum[t][d] += v;
    __syncthreads();
  }
  const long long incl = cum[t][d], excl = incl - mine;
  __syncthreads();
  if (rk >= excl && rk < incl) {          // exactly one digit per half (rank < total count)
    st[a].rank[t] = rk - excl;
    st[a].prefix[t] |= ((unsigned long long)d) << shift;
  }
  hist[a * 512 + threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    st[a].himask |= 255ull << shift;
    if (shift == 0) {
      double *out = a ? out1 : out0;
      const long long nv = st[a].nvalid;
      const double lo = key_f64(st[a].prefix[0]), hi = key_f64(st[a].prefix[1]);
      out[0] = nv == 0 ? __builtin_nan("") : ((nv & 1) ? lo : (lo + hi) / 2.0);
    }
  }
}

__global__ void k_copy2(const double *__restrict__ src, double *__restrict__ a,
                        double *__restrict__ b, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double v = src[i]; a[i] = v; b[i] = v; }
}

}  // namespace

// nanmedian of two vectors of n doubles (one sample): the multi-workgroup radix select
static int launch_nanmedian_wide(wcx_ctx *ctx, const double *a0, const double *a1, int64_t n,
                                 double *out0, double *out1) {
  void *scr = ctx->d_small;
  NmState *st = reinterpret_cast<NmState *>(scr);
  unsigned int *hist = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(scr) + 256);
  hipStream_t s = ctx->stream;
  WCX_HIP(hipMemsetAsync(st, 0, 2 * sizeof(NmState), s));
  const unsigned g = (unsigned)std::min<int64_t>(256, (n + 255) / 256 > 0 ? (n + 255) / 256 : 1);
  k_nm_count<<<dim3(g, 2), 256, 0, s>>>(a0, a1, n, st);
  k_nm_init<<<2, 256, 0, s>>>(st, hist);
  for (int shift = 56; shift >= 0; shift -= 8) {
    k_nm_hist<<<dim3(g, 2), 256, 0, s>>>(a0, a1, n, shift, st, hist);
    k_nm_decide<<<2, 512, 0, s>>>(st, hist, shift, out0, out1);
  }
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

// np.nanmedian of `count` rows of n doubles (row r at d_a + r * stride) -> d_out[count]; used by the
// PCA stage (pca.hip) for np.median(X, axis=0).
int wcx_nanmedian_rows_launch(wcx_ctx *ctx, const double *d_a, int64_t n, int64_t stride, int count,
                              double *d_out) {
  k_nanmedian<<<dim3((unsigned)count, 1), NTM, 0, ctx->stream>>>(d_a, d_a, n, stride, d_out, d_out);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

namespace {

// mean of the non-NaN entries (np.nanmean) of w[0..n), one workgroup
__global__ __launch_bounds__(1024) void k_nanmean1(const double *__restrict__ w, int64_t n,
                                                   double *__restrict__ out) {
  __shared__ double ss[16];
  __shared__ double sc[16];
  __shared__ double sb[16];
  double s = 0.0, c = 0.0, bad = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double v = w[i];
    if (v == v) { s += v; c += 1.0; }
    if (!(v - v == 0.0)) bad += 1.0;                         // NaN or +-inf
  }
  s = wcx::wave_sum(s);
  c = wcx::wave_sum(c);
  bad = wcx::wave_sum(bad);
  if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = s; sc[threadIdx.x >> 6] = c; sb[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, nb = 0.0;
    for (int q = 0; q < 16; ++q) { a += ss[q]; b += sc[q]; nb += sb[q]; }
    const double m = a / b;
    out[0] = m;
    // main.py:252-256: any NaN / inf in w / nanmean(w) -> all weights 1 ("reference too small")
    out[1] = (nb > 0.0 || !(m - m == 0.0) || m == 0.0) ? 1.0 : 0.0;
  }
}

// get_post_processed_result x3 + log_trans (predict_control.py:49-63, predict_tools.py:163-193)
// for the autosomal results of one sample, on the masked vectors; full = r | z | w over the
// unmasked bins (pre-zeroed: masked-out bins stay 0).  z is shifted by m_z and w scaled by its
// nanmean first (main.py:246-250 for a sample without gonosomal pass).
__global__ __launch_bounds__(256) void k_post_process(
    const double *__restrict__ z, const double *__restrict__ r, const double *__restrict__ nref,
    const double *__restrict__ w, int64_t B, const double *__restrict__ m_lr,
    const double *__restrict__ m_z, const double *__restrict__ wmean, double minrefbins,
    const int32_t *__restrict__ pos, int64_t n_bins, double *__restrict__ full) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  const bool keep = nref[i] >= minrefbins;
  double lr = log2(keep ? r[i] : 0.0);
  const bool good = (lr - lr == 0.0);                      // finite
  lr = good ? lr : 0.0;
  if (lr != 0.0) lr -= m_lr[0];
  const int64_t p = pos[i];
  full[p] = lr;
  full[n_bins + p] = good ? z[i] - m_z[0] : 0.0;
  full[2 * n_bins + p] = good ? (wmean[1] != 0.0 ? 1.0 : w[i] / wmean[0]) : 0.0;
}

// ---- sample preparation of a batch on the device (wcx_predict_prep_dev) -----------------------------
// total read count of a sample over the bins of this pass (predict_tools.py:46: before masking)
__global__ __launch_bounds__(1024) void k_prep_total(const int32_t *__restrict__ counts, int64_t n_bins,
                                                     double *__restrict__ total) {
  __shared__ double red[16];
  const int32_t *c = counts + (int64_t)blockIdx.x * n_bins;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n_bins; i += 1024) s += (double)c[i];   // (integers: exact below 2^53)
  s = wcx::wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += red[q];
    total[blockIdx.x] = t;
  }
}

// x[s][i] = counts[s][pos[i]] / total[s]; partial dot products of (x - mean) with the NC components
// (PSL fixed slices of the bins per sample, reduced in order by k_prep_project: deterministic)
constexpr int PSL = 64;
template <int NC>
__global__ __launch_bounds__(256) void k_prep_depth(const int32_t *__restrict__ counts, int64_t n_bins,
                                                    const double *__restrict__ total,
                                                    const int32_t *__restrict__ pos, int64_t B,
                                                    const double *__restrict__ mean,
                                                    const double *__restrict__ comps,
                                                    double *__restrict__ x, double *__restrict__ part) {
  __shared__ double red[NC][4];
  const int s = blockIdx.y, sl = blockIdx.x;
  const int64_t per = (B + PSL - 1) / PSL, lo = sl * per, hi = lo + per < B ? lo + per : B;
  const int32_t *c = counts + (int64_t)s * n_bins;
  const double tot = total[s];
  double acc[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) acc[q] = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double v = (double)c[pos[i]] / tot;
    x[(int64_t)s * B + i] = v;
    const double e = v - mean[i];
#pragma unroll
    for (int q = 0; q < NC; ++q) acc[q] += comps[(int64_t)q * B + i] * e;
  }
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    const double t = wcx::wave_sum(acc[q]);
    if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x < NC)
    part[((int64_t)s * PSL + sl) * NC + threadIdx.x] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// predict_tools.py:56-65 (scikit-learn <= 1.4 transform): x / ((t . C) + mean), t = (x - mean) . C^T
template <int NC>
__global__ __launch_bounds__(256) void k_prep_project(const double *__restrict__ part, int64_t B,
                                                      const double *__restrict__ mean,
                                                      const double *__restrict__ comps,
                                                      double *__restrict__ x) {
  const int s = blockIdx.y;
  double t[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    double a = 0.0;
    for (int sl = 0; sl < PSL; ++sl) a += part[((int64_t)s * PSL + sl) * NC + q];
    t[q] = a;
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  double rec = mean[i];
#pragma unroll
  for (int q = 0; q < NC; ++q) rec += t[q] * comps[(int64_t)q * B + i];
  x[(int64_t)s * B + i] = x[(int64_t)s * B + i] / rec;
}

// ---- A + gonosome merge + post-processing of a batch (wcx_post_process_merge_dev) ----------------
// weight statistics of one weight vector: {nanmean, count of non-NaN, non-finite entries}
__global__ __launch_bounds__(1024) void k_wstats(const double *__restrict__ w, int64_t n,
                                                 double *__restrict__ out) {
  __shared__ double ss[16], sc[16], sb[16];
  double s = 0.0, c = 0.0, bad = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double v = w[i];
    if (v == v) { s += v; c += 1.0; }
    if (!(v - v == 0.0)) bad += 1.0;
  }
  s = wcx::wave_sum(s); c = wcx::wave_sum(c); bad = wcx::wave_sum(bad);
  if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = s; sc[threadIdx.x >> 6] = c; sb[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, nb = 0.0;
    for (int q = 0; q < 16; ++q) { a += ss[q]; b += sc[q]; nb += sb[q]; }
    out[0] = a / b; out[1] = b; out[2] = nb;
  }
}

// main.py:247-256: w = append(wA * nanmean(wG), wG * nanmean(wA)); w /= nanmean(w); any NaN / inf
// left -> all ones.  One workgroup: first the nanmean of the concatenation, then the vector.
// stA / stG = k_wstats of the two parts (stG unused when BG == 0: w = wA / nanmean(wA)).
__global__ __launch_bounds__(1024) void k_wmerge(const double *__restrict__ wA, int64_t BA,
                                                 const double *__restrict__ wG, int64_t BG,
                                                 const double *__restrict__ stA,
                                                 const double *__restrict__ stG,
                                                 double *__restrict__ wout, int *__restrict__ fallback) {
  __shared__ double ss[16], sc[16];
  __shared__ double s_mean;
  __shared__ int s_bad;
  const double mA = stA[0], mG = BG ? stG[0] : 1.0;
  const double fa = BG ? mG : 1.0, fg = mA;          // scale of the autosomal / gonosomal part
  double s = 0.0, c = 0.0;
  for (int64_t i = threadIdx.x; i < BA + BG; i += 1024) {
    const double v = i < BA ? wA[i] * fa : wG[i - BA] * fg;
    if (v == v) { s += v; c += 1.0; }
  }
  s = wcx::wave_sum(s); c = wcx::wave_sum(c);
  if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = s; sc[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < 16; ++q) { a += ss[q]; b += sc[q]; }
    const double m = a / b;
    s_mean = m;
    // a non-finite weight, scale or mean anywhere makes some w / mean non-finite
    const bool bad = stA[2] > 0.0 || (BG && stG[2] > 0.0) || !(m - m == 0.0) || m == 0.0 ||
                     !(fa - fa == 0.0) || !(fg - fg == 0.0);
    s_bad = bad ? 1 : 0;
    *fallback = s_bad;
  }
  __syncthreads();
  const double m = s_mean;
  const bool bad = s_bad != 0;
  for (int64_t i = threadIdx.x; i < BA + BG; i += 1024) {
    const double v = i < BA ? wA[i] * fa : wG[i - BA] * fg;
    wout[i] = bad ? 1.0 : v / m;
  }
}

// per (sample, merged masked bin): minrefbins rule, z - m_z(A), log2 transform, inflation
__global__ __launch_bounds__(256) void k_post_merge(
    const double *__restrict__ zA, const double *__restrict__ rA, const double *__restrict__ nA,
    int64_t BA, const double *__restrict__ zG, const double *__restrict__ rG,
    const double *__restrict__ nG, int64_t BG, const double *__restrict__ wfin,
    const double *__restrict__ m_lr, const double *__restrict__ m_z, double minrefbins,
    const int32_t *__restrict__ pos, int64_t n_bins, double *__restrict__ out_r,
    double *__restrict__ out_z, double *__restrict__ out_w) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  if (i >= BA + BG) return;
  const bool aut = i < BA;
  const int64_t j = aut ? (int64_t)s * BA + i : (int64_t)s * BG + (i - BA);
  const double nref = aut ? nA[j] : nG[j], rr = aut ? rA[j] : rG[j], zz = aut ? zA[j] : zG[j];
  const bool keep = nref >= minrefbins;
  double lr = log2(keep ? rr : 0.0);
  const bool good = (lr - lr == 0.0);                      // finite
  lr = good ? lr : 0.0;
  if (lr != 0.0) lr -= m_lr[s];
  const int64_t p = (int64_t)s * n_bins + pos[i];
  out_r[p] = lr;
  out_z[p] = good && keep ? zz - m_z[s] : 0.0;
  out_w[p] = good && keep ? wfin[i] : 0.0;
}

}  // namespace

void wcx_ref_release_sel(wcx_ctx *ctx, wcx_ref *ref);   // defined next to ensure_sel below

extern "C" {

int wcx_ref_wrap_dev(wcx_ctx *ctx, const int32_t *d_idx, const double *d_dist, int64_t B, int k,
                     const int64_t *chr_cum, int n_chr, wcx_ref **out) {
  WCX_ARG(ctx && d_idx && d_dist && chr_cum && out, "NULL argument");
  WCX_ARG(B > 0 && k > 0 && n_chr > 0 && n_chr <= 32, "bad sizes (n_chr <= 32)");
  WCX_ARG(chr_cum[n_chr - 1] == B, "chr_cum[n_chr-1] must equal B");
  wcx_ref *r = new wcx_ref();
  r->d_idx = d_idx;
  r->d_dist = d_dist;
  r->owned = false;
  r->B = B;
  r->k = k;
  r->row0 = 0;
  r->nrows = B;
  r->chr_cum.assign(chr_cum, chr_cum + n_chr);
  *out = r;
  return WCX_OK;
}

int wcx_ref_upload(wcx_ctx *ctx, const int32_t *idx, const double *dist, int64_t B, int k,
                   const int64_t *chr_cum, int n_chr, wcx_ref **out) {
  WCX_ARG(ctx && idx && dist && chr_cum && out, "NULL argument");
  WCX_ARG(B > 0 && k > 0, "bad sizes");
  WCX_HIP(hipSetDevice(ctx->device));
  void *di = nullptr, *dd = nullptr;
  const size_t ib = (size_t)B * k * 4, db = (size_t)B * k * 8;
  if (hipMalloc(&di, ib) != hipSuccess || hipMalloc(&dd, db) != hipSuccess) {
    if (di) hipFree(di);
    wcx_set_error("hipMalloc failed for a %lld x %d reference", (long long)B, k);
    return WCX_ERR_NOMEM;
  }
  WCX_HIP(hipMemcpyAsync(di, idx, ib, hipMemcpyHostToDevice, ctx->stream));
  WCX_HIP(hipMemcpyAsync(dd, dist, db, hipMemcpyHostToDevice, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  int rc = wcx_ref_wrap_dev(ctx, (const int32_t *)di, (const double *)dd, B, k, chr_cum, n_chr, out);
  if (rc) { hipFree(di); hipFree(dd); return rc; }
  (*out)->owned = true;
  return WCX_OK;
}

int wcx_ref_free(wcx_ctx *ctx, wcx_ref *ref) {
  if (!ref) return WCX_OK;
  if (ctx) wcx_ref_release_sel(ctx, ref);         // back to the context's pool: no free, no sync
  else if (ref->d_sel) hipFree(ref->d_sel);       // (no context: the pool cannot be told)
  if (ref->owned) {
    if (ctx) hipStreamSynchronize(ctx->stream);
    hipFree(const_cast<int32_t *>(ref->d_idx));
    hipFree(const_cast<double *>(ref->d_dist));
  }
  delete ref;
  return WCX_OK;
}

int wcx_cutoff(wcx_ctx *ctx, const wcx_ref *ref, int repeats, double *cutoff) {
  WCX_ARG(ctx && ref && cutoff, "NULL argument");
  WCX_ARG(repeats >= 0, "repeats must be >= 0");
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t n = ref->nrows * (int64_t)ref->k;
  const int nparts = 2048;
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, (size_t)(2 * nparts + 8) * 8, &scr);
  if (rc) return rc;
  double *state = reinterpret_cast<double *>(scr);
  double *ps = state + 8, *pc = ps + nparts;
  const double init[3] = {HUGE_VAL, 0.0, 0.0};
  rc = wcx_upload_small(ctx, state, init, sizeof(init));
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "cutoff");
  if (rc) return rc;
  for (int rep = 0; rep < repeats; ++rep) {
    for (int phase = 0; phase < 2; ++phase) {
      k_cut_partial<<<nparts, NT, 0, ctx->stream>>>(ref->d_dist, n, state, phase, ps, pc);
      k_cut_final<<<1, NT, 0, ctx->stream>>>(state, phase, ps, pc, nparts);
    }
  }
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "cutoff");
  if (rc) return rc;
  WCX_HIP(hipMemcpyAsync(cutoff, state, 8, hipMemcpyDeviceToHost, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  return WCX_OK;
}

int wcx_weights_dev(wcx_ctx *ctx, const wcx_ref *ref, double *d_out) {
  WCX_ARG(ctx && ref && d_out, "NULL argument");
  WCX_HIP(hipSetDevice(ctx->device));
  int rc = wcx_timer_begin(ctx, "weights");
  if (rc) return rc;
  const unsigned grid = (unsigned)((ref->nrows + 3) / 4 < 8192 ? (ref->nrows + 3) / 4 + 1 : 8192);
  k_weights<<<grid, NT, 0, ctx->stream>>>(ref->d_dist, ref->nrows, ref->k, d_out);
  WCX_HIP(hipGetLastError());
  return wcx_timer_end(ctx, "weights");
}

int wcx_post_process_dev(wcx_ctx *ctx, const double *d_z, const double *d_r, const double *d_n,
                         const double *d_w, int64_t B, const double *d_m_lr, const double *d_m_z,
                         double minrefbins, const int32_t *d_pos, int64_t n_bins, double *out_r,
                         double *out_z, double *out_w) {
  WCX_ARG(ctx && d_z && d_r && d_n && d_w && d_m_lr && d_m_z && d_pos && out_r && out_z && out_w,
          "NULL argument");
  WCX_ARG(B > 0 && n_bins >= B, "bad sizes");
  WCX_HIP(hipSetDevice(ctx->device));
  void *scr = nullptr;
  int rc = wcx_scratch2(ctx, (size_t)n_bins * 24 + 256, &scr);
  if (rc) return rc;
  double *full = reinterpret_cast<double *>(scr);           // r | z | w, each n_bins
  double *d_wmean = full + 3 * n_bins;
  hipStream_t st = ctx->stream;
  WCX_HIP(hipMemsetAsync(full, 0, (size_t)n_bins * 24, st));
  k_nanmean1<<<1, 1024, 0, st>>>(d_w, B, d_wmean);
  k_post_process<<<(unsigned)((B + 255) / 256), 256, 0, st>>>(d_z, d_r, d_n, d_w, B, d_m_lr, d_m_z,
                                                             d_wmean, minrefbins, d_pos, n_bins, full);
  WCX_HIP(hipGetLastError());
  WCX_HIP(hipMemcpyAsync(out_r, full, (size_t)n_bins * 8, hipMemcpyDeviceToHost, st));
  WCX_HIP(hipMemcpyAsync(out_z, full + n_bins, (size_t)n_bins * 8, hipMemcpyDeviceToHost, st));
  WCX_HIP(hipMemcpyAsync(out_w, full + 2 * n_bins, (size_t)n_bins * 8, hipMemcpyDeviceToHost, st));
  WCX_HIP(hipStreamSynchronize(st));
  return WCX_OK;
}

int wcx_predict_prep_dev(wcx_ctx *ctx, const int32_t *d_counts, int n_samples, int64_t n_bins,
                         const int32_t *d_pos, int64_t B, const double *d_mean, const double *d_comps,
                         int n_comp, double *d_x) {
  WCX_ARG(ctx && d_counts && d_pos && d_mean && d_comps && d_x, "NULL argument");
  WCX_ARG(n_samples > 0 && n_bins >= B && B > 0, "bad sizes");
  WCX_ARG(n_comp == 5, "this build instantiates 5 components (newref_tools.py:138: pcacomp=5)");
  WCX_HIP(hipSetDevice(ctx->device));
  void *scr = nullptr;
  int rc = wcx_scratch2(ctx, (size_t)n_samples * (8 + (size_t)PSL * 5 * 8) + 256, &scr);
  if (rc) return rc;
  double *d_total = reinterpret_cast<double *>(scr);
  double *d_part = d_total + n_samples;
  hipStream_t st = ctx->stream;
  k_prep_total<<<(unsigned)n_samples, 1024, 0, st>>>(d_counts, n_bins, d_total);
  k_prep_depth<5><<<dim3(PSL, (unsigned)n_samples), 256, 0, st>>>(d_counts, n_bins, d_total, d_pos, B,
                                                                  d_mean, d_comps, d_x, d_part);
  k_prep_project<5><<<dim3((unsigned)((B + 255) / 256), (unsigned)n_samples), 256, 0, st>>>(
      d_part, B, d_mean, d_comps, d_x);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

int wcx_post_process_merge_dev(wcx_ctx *ctx, const double *d_zA, const double *d_rA,
                               const double *d_nA, const double *d_wA, int64_t BA,
                               const double *d_zG, const double *d_rG, const double *d_nG,
                               const double *d_wG, int64_t BG, int n_samples, const double *d_m_lr,
                               const double *d_m_z, double minrefbins, const int32_t *d_pos,
                               int64_t n_bins, double *d_out_r, double *d_out_z, double *d_out_w,
                               int *weights_fallback) {
  WCX_ARG(ctx && d_zA && d_rA && d_nA && d_wA && d_m_lr && d_m_z && d_pos && d_out_r && d_out_z && d_out_w,
          "NULL argument");
  WCX_ARG(BG == 0 || (d_zG && d_rG && d_nG && d_wG), "gonosomal part incomplete");
  WCX_ARG(BA > 0 && BG >= 0 && n_samples > 0 && n_bins >= BA + BG, "bad sizes");
  WCX_HIP(hipSetDevice(ctx->device));
  void *scr = nullptr;
  int rc = wcx_scratch2(ctx, (size_t)(BA + BG) * 8 + 512, &scr);
  if (rc) return rc;
  double *d_wfin = reinterpret_cast<double *>(scr);
  double *d_st = d_wfin + (BA + BG);                         // stA[3] | stG[3] | fallback flag
  int *d_fb = reinterpret_cast<int *>(d_st + 8);
  hipStream_t st = ctx->stream;
  WCX_HIP(hipMemsetAsync(d_out_r, 0, (size_t)n_samples * n_bins * 8, st));
  WCX_HIP(hipMemsetAsync(d_out_z, 0, (size_t)n_samples * n_bins * 8, st));
  WCX_HIP(hipMemsetAsync(d_out_w, 0, (size_t)n_samples * n_bins * 8, st));
  k_wstats<<<1, 1024, 0, st>>>(d_wA, BA, d_st);
  if (BG) k_wstats<<<1, 1024, 0, st>>>(d_wG, BG, d_st + 3);
  k_wmerge<<<1, 1024, 0, st>>>(d_wA, BA, d_wG, BG, d_st, d_st + 3, d_wfin, d_fb);
  k_post_merge<<<dim3((unsigned)((BA + BG + 255) / 256), (unsigned)n_samples), 256, 0, st>>>(
      d_zA, d_rA, d_nA, BA, d_zG, d_rG, d_nG, BG, d_wfin, d_m_lr, d_m_z, minrefbins, d_pos, n_bins,
      d_out_r, d_out_z, d_out_w);
  WCX_HIP(hipGetLastError());
  if (weights_fallback) {
    WCX_HIP(hipMemcpyAsync(weights_fallback, d_fb, 4, hipMemcpyDeviceToHost, st));
    WCX_HIP(hipStreamSynchronize(st));
  }
  return WCX_OK;
}

int wcx_weights(wcx_ctx *ctx, const wcx_ref *ref, double *out) {
  WCX_ARG(ctx && ref && out, "NULL argument");
  WCX_HIP(hipSetDevice(ctx->device));
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, (size_t)ref->nrows * 8 + 8, &scr);
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "weights");
  if (rc) return rc;
  const unsigned grid = (unsigned)((ref->nrows + 3) / 4 < 8192 ? (ref->nrows + 3) / 4 + 1 : 8192);
  k_weights<<<grid, NT, 0, ctx->stream>>>(ref->d_dist, ref->nrows, ref->k, (double *)scr);
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "weights");
  if (rc) return rc;
  if (ref->nrows)
    WCX_HIP(hipMemcpyAsync(out, scr, (size_t)ref->nrows * 8, hipMemcpyDeviceToHost, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  return WCX_OK;
}

}  // extern "C"

namespace {

int ipl_for(int k) {
  int ipl = (k + 63) / 64;                 // values per lane
  if (ipl > 8) ipl = ipl <= 16 ? 16 : 32;
  return ipl;
}

void sel_release(wcx_ctx *ctx, wcx_ref *ref) {
  if (!ref->d_sel) return;
  for (auto &sl : ctx->sel_pool)
    if (sl.p == ref->d_sel) sl.used = false;
  ref->d_sel = nullptr;
  ref->sel_bytes = 0;
}

// selection mask of the reference's rows (lent to the handle: it outlives the scratch buffers)
int ensure_sel(wcx_ctx *ctx, wcx_ref *ref, int ipl) {
  const size_t need = (size_t)ref->nrows * ipl * 8;
  if (ref->d_sel && ref->sel_bytes >= need) return WCX_OK;
  sel_release(ctx, ref);
  // the context's pool: the smallest free buffer that fits, else a new one (all work on a context is
  // ordered on its stream, so a buffer handed from one handle to the next needs no synchronisation)
  int best = -1;
  for (size_t i = 0; i < ctx->sel_pool.size(); ++i) {
    const auto &sl = ctx->sel_pool[i];
    if (!sl.used && sl.bytes >= need && (best < 0 || sl.bytes < ctx->sel_pool[best].bytes)) best = (int)i;
  }
  if (best < 0) {
    wcx_ctx::SelSlot sl{nullptr, need ? need : 8, false};
    if (hipMalloc(&sl.p, sl.bytes) != hipSuccess) {
      wcx_set_error("hipMalloc(%zu) for the selection mask failed", need);
      return WCX_ERR_NOMEM;
    }
    ctx->sel_pool.push_back(sl);
    best = (int)ctx->sel_pool.size() - 1;
  }
  ctx->sel_pool[best].used = true;
  ref->d_sel = reinterpret_cast<unsigned long long *>(ctx->sel_pool[best].p);
  ref->sel_bytes = ctx->sel_pool[best].bytes;
  return WCX_OK;
}

// One pass over the reference's rows intersected with [ct, B).
int launch_pass(wcx_ctx *ctx, const wcx_ref *ref, const double *d_x, const double *cin, double *cout,
                int n_samples, int64_t ct, bool last, double *d_z, double *d_r, double *d_n,
                double *d_lr) {
  const int64_t B = ref->B;
  const int k = ref->k;
  const int ipl = ipl_for(k);
  const int64_t lo = ct > ref->row0 ? ct : ref->row0, hi = ref->row0 + ref->nrows;
  if (lo >= hi) return WCX_OK;
  ChrTable tab;
  tab.n_chr = (int)ref->chr_cum.size();
  for (int c = 0; c < 32; ++c) tab.cum[c] = c < tab.n_chr ? ref->chr_cum[c] : B;
  const int64_t nl = hi - lo;
  dim3 grid((unsigned)((nl + 3) / 4 < 16384 ? (nl + 3) / 4 : 16384), (unsigned)n_samples);
#define WCX_NORM_LAUNCH(IPL)                                                                  \
  k_normalize_pass<IPL><<<grid, NT, 0, ctx->stream>>>(d_x, cin, cout, ref->d_idx, ref->d_sel, B, k, \
                                                      ct, lo, hi, ref->row0, tab, d_z, d_r, d_n,    \
                                                      d_lr, last ? 1 : 0)
  switch (ipl) {
    case 1: WCX_NORM_LAUNCH(1); break;
    case 2: WCX_NORM_LAUNCH(2); break;
    case 3: WCX_NORM_LAUNCH(3); break;
    case 4: WCX_NORM_LAUNCH(4); break;
    case 5: WCX_NORM_LAUNCH(5); break;
    case 6: WCX_NORM_LAUNCH(6); break;
    case 7: WCX_NORM_LAUNCH(7); break;
    case 8: WCX_NORM_LAUNCH(8); break;
    case 16: WCX_NORM_LAUNCH(16); break;
    default: WCX_NORM_LAUNCH(32); break;
  }
#undef WCX_NORM_LAUNCH
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

// One pass of the tiled (sample-minor) kernel over all rows >= ct of a whole-reference handle.
template <int T>
int launch_pass_tile(wcx_ctx *ctx, const wcx_ref *ref, const double *d_x, const double *cin,
                     double *cout, int n_samples, int NS, int64_t ct, bool last, double *d_z,
                     double *d_r, double *d_n, double *d_lr) {
  const int64_t B = ref->B;
  const int k = ref->k;
  const int ipl = ipl_for(k);
  const int64_t lo = ct, hi = B;
  if (lo >= hi) return WCX_OK;
  ChrTable tab;
  tab.n_chr = (int)ref->chr_cum.size();
  for (int c = 0; c < 32; ++c) tab.cum[c] = c < tab.n_chr ? ref->chr_cum[c] : B;
  const int64_t nl = hi - lo;
  dim3 grid((unsigned)((nl + 3) / 4 < 16384 ? (nl + 3) / 4 : 16384), (unsigned)((n_samples + T - 1) / T));
#define WCX_NORMT_LAUNCH(IPL)                                                                     \
  k_normalize_pass_tile<IPL, T><<<grid, NT, 0, ctx->stream>>>(d_x, cin, cout, ref->d_idx, ref->d_sel, \
                                                               B, k, NS, n_samples, ct, lo, hi, tab, \
                                                               d_z, d_r, d_n, d_lr, last ? 1 : 0)
  switch (ipl) {
    case 1: WCX_NORMT_LAUNCH(1); break;
    case 2: WCX_NORMT_LAUNCH(2); break;
    case 3: WCX_NORMT_LAUNCH(3); break;
    case 4: WCX_NORMT_LAUNCH(4); break;
    case 5: WCX_NORMT_LAUNCH(5); break;
    case 6: WCX_NORMT_LAUNCH(6); break;
    case 7: WCX_NORMT_LAUNCH(7); break;
    default: WCX_NORMT_LAUNCH(8); break;
  }
#undef WCX_NORMT_LAUNCH
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

int launch_select_mask(wcx_ctx *ctx, wcx_ref *ref, double cutoff, int64_t ct) {
  const int ipl = ipl_for(ref->k);
  int rc = ensure_sel(ctx, ref, ipl);
  if (rc) return rc;
  const int64_t lo = ct > ref->row0 ? ct : ref->row0, hi = ref->row0 + ref->nrows;
  if (lo >= hi) return WCX_OK;
  const int64_t nl = hi - lo;
  const unsigned gsel = (unsigned)((nl + 3) / 4 < 16384 ? (nl + 3) / 4 : 16384);
  k_select_mask<<<gsel, NT, 0, ctx->stream>>>(ref->d_dist, ref->k, ipl, cutoff, lo, hi, ref->row0,
                                              ref->d_sel);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

}  // namespace

void wcx_ref_release_sel(wcx_ctx *ctx, wcx_ref *ref) { sel_release(ctx, ref); }

extern "C" {

int wcx_predict_normalize_dev(wcx_ctx *ctx, const wcx_ref *ref_c, const double *d_x, int n_samples,
                              double cutoff, int64_t ct, int cp, double *d_out_z,
                              double *d_out_r, double *d_out_n, double *d_out_mlr,
                              double *d_out_mz) {
  wcx_ref *ref = const_cast<wcx_ref *>(ref_c);
  WCX_ARG(ctx && ref && d_x && d_out_z && d_out_r && d_out_n && d_out_mlr && d_out_mz,
          "NULL argument");
  WCX_ARG(n_samples > 0, "n_samples must be positive");
  WCX_ARG(ref->row0 == 0 && ref->nrows == ref->B, "needs a handle holding all rows");
  const int64_t B = ref->B;
  const int n_chr = (int)ref->chr_cum.size();
  WCX_ARG(cp >= 0 && cp < n_chr, "cp out of range");
  WCX_ARG(ct == (cp ? ref->chr_cum[cp - 1] : 0), "ct must be the first row of chromosome cp");
  WCX_HIP(hipSetDevice(ctx->device));
  if (ref->k > 64 * 32) {
    wcx_set_error("refsize %d too large for the normalise kernel (max 2048)", ref->k);
    return WCX_ERR_UNSUPPORTED;
  }
  const int64_t Bp = B - ct;
  if (Bp <= 0) return WCX_OK;
  // scratch: copyA | copyB | lr[n][Bp].  One sample: copies are plain [B] vectors.  A batch: the
  // copies are SAMPLE-MINOR [B][NS] (NS = n_samples rounded up to the tile) for the tiled kernel.
  constexpr int TILE = 8;
  const bool tiled = n_samples >= 2 && ipl_for(ref->k) <= 8;
  // batches of 16 or more: passes 0 and 1 (z-mask only) run lane-per-sample (k_normalize_mask_lanes;
  // 64-sample tiles: NS is rounded up to 64, and an untouched sample-minor copy of x is kept)
  static const int lanes_min = [] { const char *e = getenv("WCX_NORM_LANES_MIN"); return e && *e ? atoi(e) : 16; }();
  const bool lanes = tiled && n_samples >= lanes_min;
  const int NS = lanes ? (n_samples + 63) / 64 * 64 : tiled ? (n_samples + TILE - 1) / TILE * TILE : n_samples;
  const size_t cp_b = (size_t)NS * B * 8;
  const size_t lr_b = (size_t)n_samples * Bp * 8;
  // incremental pass 1 (k_normalize_mask_incr): pass 0's sums / counts / mask words are kept
  static const int incr_on = [] { const char *e = getenv("WCX_NORM_INCR"); return e && *e ? atoi(e) : 1; }();
  const bool incr = lanes && incr_on;
  const size_t st_b = incr ? 2 * cp_b + (size_t)NS * B * 4 + (size_t)B * (NS / 64) * 8 + 256 : 0;
  // rank medians (round 6): the last pass's medians on the samples' RANKS (null_ratios.hip:
  // k_norm_median_rank), its statistics from the incrementally updated sums; WCX_NORM_RANKMED=0: the tiled
  // kernel does the whole last pass (and for batches beyond the ranking's 128 rows)
  static const int rankmed_on = [] { const char *e = getenv("WCX_NORM_RANKMED"); return e && *e ? atoi(e) : 1; }();
  // (few target rows -- the gonosomal pass: 10 k of 195 k bins -- do not repay ranking every bin of every
  //  sample: 96 samples, tiled last pass 1.6 ms, rank path 2.5 ms)
  const bool rankmed = incr && rankmed_on && n_samples <= 128 && B < (1ll << 25) &&
                       (int64_t)n_samples * B < (1ll << 31) && Bp * 4 >= B;
  const size_t mw_b = (size_t)B * (NS / 64) * 8 + 256;
  const size_t rk_b = rankmed ? 2 * mw_b + 4 * cp_b + wcx_rank_bytes(B, n_samples) + 1024 : 0;
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, (lanes ? 3 : 2) * cp_b + lr_b + st_b + rk_b, &scr);
  if (rc) return rc;
  double *cA = reinterpret_cast<double *>(scr);
  double *cB = cA + (size_t)NS * B;
  double *lr = cB + (size_t)NS * B;
  double *xT = lanes ? lr + (size_t)n_samples * Bp : nullptr;
  double *stS1 = incr ? xT + (size_t)NS * B : nullptr;
  double *stS2 = incr ? stS1 + (size_t)NS * B : nullptr;
  unsigned long long *stM = incr ? reinterpret_cast<unsigned long long *>(stS2 + (size_t)NS * B) : nullptr;
  int *stN = incr ? reinterpret_cast<int *>(stM + (size_t)B * (NS / 64) + 8) : nullptr;
  char *rkp = rankmed ? reinterpret_cast<char *>(stN) + ((size_t)NS * B * 4 + 255) / 256 * 256 : nullptr;
  unsigned long long *stM1 = reinterpret_cast<unsigned long long *>(rkp);          // masked at pass 1
  unsigned long long *stM01 = reinterpret_cast<unsigned long long *>(rkp + mw_b);  // (unused words: spare)
  double *zT = reinterpret_cast<double *>(rkp + 2 * mw_b);
  double *nT = rankmed ? zT + (size_t)NS * B : nullptr;
  double *rT = rankmed ? nT + (size_t)NS * B : nullptr;
  double *lrT = rankmed ? rT + (size_t)NS * B : nullptr;
  char *rank_base = rankmed ? reinterpret_cast<char *>(lrT + (size_t)NS * B) : nullptr;
  (void)stM01;
  rc = wcx_timer_begin(ctx, "normalize");
  if (rc) return rc;
  WcxRankView rkv{nullptr, nullptr, 0};
  if (rankmed) {
    rc = wcx_rank_rows_launch(d_x, B, n_samples, rank_base, ctx->stream, &rkv);
    if (rc) return rc;
  }
  if (tiled) {
    k_to_sample_minor<<<dim3((unsigned)((B + 31) / 32), (unsigned)((NS + 31) / 32)), 256, 0,
                        ctx->stream>>>(d_x, B, n_samples, NS, cA, cB);
    if (lanes)
      k_to_sample_minor<<<dim3((unsigned)((B + 31) / 32), (unsigned)((NS + 31) / 32)), 256, 0,
                          ctx->stream>>>(d_x, B, n_samples, NS, xT, xT);
  } else {
    const int64_t ntot = (int64_t)n_samples * B;
    k_copy2<<<(unsigned)((ntot + 255) / 256), 256, 0, ctx->stream>>>(d_x, cA, cB, ntot);
  }
  rc = launch_select_mask(ctx, ref, cutoff, ct);
  if (rc) return rc;
  for (int pass = 0; pass < 3; ++pass) {  // predict_tools.py:99
    const double *cin = (pass & 1) ? cB : cA;
    double *cout = (pass & 1) ? cA : cB;
    if (lanes && pass < 2) {
      ChrTable tab;
      tab.n_chr = (int)ref->chr_cum.size();
      for (int c = 0; c < 32; ++c) tab.cum[c] = c < tab.n_chr ? ref->chr_cum[c] : B;
      const dim3 grid((unsigned)((Bp + 3) / 4 < 16384 ? (Bp + 3) / 4 : 16384), (unsigned)(NS / 64));
      if (incr && pass == 0) {
        // (rows below ct are never normalised here but ARE reference bins of the others: no mask bits)
        if (ct > 0) WCX_HIP(hipMemsetAsync(stM, 0, (size_t)ct * (NS / 64) * 8, ctx->stream));
        k_normalize_mask_lanes<<<grid, NT, 0, ctx->stream>>>(xT, cin, cout, ref->d_idx, ref->d_sel, B, ref->k,
                                                             ipl_for(ref->k), NS, ct, B, tab, stS1, stS2, stN, stM);
      } else if (incr) {
        if (rankmed && ct > 0) WCX_HIP(hipMemsetAsync(stM1, 0, (size_t)ct * (NS / 64) * 8, ctx->stream));
        k_normalize_mask_incr<<<grid, NT, 0, ctx->stream>>>(xT, cout, ref->d_idx, ref->d_sel, B, ref->k,
                                                            ipl_for(ref->k), NS, ct, B, tab, stS1, stS2, stN, stM,
                                                            nullptr, rankmed ? stM1 : nullptr);
      } else {
        k_normalize_mask_lanes<<<grid, NT, 0, ctx->stream>>>(xT, cin, cout, ref->d_idx, ref->d_sel, B, ref->k,
                                                             ipl_for(ref->k), NS, ct, B, tab);
      }
      WCX_HIP(hipGetLastError());
    } else if (rankmed) {
      // the last pass without the tiled kernel: statistics by one more incremental update (the bins pass 1
      // masked leave the sums), medians on ranks; all four outputs sample-minor, transposed at the end
      ChrTable tab;
      tab.n_chr = (int)ref->chr_cum.size();
      for (int c = 0; c < 32; ++c) tab.cum[c] = c < tab.n_chr ? ref->chr_cum[c] : B;
      WcxChrCum cc;
      cc.n_chr = tab.n_chr;
      for (int c = 0; c < 32; ++c) cc.cum[c] = tab.cum[c];
      const dim3 grid((unsigned)((Bp + 3) / 4 < 16384 ? (Bp + 3) / 4 : 16384), (unsigned)(NS / 64));
      k_normalize_mask_incr<<<grid, NT, 0, ctx->stream>>>(xT, const_cast<double *>(cin), ref->d_idx, ref->d_sel, B,
                                                          ref->k, ipl_for(ref->k), NS, ct, B, tab, stS1, stS2, stN,
                                                          stM1, nullptr, nullptr, zT, nT);
      WCX_HIP(hipGetLastError());
      rc = wcx_norm_rank_mark_launch(rkv, cin, B, NS, n_samples, ctx->stream);   // cin = the copy pass 1 wrote
      if (rc) return rc;
      rc = wcx_norm_median_rank_launch(rkv, ref->d_idx, ref->d_sel, xT, B, ref->k, ipl_for(ref->k), NS, n_samples,
                                       ct, cc, rT, lrT, ctx->stream);
      if (rc) return rc;
      const dim3 gt((unsigned)((Bp + 31) / 32), (unsigned)((n_samples + 31) / 32));
      k_from_sample_minor<<<gt, 256, 0, ctx->stream>>>(zT, ct, Bp, NS, n_samples, d_out_z);
      k_from_sample_minor<<<gt, 256, 0, ctx->stream>>>(nT, ct, Bp, NS, n_samples, d_out_n);
      k_from_sample_minor<<<gt, 256, 0, ctx->stream>>>(rT, ct, Bp, NS, n_samples, d_out_r);
      k_from_sample_minor<<<gt, 256, 0, ctx->stream>>>(lrT, ct, Bp, NS, n_samples, lr);
      WCX_HIP(hipGetLastError());
    } else if (tiled)
      rc = launch_pass_tile<TILE>(ctx, ref, d_x, cin, cout, n_samples, NS, ct, pass == 2, d_out_z,
                                  d_out_r, d_out_n, lr);
    else
      rc = launch_pass(ctx, ref, d_x, cin, cout, n_samples, ct, pass == 2, d_out_z, d_out_r, d_out_n,
                       lr);
    if (rc) return rc;
  }
  // m_lr = nanmedian(log2 r), m_z = nanmedian(z)   (predict_tools.py:105-106)
  if (n_samples == 1) {
    rc = launch_nanmedian_wide(ctx, lr, d_out_z, Bp, d_out_mlr, d_out_mz);
    if (rc) return rc;
  } else {
    k_nanmedian<<<dim3((unsigned)n_samples, 2), NTM, 0, ctx->stream>>>(lr, d_out_z, Bp, Bp,
                                                                        d_out_mlr, d_out_mz);
  }
  WCX_HIP(hipGetLastError());
  return wcx_timer_end(ctx, "normalize");
}

// ---- row-sharded predict (multi-GPU): the handle holds rows [row0, row0+nrows) of the reference
int wcx_ref_wrap_rows_dev(wcx_ctx *ctx, const int32_t *d_idx, const double *d_dist, int64_t B, int k,
                          const int64_t *chr_cum, int n_chr, int64_t row0, int64_t nrows,
                          wcx_ref **out) {
  WCX_ARG(row0 >= 0 && nrows >= 0 && row0 + nrows <= B, "bad row range");
  int rc = wcx_ref_wrap_dev(ctx, d_idx, d_dist, B, k, chr_cum, n_chr, out);
  if (rc) return rc;
  (*out)->row0 = row0;
  (*out)->nrows = nrows;
  return WCX_OK;
}

int wcx_cutoff_moments_dev(wcx_ctx *ctx, const wcx_ref *ref, double cutoff, double mean, int phase,
                           double *out2) {
  WCX_ARG(ctx && ref && out2 && (phase == 0 || phase == 1), "bad argument");
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t n = ref->nrows * (int64_t)ref->k;
  const int nparts = 2048;
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, (size_t)(2 * nparts + 16) * 8, &scr);
  if (rc) return rc;
  double *state = reinterpret_cast<double *>(scr);
  double *ps = state + 16, *pc = ps + nparts;
  const double init[3] = {cutoff, mean, 0.0};
  rc = wcx_upload_small(ctx, state, init, sizeof(init));
  if (rc) return rc;
  if (n > 0) k_cut_partial<<<nparts, NT, 0, ctx->stream>>>(ref->d_dist, n, state, phase, ps, pc);
  else WCX_HIP(hipMemsetAsync(ps, 0, (size_t)2 * nparts * 8, ctx->stream));
  k_sum_parts<<<1, NT, 0, ctx->stream>>>(ps, pc, nparts, state + 8);
  WCX_HIP(hipGetLastError());
  WCX_HIP(hipMemcpyAsync(out2, state + 8, 16, hipMemcpyDeviceToHost, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  return WCX_OK;
}

int wcx_predict_pass_dev(wcx_ctx *ctx, wcx_ref *ref, const double *d_x, const double *d_copy_in,
                         double *d_copy_out, double cutoff, int64_t ct, int build_mask, int last,
                         double *d_z, double *d_r, double *d_n, double *d_lr) {
  WCX_ARG(ctx && ref && d_x && d_copy_in && d_copy_out && d_z && d_r && d_n && d_lr, "NULL argument");
  WCX_HIP(hipSetDevice(ctx->device));
  if (ref->k > 64 * 32) {
    wcx_set_error("refsize %d too large for the normalise kernel (max 2048)", ref->k);
    return WCX_ERR_UNSUPPORTED;
  }
  int rc = WCX_OK;
  if (build_mask) {
    rc = wcx_timer_begin(ctx, "normalize");
    if (rc) return rc;
    rc = launch_select_mask(ctx, ref, cutoff, ct);
    if (rc) return rc;
  }
  WCX_ARG(ref->d_sel != nullptr || ref->nrows == 0, "first pass must build the selection mask");
  rc = launch_pass(ctx, ref, d_x, d_copy_in, d_copy_out, 1, ct, last != 0, d_z, d_r, d_n, d_lr);
  if (rc) return rc;
  if (last) rc = wcx_timer_end(ctx, "normalize");
  return rc;
}

int wcx_nanmedian2_dev(wcx_ctx *ctx, const double *d_a0, const double *d_a1, int64_t n,
                       double *d_out0, double *d_out1) {
  WCX_ARG(ctx && d_a0 && d_a1 && d_out0 && d_out1 && n >= 0, "bad argument");
  WCX_HIP(hipSetDevice(ctx->device));
  return launch_nanmedian_wide(ctx, d_a0, d_a1, n, d_out0, d_out1);
}

int wcx_predict_normalize(wcx_ctx *ctx, const wcx_ref *ref, const double *x, int n_samples,
                          double cutoff, int64_t ct, int cp, double *out_z, double *out_r,
                          double *out_n, double *out_mlr, double *out_mz) {
  WCX_ARG(ctx && ref && x && out_z && out_r && out_n && out_mlr && out_mz, "NULL argument");
  WCX_ARG(n_samples > 0, "n_samples must be positive");
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t B = ref->B, Bp = B - ct;
  WCX_ARG(Bp >= 0, "ct beyond the last row");
  const size_t xb = (size_t)n_samples * B * 8, ob = (size_t)n_samples * (Bp > 0 ? Bp : 0) * 8;
  void *buf = nullptr;
  int rc = wcx_scratch2(ctx, xb + 3 * ob + (size_t)n_samples * 16 + 64, &buf);
  if (rc) return rc;
  double *dx = reinterpret_cast<double *>(buf);
  double *dz = dx + (size_t)n_samples * B;
  double *dr = dz + ob / 8, *dn = dr + ob / 8;
  double *dmlr = dn + ob / 8, *dmz = dmlr + n_samples;
  WCX_HIP(hipMemcpyAsync(dx, x, xb, hipMemcpyHostToDevice, ctx->stream));
  rc = wcx_predict_normalize_dev(ctx, ref, dx, n_samples, cutoff, ct, cp, dz, dr, dn, dmlr, dmz);
  if (rc) return rc;
  if (ob) {
    WCX_HIP(hipMemcpyAsync(out_z, dz, ob, hipMemcpyDeviceToHost, ctx->stream));
    WCX_HIP(hipMemcpyAsync(out_r, dr, ob, hipMemcpyDeviceToHost, ctx->stream));
    WCX_HIP(hipMemcpyAsync(out_n, dn, ob, hipMemcpyDeviceToHost, ctx->stream));
    WCX_HIP(hipMemcpyAsync(out_mlr, dmlr, (size_t)n_samples * 8, hipMemcpyDeviceToHost, ctx->stream));
    WCX_HIP(hipMemcpyAsync(out_mz, dmz, (size_t)n_samples * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  return WCX_OK;
}

}  // extern "C"
