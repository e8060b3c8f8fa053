// Wave64-wide register bitonic sort and reductions (gfx950: wavefront = 64 lanes).
//
// A wave holds N = 64*IPL doubles, IPL per lane, element e = r*64 + lane ("lane-minor"), so
// that compare-exchange distances >= 64 stay inside a lane (register <-> register, static
// indices after unrolling) and distances < 64 are lane exchanges (ds_bpermute via __shfl_xor).
// No LDS, no barriers.  NaNs must be filtered by the caller (fmin/fmax drop them).
#pragma once
#include <hip/hip_runtime.h>

namespace wcx {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  return __shfl_xor(v, mask, 64);
}

// ---- DPP cross-lane primitives (VALU speed; ds_bpermute-based shuffles cost ~100 cycles each).
// Wave64 inclusive scan schedule on gfx9-family DPP: row_shr 1,2,4,8 inside each row of 16 lanes,
// then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3.  Lanes without a source keep
// `old` (bound_ctrl off), so `old` must be the operation's identity.
template <int CTRL, int RM>
__device__ __forceinline__ int dpp_i32(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, RM, 0xf, false);
}
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_f64(double old, double src) {
  const long long o = __double_as_longlong(old), x = __double_as_longlong(src);
  const int lo = dpp_i32<CTRL, RM>((int)(o & 0xffffffffll), (int)(x & 0xffffffffll));
  const int hi = dpp_i32<CTRL, RM>((int)(o >> 32), (int)(x >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ int wave_incl_scan_i(int v) {
  v += dpp_i32<0x111, 0xf>(0, v);
  v += dpp_i32<0x112, 0xf>(0, v);
  v += dpp_i32<0x114, 0xf>(0, v);
  v += dpp_i32<0x118, 0xf>(0, v);
  v += dpp_i32<0x142, 0xa>(0, v);
  v += dpp_i32<0x143, 0xc>(0, v);
  return v;
}

__device__ __forceinline__ int wave_sum_i(int v) {
  return __builtin_amdgcn_readlane(wave_incl_scan_i(v), 63);
}

// OR of a 32-bit mask over the wave, as a wave-uniform (scalar) value.
__device__ __forceinline__ unsigned int wave_or_u32(unsigned int m) {
  int v = (int)m;
  v |= dpp_i32<0x111, 0xf>(0, v);
  v |= dpp_i32<0x112, 0xf>(0, v);
  v |= dpp_i32<0x114, 0xf>(0, v);
  v |= dpp_i32<0x118, 0xf>(0, v);
  v |= dpp_i32<0x142, 0xa>(0, v);
  v |= dpp_i32<0x143, 0xc>(0, v);
  return (unsigned int)__builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0x111, 0xf>(0.0, v);
  v += dpp_f64<0x112, 0xf>(0.0, v);
  v += dpp_f64<0x114, 0xf>(0.0, v);
  v += dpp_f64<0x118, 0xf>(0.0, v);
  v += dpp_f64<0x142, 0xa>(0.0, v);
  v += dpp_f64<0x143, 0xc>(0.0, v);
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double wave_min_f64(double v) {   // no NaNs
  double o;
  o = dpp_f64<0x111, 0xf>(HUGE_VAL, v); v = o < v ? o : v;
  o = dpp_f64<0x112, 0xf>(HUGE_VAL, v); v = o < v ? o : v;
  o = dpp_f64<0x114, 0xf>(HUGE_VAL, v); v = o < v ? o : v;
  o = dpp_f64<0x118, 0xf>(HUGE_VAL, v); v = o < v ? o : v;
  o = dpp_f64<0x142, 0xa>(HUGE_VAL, v); v = o < v ? o : v;
  o = dpp_f64<0x143, 0xc>(HUGE_VAL, v); v = o < v ? o : v;
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double wave_max_f64(double v) { return -wave_min_f64(-v); }

template <int IPL>
__device__ __forceinline__ void wave_bitonic_sort(double (&v)[IPL]) {
  constexpr int N = 64 * IPL;
  const int lane = lane_id();
#pragma unroll
  for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride >= 1; stride >>= 1) {
      if (stride >= 64) {
        constexpr int dummy = 0;
        (void)dummy;
        const int rs = stride >> 6;  // register distance
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          if ((r & rs) == 0) {
            const int e = r * 64;  // lane bits do not matter for (e & size) when size >= 128
            const bool asc = ((e & size) == 0);
            const double a = v[r], b = v[r | rs];
            const double lo = fmin(a, b), hi = fmax(a, b);
            v[r] = asc ? lo : hi;
            v[r | rs] = asc ? hi : lo;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const int e = r * 64 + lane;
          const bool asc = ((e & size) == 0);
          const bool lower = ((lane & stride) == 0);
          const double p = shfl_xor_f64(v[r], stride);
          const double lo = fmin(v[r], p), hi = fmax(v[r], p);
          v[r] = (lower == asc) ? lo : hi;
        }
      }
    }
  }
}

// Element at sorted position p (0-based, wave-uniform) of the lane-minor layout.
template <int IPL>
__device__ __forceinline__ double wave_sorted_at(const double (&v)[IPL], int p) {
  double x = 0.0;
  const int r = p >> 6;
#pragma unroll
  for (int q = 0; q < IPL; ++q)
    if (q == r) x = v[q];
  return __shfl(x, p & 63, 64);
}

// np.median of the n smallest entries (the rest must be +inf padding): mean of the two
// middle order statistics for even n (numpy/lib/function_base.py _median), NaN for n == 0.
template <int IPL>
__device__ __forceinline__ double wave_median_sorted(const double (&v)[IPL], int n) {
  if (n <= 0) return __builtin_nan("");
  if (n & 1) return wave_sorted_at<IPL>(v, n >> 1);
  const double a = wave_sorted_at<IPL>(v, (n >> 1) - 1);
  const double b = wave_sorted_at<IPL>(v, n >> 1);
  return (a + b) / 2.0;
}

// ---------------------------------------------------------------------------------------
// Wave-wide quickselect (no sort): rank-th smallest (0-based) of the elements whose bit is
// set in `act` (bit q of lane l <-> element v[q] of lane l).  No NaNs allowed.  Control flow
// is wave-uniform; each round costs IPL compares + 2*IPL ballots, expected ~2 ln(n) rounds.
template <int IPL>
__device__ __forceinline__ double wave_quickselect(const double (&v)[IPL], unsigned int act,
                                                   int rank) {
  for (;;) {
    // pivot: first active element of the middle-most lane that still has one
    double lp = 0.0;
    bool has = false;
#pragma unroll
    for (int q = IPL - 1; q >= 0; --q)
      if ((act >> q) & 1u) { lp = v[q]; has = true; }
    const unsigned long long bal = __ballot(has);
    const unsigned long long hi = bal >> 32;
    const int src = hi ? 32 + (__ffsll((long long)hi) - 1) : 63 - __clzll((long long)bal);
    const double p = __shfl(lp, src, 64);
    int cl = 0, ce = 0;
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
      const bool a = (act >> q) & 1u;
      cl += __popcll(__ballot(a && v[q] < p));
      ce += __popcll(__ballot(a && v[q] == p));
    }
    if (rank < cl) {
#pragma unroll
      for (int q = 0; q < IPL; ++q)
        if (!(v[q] < p)) act &= ~(1u << q);
    } else if (rank < cl + ce) {
      return p;
    } else {
      rank -= cl + ce;
#pragma unroll
      for (int q = 0; q < IPL; ++q)
        if (!(v[q] > p)) act &= ~(1u << q);
    }
  }
}

__device__ __forceinline__ double readlane_f64(double x, int src) {   // src wave-uniform
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// rank-th smallest (0-based) of the active elements by one-shot bucketing: min/max, 64 equal-width
// buckets (one counter per lane, LDS atomics), prefix scan to the bucket holding the rank, exact
// rank count inside that bucket (its members compacted one per lane).  A fixed ~250 instructions
// instead of ~12 quickselect rounds; falls back to quickselect for degenerate distributions.
// hist: int[64], slots: double[64] of wave-private LDS.
template <int IPL>
__device__ __forceinline__ double wave_select_bucket_at(const double (&v)[IPL], unsigned int act,
                                                        int rank, double lo, float scale, int *hist,
                                                        double *slots);
template <int IPL>
__device__ __forceinline__ double wave_select_bucket(const double (&v)[IPL], unsigned int act,
                                                     int rank, int *hist, double *slots) {
  double lo = HUGE_VAL, hi = -HUGE_VAL;
#pragma unroll
  for (int q = 0; q < IPL; ++q)
    if ((act >> q) & 1u) { lo = v[q] < lo ? v[q] : lo; hi = v[q] > hi ? v[q] : hi; }
  lo = wave_min_f64(lo);
  hi = wave_max_f64(hi);
  const double range = hi - lo;
  if (range == 0.0) return lo;
  if (!(range > 0.0) || !(range < HUGE_VAL)) return wave_quickselect<IPL>(v, act, rank);
  const float scale = 64.0f / (float)range;
  return wave_select_bucket_at<IPL>(v, act, rank, lo, scale, hist, slots);
}

// The same selection with the bucket grid GIVEN: bucket = clamp(int((v - lo) * scale), 0, 63).  Any
// grid is exact (the bucket is monotone in v; a grid that fits the data badly only sends the call to
// the quickselect fallback) -- a caller that already knows the set's mean and spread saves the two
// wave-wide min / max reductions.
template <int IPL>
__device__ __forceinline__ double wave_select_bucket_at(const double (&v)[IPL], unsigned int act,
                                                        int rank, double lo, float scale, int *hist,
                                                        double *slots) {
  const int lane = lane_id();
  hist[lane] = 0;
  __builtin_amdgcn_wave_barrier();
  int b[IPL];
  // The two END buckets take everything beyond the grid -- a third of the values when the grid is the
  // caller's mean +- one standard deviation: as LDS atomics that is a ten-lane conflict on one address in
  // every instruction.  They are counted by ballots (scalar popcounts) instead; only the inner buckets,
  // a few values each, go through the atomics.
  int c_lo = 0, c_hi = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    b[q] = -1;
    const bool on = (act >> q) & 1u;
    int bb = (int)((float)(v[q] - lo) * scale);       // monotone in v
    bb = bb > 63 ? 63 : (bb < 0 ? 0 : bb);
    if (on) b[q] = bb;
    c_lo += __popcll(__ballot(on && bb == 0));
    c_hi += __popcll(__ballot(on && bb == 63));
    if (on && bb > 0 && bb < 63) atomicAdd(&hist[bb], 1);
  }
  __builtin_amdgcn_wave_barrier();
  const int h = lane == 0 ? c_lo : (lane == 63 ? c_hi : hist[lane]);
  const int cum = wave_incl_scan_i(h);
  const unsigned long long gt = __ballot(cum > rank);
  if (gt == 0ull) return wave_quickselect<IPL>(v, act, rank);   // cannot happen for rank < n
  const int B = __ffsll((long long)gt) - 1;
  const int before = B > 0 ? __builtin_amdgcn_readlane(cum, B - 1) : 0;
  const int need = rank - before;
  const int cB = __builtin_amdgcn_readlane(h, B);
  unsigned int inb = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q)
    if (((act >> q) & 1u) && b[q] == B) inb |= 1u << q;
  if (cB > 64) return wave_quickselect<IPL>(v, inb, need);
  int base = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const bool m = (inb >> q) & 1u;
    const unsigned long long mm = __ballot(m);
    if (m) slots[base + __popcll(mm & ((1ull << lane) - 1ull))] = v[q];
    base += __popcll(mm);
  }
  __builtin_amdgcn_wave_barrier();
  const double w = lane < cB ? slots[lane] : HUGE_VAL;
  int rk = 0;
  for (int L = 0; L < cB; ++L) {
    const double p = readlane_f64(w, L);
    rk += ((p < w) || (p == w && L < lane)) ? 1 : 0;
  }
  const unsigned long long hit = __ballot(lane < cB && rk == need);
  if (hit == 0ull) return wave_quickselect<IPL>(v, inb, need);
  return readlane_f64(w, __ffsll((long long)hit) - 1);
}

// np.median through wave_select_bucket (same contract as wave_median_select below).
// centre / spread (optional; spread > 0): the set's mean and standard deviation if the caller has them
// -- the bucket grid is then mean +- 1 spread, everything beyond it in the two end buckets (the median
// of n values lies within ~1.25 spread / sqrt(n) of the mean: 0.08 spread at n = 250, so its bucket is
// an inner one, 1/32 spread wide, holding ~3 values instead of the ~9 of a min..max grid), and the
// min / max reductions are skipped.  Any grid is exact; a bad one only costs time.
template <int IPL>
__device__ __forceinline__ double wave_median_bucket(const double (&v)[IPL], unsigned int act,
                                                     int n, int *hist, double *slots,
                                                     double centre = 0.0, double spread = 0.0) {
  if (n <= 0) return __builtin_nan("");
  const double a = spread > 0.0 && spread < HUGE_VAL
                       ? wave_select_bucket_at<IPL>(v, act, (n - 1) >> 1, centre - spread,
                                                    32.0f / (float)spread, hist, slots)
                       : wave_select_bucket<IPL>(v, act, (n - 1) >> 1, hist, slots);
  if (n & 1) return a;
  int cle = 0;
  double mn = HUGE_VAL;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const bool on = (act >> q) & 1u;
    cle += __popcll(__ballot(on && v[q] <= a));
    if (on && v[q] > a && v[q] < mn) mn = v[q];
  }
  double bb = a;
  if (cle < (n >> 1) + 1) bb = wave_min_f64(mn);
  return (a + bb) / 2.0;
}

// np.median of the n elements flagged in `act` (n = total popcount, wave-uniform, no NaNs).
template <int IPL>
__device__ __forceinline__ double wave_median_select(const double (&v)[IPL], unsigned int act,
                                                     int n) {
  if (n <= 0) return __builtin_nan("");
  const double a = wave_quickselect<IPL>(v, act, (n - 1) >> 1);
  if (n & 1) return a;
  // next order statistic: a again if enough elements are <= a, else the smallest one above a
  int cle = 0;
  double mn = HUGE_VAL;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const bool on = (act >> q) & 1u;
    cle += __popcll(__ballot(on && v[q] <= a));
    if (on && v[q] > a && v[q] < mn) mn = v[q];
  }
  double b = a;
  if (cle < (n >> 1) + 1) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const double o = shfl_xor_f64(mn, m); mn = o < mn ? o : mn; }
    b = mn;
  }
  return (a + b) / 2.0;
}

}  // namespace wcx
