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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
  return v;
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

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
