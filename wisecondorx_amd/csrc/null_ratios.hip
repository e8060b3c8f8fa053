// Null ratios (replaces newref_tools.py:210-223).
//
//   out[r][m] = log2( x_m[row_begin+r] / median_t x_m[idx[r][t]] ),  x_m = sample sid[m]
//
// The index row is applied to the FULL bin vector exactly like the reference does
// (newref_tools.py:219-221); index -1 (padding) wraps to the last bin as NumPy's negative
// indexing does.
//
// 1.8e7 medians of 300 gathered doubles each (15 kb, 100 null samples).  Selecting on doubles costs
// three instructions per compare and 80 VGPRs per wave; instead every null sample is ranked ONCE
// (own sample sort, one segment per null sample: k_rank_*), and the medians are selected on the
// 32-bit RANKS -- exact, because rank order is value order and equal values give equal medians
// whichever of them is picked:
//   k_rank_splitters / _bucket / _scan / _scatter / _sort / _pieces   ranks (pieces Rg) and the values by rank V
//   k_null_ratios   one wave per (row, 8 samples): the 8 samples' ranks of a bin share one 32-byte
//                   piece (Rg[group][bin][8]); per sample: min/max, 64 buckets (LDS atomics), scan
//                   to the bucket holding the median rank, exact ranks inside it; the median value
//                   is V[rank] (mean of the two middle ones).
// Roofline: gather bound in principle (k * 32 B per row and sample group through L2 = 22 GB at
// 15 kb); the selection arithmetic still dominates -- see DESIGN.md.
#include <algorithm>

#include "wave_sort.h"
#include "wcx_common.h"


namespace {

constexpr int NT = 256;        // 4 waves per workgroup
constexpr int BIN_BITS = 25;   // bins per call (historic payload width; keeps the 2^31 element bound company)

// order-preserving 64-bit image of a double; NaN sorts last, -0 == +0
__device__ __forceinline__ unsigned long long dkey(double x) {
  if (x != x) return ~0ull;
  x = x + 0.0;
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// ---- Ranking of the null samples: every sample's B values sorted (own kernels; replaces a library
// radix sort).  n_ids independent segments of B keys -> a SAMPLE SORT per segment:
//   k_rank_splitters  one workgroup per sample: 4096 evenly spaced elements sorted in LDS (bitonic),
//                     every fourth = one of 1023 splitters
//   k_rank_bucket     every element: its bucket by binary search among the splitters (LDS), counted
//   k_rank_scan       exclusive scan of the 1024 bucket sizes of every sample
//   k_rank_scatter    elements -> their bucket's range (one device-scope reservation per (tile, bucket))
//   k_rank_sort       one wave per bucket (~178 elements; <= 1024 from LDS, anything bigger from memory):
//                     rank = bucket start + number of smaller keys in the bucket; writes the rank pieces
//                     Rg[group][bin][8] and the values by rank V
// The BUCKET key is the COMPOSITE (order-preserving image of the value, bin): all keys are distinct, so
// equal values cannot pile into one bucket whatever the data (constant samples, integer counts); inside
// a bucket equal values share a rank (see rank_bucket_counted).
constexpr int RK_NS = 4096;        // sampled elements per segment
constexpr int RK_NB = 1024;        // buckets per segment
constexpr int RK_WMAX = 1024;      // bucket size a wave ranks from its LDS slice (16 keys per lane)
constexpr int RK_TILE = 4096;      // bins of one sample per workgroup of the bucket / scatter passes

__device__ __forceinline__ bool rk_less(unsigned long long ka, unsigned int ba, unsigned long long kb,
                                        unsigned int bb) {
  return ka < kb || (ka == kb && ba < bb);
}

__global__ __launch_bounds__(1024) void k_rank_splitters(const double *__restrict__ Xs, int64_t B,
                                                         const int32_t *__restrict__ sids,
                                                         unsigned long long *__restrict__ spk,
                                                         unsigned int *__restrict__ spb) {
  __shared__ unsigned long long sk[RK_NS];
  __shared__ unsigned int sb[RK_NS];
  const int m = blockIdx.x;
  const double *x = Xs + (int64_t)sids[m] * B;
  for (int j = threadIdx.x; j < RK_NS; j += 1024) {
    const int64_t b = (int64_t)j * B / RK_NS;                    // evenly spaced (B may be < RK_NS: repeats)
    sk[j] = dkey(x[b]);
    sb[j] = (unsigned int)b;
  }
  for (int size = 2; size <= RK_NS; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < RK_NS / 2; t += 1024) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool asc = (lo & size) == 0;
        const unsigned long long kl = sk[lo], kh = sk[hi];
        const unsigned int bl = sb[lo], bh = sb[hi];
        if (rk_less(kh, bh, kl, bl) == asc) { sk[lo] = kh; sk[hi] = kl; sb[lo] = bh; sb[hi] = bl; }
      }
    }
  __syncthreads();
  for (int j = threadIdx.x; j < RK_NB - 1; j += 1024) {
    spk[(int64_t)m * RK_NB + j] = sk[(RK_NS / RK_NB) * (j + 1) - 1];
    spb[(int64_t)m * RK_NB + j] = sb[(RK_NS / RK_NB) * (j + 1) - 1];
  }
}

// inverse of dkey() (-0 comes back as +0, the NaN key as a NaN)
__device__ __forceinline__ double dkey_inv(unsigned long long key) {
  const unsigned long long u = (key >> 63) ? (key ^ 0x8000000000000000ull) : ~key;
  return __longlong_as_double((long long)u);
}

// bucket of an element = number of splitters strictly below it (composite order).  One workgroup per
// RK_TILE consecutive bins of one sample: the splitters are read once per 4096 elements and the counts
// reach the sample's histogram as one atomic per (tile, bucket) -- ~4 elements each.
__global__ __launch_bounds__(NT) void k_rank_bucket(const double *__restrict__ Xs, int64_t B,
                                                    const int32_t *__restrict__ sids,
                                                    const unsigned long long *__restrict__ spk,
                                                    const unsigned int *__restrict__ spb,
                                                    unsigned short *__restrict__ bkt,
                                                    int *__restrict__ cnt, int *__restrict__ n_nan) {
  __shared__ unsigned long long sk[RK_NB];
  __shared__ unsigned int sb[RK_NB];
  __shared__ int hist[RK_NB];
  const int m = blockIdx.x;                            // samples fastest: all of them in flight at any time
  for (int j = threadIdx.x; j < RK_NB; j += NT) {
    sk[j] = j < RK_NB - 1 ? spk[(int64_t)m * RK_NB + j] : ~0ull;
    sb[j] = j < RK_NB - 1 ? spb[(int64_t)m * RK_NB + j] : ~0u;
    hist[j] = 0;
  }
  __syncthreads();
  const double *x = Xs + (int64_t)sids[m] * B;
  int nan_here = 0;
#pragma unroll 4
  for (int i = 0; i < RK_TILE / NT; ++i) {
    const int64_t b = (int64_t)blockIdx.y * RK_TILE + i * NT + threadIdx.x;
    if (b < B) {
      const double xv = x[b];
      const unsigned long long key = dkey(xv);
      int lo = 0, hi = RK_NB - 1;                       // first splitter that is not below the element
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (rk_less(sk[mid], sb[mid], key, (unsigned int)b)) lo = mid + 1; else hi = mid;
      }
      bkt[(int64_t)m * B + b] = (unsigned short)lo;
      atomicAdd(&hist[lo], 1);
      nan_here += xv != xv ? 1 : 0;
    }
  }
  if (nan_here) atomicAdd(&n_nan[m], nan_here);
  __syncthreads();
  for (int j = threadIdx.x; j < RK_NB; j += NT)
    if (hist[j]) atomicAdd(&cnt[(int64_t)m * RK_NB + j], hist[j]);
}

__global__ __launch_bounds__(RK_NB) void k_rank_scan(const int *__restrict__ cnt, int *__restrict__ start,
                                                     int *__restrict__ cursor) {
  __shared__ int part[RK_NB];
  const int m = blockIdx.x, t = threadIdx.x;
  const int c = cnt[(int64_t)m * RK_NB + t];
  part[t] = c;
  __syncthreads();
  for (int off = 1; off < RK_NB; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  start[(int64_t)m * RK_NB + t] = part[t] - c;
  cursor[(int64_t)m * RK_NB + t] = part[t] - c;
}

// Elements -> their bucket's range.  Per tile: positions inside the tile's share of a bucket from an
// LDS counter, the share itself reserved by ONE returning device atomic per (tile, bucket).
__global__ __launch_bounds__(NT) void k_rank_scatter(const double *__restrict__ Xs, int64_t B,
                                                     const int32_t *__restrict__ sids,
                                                     const unsigned short *__restrict__ bkt,
                                                     int *__restrict__ cursor,
                                                     unsigned long long *__restrict__ tk,
                                                     unsigned int *__restrict__ tb) {
  __shared__ int hist[RK_NB];
  const int m = blockIdx.x;
  for (int j = threadIdx.x; j < RK_NB; j += NT) hist[j] = 0;
  __syncthreads();
  const double *x = Xs + (int64_t)sids[m] * B;
  unsigned short bk[RK_TILE / NT], off[RK_TILE / NT];
#pragma unroll
  for (int i = 0; i < RK_TILE / NT; ++i) {
    const int64_t b = (int64_t)blockIdx.y * RK_TILE + i * NT + threadIdx.x;
    bk[i] = 0;
    off[i] = 0;
    if (b < B) {
      bk[i] = bkt[(int64_t)m * B + b];
      off[i] = (unsigned short)atomicAdd(&hist[bk[i]], 1);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < RK_NB; j += NT) {
    const int h = hist[j];
    if (h) hist[j] = atomicAdd(&cursor[(int64_t)m * RK_NB + j], h);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RK_TILE / NT; ++i) {
    const int64_t b = (int64_t)blockIdx.y * RK_TILE + i * NT + threadIdx.x;
    if (b < B) {
      const int pos = hist[bk[i]] + off[i];
      tk[(int64_t)m * B + pos] = dkey(x[b]);
      tb[(int64_t)m * B + pos] = (unsigned int)b;
    }
  }
}

// Ranks inside one bucket BY COUNTING: rank = start of the bucket + the number of its keys below the
// element's (the bucket's keys broadcast from wave-private LDS, two per read; a loop of a few
// instructions instead of the ~2 500 straight-line ones of a register bitonic sort, which ran out of
// the instruction cache: 3.4 ms for the 102 400 buckets of 100 samples at 15 kb).  Equal VALUES get
// equal ranks -- the selection of k_null_ratios handles repeated ranks (an index row may repeat a bin
// anyway), and V[rank] is that value whichever of them wrote it.  V = the inverse image of the key:
// the value itself but for -0 -> +0, which the median's "+ 0.0" does to it anyway.
template <int IPL>
__device__ __forceinline__ void rank_bucket_counted(const unsigned long long *__restrict__ tk,
                                                    const unsigned int *__restrict__ tb, int n, int64_t base,
                                                    int m, int64_t B, unsigned long long *lk,
                                                    unsigned int *__restrict__ R, double *__restrict__ V) {
  const int lane = wcx::lane_id();
  unsigned long long kk[IPL];
  unsigned int bb[IPL];
  int rk[IPL];
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int e = q * 64 + lane;
    kk[q] = e < n ? tk[e] : ~0ull;                    // (no key is below the padding)
    bb[q] = e < n ? tb[e] : 0u;
    lk[e] = kk[q];
    rk[q] = 0;
  }
  __builtin_amdgcn_wave_barrier();
  const int n2 = (n + 1) & ~1;
#pragma unroll 4
  for (int j = 0; j < n2; j += 2) {
    const ulonglong2 p = *reinterpret_cast<const ulonglong2 *>(lk + j);
#pragma unroll
    for (int q = 0; q < IPL; ++q) rk[q] += (p.x < kk[q] ? 1 : 0) + (p.y < kk[q] ? 1 : 0);
  }
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int e = q * 64 + lane;
    if (e < n) {
      const int64_t r = base + rk[q];
      R[(int64_t)m * B + bb[q]] = (unsigned int)r;
      V[(int64_t)m * B + r] = dkey_inv(kk[q]);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// Workgroups go to the 8 XCDs round robin by their linear id; every XCD has its own L2.  All work of
// ONE sample is given to ONE XCD (linear id l: XCD l % 8 takes the samples = l % 8 mod 8, one after the
// other), so that the scattered 4-byte rank stores of a sample -- 730 KB at 15 kb, every 64-byte line of
// it hit 16 times in random order -- meet in that one L2 and leave it as full lines.  (Ranks straight
// into the pieces Rg[group][bin][8] were 18 M partial-line writes to memory: 3.4 ms whatever the
// arithmetic.)  Only the speed depends on the dispatch order.
__device__ __forceinline__ bool xcd_sample_slot(int per_sample, int n_ids, int &m, int &slot) {
  const int l = blockIdx.x, s = l >> 3;
  m = ((s / per_sample) << 3) + (l & 7);
  slot = s % per_sample;
  return m < n_ids;
}

__global__ __launch_bounds__(NT) void k_rank_sort(int64_t B, int n_ids,
                                                  const int *__restrict__ start, const int *__restrict__ cnt,
                                                  const unsigned long long *__restrict__ tk,
                                                  const unsigned int *__restrict__ tb,
                                                  unsigned int *__restrict__ R, double *__restrict__ V) {
  __shared__ __attribute__((aligned(16))) unsigned long long s_keys[NT / 64][RK_WMAX];
  const int lane = wcx::lane_id();
  int m, slot;
  if (!xcd_sample_slot(RK_NB / (NT / 64), n_ids, m, slot)) return;
  const int wave = threadIdx.x >> 6;
  const int bk = slot * (NT / 64) + wave;
  const int n = cnt[(int64_t)m * RK_NB + bk];
  if (n == 0) return;
  const int64_t base = start[(int64_t)m * RK_NB + bk];
  const unsigned long long *k0 = tk + (int64_t)m * B + base;
  const unsigned int *b0 = tb + (int64_t)m * B + base;
  unsigned long long *lk = s_keys[wave];
  if (n <= 64) rank_bucket_counted<1>(k0, b0, n, base, m, B, lk, R, V);
  else if (n <= 128) rank_bucket_counted<2>(k0, b0, n, base, m, B, lk, R, V);
  else if (n <= 192) rank_bucket_counted<3>(k0, b0, n, base, m, B, lk, R, V);
  else if (n <= 256) rank_bucket_counted<4>(k0, b0, n, base, m, B, lk, R, V);
  else if (n <= 384) rank_bucket_counted<6>(k0, b0, n, base, m, B, lk, R, V);
  else if (n <= 512) rank_bucket_counted<8>(k0, b0, n, base, m, B, lk, R, V);
  else if (n <= RK_WMAX) rank_bucket_counted<16>(k0, b0, n, base, m, B, lk, R, V);
  else {
    // a bucket the sample did not predict (never seen; P ~ 1e-7 per bucket): the same count from memory
    for (int e = lane; e < n; e += 64) {
      const unsigned long long ke = k0[e];
      int r = 0;
      for (int j = 0; j < n; ++j) r += k0[j] < ke ? 1 : 0;
      R[(int64_t)m * B + b0[e]] = (unsigned int)(base + r);
      V[(int64_t)m * B + base + r] = dkey_inv(ke);
    }
  }
}

// R[sample][bin] -> the pieces Rg[group][bin][8] the selection gathers (samples beyond n_ids: 0)
__global__ __launch_bounds__(NT) void k_rank_pieces(const unsigned int *__restrict__ R, int64_t B, int n_ids,
                                                    unsigned int *__restrict__ Rg) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int g = blockIdx.y;
  if (b >= B) return;
  unsigned int r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = g * 8 + j < n_ids ? R[(int64_t)(g * 8 + j) * B + b] : 0u;
  uint4 *dst = reinterpret_cast<uint4 *>(Rg + ((int64_t)g * B + b) * 8);
  dst[0] = make_uint4(r[0], r[1], r[2], r[3]);
  dst[1] = make_uint4(r[4], r[5], r[6], r[7]);
}

__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v) {
  using wcx::dpp_i32;
  unsigned int o;
  o = (unsigned int)dpp_i32<0x111, 0xf>(-1, (int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<0x112, 0xf>(-1, (int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<0x114, 0xf>(-1, (int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<0x118, 0xf>(-1, (int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<0x142, 0xa>(-1, (int)v); v = o < v ? o : v;
  o = (unsigned int)dpp_i32<0x143, 0xc>(-1, (int)v); v = o < v ? o : v;
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
  using wcx::dpp_i32;
  unsigned int o;
  o = (unsigned int)dpp_i32<0x111, 0xf>(0, (int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<0x112, 0xf>(0, (int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<0x114, 0xf>(0, (int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<0x118, 0xf>(0, (int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<0x142, 0xa>(0, (int)v); v = o > v ? o : v;
  o = (unsigned int)dpp_i32<0x143, 0xc>(0, (int)v); v = o > v ? o : v;
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

// rank-th smallest (0-based, rank < n) of the active 32-bit values by bitwise bisection: exact for
// any input (duplicates included).  Rare path (a bucket with more than 64 members).
template <int IPL>
__device__ __forceinline__ unsigned int select_u32_bisect(const unsigned int (&v)[IPL],
                                                                    unsigned int act, int rank) {
  unsigned int prefix = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned int trial = prefix | (1u << bit);
    int c = 0;
#pragma unroll
    for (int q = 0; q < IPL; ++q) c += __popcll(__ballot(((act >> q) & 1u) && v[q] < trial));
    if (c <= rank) prefix = trial;
  }
  return prefix;
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// The two middle order statistics (ranks r0 = (n-1)/2 and r1 = n/2) of the active values: entry
// q * 64 + lane is active iff it is < n (one compare against a scalar; `act` is that as a per-lane
// bit mask, for the general path).
// hist: int[64], slots: unsigned[64], wave-private LDS.  hi_out = the maximum (NaN detection).
// Slices of 64 list entries that are full whatever the refsize, given the launch rule IPL = the
// instantiated size for ceil(k / 64) (1 .. 6, 8, 16, 32): k > 64 NFULL<IPL>.  Their "is this entry
// active?" test is a compile-time true -- at k = 300 four of five slices lose their compare + select +
// mask arithmetic in every loop below (the predicates were a third of the kernel's vector instructions).
template <int IPL>
struct NFull { static constexpr int value = IPL <= 6 ? IPL - 1 : (IPL == 8 ? 6 : IPL / 2); };
// The launch rule (WCX_NR_LAUNCH / WCX_NRH_LAUNCH below): ceil(k / 64) in (P, IPL] runs instantiation IPL, with
// P the next smaller instantiated size -- so k >= 64 P + 1, and NFull<IPL> <= P is what the kernels rely on.
static_assert(NFull<1>::value <= 0 && NFull<2>::value <= 1 && NFull<3>::value <= 2 && NFull<4>::value <= 3 &&
              NFull<5>::value <= 4 && NFull<6>::value <= 5 && NFull<8>::value <= 6 && NFull<16>::value <= 8 &&
              NFull<32>::value <= 16, "NFull must stay below the smallest refsize an instantiation is launched for");

template <int IPL>
__device__ __forceinline__ void wave_middle_u32(const unsigned int (&v)[IPL], unsigned int act,
                                                int n, int *hist, unsigned int *slots,
                                                unsigned int &a0, unsigned int &a1,
                                                unsigned int &hi_out) {
  const int lane = wcx::lane_id();
  constexpr int NF = NFull<IPL>::value;                    // (callers: n > 64 NF)
  const int r0 = (n - 1) >> 1, r1 = n >> 1;
  unsigned int lo = 0xffffffffu, mx = 0u;
#pragma unroll
  for (int q = 0; q < IPL; ++q)
    if (q < NF || q * 64 + lane < n) { lo = v[q] < lo ? v[q] : lo; mx = v[q] > mx ? v[q] : mx; }
  lo = wave_min_u32(lo);
  const unsigned int hi = wave_max_u32(mx);
  hi_out = hi;
  if (hi == lo) { a0 = lo; a1 = lo; return; }
  // (any monotone map into 0 .. 63 will do -- the order statistics below are exact whatever the buckets --,
  //  so the hardware reciprocal, 1 ulp, replaces an IEEE division: 2 instead of 11 vector instructions)
  const float scale = 64.0f * __builtin_amdgcn_rcpf((float)(hi - lo) * 1.0000002f + 1.0f);   // bucket(hi) clamped to 63
  hist[lane] = 0;
  __builtin_amdgcn_wave_barrier();
  int b[IPL];                                               // bucket of the entry; -1: not active
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    b[q] = -1;
    if (q < NF || q * 64 + lane < n) {
      int bb = (int)((float)(v[q] - lo) * scale);          // monotone in v
      bb = bb > 63 ? 63 : bb;
      b[q] = bb;
      atomicAdd(&hist[bb], 1);
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int h = hist[lane];
  const int cum = wcx::wave_incl_scan_i(h);
  const unsigned long long gt = __ballot(cum > r0);
  const int B0 = __ffsll((long long)gt) - 1;                 // gt != 0: cum[63] = n > r0
  const int before = B0 > 0 ? __builtin_amdgcn_readlane(cum, B0 - 1) : 0;
  const int need = r0 - before;
  const int cB = __builtin_amdgcn_readlane(h, B0);
  if (cB > 64) {                                             // heavy duplicates: general path
    a0 = select_u32_bisect<IPL>(v, act, r0);
    a1 = r1 == r0 ? a0 : select_u32_bisect<IPL>(v, act, r1);
    return;
  }
  int base = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const bool m = b[q] == B0;
    const unsigned long long mm = __ballot(m);
    if (m) slots[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mm, 0u))] = v[q];
    base += __popcll(mm);
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned int w = lane < cB ? slots[lane] : 0xffffffffu;
  int rk = 0;
  for (int L = 0; L < cB; ++L) {
    const unsigned int p = (unsigned int)__builtin_amdgcn_readlane((int)w, L);
    rk += ((p < w) || (p == w && L < lane)) ? 1 : 0;
  }
  const unsigned long long hit = __ballot(lane < cB && rk == need);
  a0 = (unsigned int)__builtin_amdgcn_readlane((int)w, __ffsll((long long)hit) - 1);
  a1 = a0;
  if (r1 != r0) {
    if (need + 1 < cB) {
      const unsigned long long hit1 = __ballot(lane < cB && rk == need + 1);
      a1 = (unsigned int)__builtin_amdgcn_readlane((int)w, __ffsll((long long)hit1) - 1);
    } else {                         // next order statistic = smallest value of the later buckets
      unsigned int mn = 0xffffffffu;
#pragma unroll
      for (int q = 0; q < IPL; ++q)
        if (b[q] > B0 && v[q] < mn) mn = v[q];
      a1 = wave_min_u32(mn);
    }
  }
}

// The two middle order statistics of an ARBITRARY active set (bit q of `act` <-> element v[q] of this
// lane; n = its size, wave-uniform, >= 1): wave_middle_u32 for sets that are not a dense prefix -- the
// reference bins of a normalisation pass that are selected (distance below the cut-off) and not masked.
template <int IPL>
__device__ __forceinline__ void wave_middle_u32_act(const unsigned int (&v)[IPL], unsigned int act, int n,
                                                    int *hist, unsigned int *slots, unsigned int &a0,
                                                    unsigned int &a1) {
  const int lane = wcx::lane_id();
  const int r0 = (n - 1) >> 1, r1 = n >> 1;
  unsigned int lo = 0xffffffffu, mx = 0u;
#pragma unroll
  for (int q = 0; q < IPL; ++q)
    if ((act >> q) & 1u) { lo = v[q] < lo ? v[q] : lo; mx = v[q] > mx ? v[q] : mx; }
  lo = wave_min_u32(lo);
  const unsigned int hi = wave_max_u32(mx);
  if (hi == lo) { a0 = lo; a1 = lo; return; }
  // (any monotone map into 0 .. 63 will do -- the order statistics below are exact whatever the buckets --,
  //  so the hardware reciprocal, 1 ulp, replaces an IEEE division: 2 instead of 11 vector instructions)
  const float scale = 64.0f * __builtin_amdgcn_rcpf((float)(hi - lo) * 1.0000002f + 1.0f);   // bucket(hi) clamped to 63
  hist[lane] = 0;
  __builtin_amdgcn_wave_barrier();
  int b[IPL];
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    b[q] = -1;
    if ((act >> q) & 1u) {
      int bb = (int)((float)(v[q] - lo) * scale);          // monotone in v
      bb = bb > 63 ? 63 : bb;
      b[q] = bb;
      atomicAdd(&hist[bb], 1);
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int h = hist[lane];
  const int cum = wcx::wave_incl_scan_i(h);
  const unsigned long long gt = __ballot(cum > r0);
  const int B0 = __ffsll((long long)gt) - 1;                 // gt != 0: cum[63] = n > r0
  const int before = B0 > 0 ? __builtin_amdgcn_readlane(cum, B0 - 1) : 0;
  const int need = r0 - before;
  const int cB = __builtin_amdgcn_readlane(h, B0);
  if (cB > 64) {                                             // heavy duplicates: general path
    a0 = select_u32_bisect<IPL>(v, act, r0);
    a1 = r1 == r0 ? a0 : select_u32_bisect<IPL>(v, act, r1);
    return;
  }
  int base = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const bool m = b[q] == B0;
    const unsigned long long mm = __ballot(m);
    if (m) slots[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mm, 0u))] = v[q];
    base += __popcll(mm);
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned int w = lane < cB ? slots[lane] : 0xffffffffu;
  int rk = 0;
  for (int L = 0; L < cB; ++L) {
    const unsigned int p = (unsigned int)__builtin_amdgcn_readlane((int)w, L);
    rk += ((p < w) || (p == w && L < lane)) ? 1 : 0;
  }
  const unsigned long long hit = __ballot(lane < cB && rk == need);
  a0 = (unsigned int)__builtin_amdgcn_readlane((int)w, __ffsll((long long)hit) - 1);
  a1 = a0;
  if (r1 != r0) {
    if (need + 1 < cB) {
      const unsigned long long hit1 = __ballot(lane < cB && rk == need + 1);
      a1 = (unsigned int)__builtin_amdgcn_readlane((int)w, __ffsll((long long)hit1) - 1);
    } else {                         // next order statistic = smallest value of the later buckets
      unsigned int mn = 0xffffffffu;
#pragma unroll
      for (int q = 0; q < IPL; ++q)
        if (b[q] > B0 && v[q] < mn) mn = v[q];
      a1 = wave_min_u32(mn);
    }
  }
}

// ---- The medians of the LAST normalisation pass of a batch (predict_tools.py:137: the ratio's divisor)
// on RANKS.  A batch's samples are ranked once (k_rank_*: the same pipeline, the batch's [n][B] matrix of
// projected coverages as "the null samples"); k_norm_rank_mark sets bit 31 of the rank of every (bin,
// sample) the first two passes masked (or that fails the >= 0 test anyway: predict_tools.py:134); then one
// wave per (bin, 8 samples) gathers the 32-byte rank pieces of the bin's selected reference bins -- the
// gathers k_null_ratios does -- and takes the middle of the unmasked ones: exact (rank order is value
// order), 4 bytes and one-instruction compares per element where the tiled kernel selects on doubles.
constexpr unsigned int RANK_MASKED = 0x80000000u;
__global__ __launch_bounds__(NT) void k_norm_rank_mark(unsigned int *__restrict__ Rg,
                                                       const double *__restrict__ copyT, int64_t B, int NS,
                                                       int n_samples) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int g = blockIdx.y;
  if (b >= B) return;
  uint4 *p = reinterpret_cast<uint4 *>(Rg + ((int64_t)g * B + b) * 8);
  uint4 r0 = p[0], r1 = p[1];
  unsigned int r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  bool any = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int s = g * 8 + j;
    if (s < n_samples && !(copyT[b * NS + s] >= 0.0)) { r[j] |= RANK_MASKED; any = true; }
  }
  if (any) {
    p[0] = make_uint4(r[0], r[1], r[2], r[3]);
    p[1] = make_uint4(r[4], r[5], r[6], r[7]);
  }
}

template <int IPL>
__global__ __launch_bounds__(NT) void k_norm_median_rank(
    const unsigned int *__restrict__ Rg, const double *__restrict__ V, const int32_t *__restrict__ idx,
    const unsigned long long *__restrict__ sel, const double *__restrict__ xT, int64_t B, int k, int NS,
    int n_samples, int64_t ct, WcxChrCum chr, double *__restrict__ rT, double *__restrict__ lrT) {
  const int lane = wcx::lane_id();
  const int wave = threadIdx.x >> 6;
  __shared__ int s_hist[NT / 64][64];
  __shared__ unsigned int s_slots[NT / 64][64];
  const int64_t i = ct + (int64_t)blockIdx.x * (NT / 64) + wave;
  if (i >= B) return;
  const int sg = blockIdx.y;
  const uint4 *slab = reinterpret_cast<const uint4 *>(Rg + (int64_t)sg * B * 8);
  int64_t cs = 0, ce = chr.cum[0];
  for (int c = 1; c < chr.n_chr && i >= ce; ++c) { cs = ce; ce = chr.cum[c]; }
  const int64_t own = ce - cs;
  const int64_t len_cd = B - own;                 // len(chr_data), predict_tools.py:125-130
  unsigned int v[8][IPL];
  unsigned int selmask = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int t = q * 64 + lane;
    const bool selq = (t < k) && ((sel[i * IPL + q] >> lane) & 1ull);
    uint4 a0 = make_uint4(RANK_MASKED, RANK_MASKED, RANK_MASKED, RANK_MASKED), a1 = a0;
    if (selq) {
      int64_t c = idx[i * (int64_t)k + t];
      if (c < 0) c += len_cd;                      // NumPy negative index
      const int64_t g = c < cs ? c : c + own;      // chr_data index -> row
      a0 = slab[g * 2];
      a1 = slab[g * 2 + 1];
      selmask |= 1u << q;
    }
    v[0][q] = a0.x; v[1][q] = a0.y; v[2][q] = a0.z; v[3][q] = a0.w;
    v[4][q] = a1.x; v[5][q] = a1.y; v[6][q] = a1.z; v[7][q] = a1.w;
  }
  unsigned int my_a0 = 0, my_a1 = 0;
  int my_n = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (sg * 8 + s >= n_samples) break;            // (wave-uniform: padding samples of the last group)
    unsigned int act = 0;
    int n = 0;
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
      const bool on = !(v[s][q] & RANK_MASKED);    // selected (unselected slots carry the bit) and kept
      act |= on ? (1u << q) : 0u;
      n += __popcll(__ballot(on));
    }
    unsigned int a0 = 0, a1 = 0;
    if (n > 0) wave_middle_u32_act<IPL>(v[s], act, n, s_hist[wave], s_slots[wave], a0, a1);
    if (lane == s) { my_a0 = a0; my_a1 = a1; my_n = n; }
  }
  const int m = sg * 8 + lane;
  if (lane < 8 && m < n_samples) {
    const double *Vm = V + (int64_t)m * B;
    double med = __builtin_nan("");                // np.median of nothing
    if (my_n > 0) {
      const double a = Vm[my_a0];
      med = (my_n & 1) ? a : (a + Vm[my_a1]) / 2.0;
    }
    const double r = xT[i * NS + m] / med;         // predict_tools.py:137
    rT[i * NS + m] = r;
    lrT[i * NS + m] = log2(r);
  }
}

__global__ void k_iota_i32(int32_t *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// ---- FEW target rows: selection on the HIGH HALVES of the values' keys (no ranking at all) -------
// Ranking every bin of every null sample (a sample sort of n_ids * B elements: 1.3 ms alone on the device,
// 3-4 ms beside the refine it shares the chip with -- when round 2 measured this it was a library sort: 6.7 ms of device
// work at 15 kb) is worth it when ~all rows are targets and the sort hides beside the refine; for
// the chrX / chrY rows of a gonosomal pass or a rank's shard of an 8-GPU build it is not.
// hi32(dkey(x)) is monotone in x (not strictly): the bucket selection on it finds the 32-bit keys a0,
// a1 of the two middle order statistics; which ELEMENT carries them is then settled exactly on the
// full doubles of the (almost always single) candidates that share that high half.  Needs only the
// key pieces K[group][bin][8] -- one streaming pass over the null samples' rows.  Same bits as the
// rank path.  54 ns per (row, 100 samples) against 35 ns + the ranking.
__global__ __launch_bounds__(NT) void k_nr_hikeys(const double *__restrict__ Xs, int64_t B,
                                                  const int32_t *__restrict__ sids, int n_ids,
                                                  unsigned int *__restrict__ Kg) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int g = blockIdx.y;
  if (b >= B) return;
  unsigned int kk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int m = g * 8 + j;
    kk[j] = m < n_ids ? (unsigned int)(dkey(Xs[(int64_t)sids[m] * B + b]) >> 32) : 0u;
  }
  uint4 *dst = reinterpret_cast<uint4 *>(Kg + ((int64_t)g * B + b) * 8);
  dst[0] = make_uint4(kk[0], kk[1], kk[2], kk[3]);
  dst[1] = make_uint4(kk[4], kk[5], kk[6], kk[7]);
}

// The j-th smallest (0-based) FULL value among the active elements whose high key equals `key`
// (bins in cc[], sample row x).  cand: int[64] wave-private LDS.
template <int IPL>
__device__ __forceinline__ double pick_among_ties(const unsigned int (&v)[IPL], const int (&cc)[IPL],
                                                  unsigned int act, unsigned int key, int j,
                                                  const double *__restrict__ x, int *cand) {
  const int lane = wcx::lane_id();
  int t = 0;
  unsigned int tied = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const bool m = ((act >> q) & 1u) && v[q] == key;
    const unsigned long long mm = __ballot(m);
    const int pos = t + __popcll(mm & ((1ull << lane) - 1ull));
    if (m && pos < 64) cand[pos] = cc[q];
    tied |= m ? (1u << q) : 0u;
    t += __popcll(mm);
  }
  __builtin_amdgcn_wave_barrier();
  double res;
  if (t == 1) {
    res = x[cand[0]];                                   // the usual case: one broadcast load
  } else if (t <= 64) {
    const double xv = lane < t ? x[cand[lane]] : 0.0;
    const unsigned long long w = lane < t ? dkey(xv) : ~0ull;
    int rk = 0;
    for (int L = 0; L < t; ++L) {
      const unsigned long long p = readlane_u64(w, L);
      rk += ((p < w) || (p == w && L < lane)) ? 1 : 0;
    }
    const unsigned long long hit = __ballot(lane < t && rk == j);
    res = dkey_inv(readlane_u64(w, __ffsll((long long)hit) - 1));
  } else {
    // more than 64 elements share the high half (an index row that repeats a bin): select on the
    // low halves of the tied elements
    unsigned int lo[IPL];
#pragma unroll
    for (int q = 0; q < IPL; ++q) lo[q] = ((tied >> q) & 1u) ? (unsigned int)dkey(x[cc[q]]) : 0u;
    const unsigned int sel = select_u32_bisect<IPL>(lo, tied, j);
    res = dkey_inv(((unsigned long long)key << 32) | sel);
  }
  __builtin_amdgcn_wave_barrier();
  return res;
}

template <int IPL>
__global__ __launch_bounds__(NT) void k_null_ratios_hi(
    const unsigned int *__restrict__ Kg, const double *__restrict__ Xs,
    const int32_t *__restrict__ sids, int64_t B, const int32_t *__restrict__ idx,
    int64_t row_begin, int64_t n_rows, int k, int n_ids, double *__restrict__ out) {
  const int lane = wcx::lane_id();
  const int wave = threadIdx.x >> 6;
  __shared__ int s_hist[NT / 64][64];
  __shared__ unsigned int s_slots[NT / 64][64];
  __shared__ int s_cand[NT / 64][64];
  const int64_t r = (int64_t)blockIdx.x * (NT / 64) + wave;
  if (r >= n_rows) return;
  const int sg = blockIdx.y;
  const uint4 *slab = reinterpret_cast<const uint4 *>(Kg + (int64_t)sg * B * 8);
  unsigned int v[8][IPL];
  int cc[IPL];
  unsigned int act = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int t = q * 64 + lane;
    const bool valid = q < NFull<IPL>::value || t < k;      // (k > 64 NFull: launch rule)
    int64_t c = valid ? (int64_t)idx[r * (int64_t)k + t] : 0;
    if (c < 0) c += B;  // NumPy negative index
    cc[q] = (int)c;
    act |= valid ? (1u << q) : 0u;
    const uint4 a0 = slab[c * 2], a1 = slab[c * 2 + 1];
    v[0][q] = a0.x; v[1][q] = a0.y; v[2][q] = a0.z; v[3][q] = a0.w;
    v[4][q] = a1.x; v[5][q] = a1.y; v[6][q] = a1.z; v[7][q] = a1.w;
  }
  const int r0 = (k - 1) >> 1, r1 = k >> 1;
  // Per sample: the high keys a0 <= a1 of the two middle order statistics.  Nearly always each is
  // carried by ONE element: its bin goes to lane s, and the eight lanes fetch their two doubles
  // together at the end (no dependent load inside the loop).  Shared high halves take the exact
  // path at once.
  int my_c0 = 0, my_c1 = 0;
  double my_v0 = 0.0, my_v1 = 0.0;
  bool my_slow = false, my_nan = false;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int m = sg * 8 + s;
    if (m >= n_ids) break;                                   // wave-uniform
    unsigned int a0, a1, hi;
    wave_middle_u32<IPL>(v[s], act, k, s_hist[wave], s_slots[wave], a0, a1, hi);
    int below = 0, t0 = 0, t1 = 0, c0 = 0, c1 = 0;
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
      const bool on = q < NFull<IPL>::value || q * 64 + lane < k;
      below += __popcll(__ballot(on && v[s][q] < a0));
      const unsigned long long e0 = __ballot(on && v[s][q] == a0);
      if (e0) {
        t0 += __popcll(e0);
        c0 = __builtin_amdgcn_readlane(cc[q], __builtin_amdgcn_readfirstlane(__ffsll((long long)e0) - 1));
      }
      if (a1 != a0) {
        const unsigned long long e1 = __ballot(on && v[s][q] == a1);
        if (e1) {
          t1 += __popcll(e1);
          c1 = __builtin_amdgcn_readlane(cc[q], __builtin_amdgcn_readfirstlane(__ffsll((long long)e1) - 1));
        }
      }
    }
    const bool fast = t0 == 1 && (r1 == r0 || (a1 != a0 && t1 == 1));
    if (fast) {
      if (lane == s) { my_c0 = c0; my_c1 = r1 == r0 ? c0 : c1; }
    } else {
      const double *x = Xs + (int64_t)sids[m] * B;
      const double v0 = pick_among_ties<IPL>(v[s], cc, act, a0, r0 - below, x, s_cand[wave]);
      double v1 = v0;
      if (r1 != r0)
        v1 = pick_among_ties<IPL>(v[s], cc, act, a1, a1 == a0 ? r1 - below : 0, x, s_cand[wave]);
      if (lane == s) { my_v0 = v0; my_v1 = v1; my_slow = true; }
    }
    if (lane == s) my_nan = hi == 0xffffffffu;               // only NaN has that high half
  }
  const int m = sg * 8 + lane;
  if (lane < 8 && m < n_ids) {
    const double *x = Xs + (int64_t)sids[m] * B;
    if (!my_slow) { my_v0 = x[my_c0]; my_v1 = x[my_c1]; }
    double med = (my_v0 + my_v1) / 2.0 + 0.0;                // (np.median's mean: a zero median is +0)
    if (my_nan) med = __builtin_nan("");                     // np.median propagates NaN
    out[r * (int64_t)n_ids + m] = log2(x[row_begin + r] / med);
  }
}

// One wave per (row, group of 8 samples).  blockIdx.x (fastest in dispatch order) walks the rows,
// blockIdx.y the sample groups: at any moment the whole chip gathers from ONE 32*B-byte slab.
template <int IPL>
__global__ __launch_bounds__(NT) void k_null_ratios(
    const unsigned int *__restrict__ Rg, const double *__restrict__ V,
    const int *__restrict__ n_nan, const double *__restrict__ Xs,
    const int32_t *__restrict__ sids, int64_t B, const int32_t *__restrict__ idx,
    int64_t row_begin, int64_t n_rows, int k, int n_ids, double *__restrict__ out, int dbg) {
  const int lane = wcx::lane_id();
  const int wave = threadIdx.x >> 6;
  __shared__ int s_hist[NT / 64][64];
  __shared__ unsigned int s_slots[NT / 64][64];
  const int64_t r = (int64_t)blockIdx.x * (NT / 64) + wave;
  if (r >= n_rows) return;
  const int sg = blockIdx.y;
  const uint4 *slab = reinterpret_cast<const uint4 *>(Rg + (int64_t)sg * B * 8);
  unsigned int v[8][IPL];
  unsigned int act = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int t = q * 64 + lane;
    const bool valid = q < NFull<IPL>::value || t < k;      // (k > 64 NFull: launch rule)
    int64_t c = valid ? (int64_t)idx[r * (int64_t)k + t] : 0;
    if (c < 0) c += B;  // NumPy negative index
    act |= valid ? (1u << q) : 0u;
    uint4 a0, a1;
    if (dbg & 16) {            // ablation: no gathers
      a0 = make_uint4((unsigned)c, (unsigned)c * 3u, (unsigned)c ^ 0x55u, (unsigned)c + 7u);
      a1 = make_uint4((unsigned)c * 5u, (unsigned)c + 1u, (unsigned)c ^ 0x33u, (unsigned)c + 9u);
    } else { a0 = slab[c * 2]; a1 = slab[c * 2 + 1]; }
    v[0][q] = a0.x; v[1][q] = a0.y; v[2][q] = a0.z; v[3][q] = a0.w;
    v[4][q] = a1.x; v[5][q] = a1.y; v[6][q] = a1.z; v[7][q] = a1.w;
  }
  unsigned int my_a0 = 0, my_a1 = 0, my_hi = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    unsigned int a0, a1, hi;
    if (dbg & 64) {            // ablation: no selection
      a0 = 0;
#pragma unroll
      for (int q = 0; q < IPL; ++q) a0 ^= v[s][q];
      a0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)a0) % (unsigned int)B; a1 = a0; hi = 0;
    } else
    wave_middle_u32<IPL>(v[s], act, k, s_hist[wave], s_slots[wave], a0, a1, hi);
    if (lane == s) { my_a0 = a0; my_a1 = a1; my_hi = hi; }
  }
  const int m = sg * 8 + lane;
  if (lane < 8 && m < n_ids) {
    const double *Vm = V + (int64_t)m * B;
    if (dbg & 16) { my_a0 %= (unsigned int)B; my_a1 %= (unsigned int)B; }
    double med = (Vm[my_a0] + Vm[my_a1]) / 2.0 + 0.0;   // (np.median's mean: a zero median is +0)
    if ((int64_t)my_hi >= B - n_nan[m]) med = __builtin_nan("");   // np.median propagates NaN
    const double xr = Xs[(int64_t)sids[m] * B + row_begin + r];
    out[r * (int64_t)n_ids + m] = log2(xr / med);
  }
}

}  // namespace

extern "C" {

// Ranking of every null sample (k_rank_*: ranks, rank pieces, values by rank)
// on stream `st` into the buffer at `base` (layout from rank_layout()).
struct RankLayout {
  size_t o_sid, o_nan, o_cnt, o_start, o_cursor, o_spk, o_spb, o_bkt, o_tk, o_tb, o_r, o_rg, o_v, total;
  int n_sg;
};

static int rank_layout(int64_t B, int n_ids, hipStream_t, RankLayout &L) {
  const int64_t n = (int64_t)n_ids * B;
  L.n_sg = (n_ids + 7) / 8;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  L.o_sid = carve((size_t)n_ids * 4);
  L.o_nan = carve((size_t)n_ids * 4);
  L.o_cnt = carve((size_t)n_ids * RK_NB * 4);
  L.o_start = carve((size_t)n_ids * RK_NB * 4);
  L.o_cursor = carve((size_t)n_ids * RK_NB * 4);
  L.o_spk = carve((size_t)n_ids * RK_NB * 8);
  L.o_spb = carve((size_t)n_ids * RK_NB * 4);
  L.o_bkt = carve((size_t)n * 2);
  L.o_tk = carve((size_t)n * 8);
  L.o_tb = carve((size_t)n * 4);
  L.o_r = carve((size_t)n * 4);
  L.o_rg = carve((size_t)L.n_sg * B * 32);
  L.o_v = carve((size_t)n * 8);
  L.total = off;
  return WCX_OK;
}

static int rank_run(const double *dXs, int64_t B, int n_ids, const RankLayout &L, char *base,
                    hipStream_t st) {
  int32_t *d_sids = reinterpret_cast<int32_t *>(base + L.o_sid);
  int *d_nan = reinterpret_cast<int *>(base + L.o_nan);
  int *cnt = reinterpret_cast<int *>(base + L.o_cnt);
  int *start = reinterpret_cast<int *>(base + L.o_start);
  int *cursor = reinterpret_cast<int *>(base + L.o_cursor);
  unsigned long long *spk = reinterpret_cast<unsigned long long *>(base + L.o_spk);
  unsigned int *spb = reinterpret_cast<unsigned int *>(base + L.o_spb);
  unsigned short *bkt = reinterpret_cast<unsigned short *>(base + L.o_bkt);
  unsigned long long *tk = reinterpret_cast<unsigned long long *>(base + L.o_tk);
  unsigned int *tb = reinterpret_cast<unsigned int *>(base + L.o_tb);
  unsigned int *R = reinterpret_cast<unsigned int *>(base + L.o_r);
  unsigned int *Rg = reinterpret_cast<unsigned int *>(base + L.o_rg);
  double *V = reinterpret_cast<double *>(base + L.o_v);
  WCX_HIP(hipMemsetAsync(d_nan, 0, (size_t)n_ids * 4, st));
  WCX_HIP(hipMemsetAsync(cnt, 0, (size_t)n_ids * RK_NB * 4, st));
  const unsigned gb = (unsigned)((B + RK_TILE - 1) / RK_TILE);
  k_rank_splitters<<<(unsigned)n_ids, 1024, 0, st>>>(dXs, B, d_sids, spk, spb);
  k_rank_bucket<<<dim3((unsigned)n_ids, gb), NT, 0, st>>>(dXs, B, d_sids, spk, spb, bkt, cnt, d_nan);
  k_rank_scan<<<(unsigned)n_ids, RK_NB, 0, st>>>(cnt, start, cursor);
  k_rank_scatter<<<dim3((unsigned)n_ids, gb), NT, 0, st>>>(dXs, B, d_sids, bkt, cursor, tk, tb);
  k_rank_sort<<<(unsigned)(L.n_sg * 8 * (RK_NB / (NT / 64))), NT, 0, st>>>(B, n_ids, start, cnt, tk, tb, R, V);
  k_rank_pieces<<<dim3((unsigned)((B + NT - 1) / NT), (unsigned)L.n_sg), NT, 0, st>>>(R, B, n_ids, Rg);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

}  // extern "C"

size_t wcx_rank_bytes(int64_t B, int n) {
  RankLayout L;
  rank_layout(B, n, nullptr, L);
  return L.total;
}

int wcx_rank_rows_launch(const double *d_x, int64_t B, int n, char *base, hipStream_t st, WcxRankView *out) {
  WCX_ARG(n > 0 && n <= 128 && B < (1ll << BIN_BITS) && (int64_t)n * B < (1ll << 31), "too many rows / bins to rank");
  RankLayout L;
  rank_layout(B, n, st, L);
  k_iota_i32<<<1, 128, 0, st>>>(reinterpret_cast<int32_t *>(base + L.o_sid), n);
  const int rc = rank_run(d_x, B, n, L, base, st);
  if (rc) return rc;
  out->Rg = reinterpret_cast<unsigned int *>(base + L.o_rg);
  out->V = reinterpret_cast<const double *>(base + L.o_v);
  out->n_sg = L.n_sg;
  return WCX_OK;
}

int wcx_norm_rank_mark_launch(const WcxRankView &rk, const double *copyT, int64_t B, int NS, int n_samples,
                              hipStream_t st) {
  k_norm_rank_mark<<<dim3((unsigned)((B + NT - 1) / NT), (unsigned)rk.n_sg), NT, 0, st>>>(rk.Rg, copyT, B, NS,
                                                                                          n_samples);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

int wcx_norm_median_rank_launch(const WcxRankView &rk, const int32_t *d_idx, const unsigned long long *d_sel,
                                const double *xT, int64_t B, int k, int ipl, int NS, int n_samples, int64_t ct,
                                const WcxChrCum &chr, double *rT, double *lrT, hipStream_t st) {
  const int64_t n_rows = B - ct;
  if (n_rows <= 0) return WCX_OK;
  const dim3 grid((unsigned)((n_rows + NT / 64 - 1) / (NT / 64)), (unsigned)rk.n_sg);
#define WCX_NMR_LAUNCH(IPL)                                                                            \
  k_norm_median_rank<IPL><<<grid, NT, 0, st>>>(rk.Rg, rk.V, d_idx, d_sel, xT, B, k, NS, n_samples, ct, chr, rT, lrT)
  switch (ipl) {
    case 1: WCX_NMR_LAUNCH(1); break;
    case 2: WCX_NMR_LAUNCH(2); break;
    case 3: WCX_NMR_LAUNCH(3); break;
    case 4: WCX_NMR_LAUNCH(4); break;
    case 5: WCX_NMR_LAUNCH(5); break;
    case 6: WCX_NMR_LAUNCH(6); break;
    case 7: WCX_NMR_LAUNCH(7); break;
    case 8: WCX_NMR_LAUNCH(8); break;
    default:
      wcx_set_error("rank medians: %d values per lane are not instantiated", ipl);
      return WCX_ERR_UNSUPPORTED;
  }
#undef WCX_NMR_LAUNCH
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

extern "C" {

static int check_ids(const int32_t *sample_ids, int n_ids, int S) {
  for (int i = 0; i < n_ids; ++i)
    WCX_ARG(sample_ids[i] >= 0 && sample_ids[i] < S, "sample id out of range");
  return WCX_OK;
}

// Optional head start: rank the null samples on the context's AUXILIARY stream while the search
// (which needs nothing of this) runs on the main one; the next wcx_null_ratios_dev with the same
// matrix and ids picks the result up.  The ranking depends on X only, not on the row shard: in a
// multi-GPU build it is per-rank fixed cost, so it is worth hiding.
int wcx_null_rank_prepare_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                              const int32_t *sample_ids, int n_ids) {
  WCX_ARG(ctx && dXs && sample_ids, "NULL argument");
  WCX_ARG(B > 0 && S > 0 && n_ids > 0, "bad sizes");
  WCX_ARG(B < (1ll << BIN_BITS) && n_ids <= 128, "too many bins / null samples");
  WCX_ARG((int64_t)n_ids * B < (1ll << 31), "null samples x bins exceeds 2^31");
  int rc = check_ids(sample_ids, n_ids, S);
  if (rc) return rc;
  WCX_HIP(hipSetDevice(ctx->device));
  if (!ctx->aux_stream) {
    WCX_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    WCX_HIP(hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
    WCX_HIP(hipEventCreateWithFlags(&ctx->ev_rank, hipEventDisableTiming));
  }
  RankLayout L;
  rc = rank_layout(B, n_ids, ctx->aux_stream, L);
  if (rc) return rc;
  if (ctx->rank_bytes < L.total) {
    WCX_HIP(hipStreamSynchronize(ctx->aux_stream));
    WCX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_rank) { WCX_HIP(hipFree(ctx->d_rank)); ctx->d_rank = nullptr; ctx->rank_bytes = 0; }
    if (hipMalloc(&ctx->d_rank, L.total) != hipSuccess) {
      wcx_set_error("hipMalloc(%zu) for the null-sample ranks failed", L.total);
      return WCX_ERR_NOMEM;
    }
    ctx->rank_bytes = L.total;
  }
  ctx->rank_ids.assign(sample_ids, sample_ids + n_ids);
  ctx->rank_X = dXs;
  ctx->rank_B = B;
  ctx->rank_pending = true;   // started by wcx_aux_kick (inside the search)
  return WCX_OK;
}

}  // extern "C"

// Rows whose reference-bin row is the gonosomal passes' dummy (all indices 0, newref_tools.py:186-191):
// the median of k copies of x[0] is x[0], so out[r][m] = log2(x[row] / x[0]) -- no gather, no selection.
// Null ratios of rows whose reference list is the reference's dummy (all indexes 0: the autosomal target rows
// of a gonosomal pass, newref_tools.py:186-191 + 219-221): log2(x[row] / x[0]) per null sample.  A workgroup
// takes 32 rows: every wave reads its samples' values of those rows (sample-major: contiguous), the ratios
// meet in LDS and leave as the rows' contiguous n_ids doubles (one thread per (row, sample) wrote 8 bytes
// every 800: 0.8 TB/s).
constexpr int ND_ROWS = 32;
__global__ __launch_bounds__(256) void k_null_dummy(const double *__restrict__ Xs, int64_t B,
                                                    const int32_t *__restrict__ sids, int n_ids,
                                                    int64_t row_begin, int64_t n_rows,
                                                    double *__restrict__ out) {
  __shared__ double tile[ND_ROWS][129];
  const int64_t r0 = (int64_t)blockIdx.x * ND_ROWS;
  const int rl = threadIdx.x & (ND_ROWS - 1), part = threadIdx.x / ND_ROWS;      // 8 sample slots
  const int64_t r = r0 + rl;
  for (int m = part; m < n_ids; m += 256 / ND_ROWS) {
    const double *x = Xs + (int64_t)sids[m] * B;
    if (r < n_rows) tile[rl][m] = log2(x[row_begin + r] / x[0]);
  }
  __syncthreads();
  const int64_t rows_here = n_rows - r0 < ND_ROWS ? n_rows - r0 : ND_ROWS;
  const int64_t total = rows_here * n_ids;
  double *dst = out + r0 * n_ids;
  for (int64_t e = threadIdx.x; e < total; e += 256) {
    const int rr = (int)(e / n_ids), mm = (int)(e - (int64_t)rr * n_ids);
    dst[e] = tile[rr][mm];
  }
}

// Ranking cost ~0.37 ns per (null sample, bin) when this was measured (round 3; 0.07 ns alone on the
// device since round 5) whatever the number of target rows -- mostly hidden
// beside the refine on the auxiliary stream when the refine is long; the high-key selection costs
// 54 ns per (row, 100 samples) against the rank kernel's 35 ns and needs no ranking.  Measured at
// 15 kb (scripts/sweep_shard_nr.sh, bench.py): the chrX / chrY rows of a gonosomal pass (B / rows =
// 16-20): null ratios 1.4 / 1.9 ms (a 64-bit-key variant) -> 0.6 / 0.7 ms, and no sort beside the
// search; a rank's shard of an 8-GPU build (B / rows = 8): shard wall 14.7 -> 12.8 ms (S = 500),
// 7.8 -> 5.8 ms (S = 100); of a 4-GPU build: 20.4 -> 19.1 ms, 9.9 -> 9.8 ms; from half of the rows
// on the rank path wins.  Hence: no ranking from B / rows >= 4 (1.5 % slack for uneven shards;
// WCX_NR_DIRECT_RATIO overrides, 0 = always rank).
bool wcx_null_ratios_direct_pays(int64_t B, int64_t n_rows) {
  static const int ratio = [] {
    const char *e = getenv("WCX_NR_DIRECT_RATIO");
    return e && *e ? atoi(e) : 4;
  }();
  return ratio > 0 && n_rows * ratio <= B + B / 64;
}

// The search knows how many rows it really searches: if they are few, the ranking announced by
// wcx_null_rank_prepare_dev is dropped before it starts (wcx_null_ratios_dev then selects directly).
void wcx_aux_cancel_if_few_rows(wcx_ctx *ctx, int64_t B, int64_t n_rows) {
  if (ctx->rank_pending && wcx_null_ratios_direct_pays(B, n_rows)) {
    ctx->rank_pending = false;
    ctx->rank_X = nullptr;
  }
}

// Starts the pending ranking on the auxiliary stream, behind the main stream's current position.
// The search calls this between its sweep and its refine (see wcx_topk_screen_launch for the
// measurements behind that choice); wcx_null_ratios_dev calls it as a catch-all.
int wcx_aux_kick(wcx_ctx *ctx) {
  if (!ctx->rank_pending) return WCX_OK;
  ctx->rank_pending = false;
  const int n_ids = (int)ctx->rank_ids.size();
  RankLayout L;
  int rc = rank_layout(ctx->rank_B, n_ids, ctx->aux_stream, L);
  if (rc) return rc;
  char *base = reinterpret_cast<char *>(ctx->d_rank);
  WCX_HIP(hipEventRecord(ctx->ev_main, ctx->stream));
  WCX_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_main, 0));
  WCX_HIP(hipMemcpyAsync(base + L.o_sid, ctx->rank_ids.data(), (size_t)n_ids * 4, hipMemcpyHostToDevice,
                         ctx->aux_stream));
  rc = rank_run(ctx->rank_X, ctx->rank_B, n_ids, L, base, ctx->aux_stream);
  if (rc) return rc;
  WCX_HIP(hipEventRecord(ctx->ev_rank, ctx->aux_stream));
  return WCX_OK;
}

extern "C" {

int wcx_null_ratios_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                        const int32_t *d_idx, int64_t row_begin, int64_t row_end, int k,
                        const int32_t *sample_ids, int n_ids, double *d_out) {
  WCX_ARG(ctx && dXs && d_idx && sample_ids && d_out, "NULL argument");
  WCX_ARG(B > 0 && S > 0 && k > 0 && n_ids >= 0, "bad sizes");
  WCX_ARG(0 <= row_begin && row_begin <= row_end && row_end <= B, "bad row range");
  WCX_ARG(B < (1ll << BIN_BITS) && n_ids <= 128, "too many bins / null samples");
  WCX_ARG((int64_t)n_ids * B < (1ll << 31), "null samples x bins exceeds 2^31");
  int rc = check_ids(sample_ids, n_ids, S);
  if (rc) return rc;
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t n_rows = row_end - row_begin;
  if (n_rows == 0 || n_ids == 0) return WCX_OK;
  hipStream_t st = ctx->stream;
  RankLayout L;
  rc = rank_layout(B, n_ids, st, L);
  if (rc) return rc;
  char *base = nullptr;
  const bool prepared = ctx->rank_X == dXs && ctx->rank_B == B && (int)ctx->rank_ids.size() == n_ids &&
                        std::equal(ctx->rank_ids.begin(), ctx->rank_ids.end(), sample_ids);
  rc = wcx_timer_begin(ctx, "null_ratios");
  if (rc) return rc;
  if (!prepared && wcx_null_ratios_direct_pays(B, n_rows)) {
    // few rows: no ranking of the n_ids x B matrix, selection on the high key halves
    ctx->rank_pending = false;
    ctx->rank_X = nullptr;
    const int n_sg = (n_ids + 7) / 8;
    void *scr = nullptr;
    rc = wcx_scratch(ctx, (size_t)n_sg * B * 32 + (size_t)n_ids * 4 + 512, &scr);
    if (rc) return rc;
    unsigned int *Kg = reinterpret_cast<unsigned int *>(scr);
    int32_t *d_sids = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(scr) + (size_t)n_sg * B * 32 + 256);
    rc = wcx_upload_small(ctx, d_sids, sample_ids, (size_t)n_ids * 4);
    if (rc) return rc;
    k_nr_hikeys<<<dim3((unsigned)((B + NT - 1) / NT), (unsigned)n_sg), NT, 0, st>>>(dXs, B, d_sids, n_ids, Kg);
    const dim3 grid((unsigned)((n_rows + NT / 64 - 1) / (NT / 64)), (unsigned)n_sg);
#define WCX_NRH_LAUNCH(IPL)                                                                        \
  k_null_ratios_hi<IPL><<<grid, NT, 0, st>>>(Kg, dXs, d_sids, B, d_idx, row_begin, n_rows, k, n_ids, d_out)
    const int ipl = (k + 63) / 64;
    if (ipl <= 1) WCX_NRH_LAUNCH(1);
    else if (ipl <= 2) WCX_NRH_LAUNCH(2);
    else if (ipl <= 3) WCX_NRH_LAUNCH(3);
    else if (ipl <= 4) WCX_NRH_LAUNCH(4);
    else if (ipl <= 5) WCX_NRH_LAUNCH(5);
    else if (ipl <= 6) WCX_NRH_LAUNCH(6);
    else if (ipl <= 8) WCX_NRH_LAUNCH(8);
    else if (ipl <= 16) WCX_NRH_LAUNCH(16);
    else if (ipl <= 32) WCX_NRH_LAUNCH(32);
    else {
      wcx_set_error("refsize %d too large for the null-ratio kernel (max 2048)", k);
      return WCX_ERR_UNSUPPORTED;
    }
#undef WCX_NRH_LAUNCH
    WCX_HIP(hipGetLastError());
    return wcx_timer_end(ctx, "null_ratios");
  }
  if (prepared) {          // ranked ahead on the auxiliary stream (wcx_null_rank_prepare_dev)
    rc = wcx_aux_kick(ctx);
    if (rc) return rc;
    base = reinterpret_cast<char *>(ctx->d_rank);
    WCX_HIP(hipStreamWaitEvent(st, ctx->ev_rank, 0));
    ctx->rank_X = nullptr;
  } else {
    ctx->rank_pending = false;
    void *scr = nullptr;
    rc = wcx_scratch(ctx, L.total, &scr);
    if (rc) return rc;
    base = reinterpret_cast<char *>(scr);
    rc = wcx_upload_small(ctx, base + L.o_sid, sample_ids, (size_t)n_ids * 4);
    if (rc) return rc;
    rc = rank_run(dXs, B, n_ids, L, base, st);
    if (rc) return rc;
  }
  const int32_t *d_sids = reinterpret_cast<const int32_t *>(base + L.o_sid);
  const int *d_nan = reinterpret_cast<const int *>(base + L.o_nan);
  const unsigned int *Rg = reinterpret_cast<const unsigned int *>(base + L.o_rg);
  const double *V = reinterpret_cast<const double *>(base + L.o_v);
  const int ipl = (k + 63) / 64;
  const int n_sg = L.n_sg;
  const dim3 grid((unsigned)((n_rows + NT / 64 - 1) / (NT / 64)), (unsigned)n_sg);
#define WCX_NR_LAUNCH(IPL)                                                                      \
  k_null_ratios<IPL><<<grid, NT, 0, st>>>(Rg, V, d_nan, dXs, d_sids, B, d_idx, row_begin, n_rows, \
                                          k, n_ids, d_out, ctx->debug_flags)
  if (ipl <= 1) WCX_NR_LAUNCH(1);
  else if (ipl <= 2) WCX_NR_LAUNCH(2);
  else if (ipl <= 3) WCX_NR_LAUNCH(3);
  else if (ipl <= 4) WCX_NR_LAUNCH(4);
  else if (ipl <= 5) WCX_NR_LAUNCH(5);
  else if (ipl <= 6) WCX_NR_LAUNCH(6);
  else if (ipl <= 8) WCX_NR_LAUNCH(8);
  else if (ipl <= 16) WCX_NR_LAUNCH(16);
  else if (ipl <= 32) WCX_NR_LAUNCH(32);
  else {
    wcx_set_error("refsize %d too large for the null-ratio kernel (max 2048)", k);
    return WCX_ERR_UNSUPPORTED;
  }
#undef WCX_NR_LAUNCH
  WCX_HIP(hipGetLastError());
  return wcx_timer_end(ctx, "null_ratios");
}

int wcx_null_ratios_dummy_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S, int64_t row_begin,
                              int64_t row_end, const int32_t *sample_ids, int n_ids, double *d_out) {
  WCX_ARG(ctx && dXs && sample_ids && d_out, "NULL argument");
  WCX_ARG(B > 0 && S > 0 && n_ids >= 0 && n_ids <= 128, "bad sizes");
  WCX_ARG(0 <= row_begin && row_begin <= row_end && row_end <= B, "bad row range");
  int rc = check_ids(sample_ids, n_ids, S);
  if (rc) return rc;
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t n_rows = row_end - row_begin;
  if (n_rows == 0 || n_ids == 0) return WCX_OK;
  void *scr = nullptr;
  rc = wcx_scratch2(ctx, 1024, &scr);
  if (rc) return rc;
  rc = wcx_upload_small(ctx, scr, sample_ids, (size_t)n_ids * 4);
  if (rc) return rc;
  k_null_dummy<<<dim3((unsigned)((n_rows + ND_ROWS - 1) / ND_ROWS)), 256, 0, ctx->stream>>>(
      dXs, B, reinterpret_cast<const int32_t *>(scr), n_ids, row_begin, n_rows, d_out);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

int wcx_null_ratios(wcx_ctx *ctx, const double *Xs, int64_t B, int S, const int32_t *idx,
                    int64_t row_begin, int64_t row_end, int k, const int32_t *sample_ids,
                    int n_ids, double *out) {
  WCX_ARG(ctx && Xs && idx && out, "NULL argument");
  WCX_ARG(B > 0 && S > 0 && k > 0 && row_end >= row_begin, "bad sizes");
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t n_rows = row_end - row_begin;
  const size_t xb = (size_t)B * S * 8, ib = (size_t)n_rows * k * 4,
               ob = (size_t)n_rows * (size_t)n_ids * 8;
  void *buf = nullptr;
  int rc = wcx_scratch2(ctx, xb + ob + ib + 64, &buf);
  if (rc) return rc;
  double *dX = reinterpret_cast<double *>(buf);
  double *dO = reinterpret_cast<double *>(reinterpret_cast<char *>(buf) + xb);
  int32_t *dI = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(buf) + xb + ob);
  WCX_HIP(hipMemcpyAsync(dX, Xs, xb, hipMemcpyHostToDevice, ctx->stream));
  if (ib) WCX_HIP(hipMemcpyAsync(dI, idx, ib, hipMemcpyHostToDevice, ctx->stream));
  rc = wcx_null_ratios_dev(ctx, dX, B, S, dI, row_begin, row_end, k, sample_ids, n_ids, dO);
  if (rc) return rc;
  if (ob) WCX_HIP(hipMemcpyAsync(out, dO, ob, hipMemcpyDeviceToHost, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  return WCX_OK;
}

}  // extern "C"
