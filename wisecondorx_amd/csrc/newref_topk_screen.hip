// MFMA screen + exact fp64 refine for the reference-bin search (SURVEY.md §8a row a6;
// replaces newref_tools.py:255-278).  Results are identical to the exact kernel
// (newref_topk_exact.hip): the screen only decides WHICH candidates get the exact treatment.
//
// Pipeline (all on one stream):
//   k_col_stats    per-sample mean c_j and max |x - c_j|           (centring + fp16 scale)
//   k_transpose    Xs [S][B] -> Xr [B][S]                           (rows contiguous for refine)
//   k_row_norm / k_row_hist / k_scan_cells / k_scatter / k_group_mask
//                  best-first sweep order: counting sort by (norm bucket, chromosome)
//   k_screen_prep  a~ = fp16(2^p (x - c)) in MFMA-fragment order + four augmented k-columns that
//                  carry |a~|^2; per row: representation-error norm      (rigorous error budget)
//   k_screen       -2 a~.b~ Gram tiles on the matrix cores: v_mfma_f32_32x32x16_f16, fp32
//                  accumulate; targets stay in registers as the B operand (their augmented
//                  columns carry the current threshold), candidates stream through LDS as the A
//                  operand; the screen test is the SIGN BIT of the accumulator; passing pairs are
//                  appended to the target's shortlist, which is cut back by a wave-level bitwise
//                  bisection select whenever it nears capacity.
//   k_merge_segments (row shards with few target blocks only) joins per-segment shortlists.
//   k_refine       exact sequential fp64 distance (newref_tools.py:260 arithmetic) of every
//                  shortlisted pair, sort by (distance, index), emit the first k.
//   rows whose shortlist overflowed (never seen on real data) are redone by the exact kernel.
//
// Error budget (t = nb' - 2 g~, screen distance d~ = t + |a~|^2, all in scaled units):
//   sqrt(d) in [sqrt(dh) - E, sqrt(dh) + E],  dh = |a~ - b~|^2,  E = e_a + e_max  (e = |a - a~|)
//   |d~ - dh| <= Q  (fp32 accumulation of the 16 NK products incl. the augmented ones,
//     gamma = n 2^-23; fp32 roundings; nb - nb'): see row_budget()
//   T = (sqrt(d~_(k) + Q) + E)^2 bounds the true k-th distance; a pair can be among the k nearest
//   only if d~ <= F = (sqrt(T) + E)^2 + Q.  Everything with d~ <= F is kept and refined exactly.
//
// Roofline: MFMA bound, 2*32*32*16 flop per instruction, dense f16 peak ~2.5 PFLOP/s.
#include <cstdlib>

#include "wave_sort.h"
#include "wcx_common.h"
#include "screen_common.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ unsigned int f32_key(float t) {
  const unsigned int u = __float_as_uint(t);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned int kx) {
  const unsigned int u = (kx & 0x80000000u) ? (kx & 0x7fffffffu) : ~kx;
  return __uint_as_float(u);
}
__device__ __forceinline__ float up(float v) {  // a float strictly above v (v >= 0, finite)
  return v * 1.0000005f + 1e-37f;
}

// ------------------------------------------------------------------------------------------
constexpr int CSPLIT = 16;   // workgroups per sample column

// per-sample mean over finite entries: partial sums, fp64 atomics
__device__ __forceinline__ unsigned long long dord(double x) {   // order-preserving image
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dord_inv(unsigned long long k) {
  return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

// ONE sweep of X: per-sample sum, count, min and max over finite entries (max |x - mean| follows
// from min and max: it is attained at one of them, with the same rounding as fabs(x - mean)).
__global__ __launch_bounds__(NT) void k_col_sum(const double *__restrict__ Xs, int64_t B,
                                                double *__restrict__ csum,
                                                double *__restrict__ ccnt,
                                                unsigned long long *__restrict__ cmin,
                                                unsigned long long *__restrict__ cmax) {
  const double *x = Xs + (int64_t)blockIdx.x * B;
  double s = 0.0, c = 0.0, mn = HUGE_VAL, mx = -HUGE_VAL;
  for (int64_t i = (int64_t)blockIdx.y * NT + threadIdx.x; i < B; i += (int64_t)NT * CSPLIT) {
    const double v = x[i];
    if (fabs(v) < HUGE_VAL) { s += v; c += 1.0; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }  // finite only
  }
  s = wcx::wave_sum(s);
  c = wcx::wave_sum(c);
  mn = wcx::wave_min_f64(mn);
  mx = wcx::wave_max_f64(mx);
  if ((threadIdx.x & 63) == 0 && c > 0.0) {
    atomicAdd(&csum[blockIdx.x], s);
    atomicAdd(&ccnt[blockIdx.x], c);
    atomicMin(&cmin[blockIdx.x], dord(mn));
    atomicMax(&cmax[blockIdx.x], dord(mx));
  }
}

// cmean[j] = csum/ccnt; global max |x - mean| over finite entries
__global__ void k_col_stats(int S, const double *__restrict__ csum, const double *__restrict__ ccnt,
                            const unsigned long long *__restrict__ cmin,
                            const unsigned long long *__restrict__ cmax,
                            double *__restrict__ cmean, ScreenGlobals *__restrict__ glob) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S) return;
  const double cc = ccnt[j];
  const double m = cc > 0 ? csum[j] / cc : 0.0;
  cmean[j] = m;
  if (cc > 0) {
    const double a0 = fabs(dord_inv(cmax[j]) - m), a1 = fabs(dord_inv(cmin[j]) - m);
    const double mx = a0 > a1 ? a0 : a1;
    if (mx < HUGE_VAL) atomicMax(&glob->amax_bits, (unsigned long long)__double_as_longlong(mx));
  }
}

__global__ void k_transpose(const double *__restrict__ Xs, int64_t B, int S, int Sp,
                            double *__restrict__ Xr) {
  // Xs [S][B] -> Xr [B][Sp], Sp = S rounded up to 4 doubles (32-byte aligned rows, zero pad)
  __shared__ double tile[32][33];
  const int64_t b0 = (int64_t)blockIdx.x * 32;
  const int j0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int r = ty; r < 32; r += 8) {
    const int j = j0 + r;
    const int64_t b = b0 + tx;
    tile[r][tx] = (j < S && b < B) ? Xs[(int64_t)j * B + b] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t b = b0 + r;
    const int j = j0 + tx;
    if (b < B && j < Sp) Xr[b * Sp + j] = tile[tx][r];
  }
}

// One thread per (padded) row: centre, scale, split into fp16 hi/lo in MFMA fragment order.
// Fragment array F: half8[tile = row/32][ks][plane][lane'], lane' = (row%32) + 32*(k/8 % 2),
// the 8 halfs are k = ks*16 + 8*(lane'/32) + 0..7 -- exactly the A/B operand of
// v_mfma_f32_32x32x16_f16, so one wave-wide 16-byte load per (ks, plane) is fully coalesced.
// ------------------------------------------------------------------------------------------
// Best-first candidate order.  Nearest neighbours are overwhelmingly low-noise bins (small
// centred norm), so the candidates are swept in order of increasing norm bucket: the top-k
// thresholds are near-final after the first few per cent of the sweep and almost nothing
// passes the screen afterwards.  Order = counting sort by (norm bucket, chromosome); groups of
// 64 positions are therefore chromosome-pure except at cell borders, which lets the sweep skip
// own-chromosome groups wholesale (gmask) and mask rows only in the rare mixed groups.
constexpr int NBUCKET = 128;            // norm buckets: float bits >> 20 (12.5 % steps)
constexpr int NCELL = NBUCKET * 32;     // (bucket, chromosome) cells

__global__ __launch_bounds__(NT) void k_row_norm(const double *__restrict__ Xs, int64_t B, int S,
                                                 int Sp, const double *__restrict__ cmean,
                                                 ChrTab chr, ScreenGlobals *__restrict__ glob,
                                                 unsigned int *__restrict__ rbits,
                                                 int *__restrict__ rchr) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int j = 0; j < S; ++j) {
    const float a = (float)(Xs[(int64_t)j * B + b] - cmean[j]);   // sample-major: coalesced
    s += a * a;
  }
  int c = 0;
  while (c < chr.n_chr - 1 && b >= chr.cum[c]) ++c;
  rchr[b] = c;
  unsigned int u = 0xffffffffu;                     // non-finite rows go last
  if (s < HUGE_VALF) {
    u = __float_as_uint(s) >> 20;
    atomicMax(&glob->uinv, 0xffffffffu - u);
  }
  rbits[b] = u;
}

// Cell histogram with workgroup-private LDS counters (the rows pile into a few dozen cells: global
// atomics straight from every row serialise).
__global__ __launch_bounds__(NT) void k_row_hist(const unsigned int *__restrict__ rbits,
                                                 const int *__restrict__ rchr, int64_t B,
                                                 const ScreenGlobals *__restrict__ glob,
                                                 int *__restrict__ rkey, int *__restrict__ cellcnt) {
  __shared__ int lh[NCELL];
  for (int i = threadIdx.x; i < NCELL; i += NT) lh[i] = 0;
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b < B) {
    const unsigned int umin = 0xffffffffu - glob->uinv;
    const unsigned int u = rbits[b];
    unsigned int bucket = NBUCKET - 1;
    if (u != 0xffffffffu && u >= umin && u - umin < NBUCKET - 1) bucket = u - umin;
    const int key = (int)bucket * 32 + rchr[b];
    rkey[b] = key;
    atomicAdd(&lh[key], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NCELL; i += NT)
    if (lh[i]) atomicAdd(&cellcnt[i], lh[i]);
}

__global__ __launch_bounds__(1024) void k_scan_cells(const int *__restrict__ cellcnt,
                                                     int *__restrict__ cursor) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  constexpr int PER = NCELL / 1024;
  int loc[PER], s = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) { loc[i] = s; s += cellcnt[t * PER + i]; }
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  const int base = part[t] - s;
#pragma unroll
  for (int i = 0; i < PER; ++i) cursor[t * PER + i] = base + loc[i];
}

// Rows -> sweep positions: each workgroup reserves one range per cell it touches.
__global__ __launch_bounds__(NT) void k_scatter(const int *__restrict__ rkey, int64_t B,
                                                int *__restrict__ cursor, int *__restrict__ perm,
                                                int *__restrict__ rowpos) {
  __shared__ int lh[NCELL];
  for (int i = threadIdx.x; i < NCELL; i += NT) lh[i] = 0;
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  int key = 0, local = 0;
  if (b < B) { key = rkey[b]; local = atomicAdd(&lh[key], 1); }
  __syncthreads();
  for (int i = threadIdx.x; i < NCELL; i += NT)
    if (lh[i]) lh[i] = atomicAdd(&cursor[i], lh[i]);        // count -> base of this workgroup's range
  __syncthreads();
  if (b < B) {
    const int pos = lh[key] + local;
    perm[pos] = (int)b;
    rowpos[b] = pos;
  }
}

__global__ __launch_bounds__(NT) void k_group_mask(const int *__restrict__ perm,
                                                   const int *__restrict__ rchr, int64_t n_groups,
                                                   unsigned int *__restrict__ gmask) {
  const int64_t g = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (g >= n_groups) return;
  unsigned int m = 0;
  for (int i = 0; i < CT; ++i) {
    const int row = perm[g * CT + i];
    if (row >= 0) m |= 1u << rchr[row];
  }
  gmask[g] = m;
}

// Two fp16 values h1 + h2 bracketing x from below (up = false) or above (up = true), with
// sum = h1 + h2 EXACT in fp32 and no fp16 subnormals (quantum >= 2^-14; the matrix pipe may
// flush them).  |x| <= 65000.  Error |x - sum| < max(2^-21 |x|, 2^-14).
__device__ __forceinline__ void split16(float x, bool up, _Float16 &h1, _Float16 &h2, float &sum) {
  h1 = (_Float16)x;                       // round to nearest
  const float f1 = (float)h1;
  const float r = x - f1;                 // exact (Sterbenz)
  int eb = (int)((__float_as_uint(f1) >> 23) & 0xffu) - 21;
  if (eb < 113) eb = 113;
  const float q = __uint_as_float((unsigned int)eb << 23);
  const float m = up ? ceilf(r / q) : floorf(r / q);   // |m| <= 1024
  const float f2 = m * q;
  h2 = (_Float16)f2;                      // exact
  sum = f1 + f2;                          // exact: a multiple of q below 2^(e+1)
}

constexpr float AUG = 32768.f;            // the constant factor of the augmented products
constexpr float GMAX = 4.0e9f;            // thresholds at or above this count as "none yet"
constexpr float SLOW_OFF = 3.75e9f;       // slow-path pass offset: real rows pass, padding fails

// One thread per sweep POSITION p (the row it holds is perm[p], -1 = padding): fp16 image of the
// centred, scaled row in MFMA-fragment order + the augmented columns.
//
// Augmented columns (the last four of the padded K = 16 NK; requires S <= K - 4): a candidate row
// carries (-u1, -u2, AUG, AUG) with 65536 (u1 + u2) = nb' <= |b~|^2, a target row carries
// (AUG, AUG, w1, w2) with 65536 (w1 + w2) = G' >= G (patched in registers by k_screen), so the
// matrix product itself delivers  acc = g~ - nb'/2 + G'/2  and the screen test  nb' - 2 g~ <= G'
// is the sign bit of acc: no per-output VALU work besides collecting that bit.
template <int NK>
__global__ __launch_bounds__(NT) void k_screen_prep(
    const double *__restrict__ Xr, int64_t Bpad, int S, int Sp,
    const double *__restrict__ cmean, const int *__restrict__ perm,
    ScreenGlobals *__restrict__ glob, half8 *__restrict__ F, RowInfo *__restrict__ info) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b >= Bpad) return;
  const int64_t row = perm[b];
  // scale 2^p so that the largest |a| lands in [1024, 2048): |a~|^2 <= 508 * 2^22 < 2^31.1 keeps
  // nb/65536 and any threshold/65536 inside the fp16 range
  const double amax = __longlong_as_double((long long)glob->amax_bits);
  int ex = 0;
  double scale = 1.0;
  if (amax > 0.0) {
    frexp(amax, &ex);            // amax = m 2^ex, m in [0.5,1)
    scale = ldexp(1.0, 11 - ex);
  }
  const int64_t tile = b >> 5;
  const int rl = (int)(b & 31);
  bool bad = false;
  if (row >= 0)
    for (int j = 0; j < S; ++j) bad |= !(fabs(Xr[row * Sp + j]) < HUGE_VAL);
  const bool zero = row < 0 || bad;          // padding / NaN-inf rows: all-zero image
  double n2 = 0.0, e2 = 0.0;
#pragma unroll 1
  for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      half8 hi;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = ks * 16 + h * 8 + e;
        double a = 0.0;
        if (!zero && j < S) a = (Xr[row * Sp + j] - cmean[j]) * scale;
        _Float16 hh = (_Float16)a;
        if (fabs((double)hh) < 6.103515625e-05) hh = (_Float16)0;   // no fp16 subnormals
        const double res = a - (double)hh;
        n2 += (double)hh * (double)hh;
        e2 += res * res;
        hi[e] = hh;
      }
      if (ks < NK - 1 || h == 0) F[(tile * NK + ks) * 64 + rl + 32 * h] = hi;
      else {
        // last half fragment: entries 4..7 are the augmented columns (filled below)
        float nbf = (float)n2;
        if ((double)nbf > n2) nbf = __uint_as_float(__float_as_uint(nbf) - 1u);
        _Float16 u1, u2;
        float usum;
        split16(nbf * (1.f / 65536.f), false, u1, u2, usum);
        if (zero) { u1 = (_Float16)65504.f; u2 = (_Float16)65504.f; usum = 131008.f; }
        hi[4] = -u1; hi[5] = -u2; hi[6] = (_Float16)AUG; hi[7] = (_Float16)AUG;
        F[(tile * NK + ks) * 64 + rl + 32 * h] = hi;
        RowInfo ri;
        ri.nb = usum * 65536.f;               // nb' exactly as the matrix pipe sees it
        ri.L = 0.f;
        if (zero) { ri.e = 0.f; ri.N = 0.f; }
        else {
          // representation error also covers the fp64 rounding of (x - c) * scale
          ri.e = up((float)(sqrt(e2) + 1e-15 * sqrt(n2)));
          ri.N = up((float)sqrt(n2));
          atomicMax(&glob->e_max, __float_as_uint(ri.e));
          atomicMax(&glob->N_max, __float_as_uint(ri.N));
        }
        info[b] = ri;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
struct ScreenBlock {
  int64_t row0;
  int32_t nrows;  // <= TGT
  int32_t chr;    // chromosome index of the target rows
  int64_t cs, ce;
};

// Error budget of one target row (see the file header): na = |a~|^2 as encoded, E, Q.
// |computed t - exact hi-plane t| <= Q: fp32 accumulation of the 16 NK products (data columns
// + the nb'/2 and G'/2 columns: sum |x y| <= N_a N_max + (N_a + N_max)^2), the rounding of
// t = G' - 2 acc, and nb - nb' < 2^-21 nb + 4.  gamma = (16 NK + 8) 2^-23.
__device__ __forceinline__ void row_budget(const RowInfo &ti, float e_max, float N_max, float gamma,
                                           float &na, float &E, float &Q) {
  na = ti.nb;
  E = up(ti.e + e_max);
  const float nsum = ti.N + N_max;
  Q = up(2.f * gamma * ti.N * N_max + 4.8e-7f * (ti.N * ti.N + 2.f * N_max * N_max) +
         2.2f * gamma * nsum * nsum + 4.f);
}
constexpr float G_INIT = 3.0e38f;   // "no threshold yet" (finite on purpose)

// Wave-level shortlist compaction of one target of this wave, in two halves so that a burst of
// compactions can have the NEXT target's 8 KB in flight while the current one is processed.
// Shortlist entries are (float bits of t, sweep position).
//
// All CAP slots exist in memory: load unconditionally (16 independent loads in flight; a load
// under `if (e < n)` made the compiler wait for each one in turn -- 16 serial round trips to HBM,
// ~40k cycles per compaction) and mask afterwards.
__device__ __forceinline__ void load_shortlist(const uint2 *__restrict__ sl_row, uint2 (&raw)[CAP / 64]) {
  const int lane = wcx::lane_id();
#pragma unroll
  for (int q = 0; q < CAP / 64; ++q) raw[q] = sl_row[q * 64 + lane];
}

// Returns the new threshold G (t-space) and the new count (n_out; 0 + overflow flag when the list
// cannot be cut below LIM).  exact: resolve the k-th key to the last bit (final cut), else to
// 2^-11 relative -- the resolution of the fp16 screen itself -- rounded up (in-sweep cuts: a tight
// threshold matters, every later candidate passes with probability ~ rank/n).
__device__ __forceinline__ float compact_loaded(const uint2 (&raw)[CAP / 64], int n,
                                                uint2 *__restrict__ sl_row, int k, float na, float E,
                                                float Q, float G_old, unsigned int *overflow_flag,
                                                bool exact, int &n_out) {
  const int lane = wcx::lane_id();
  unsigned int key[CAP / 64], idx[CAP / 64];
#pragma unroll
  for (int q = 0; q < CAP / 64; ++q) {
    const bool in = q * 64 + lane < n;
    key[q] = in ? f32_key(__uint_as_float(raw[q].x)) : 0xffffffffu;
    idx[q] = in ? raw[q].y : 0u;
  }
  float G = G_old;
  if (n >= k) {
    // k-th smallest key by bitwise bisection (ballot counts): largest v with #(key < v) < k.
    // The keys share their leading bits (sign, exponent, ...): start below the common prefix.
    const unsigned int kref = (unsigned int)__builtin_amdgcn_readfirstlane((int)key[0]);
    unsigned int x = 0;
#pragma unroll
    for (int q = 0; q < CAP / 64; ++q) x |= (q * 64 + lane < n) ? (key[q] ^ kref) : 0u;
    x = wcx::wave_or_u32(x);
    const int hb = 31 - __builtin_clz(x | 1u);              // highest differing bit (0 if none)
    unsigned int prefix = kref & ~((2u << hb) - 1u);
    const int low = exact ? 0 : 12;
    for (int bit = hb; bit >= low; --bit) {
      const unsigned int trial = prefix | (1u << bit);
      int c = 0;
#pragma unroll
      for (int q = 0; q < CAP / 64; ++q) c += __popcll(__ballot(key[q] < trial));
      if (c < k) prefix = trial;
    }
    if (!exact && hb >= 12) prefix |= 0xfffu;
    if (!exact && hb < 12) prefix |= (2u << hb) - 1u;       // all keys within the low bits: upper end
    const float tk = key_f32(prefix);
    // T-space -> distance space -> filter bound F -> back to t-space, rounded outwards
    float dk = tk + na;
    dk = dk > 0.f ? dk : 0.f;
    const float rt = sqrtf(up(dk + Q)) * 1.0000005f + 2.f * E;
    const float Fb = up(up(rt * rt) + Q);
    const float Gn = (Fb - na) + 4e-7f * (Fb + na);
    if (tk < HUGE_VALF && Gn < G_old) G = Gn;   // (NaN / inf bound: keep the old threshold)
  }
  // keep entries with t <= G
  const unsigned int gkey = f32_key(G);
  int base = 0;
#pragma unroll
  for (int q = 0; q < CAP / 64; ++q) {
    const bool keep = (q * 64 + lane < n) && (key[q] <= gkey);
    const unsigned long long m = __ballot(keep);
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) sl_row[pos] = make_uint2(__float_as_uint(key_f32(key[q])), idx[q]);
    base += __popcll(m);
  }
  if (base > LIM) {   // cannot make room: hand the row to the exact kernel
    if (lane == 0) *overflow_flag = 1u;
    n_out = 0;
    return -HUGE_VALF;
  }
  n_out = base;
  return G;
}

// Threshold -> the two fp16 values of the target's augmented columns and the value G' they encode.
__device__ __forceinline__ void encode_threshold(float G, _Float16 &w1, _Float16 &w2, float &Gp) {
  if (G < -GMAX) {            // nothing may pass (unused target lane, overflowed row)
    w1 = (_Float16)-65504.f; w2 = (_Float16)-65504.f; Gp = -131008.f * 65536.f;
  } else if (G < GMAX) {
    float s;
    split16(G * (1.f / 65536.f), true, w1, w2, s);
    Gp = s * 65536.f;
  } else {                    // no threshold yet: columns off, the slow path passes every real row
    w1 = (_Float16)0; w2 = (_Float16)0; Gp = 0.f;
  }
}

// NK = k-steps of 16 (K = 16 NK >= S + 4), CTG = candidate sub-tiles of 32 rows per iteration.
// PROF = per-phase s_memtime accounting into stats[8..13] (diagnostics, debug flag 4).
template <int NK, int CTG, bool PROF = false>
__global__ __launch_bounds__(NT, (NK <= 8 ? 3 : 2)) void k_screen(
    const half8 *__restrict__ F, const RowInfo *__restrict__ info,
    const ScreenGlobals *__restrict__ glob,
    const int *__restrict__ perm, const int *__restrict__ rowpos,
    const unsigned int *__restrict__ gmask,
    const ScreenBlock *__restrict__ blocks, int k, int64_t row_begin,
    uint2 *__restrict__ sl, int *__restrict__ cnt_out, unsigned int *__restrict__ flags,
    float *__restrict__ g_state, int64_t gi_begin, int64_t gi_end, int first, int last,
    unsigned long long *__restrict__ stats, int dbg, int n_seg, int64_t n_rows_all) {
  // The candidate sweep is cut into chunks [gi_begin,gi_end) of groups, one launch per chunk:
  // every workgroup of a launch streams the SAME few MB of candidate fragments, which therefore
  // come out of the XCD L2s instead of HBM/MALL.  Per-target state (threshold G, shortlist
  // count) lives in g_state/cnt_out between launches; the shortlists are in HBM anyway.
  constexpr int GR = CTG * 32;                      // candidate rows per iteration
  constexpr int TILE_H8 = CTG * NK * 64;            // half8 elements per staged candidate group
  constexpr int NPT = (TILE_H8 + NT - 1) / NT;      // 16-byte pieces per thread
  constexpr int NOUT = CTG * 16;                    // screen outputs per lane per iteration
  extern __shared__ __align__(16) unsigned char smem[];
  half8 *sbuf = reinterpret_cast<half8 *>(smem);                       // [2][TILE_H8]
  int *glist = reinterpret_cast<int *>(smem + 2 * TILE_H8 * 16);       // [groups of the chunk]
  __shared__ int s_nlist;

  // Candidate segments: with few target blocks (a row shard of a multi-GPU build) every block is
  // issued n_seg times; copy `seg` sweeps the candidate groups g = seg (mod n_seg) into its own
  // shortlists / thresholds (arrays offset by seg * n_rows_all); k_merge_segments joins them.
  const int n_blocks = (int)gridDim.x / n_seg;
  const int seg = (int)blockIdx.x / n_blocks;
  const ScreenBlock blk = blocks[(int)blockIdx.x - seg * n_blocks];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tl = wave * 32 + (lane & 31);        // local target of this lane
  const int hf = lane >> 5;
  const bool tvalid = tl < blk.nrows;
  const int64_t trow = tvalid ? blk.row0 + tl : blk.row0;
  const int64_t soff = (int64_t)seg * n_rows_all;
  const int64_t srow = trow - row_begin + soff;
  const int64_t wg_srow = blk.row0 - row_begin + soff;
  uint2 *wg_sl = sl + wg_srow * (int64_t)CAP;    // this workgroup's TGT shortlists (uniform base)

  // Visit list of this launch's chunk, built once per workgroup in LDS: groups holding only
  // own-chromosome rows are skipped (gmask = chromosomes present per 64 rows); bit 31 marks groups
  // that also contain own-chromosome rows.
  const unsigned int blkbit = 1u << blk.chr;
  if (wave == 0) {
    int count = 0;
    for (int64_t g0 = gi_begin; g0 < gi_end; g0 += 64) {
      const int64_t g = g0 + lane;
      unsigned int m = blkbit;
      if (g < gi_end) m = gmask[(g * GR) >> 6];
      const bool keep = (g < gi_end) && m != blkbit && ((int)g & (n_seg - 1)) == seg;
      const unsigned long long bal = __ballot(keep);
      if (keep) glist[count + __popcll(bal & ((1ull << lane) - 1ull))] =
          (int)g | ((m & blkbit) ? (int)0x80000000 : 0);
      count += __popcll(bal);
    }
    if (lane == 0) s_nlist = count;
  }

  // target operand (B operand of the MFMA) stays in registers for the whole sweep
  const RowInfo ti = info[rowpos[trow]];
  half8 th[NK];
  {
    const int64_t tpos = rowpos[trow];        // sweep position of the target row
    const int64_t ttile = tpos >> 5;
    const int trl = (int)(tpos & 31);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) th[ks] = F[(ttile * NK + ks) * 64 + trl + 32 * hf];
  }
  const float e_max = __uint_as_float(glob->e_max), N_max = __uint_as_float(glob->N_max);
  float na, E, Q;
  row_budget(ti, e_max, N_max, (float)(16 * NK + 8) * 1.1920929e-7f, na, E, Q);
  float G = tvalid ? (first ? G_INIT : g_state[srow]) : -HUGE_VALF;
  float Gp;
  {
    _Float16 w1, w2;
    encode_threshold(G, w1, w2, Gp);
    if (hf) { th[NK - 1][4] = (_Float16)AUG; th[NK - 1][5] = (_Float16)AUG;
              th[NK - 1][6] = w1; th[NK - 1][7] = w2; }
  }
  int n_compact = 0, n_app = 0;
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, tp = 0;
  auto stamp = [&](int ph) {
    if (PROF) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      pt[ph] += now - tp;
      tp = now;
    }
  };

  half8 pre[NPT];
  auto fetch = [&](int gix) {
    const half8 *src = F + (int64_t)gix * TILE_H8;
#pragma unroll
    for (int p = 0; p < NPT; ++p)
      if ((p + 1) * NT <= TILE_H8 || p * NT + tid < TILE_H8) pre[p] = src[p * NT + tid];
  };
  int buf = 0;
  __syncthreads();
  const int n_list = s_nlist;
  int cur = n_list > 0 ? glist[0] : 0;
  if (n_list > 0) {
    fetch(cur & 0x7fffffff);
#pragma unroll
    for (int p = 0; p < NPT; ++p)
      if ((p + 1) * NT <= TILE_H8 || p * NT + tid < TILE_H8) sbuf[p * NT + tid] = pre[p];
  }
  // shortlist count of my target, kept in a register (identical in the target's two lanes)
  int cntr = (first || !tvalid) ? 0 : cnt_out[srow];
  __syncthreads();
  bool fast = false;
  // A burst of compactions (targets of this wave flagged in `need`): the next target's shortlist
  // is loaded while the current one is selected and written back.
  auto run_compactions = [&](unsigned int need, bool exact) {
    const float G_before = G;
    // this wave's appends must be visible before they are re-read
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    int c = __ffs((int)need) - 1;
    need &= need - 1;
    uint2 raw[CAP / 64];
    load_shortlist(sl + (wg_srow + wave * 32 + c) * (int64_t)CAP, raw);
    for (;;) {
      const int cn = need ? __ffs((int)need) - 1 : -1;
      need &= need - 1;
      uint2 rawn[CAP / 64];
      if (NK <= 8 && cn >= 0) load_shortlist(sl + (wg_srow + wave * 32 + cn) * (int64_t)CAP, rawn);
      const int64_t crow_s = wg_srow + wave * 32 + c;
      const float na_c = __shfl(na, c, 64), E_c = __shfl(E, c, 64), Q_c = __shfl(Q, c, 64),
                  G_c = __shfl(G, c, 64);
      const int n_c = __builtin_amdgcn_readlane(cntr, c);
      int n_new;
      const float Gn = compact_loaded(raw, n_c, sl + crow_s * (int64_t)CAP, k, na_c, E_c, Q_c, G_c,
                                      &flags[crow_s], exact, n_new);
      if ((lane & 31) == c) { G = Gn; cntr = n_new; }
      ++n_compact;
      if (cn < 0) break;
      c = cn;
      if (NK <= 8) {
#pragma unroll
        for (int q = 0; q < CAP / 64; ++q) raw[q] = rawn[q];
      } else {                       // large K: no registers to spare for the look-ahead
        load_shortlist(sl + (wg_srow + wave * 32 + c) * (int64_t)CAP, raw);
      }
    }
    if (G != G_before) {
      _Float16 w1, w2;
      encode_threshold(G, w1, w2, Gp);
      if (hf) { th[NK - 1][6] = w1; th[NK - 1][7] = w2; }
    }
  };
  for (int j = 0; j < n_list; ++j) {
    // Order inside an iteration: issue the next group's global loads, run the MFMA block on the
    // current LDS buffer, park the loaded group in the other buffer, barrier, THEN do the
    // shortlist appends.  The appends' stores share the vmcnt counter with the loads; in this order
    // nobody waits for a store until a whole MFMA block later.
    const int nxt = (j + 1 < n_list) ? glist[j + 1] : 0;
    half8 *sb = sbuf + buf * TILE_H8;
    const int gix = cur & 0x7fffffff;
    const bool mixed = cur < 0;                  // some own-chromosome rows in this group
    if (PROF) tp = __builtin_amdgcn_s_memtime();
    if (j + 1 < n_list) fetch(nxt & 0x7fffffff);

    // acc = g~ - nb'/2 + G'/2 straight out of the matrix pipe (see k_screen_prep); A fragments
    // are read lane-linearly (conflict-free ds_read_b128), a few reads ahead of their MFMA.
    f32x16 acc[CTG];
#pragma unroll
    for (int sub = 0; sub < CTG; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[sub][r] = 0.f;
    {
      half8 a[NK][CTG];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int sub = 0; sub < CTG; ++sub) a[ks][sub] = sb[(sub * NK + ks) * 64 + lane];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int sub = 0; sub < CTG; ++sub)
          acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][sub], th[ks], acc[sub], 0, 0, 0);
      // schedule: PRE reads up front, then one read per MFMA, the last PRE MFMAs back to back
      constexpr int NM = NK * CTG, PRE = NM < 6 ? NM : 6;
      __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
#pragma unroll
      for (int i = 0; i < NM - PRE; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, PRE, 0);
    }
    stamp(0);                                    // loads issued + MFMA block issued
    if (j + 1 < n_list) {
      half8 *so = sbuf + (buf ^ 1) * TILE_H8;
#pragma unroll
      for (int p = 0; p < NPT; ++p)
        if ((p + 1) * NT <= TILE_H8 || p * NT + tid < TILE_H8) so[p * NT + tid] = pre[p];
    }
    stamp(1);                                    // wait for the loads + LDS writes
    __syncthreads();
    stamp(2);                                    // barrier
    // C[row = candidate][col = target]; output rr = sub*16 + r is candidate row
    // loc(rr) = sub*32 + 8*(r>>2) + 4*(lane>>5) + (r&3) of this group.  Bit (31-rr) of pmask.
    if (!fast) fast = __all(G < GMAX);           // G only ever decreases
    unsigned int negs[CTG];                      // one dependent chain per sub-tile
    if (fast) {
#pragma unroll
      for (int sub = 0; sub < CTG; ++sub) {
        negs[sub] = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          negs[sub] = __builtin_amdgcn_alignbit(negs[sub], __float_as_uint(acc[sub][r]), 31);
      }
    } else {
      asm volatile("; slow path" ::: "memory");
      const float off = (G < GMAX) ? 0.f : SLOW_OFF;   // lanes without a threshold: pass real rows
#pragma unroll
      for (int sub = 0; sub < CTG; ++sub) {
        negs[sub] = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          negs[sub] = __builtin_amdgcn_alignbit(negs[sub], __float_as_uint(acc[sub][r] + off), 31);
      }
    }
    unsigned int neg = negs[0];
    if (CTG == 2) neg = (negs[0] << 16) | (negs[CTG - 1] & 0xffffu);
    unsigned int pmask = (~neg) << (32 - NOUT);
    if (mixed) {   // rare: mask the own-chromosome rows of a mixed group
      asm volatile("; mixed group" ::: "memory");   // keep this a branch (no if-conversion)
      const int cs32 = (int)blk.cs, ce32 = (int)blk.ce;
#pragma unroll 1
      for (int rr = 0; rr < NOUT; ++rr) {
        const int loc = (rr >> 4) * 32 + 8 * ((rr >> 2) & 3) + 4 * hf + (rr & 3);
        const int g = perm[(int64_t)gix * GR + loc];
        if (g >= cs32 && g < ce32) pmask &= ~(0x80000000u >> rr);
      }
    }
    if (dbg & 1) pmask = 0;
    const unsigned int anym = wcx::wave_or_u32(pmask);       // wave-uniform
    stamp(3);                                    // MFMA completion + sign bits + OR
    if (anym) {
      // slot reservation without LDS: the target's two lanes (l, l+32) swap their pass counts
      // (appends staged in LDS and flushed as whole 128-byte lines were measured: no gain)
      const unsigned int pc = (unsigned int)__popc(pmask);
      const auto pcs = __builtin_amdgcn_permlane32_swap(pc, pc, false, false);   // {low, high} lane's
      unsigned int ofs = (unsigned int)(tl * CAP + cntr + (hf ? (int)pcs[0] : 0));
      cntr += (int)(pcs[0] + pcs[1]);
      n_app += (int)pc;
      // cntr <= LIM + CT = CAP: the slots exist (counts are cut back to <= LIM below)
      const unsigned int pbase = (unsigned int)(gix * GR + 4 * hf);
#pragma unroll
      for (int rr = 0; rr < NOUT; ++rr) {
        if (anym & (0x80000000u >> rr)) {                    // scalar branch: skip empty outputs
          asm volatile("" ::: "memory");                     // (keeps the two tests separate)
          if (pmask & (0x80000000u >> rr)) {
            const int loc0 = (rr >> 4) * 32 + 8 * ((rr >> 2) & 3) + (rr & 3);
            const float t = fmaf(-2.f, acc[rr >> 4][rr & 15], Gp);
            wg_sl[ofs] = make_uint2(__float_as_uint(t), pbase + loc0);
            ++ofs;
          }
        }
      }
      stamp(4);                                  // appends
      // shortlist maintenance: wave-private (this wave's 32 targets); counts only change here
      const int trig = (dbg >> 8) ? (dbg >> 8) : LIM;         // (diagnostics: earlier cuts)
      const unsigned int need = (unsigned int)__ballot(tvalid && cntr > trig);   // low half = targets
      if (need) run_compactions(need, false);
    }
    stamp(5);                                    // maintenance (compactions)
    buf ^= 1;
    cur = nxt;
  }
  if (last) {
    // final cut of every target's shortlist with its final threshold (exact k-th key)
    const int nv = blk.nrows - wave * 32;
    if (nv > 0) run_compactions(nv >= 32 ? 0xffffffffu : ((1u << nv) - 1u), true);
  } else if (tvalid && hf == 0) {
    g_state[srow] = G;
  }
  if (tvalid && hf == 0) cnt_out[srow] = cntr;
  if (stats) {
    const int tot_c = wcx::wave_sum_i(n_compact), tot_a = wcx::wave_sum_i(n_app);
    if (lane == 0) {
      atomicAdd(&stats[2], (unsigned long long)(tot_c / 64));
      atomicAdd(&stats[4], (unsigned long long)tot_a);
      if (PROF)
        for (int i = 0; i < 6; ++i) atomicAdd(&stats[8 + i], pt[i]);
    }
  }
}

__global__ void k_mark(unsigned char *searched, const ScreenBlock *__restrict__ blocks,
                       int64_t row_begin) {
  const ScreenBlock b = blocks[blockIdx.x];
  if ((int)threadIdx.x < b.nrows) searched[b.row0 - row_begin + threadIdx.x] = 1;
}

// Joins the shortlists of candidate segments s0 and s1 of every row into s0's: exact cut of the
// union at its k-th smallest screen value (every row's true neighbours are in the union of the
// segments' own top lists).  One wave per row.
__global__ __launch_bounds__(NT) void k_merge_segments(
    const RowInfo *__restrict__ info, const ScreenGlobals *__restrict__ glob,
    const int *__restrict__ rowpos, int64_t row_begin, int64_t n_rows,
    const unsigned char *__restrict__ searched, uint2 *__restrict__ sl, int *__restrict__ cnt,
    unsigned int *__restrict__ flags, int s0, int s1, int k, float gamma) {
  const int lane = wcx::lane_id();
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  const float e_max = __uint_as_float(glob->e_max), N_max = __uint_as_float(glob->N_max);
  for (int64_t r = w0; r < n_rows; r += nw) {
    if (!searched[r]) continue;
    const int64_t i0 = (int64_t)s0 * n_rows + r, i1 = (int64_t)s1 * n_rows + r;
    const int n0 = cnt[i0], n1 = cnt[i1];
    if (flags[i0] | flags[i1] | (unsigned int)(n0 + n1 > CAP)) {   // -> exact fallback
      if (lane == 0) flags[i0] = 1u;
      continue;
    }
    uint2 raw[CAP / 64];
#pragma unroll
    for (int q = 0; q < CAP / 64; ++q) {
      const int e = q * 64 + lane;
      raw[q] = e < n0 ? sl[i0 * CAP + e] : sl[i1 * CAP + (e - n0 < CAP ? e - n0 : 0)];
    }
    float na, E, Q;
    row_budget(info[rowpos[row_begin + r]], e_max, N_max, gamma, na, E, Q);
    int n_new;
    (void)compact_loaded(raw, n0 + n1, sl + i0 * CAP, k, na, E, Q, G_INIT, &flags[i0], true, n_new);
    if (lane == 0) cnt[i0] = n_new;
  }
}

}  // namespace

// Host side --------------------------------------------------------------------------------
// dst[c][r] = src[r][c] for a row-major double matrix (rows x cols), on the context's stream.
int wcx_transpose_launch(wcx_ctx *ctx, const double *src, int64_t rows, int64_t cols, double *dst) {
  // k_transpose reads "Xs" [S'][B'] and writes "Xr" [B'][Sp]: S' = rows, B' = cols, Sp = rows
  k_transpose<<<dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), 256, 0,
                ctx->stream>>>(src, cols, (int)rows, (int)rows, dst);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

int wcx_debug_value = 0;   // diagnostics only (wcx_debug_flags): ablation switches for profiling
bool wcx_screen_supported(int64_t B, int S, int k) {
  return S <= 508 && k <= 512 && k <= LIM && B >= 2048;
}

int wcx_topk_screen_launch(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                           const int64_t *chr_cum, int n_chr,
                           const std::vector<TopkBlock> &exact_blocks, int64_t row_begin,
                           int64_t n_rows, int k, int32_t *d_out_idx, double *d_out_dist) {
  if (exact_blocks.empty()) return WCX_OK;
  // One fp16 plane (its 2^-11 representation error only widens the shortlists by a few dozen
  // entries; a hi+lo three-product form was measured 35 % slower end to end).  K = 16 NK holds the
  // S data columns + 4 augmented columns (see k_screen_prep); NK is rounded up to an instantiated
  // value.
  static const int nk_list[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32};
  int NK = 32;
  for (int v : nk_list)
    if (16 * v >= S + 4) { NK = v; break; }
  const int CTG = NK <= 16 ? 2 : 1;
  const int64_t Bpad = (B + CT - 1) / CT * CT;
  // regroup the searched row ranges into workgroups of <= 128 rows (same chromosome)
  std::vector<ScreenBlock> blocks;
  {
    size_t i = 0;
    while (i < exact_blocks.size()) {
      ScreenBlock sb;
      sb.row0 = exact_blocks[i].row0;
      sb.nrows = exact_blocks[i].nrows;
      sb.chr = 0;
      for (int c = 0; c < n_chr; ++c)
        if (chr_cum[c] == exact_blocks[i].ce && (c ? chr_cum[c - 1] : 0) == exact_blocks[i].cs) sb.chr = c;
      sb.cs = exact_blocks[i].cs;
      sb.ce = exact_blocks[i].ce;
      size_t j = i + 1;
      while (j < exact_blocks.size() && exact_blocks[j].cs == sb.cs &&
             exact_blocks[j].row0 == sb.row0 + sb.nrows && sb.nrows + exact_blocks[j].nrows <= TGT) {
        sb.nrows += exact_blocks[j].nrows;
        ++j;
      }
      blocks.push_back(sb);
      i = j;
    }
  }
  // candidate segments: fill the chip when a row shard has few target blocks (multi-GPU builds)
  int n_seg = 1;
  if (blocks.size() < 512) n_seg = 2;      // measured: 2 helps below ~500 blocks, 4 never does
  if (const char *e = getenv("WCX_SCREEN_SEGMENTS")) {   // testing / tuning: 1, 2 or 4
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4) n_seg = v;
  }
  // scratch layout
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_glob = carve(sizeof(ScreenGlobals));
  const size_t o_mean = carve((size_t)S * 8 * 5);   // mean | sum | count | min | max
  const int Sp = (S + 3) & ~3;
  const size_t o_xr = carve((size_t)B * Sp * 8 + 256);   // + slack: refine loads whole 128-B chunks
  const size_t o_F = carve((size_t)Bpad * NK * 32);  // Bpad/32 tiles * NK * 1 KiB
  const size_t o_info = carve((size_t)Bpad * sizeof(RowInfo));
  const int64_t n_groups = Bpad / CT;
  const size_t o_perm = carve((size_t)Bpad * 4);
  const size_t o_rpos = carve((size_t)B * 4);
  const size_t o_rbit = carve((size_t)B * 4);
  const size_t o_rchr = carve((size_t)B * 4);
  const size_t o_rkey = carve((size_t)B * 4);
  const size_t o_cell = carve((size_t)NCELL * 4);
  const size_t o_curs = carve((size_t)NCELL * 4);
  const size_t o_gmsk = carve((size_t)n_groups * 4);
  const size_t o_sl = carve((size_t)n_seg * n_rows * CAP * 8);
  const size_t o_cnt = carve((size_t)n_seg * n_rows * 4);
  const size_t o_gst = carve((size_t)n_seg * n_rows * 4);
  const size_t o_flag = carve((size_t)n_seg * n_rows * 4);
  const size_t o_srch = carve((size_t)n_rows);
  const size_t o_blk = carve(blocks.size() * sizeof(ScreenBlock));
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, off, &scr);
  if (rc) return rc;
  char *base = reinterpret_cast<char *>(scr);
  ScreenGlobals *glob = reinterpret_cast<ScreenGlobals *>(base + o_glob);
  double *cmean = reinterpret_cast<double *>(base + o_mean);
  double *Xr = reinterpret_cast<double *>(base + o_xr);
  half8 *F = reinterpret_cast<half8 *>(base + o_F);
  RowInfo *info = reinterpret_cast<RowInfo *>(base + o_info);
  int *perm = reinterpret_cast<int *>(base + o_perm);
  int *rowpos = reinterpret_cast<int *>(base + o_rpos);
  unsigned int *rbits = reinterpret_cast<unsigned int *>(base + o_rbit);
  int *rchr = reinterpret_cast<int *>(base + o_rchr);
  int *rkey = reinterpret_cast<int *>(base + o_rkey);
  int *cellcnt = reinterpret_cast<int *>(base + o_cell);
  int *cursor = reinterpret_cast<int *>(base + o_curs);
  unsigned int *gmask = reinterpret_cast<unsigned int *>(base + o_gmsk);
  uint2 *sl = reinterpret_cast<uint2 *>(base + o_sl);
  int *cnt_out = reinterpret_cast<int *>(base + o_cnt);
  float *g_state = reinterpret_cast<float *>(base + o_gst);
  unsigned int *flags = reinterpret_cast<unsigned int *>(base + o_flag);
  unsigned char *searched = reinterpret_cast<unsigned char *>(base + o_srch);
  ScreenBlock *d_blocks = reinterpret_cast<ScreenBlock *>(base + o_blk);

  hipStream_t st = ctx->stream;
  WCX_HIP(hipMemsetAsync(glob, 0, sizeof(ScreenGlobals), st));
  WCX_HIP(hipMemsetAsync(cnt_out, 0, (size_t)n_seg * n_rows * 4, st));
  WCX_HIP(hipMemsetAsync(flags, 0, (size_t)n_seg * n_rows * 4, st));
  WCX_HIP(hipMemsetAsync(searched, 0, (size_t)n_rows, st));
  WCX_HIP(hipMemsetAsync(ctx->d_stats, 0, 128, st));
  rc = wcx_upload_small(ctx, d_blocks, blocks.data(), blocks.size() * sizeof(ScreenBlock));
  if (rc) return rc;
  k_mark<<<(unsigned)blocks.size(), TGT, 0, st>>>(searched, d_blocks, row_begin);

  rc = wcx_timer_begin(ctx, "topk");
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "topk_prep");
  if (rc) return rc;
  unsigned long long *cmin = reinterpret_cast<unsigned long long *>(cmean + 3 * S);
  unsigned long long *cmax = cmin + S;
  WCX_HIP(hipMemsetAsync(cmean + S, 0, (size_t)S * 16, st));
  WCX_HIP(hipMemsetAsync(cmin, 0xff, (size_t)S * 8, st));
  WCX_HIP(hipMemsetAsync(cmax, 0, (size_t)S * 8, st));
  k_col_sum<<<dim3((unsigned)S, CSPLIT), NT, 0, st>>>(dXs, B, cmean + S, cmean + 2 * S, cmin, cmax);
  k_col_stats<<<(unsigned)((S + 63) / 64), 64, 0, st>>>(S, cmean + S, cmean + 2 * S, cmin, cmax, cmean,
                                                         glob);
  k_transpose<<<dim3((unsigned)((B + 31) / 32), (unsigned)((Sp + 31) / 32)), 256, 0, st>>>(dXs, B, S, Sp, Xr);
  {
    ChrTab tab0;
    tab0.n_chr = n_chr;
    for (int c = 0; c < 32; ++c) tab0.cum[c] = c < n_chr ? chr_cum[c] : B;
    const unsigned gb = (unsigned)((B + NT - 1) / NT);
    WCX_HIP(hipMemsetAsync(cellcnt, 0, (size_t)NCELL * 4, st));
    WCX_HIP(hipMemsetAsync(perm, 0xff, (size_t)Bpad * 4, st));
    k_row_norm<<<gb, NT, 0, st>>>(dXs, B, S, Sp, cmean, tab0, glob, rbits, rchr);
    k_row_hist<<<gb, NT, 0, st>>>(rbits, rchr, B, glob, rkey, cellcnt);
    k_scan_cells<<<1, 1024, 0, st>>>(cellcnt, cursor);
    k_scatter<<<gb, NT, 0, st>>>(rkey, B, cursor, perm, rowpos);
    k_group_mask<<<(unsigned)((n_groups + NT - 1) / NT), NT, 0, st>>>(perm, rchr, n_groups, gmask);
  }
  const unsigned gprep = (unsigned)((Bpad + NT - 1) / NT);
  const int GRr = CTG * 32;
  // candidate chunk per launch: ~3 MB of fragments (fits the 4 MB XCD L2)
  const int64_t n_iter_groups = Bpad / GRr;
  const int64_t group_bytes = (int64_t)GRr * NK * 32;
  int64_t chunk_groups = (3 << 20) / group_bytes;
  if (chunk_groups < 16) chunk_groups = 16;
  if (chunk_groups > 4096) chunk_groups = 4096;
  const size_t lds = 2 * (size_t)(CTG * NK * 64) * 16 +
                     (size_t)(chunk_groups + 64) * 4;   // + the chunk's visit list
#define WCX_SCREEN_CASE(N, G)                                                                  \
  case N: {                                                                                    \
    k_screen_prep<N><<<gprep, NT, 0, st>>>(Xr, Bpad, S, Sp, cmean, perm, glob, F, info);        \
    rc = wcx_timer_end(ctx, "topk_prep");                                                      \
    if (rc) return rc;                                                                         \
    rc = wcx_timer_begin(ctx, "topk_screen");                                                  \
    if (rc) return rc;                                                                         \
    WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_screen<N, G>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
    for (int64_t g0 = 0; g0 < n_iter_groups; g0 += chunk_groups) {                             \
      const int64_t g1 = g0 + chunk_groups < n_iter_groups ? g0 + chunk_groups : n_iter_groups; \
      k_screen<N, G><<<(unsigned)(blocks.size() * n_seg), NT, lds, st>>>(                                \
          F, info, glob, perm, rowpos, gmask, d_blocks, k, row_begin, sl, cnt_out, flags,      \
          g_state, g0, g1, g0 == 0, g1 == n_iter_groups, ctx->d_stats, wcx_debug_value, n_seg,   \
          n_rows);                                                                             \
    }                                                                                          \
  } break;
  if ((wcx_debug_value & 4) && NK == 7) {
    k_screen_prep<7><<<gprep, NT, 0, st>>>(Xr, Bpad, S, Sp, cmean, perm, glob, F, info);
    rc = wcx_timer_end(ctx, "topk_prep");
    if (rc) return rc;
    rc = wcx_timer_begin(ctx, "topk_screen");
    if (rc) return rc;
    WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_screen<7, 2, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int64_t g0 = 0; g0 < n_iter_groups; g0 += chunk_groups) {
      const int64_t g1 = g0 + chunk_groups < n_iter_groups ? g0 + chunk_groups : n_iter_groups;
      k_screen<7, 2, true><<<(unsigned)(blocks.size() * n_seg), NT, lds, st>>>(
          F, info, glob, perm, rowpos, gmask, d_blocks, k, row_begin, sl, cnt_out, flags,
          g_state, g0, g1, g0 == 0, g1 == n_iter_groups, ctx->d_stats, wcx_debug_value, n_seg,
          n_rows);
    }
  } else
  switch (NK) {
    WCX_SCREEN_CASE(1, 2) WCX_SCREEN_CASE(2, 2) WCX_SCREEN_CASE(3, 2) WCX_SCREEN_CASE(4, 2)
    WCX_SCREEN_CASE(5, 2) WCX_SCREEN_CASE(6, 2) WCX_SCREEN_CASE(7, 2) WCX_SCREEN_CASE(8, 2)
    WCX_SCREEN_CASE(10, 2) WCX_SCREEN_CASE(12, 2) WCX_SCREEN_CASE(14, 2) WCX_SCREEN_CASE(16, 2)
    WCX_SCREEN_CASE(20, 1) WCX_SCREEN_CASE(24, 1) WCX_SCREEN_CASE(28, 1)
    default: WCX_SCREEN_CASE(32, 1)
  }
#undef WCX_SCREEN_CASE
  if (n_seg > 1) {
    const float gamma = (float)(16 * NK + 8) * 1.1920929e-7f;
    const unsigned gm = (unsigned)((n_rows + 3) / 4 < 65536 ? (n_rows + 3) / 4 : 65536);
    for (int step = 1; step < n_seg; step *= 2)
      for (int s0 = 0; s0 + step < n_seg; s0 += 2 * step)
        k_merge_segments<<<gm, NT, 0, st>>>(info, glob, rowpos, row_begin, n_rows, searched, sl,
                                            cnt_out, flags, s0, s0 + step, k, gamma);
  }
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "topk_screen");
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "topk_refine");
  if (rc) return rc;
  ChrTab tab;
  tab.n_chr = n_chr;
  for (int c = 0; c < 32; ++c) tab.cum[c] = c < n_chr ? chr_cum[c] : B;
  rc = wcx_refine_launch(ctx, Xr, S, Sp, tab, row_begin, n_rows, searched, sl, cnt_out, flags, perm,
                         k, d_out_idx, d_out_dist, glob);
  if (rc) return rc;
  rc = wcx_timer_end(ctx, "topk_refine");
  if (rc) return rc;
  rc = wcx_timer_end(ctx, "topk");
  if (rc) return rc;

  // overflowed rows (if any) are redone exactly
  ScreenGlobals hg;
  WCX_HIP(hipMemcpyAsync(&hg, glob, sizeof(hg), hipMemcpyDeviceToHost, st));
  WCX_HIP(hipStreamSynchronize(st));
  ctx->stage.clear();
  if (hg.n_overflow) {
    std::vector<unsigned int> hflags((size_t)n_rows);
    WCX_HIP(hipMemcpyAsync(hflags.data(), flags, (size_t)n_rows * 4, hipMemcpyDeviceToHost, st));
    WCX_HIP(hipStreamSynchronize(st));
    std::vector<TopkBlock> redo;
    for (const TopkBlock &eb : exact_blocks)
      for (int r = 0; r < eb.nrows; ++r)
        if (hflags[(size_t)(eb.row0 + r - row_begin)]) {
          TopkBlock one = eb;
          one.row0 = eb.row0 + r;
          one.nrows = 1;
          redo.push_back(one);
        }
    // NOTE: wcx_topk_exact_launch re-uses ctx->scratch; the screen scratch is dead by now.
    rc = wcx_topk_exact_launch(ctx, dXs, B, S, redo, row_begin, n_rows, k, d_out_idx, d_out_dist);
    if (rc) return rc;
    WCX_HIP(hipStreamSynchronize(st));
    unsigned long long fb = redo.size();
    WCX_HIP(hipMemcpyAsync(ctx->d_stats + 3, &fb, 8, hipMemcpyHostToDevice, st));
    WCX_HIP(hipStreamSynchronize(st));
  }
  return WCX_OK;
}
