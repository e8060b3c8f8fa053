// MFMA screen + exact fp64 refine for the reference-bin search (SURVEY.md §8a row a6;
// replaces newref_tools.py:255-278).  Results are identical to the exact kernel
// (newref_topk_exact.hip): the screen only decides WHICH candidates get the exact treatment.
//
// Pipeline (all on one stream):
//   k_col_stats    per-sample mean c_j and max |x - c_j|           (centring + fp16 scale)
//   k_transpose    Xs [S][B] -> Xr [B][S]                           (rows contiguous for refine)
//   k_screen_prep  a = 2^p (x - c) split into fp16 hi + lo, written in MFMA-fragment order;
//                  per row: |a~|^2, representation error norms      (rigorous error budget)
//   k_screen       -2 a~.b~ Gram tiles on the matrix cores: v_mfma_f32_32x32x16_f16 (hi plane;
//                  optionally hi.hi + hi.lo + lo.hi), fp32 accumulate; targets stay in registers as the
//                  B operand, candidates stream through LDS as the A operand; the epilogue
//                  adds the squared norms and appends (screen distance, index) to the target's
//                  shortlist whenever the pair could still be among the k nearest given the
//                  error budget; shortlists are cut back by a wave-level bisection select.
//   k_refine       exact sequential fp64 distance (newref_tools.py:260 arithmetic) of every
//                  shortlisted pair, sort by (distance, index), emit the first k.
//   rows whose shortlist overflowed (never seen on real data) are redone by the exact kernel.
//
// Error budget (t = |b~|^2 - 2 g~, screen distance d~ = t + |a~|^2, all in scaled units):
//   sqrt(d) in [sqrt(dh) - E, sqrt(dh) + E],  dh = |a~ - b~|^2,  E = e_a + e_max  (e = |a - a~|)
//   |d~ - dh| <= Q = 2 L_a L_max + 2 gamma N_a N_max + 2^-21 (N_a^2 + N_max^2)
//     (dropped lo.lo term; fp32 accumulation of 3*16*NK products, gamma = n 2^-23; fp32 roundings)
//   T = (sqrt(d~_(k) + Q) + E)^2 bounds the true k-th distance; a pair can be among the k nearest
//   only if d~ <= F = (sqrt(T) + E)^2 + Q.  Everything with d~ <= F is kept and refined exactly.
//
// Roofline: MFMA bound, 3 * 2*32*32*16 flop per instruction, dense f16 peak ~2.5 PFLOP/s.
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
__global__ __launch_bounds__(NT) void k_col_sum(const double *__restrict__ Xs, int64_t B,
                                                double *__restrict__ csum,
                                                double *__restrict__ ccnt) {
  const double *x = Xs + (int64_t)blockIdx.x * B;
  double s = 0.0, c = 0.0;
  for (int64_t i = (int64_t)blockIdx.y * NT + threadIdx.x; i < B; i += (int64_t)NT * CSPLIT) {
    const double v = x[i];
    if (fabs(v) < HUGE_VAL) { s += v; c += 1.0; }  // finite only
  }
  s = wcx::wave_sum(s);
  c = wcx::wave_sum(c);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&csum[blockIdx.x], s); atomicAdd(&ccnt[blockIdx.x], c); }
}

// cmean[j] = csum/ccnt; global max |x - mean| over finite entries
__global__ __launch_bounds__(NT) void k_col_stats(const double *__restrict__ Xs, int64_t B,
                                                  const double *__restrict__ csum,
                                                  const double *__restrict__ ccnt,
                                                  double *__restrict__ cmean,
                                                  ScreenGlobals *__restrict__ glob) {
  const double *x = Xs + (int64_t)blockIdx.x * B;
  const double cc = ccnt[blockIdx.x];
  const double m = cc > 0 ? csum[blockIdx.x] / cc : 0.0;
  if (blockIdx.y == 0 && threadIdx.x == 0) cmean[blockIdx.x] = m;
  double mx = 0.0;
  for (int64_t i = (int64_t)blockIdx.y * NT + threadIdx.x; i < B; i += (int64_t)NT * CSPLIT) {
    const double a = fabs(x[i] - m);
    if (a < HUGE_VAL && a > mx) mx = a;
  }
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) { const double o = wcx::shfl_xor_f64(mx, k); mx = o > mx ? o : mx; }
  if ((threadIdx.x & 63) == 0)
    atomicMax(&glob->amax_bits, (unsigned long long)__double_as_longlong(mx));
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

__global__ __launch_bounds__(NT) void k_row_norm(const double *__restrict__ Xr, int64_t B, int S,
                                                 int Sp, const double *__restrict__ cmean,
                                                 ChrTab chr, ScreenGlobals *__restrict__ glob,
                                                 unsigned int *__restrict__ rbits,
                                                 int *__restrict__ rchr) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int j = 0; j < S; ++j) {
    const float a = (float)(Xr[b * Sp + j] - cmean[j]);
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

__global__ __launch_bounds__(NT) void k_row_hist(const unsigned int *__restrict__ rbits,
                                                 const int *__restrict__ rchr, int64_t B,
                                                 const ScreenGlobals *__restrict__ glob,
                                                 int *__restrict__ rkey, int *__restrict__ cellcnt) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b >= B) return;
  const unsigned int umin = 0xffffffffu - glob->uinv;
  const unsigned int u = rbits[b];
  unsigned int bucket = NBUCKET - 1;
  if (u != 0xffffffffu && u >= umin && u - umin < NBUCKET - 1) bucket = u - umin;
  const int key = (int)bucket * 32 + rchr[b];
  rkey[b] = key;
  atomicAdd(&cellcnt[key], 1);
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

__global__ __launch_bounds__(NT) void k_scatter(const int *__restrict__ rkey, int64_t B,
                                                int *__restrict__ cursor, int *__restrict__ perm,
                                                int *__restrict__ rowpos) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b >= B) return;
  const int pos = atomicAdd(&cursor[rkey[b]], 1);
  perm[pos] = (int)b;
  rowpos[b] = pos;
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

template <int NK, int PL>
__global__ __launch_bounds__(NT) void k_screen_prep(
    const double *__restrict__ Xr, int64_t Bpad, int S, int Sp,
    const double *__restrict__ cmean, const int *__restrict__ perm,
    ScreenGlobals *__restrict__ glob, half8 *__restrict__ F, RowInfo *__restrict__ info) {
  // one thread per sweep POSITION p; the row it holds is perm[p] (-1 = padding)
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b >= Bpad) return;
  const int64_t row = perm[b];
  // scale 2^p so that the largest |a| lands in [8192, 16384)  (fp16 max 65504)
  const double amax = __longlong_as_double((long long)glob->amax_bits);
  int ex = 0;
  double scale = 1.0;
  if (amax > 0.0) {
    frexp(amax, &ex);            // amax = m 2^ex, m in [0.5,1)
    scale = ldexp(1.0, 14 - ex);
  }
  const int64_t tile = b >> 5;
  const int rl = (int)(b & 31);
  double n2 = 0.0, e2 = 0.0, l2 = 0.0;
  bool bad = false;
  for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      half8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = ks * 16 + h * 8 + e;
        double a = 0.0;
        if (row >= 0 && j < S) a = (Xr[row * Sp + j] - cmean[j]) * scale;
        const _Float16 hh = (_Float16)a;
        const double r1 = a - (double)hh;
        const _Float16 ll = PL == 2 ? (_Float16)r1 : (_Float16)0;
        const double res = r1 - (double)ll;
        const double at = (double)hh + (double)ll;
        n2 += at * at;
        e2 += res * res;
        l2 += (double)ll * (double)ll;
        bad |= !(fabs(a) < HUGE_VAL);
        hi[e] = hh;
        lo[e] = ll;
      }
      const int64_t base = ((tile * NK + ks) * PL) * 64 + rl + 32 * h;
      F[base] = hi;
      if (PL == 2) F[base + 64] = lo;
    }
  }
  RowInfo ri;
  if (row < 0) {           // padding position: +inf norm -> every screen test fails
    ri.nb = HUGE_VALF; ri.e = 0.f; ri.L = 0.f; ri.N = 0.f;
  } else if (bad) {        // NaN/inf row: as in the reference it is never admitted / finds nothing
    ri.nb = HUGE_VALF; ri.e = 0.f; ri.L = 0.f; ri.N = 0.f;
  } else {
    ri.nb = (float)n2;
    // representation error also covers the fp64 rounding of (x - c) * scale
    ri.e = up((float)(sqrt(e2) + 1e-15 * sqrt(n2)));
    ri.L = up((float)sqrt(l2));
    ri.N = up((float)sqrt(n2));
    atomicMax(&glob->e_max, __float_as_uint(ri.e));
    atomicMax(&glob->L_max, __float_as_uint(ri.L));
    atomicMax(&glob->N_max, __float_as_uint(ri.N));
  }
  info[b] = ri;
}

// ------------------------------------------------------------------------------------------
struct ScreenBlock {
  int64_t row0;
  int32_t nrows;  // <= TGT
  int32_t chr;    // chromosome index of the target rows
  int64_t cs, ce;
};

// Wave-level shortlist compaction of target `c` (0..31) of this wave.
// Returns the new threshold G (t-space) for that target; updates cnt in LDS.
__device__ __attribute__((noinline)) float compact_target(uint2 *__restrict__ sl_row, int *cnt_p, int k,
                                                float na, float E, float Q, float G_old,
                                                unsigned int *overflow_flag, bool exact) {
  const int lane = wcx::lane_id();
  // the entries were stored by (other lanes of) this wave: make them visible before re-reading
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  const int n = *cnt_p;
  unsigned int key[CAP / 64], idx[CAP / 64];
#pragma unroll
  for (int q = 0; q < CAP / 64; ++q) {
    const int e = q * 64 + lane;
    uint2 v = make_uint2(0xffffffffu, 0u);
    if (e < n) v = sl_row[e];
    key[q] = v.x;
    idx[q] = v.y;
  }
  float G = G_old;
  if (n >= k) {
    unsigned int prefix = 0;
    if (exact) {
      // k-th smallest key by bitwise bisection (ballot counts): largest v with #(key < v) < k
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned int trial = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int q = 0; q < CAP / 64; ++q) c += __popcll(__ballot(key[q] < trial));
        if (c < k) prefix = trial;
      }
    } else {
      // Cheap upper bound of the k-th smallest key: bisect a 64-entry strided sample to 16-bit
      // resolution at a rank a little above k/n, then VERIFY by an exact count (any value with
      // >= k keys at or below it is a valid bound); raise the sample rank until it holds.
      unsigned int smp = key[0];
#pragma unroll
      for (int q = 1; q < CAP / 64; ++q)
        if ((lane & (CAP / 64 - 1)) == q) smp = key[q];
      const bool smp_ok = ((lane & (CAP / 64 - 1)) * 64 + lane) < n;
      if (!smp_ok) smp = 0xffffffffu;
      int rs = (64 * k + n - 1) / n;
      rs += (rs >> 2) + 3;
      for (;;) {
        if (rs > 64) rs = 64;
        unsigned int p = 0;
        for (int bit = 31; bit >= 16; --bit) {
          const unsigned int trial = p | (1u << bit);
          if (__popcll(__ballot(smp < trial)) < rs) p = trial;
        }
        p |= 0xffffu;
        if (rs >= 64) p = 0xfffffffeu;   // everything valid
        int c = 0;
#pragma unroll
        for (int q = 0; q < CAP / 64; ++q) c += __popcll(__ballot(key[q] <= p));
        if (c >= k) { prefix = p; break; }
        rs += 8;
      }
    }
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
    if (keep) sl_row[pos] = make_uint2(key[q], idx[q]);
    base += __popcll(m);
  }
  if (base > LIM) {   // cannot make room: hand the row to the exact kernel
    if (lane == 0) { *overflow_flag = 1u; *cnt_p = 0; }
    return -HUGE_VALF;
  }
  if (lane == 0) *cnt_p = base;
  return G;
}

// NK = k-steps of 16 (K padded), PL = fp16 planes (2: hi+lo, three products; 1: hi only, one
// product -- larger shortlists, used when the targets' hi+lo fragments would not fit the
// register file), CTG = candidate sub-tiles of 32 rows per main-loop iteration.
template <int NK, int PL, int CTG>
__global__ __launch_bounds__(NT, (NK * PL <= 8 ? 3 : 2)) void k_screen(
    const half8 *__restrict__ F, const RowInfo *__restrict__ info,
    const ScreenGlobals *__restrict__ glob, int64_t B, int64_t Bpad,
    const int *__restrict__ perm, const int *__restrict__ rowpos,
    const unsigned int *__restrict__ gmask,
    const ScreenBlock *__restrict__ blocks, int k, int64_t row_begin,
    uint2 *__restrict__ sl, int *__restrict__ cnt_out, unsigned int *__restrict__ flags,
    float *__restrict__ g_state, int64_t gi_begin, int64_t gi_end, int first, int last,
    unsigned long long *__restrict__ stats, int dbg) {
  // The candidate sweep is cut into chunks [gi_begin,gi_end) of groups, one launch per chunk:
  // every workgroup of a launch streams the SAME few MB of candidate fragments, which therefore
  // come out of the XCD L2s instead of HBM/MALL.  Per-target state (threshold G, shortlist
  // count) lives in g_state/cnt_out between launches; the shortlists are in HBM anyway.
  constexpr int GR = CTG * 32;                      // candidate rows per iteration
  constexpr int TILE_H8 = CTG * NK * PL * 64;       // half8 elements per staged candidate group
  constexpr int NPT = TILE_H8 / NT;                 // 16-byte pieces per thread
  static_assert(TILE_H8 % NT == 0, "staging must divide evenly over the workgroup");
  constexpr int NOUT = CTG * 16;                    // screen outputs per lane per iteration
  extern __shared__ __align__(16) unsigned char smem[];
  half8 *sbuf = reinterpret_cast<half8 *>(smem);                       // [2][TILE_H8]
  float *snb = reinterpret_cast<float *>(smem + 2 * TILE_H8 * 16);     // [2][GR]
  int *cnt = reinterpret_cast<int *>(smem + 2 * TILE_H8 * 16 + 2 * GR * 4);  // [TGT]
  int *sperm = cnt + TGT;                                              // [2][GR] rows of the group

  const ScreenBlock blk = blocks[blockIdx.x];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tl = wave * 32 + (lane & 31);        // local target of this lane
  const int hf = lane >> 5;
  const int64_t own = blk.ce - blk.cs;
  const bool tvalid = tl < blk.nrows;
  const int64_t trow = tvalid ? blk.row0 + tl : blk.row0;
  const int64_t srow = trow - row_begin;

  if (tid < TGT) cnt[tid] = (first || tid >= blk.nrows) ? 0 : cnt_out[blk.row0 + tid - row_begin];

  // target operand (B operand of the MFMA) stays in registers for the whole sweep
  half8 th[NK], tlo[PL == 2 ? NK : 1];
  {
    const int64_t tpos = rowpos[trow];        // sweep position of the target row
    const int64_t ttile = tpos >> 5;
    const int trl = (int)(tpos & 31);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int64_t base = ((ttile * NK + ks) * PL) * 64 + trl + 32 * hf;
      th[ks] = F[base];
      if (PL == 2) tlo[ks] = F[base + 64];
    }
  }
  const RowInfo ti = info[rowpos[trow]];
  const float e_max = __uint_as_float(glob->e_max), L_max = __uint_as_float(glob->L_max),
              N_max = __uint_as_float(glob->N_max);
  const float na = ti.nb;
  const float E = up(ti.e + e_max);
  const float gamma = (float)((PL == 2 ? 3 : 1) * 16 * NK + 8) * 1.1920929e-7f;   // n * 2^-23
  // Q also covers the threshold folded into the accumulator (see the main loop) and the
  // reconstruction t = G - 2 acc:  2.2 gamma (N_a + N_max)^2
  const float nsum = ti.N + N_max;
  const float Q = up(2.f * ti.L * L_max + 2.f * gamma * ti.N * N_max +
                     4.8e-7f * (ti.N * ti.N + N_max * N_max) + 2.2f * gamma * nsum * nsum);
  constexpr float G_INIT = 3.0e38f;   // "no threshold yet" (finite on purpose)
  float G = tvalid ? (first ? G_INIT : g_state[srow]) : -HUGE_VALF;
  uint2 *sl_row = sl + srow * (int64_t)CAP;
  unsigned long long n_compact = 0;

  // Visit list of this launch's chunk, built once per workgroup in LDS: groups holding only
  // own-chromosome rows are skipped (gmask = chromosomes present per 64 rows); bit 31 marks groups
  // that also contain own-chromosome rows.  (Reading gmask from global memory inside the loop put
  // two dependent scalar-load latencies on every iteration: 7 ms of a 20 ms sweep.)
  const unsigned int blkbit = 1u << blk.chr;
  int *glist = sperm + 2 * GR;                 // [gi_end - gi_begin + 1]
  __shared__ int s_nlist;
  if (wave == 0) {
    int count = 0;
    for (int64_t g0 = gi_begin; g0 < gi_end; g0 += 64) {
      const int64_t g = g0 + lane;
      unsigned int m = blkbit;
      if (g < gi_end) m = gmask[(g * GR) >> 6];
      const bool keep = (g < gi_end) && m != blkbit;
      const unsigned long long bal = __ballot(keep);
      if (keep) glist[count + __popcll(bal & ((1ull << lane) - 1ull))] =
          (int)g | ((m & blkbit) ? (int)0x80000000 : 0);
      count += __popcll(bal);
    }
    if (lane == 0) s_nlist = count;
  }
  half8 pre[NPT];
  float pre_nb = 0.f;
  int pre_row = -1;
  auto fetch = [&](int64_t gix) {
    const half8 *src = F + gix * (int64_t)TILE_H8;
#pragma unroll
    for (int p = 0; p < NPT; ++p) pre[p] = src[p * NT + tid];
    if (tid < GR) { pre_nb = info[gix * GR + tid].nb; pre_row = perm[gix * GR + tid]; }
  };
  int buf = 0;
  __syncthreads();
  const int n_list = s_nlist;
  int cur = n_list > 0 ? glist[0] : 0;
  if (n_list > 0) fetch(cur & 0x7fffffff);
  bool fast = false;
  for (int j = 0; j < n_list; ++j) {
    const int nxt = (j + 1 < n_list) ? glist[j + 1] : 0;   // LDS read, used after the barrier
    half8 *sb = sbuf + buf * TILE_H8;
    float *nbb = snb + buf * GR;
    int *prow = sperm + buf * GR;
#pragma unroll
    for (int p = 0; p < NPT; ++p) sb[p * NT + tid] = pre[p];
    if (tid < GR) { nbb[tid] = pre_nb; prow[tid] = pre_row; }
    __syncthreads();
    const bool mixed = cur < 0;                  // some own-chromosome rows in this group
    if (j + 1 < n_list && !(dbg & 8)) fetch(nxt & 0x7fffffff);

    // Once every target of the wave has a finite threshold the test is folded into the MFMA:
    // acc starts at (G - |b~|^2)/2, so after the products acc = g~ - (|b~|^2 - G)/2 and the pair
    // passes (t = |b~|^2 - 2 g~ <= G) iff acc >= 0 -- one sign bit per output, no extra VALU.
    if (!fast) fast = __all(G < 1.0e37f);        // G only ever decreases
    f32x16 acc[CTG];
    unsigned int pmask = 0;
    auto products = [&]() {
      if (!(dbg & 2))
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        half8 ch[CTG], cl[PL == 2 ? CTG : 1];
#pragma unroll
        for (int sub = 0; sub < CTG; ++sub) {
          ch[sub] = sb[((sub * NK + ks) * PL + 0) * 64 + lane];
          if (PL == 2) cl[sub] = sb[((sub * NK + ks) * PL + 1) * 64 + lane];
        }
#pragma unroll
        for (int sub = 0; sub < CTG; ++sub)
          acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[sub], th[ks], acc[sub], 0, 0, 0);
        if (PL == 2) {
#pragma unroll
          for (int sub = 0; sub < CTG; ++sub)
            acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[sub], tlo[ks], acc[sub], 0, 0, 0);
#pragma unroll
          for (int sub = 0; sub < CTG; ++sub)
            acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl[sub], th[ks], acc[sub], 0, 0, 0);
        }
      }
    };
    // C[row = candidate][col = target]; output rr = sub*16 + r is candidate row
    // loc(rr) = sub*32 + 8*(r>>2) + 4*(lane>>5) + (r&3) of this group.  Bit (31-rr) of pmask.
    if (fast) {
      const float halfG = 0.5f * G;
#pragma unroll
      for (int sub = 0; sub < CTG; ++sub)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 nb4 = *reinterpret_cast<const float4 *>(&nbb[sub * 32 + 8 * j + 4 * hf]);
          acc[sub][4 * j + 0] = fmaf(-0.5f, nb4.x, halfG);
          acc[sub][4 * j + 1] = fmaf(-0.5f, nb4.y, halfG);
          acc[sub][4 * j + 2] = fmaf(-0.5f, nb4.z, halfG);
          acc[sub][4 * j + 3] = fmaf(-0.5f, nb4.w, halfG);
        }
      products();
      unsigned int neg = 0;
#pragma unroll
      for (int sub = 0; sub < CTG; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) neg = (neg << 1) | (__float_as_uint(acc[sub][r]) >> 31);
      pmask = (~neg) << (32 - NOUT);
    } else {
      asm volatile("; slow path" ::: "memory");
#pragma unroll
      for (int sub = 0; sub < CTG; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[sub][r] = 0.f;
      products();
#pragma unroll
      for (int sub = 0; sub < CTG; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float nbr = nbb[sub * 32 + 8 * (r >> 2) + 4 * hf + (r & 3)];
          const bool ok = fmaf(-2.f, acc[sub][r], nbr) <= G;
          pmask |= ok ? (0x80000000u >> (sub * 16 + r)) : 0u;
        }
    }
    if (mixed) {   // rare: mask the own-chromosome rows of a mixed group
      asm volatile("; mixed group" ::: "memory");   // keep this a branch (no if-conversion)
      const int cs32 = (int)blk.cs, ce32 = (int)blk.ce;
#pragma unroll
      for (int rr = 0; rr < NOUT; ++rr) {
        const int loc = (rr >> 4) * 32 + 8 * ((rr >> 2) & 3) + 4 * hf + (rr & 3);
        const int g = prow[loc];
        if (g >= cs32 && g < ce32) pmask &= ~(0x80000000u >> rr);
      }
    }
    if (dbg & 1) pmask = 0;
    if (__any(pmask != 0u)) {
      int pos = 0;
      if (pmask) pos = atomicAdd(&cnt[tl], __popc(pmask));
#pragma unroll
      for (int rr = 0; rr < NOUT; ++rr) {
        if (pmask & (0x80000000u >> rr)) {
          const int loc = (rr >> 4) * 32 + 8 * ((rr >> 2) & 3) + 4 * hf + (rr & 3);
          const int64_t g = prow[loc];
          const float av = acc[rr >> 4][rr & 15];
          const float t = fast ? fmaf(-2.f, av, G) : fmaf(-2.f, av, nbb[loc]);
          if (pos < CAP)
            sl_row[pos] = make_uint2(f32_key(t), (unsigned int)(g < blk.cs ? g : g - own));
          ++pos;
        }
      }
      // shortlist maintenance: wave-private (this wave's 32 targets); counts only change here
        {
        const int my_cnt = cnt[tl];
        unsigned long long need = __ballot(tvalid && my_cnt > LIM) & 0xffffffffull;
        while (need) {
          const int c = __ffsll((long long)need) - 1;
          need &= need - 1;
          const int64_t crow_s = (blk.row0 + wave * 32 + c) - row_begin;
          const float na_c = __shfl(na, c, 64), E_c = __shfl(E, c, 64), Q_c = __shfl(Q, c, 64),
                      G_c = __shfl(G, c, 64);
          const float Gn = compact_target(sl + crow_s * (int64_t)CAP, &cnt[wave * 32 + c], k, na_c,
                                          E_c, Q_c, G_c, &flags[crow_s], false);
          if ((lane & 31) == c) G = Gn;
          ++n_compact;
        }
      }
    }
    buf ^= 1;
    cur = nxt;
  }
  if (last) {
    // final cut of every target's shortlist with its final threshold
    for (int c = 0; c < 32; ++c) {
      if (wave * 32 + c >= blk.nrows) break;
      const int64_t crow_s = (blk.row0 + wave * 32 + c) - row_begin;
      const float na_c = __shfl(na, c, 64), E_c = __shfl(E, c, 64), Q_c = __shfl(Q, c, 64),
                  G_c = __shfl(G, c, 64);
      (void)compact_target(sl + crow_s * (int64_t)CAP, &cnt[wave * 32 + c], k, na_c, E_c, Q_c,
                           G_c, &flags[crow_s], true);
    }
  } else if (tvalid && hf == 0) {
    g_state[srow] = G;
  }
  if (tvalid && hf == 0) cnt_out[srow] = cnt[tl];
  if (lane == 0 && stats) atomicAdd(&stats[2], n_compact);
}

__global__ void k_mark(unsigned char *searched, const ScreenBlock *__restrict__ blocks,
                       int64_t row_begin) {
  const ScreenBlock b = blocks[blockIdx.x];
  if ((int)threadIdx.x < b.nrows) searched[b.row0 - row_begin + threadIdx.x] = 1;
}

}  // namespace

// Host side --------------------------------------------------------------------------------
int wcx_debug_value = 0;   // diagnostics only (wcx_debug_flags): ablation switches for profiling
bool wcx_screen_supported(int64_t B, int S, int k) {
  return S <= 512 && k <= 512 && k <= LIM && B >= 2048;
}

int wcx_topk_screen_launch(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                           const int64_t *chr_cum, int n_chr,
                           const std::vector<TopkBlock> &exact_blocks, int64_t row_begin,
                           int64_t n_rows, int k, int32_t *d_out_idx, double *d_out_dist) {
  if (exact_blocks.empty()) return WCX_OK;
  // Default: hi plane only (one fp16 product).  Its representation error (2^-11 relative) only
  // widens the shortlists by a few dozen entries, while a third of the MFMA work and half the
  // LDS traffic of the hi+lo form (three products; kept behind debug flag 32, S <= 128 only) is
  // enough -- measured 24.5 ms vs 33.0 ms at 15 kb / S=100, refine +1.2 ms.
  // K is padded so that the staging divides evenly over the workgroup.
  const bool two_planes = (wcx_debug_value & 32) && S <= 128;
  const int PL = two_planes ? 2 : 1;
  int NK = (S + 15) / 16;
  if (!two_planes) {
    if (NK <= 8) NK = (NK + 1) & ~1;
    else NK = NK <= 16 ? 16 : (NK <= 24 ? 24 : 32);
  }
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
  // scratch layout
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_glob = carve(sizeof(ScreenGlobals));
  const size_t o_mean = carve((size_t)S * 8 * 3);   // mean | sum | count
  const int Sp = (S + 3) & ~3;
  const size_t o_xr = carve((size_t)B * Sp * 8 + 256);   // + slack: refine loads whole 128-B chunks
  const size_t o_F = carve((size_t)Bpad * NK * PL * 32);  // Bpad/32 tiles * NK * PL planes * 1 KiB
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
  const size_t o_sl = carve((size_t)n_rows * CAP * 8);
  const size_t o_cnt = carve((size_t)n_rows * 4);
  const size_t o_gst = carve((size_t)n_rows * 4);
  const size_t o_flag = carve((size_t)n_rows * 4);
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
  WCX_HIP(hipMemsetAsync(cnt_out, 0, (size_t)n_rows * 4, st));
  WCX_HIP(hipMemsetAsync(flags, 0, (size_t)n_rows * 4, st));
  WCX_HIP(hipMemsetAsync(searched, 0, (size_t)n_rows, st));
  WCX_HIP(hipMemsetAsync(ctx->d_stats, 0, 32, st));
  rc = wcx_upload_small(ctx, d_blocks, blocks.data(), blocks.size() * sizeof(ScreenBlock));
  if (rc) return rc;
  k_mark<<<(unsigned)blocks.size(), TGT, 0, st>>>(searched, d_blocks, row_begin);

  rc = wcx_timer_begin(ctx, "topk");
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "topk_prep");
  if (rc) return rc;
  WCX_HIP(hipMemsetAsync(cmean + S, 0, (size_t)S * 16, st));
  k_col_sum<<<dim3((unsigned)S, CSPLIT), NT, 0, st>>>(dXs, B, cmean + S, cmean + 2 * S);
  k_col_stats<<<dim3((unsigned)S, CSPLIT), NT, 0, st>>>(dXs, B, cmean + S, cmean + 2 * S, cmean, glob);
  k_transpose<<<dim3((unsigned)((B + 31) / 32), (unsigned)((Sp + 31) / 32)), 256, 0, st>>>(dXs, B, S, Sp, Xr);
  {
    ChrTab tab0;
    tab0.n_chr = n_chr;
    for (int c = 0; c < 32; ++c) tab0.cum[c] = c < n_chr ? chr_cum[c] : B;
    const unsigned gb = (unsigned)((B + NT - 1) / NT);
    WCX_HIP(hipMemsetAsync(cellcnt, 0, (size_t)NCELL * 4, st));
    WCX_HIP(hipMemsetAsync(perm, 0xff, (size_t)Bpad * 4, st));
    k_row_norm<<<gb, NT, 0, st>>>(Xr, B, S, Sp, cmean, tab0, glob, rbits, rchr);
    k_row_hist<<<gb, NT, 0, st>>>(rbits, rchr, B, glob, rkey, cellcnt);
    k_scan_cells<<<1, 1024, 0, st>>>(cellcnt, cursor);
    k_scatter<<<gb, NT, 0, st>>>(rkey, B, cursor, perm, rowpos);
    k_group_mask<<<(unsigned)((n_groups + NT - 1) / NT), NT, 0, st>>>(perm, rchr, n_groups, gmask);
  }
  const unsigned gprep = (unsigned)((Bpad + NT - 1) / NT);
  const int GRr = CTG * 32;
  // candidate chunk per launch: ~3 MB of fragments (fits the 4 MB XCD L2)
  const int64_t n_iter_groups = Bpad / GRr;
  const int64_t group_bytes = (int64_t)GRr * NK * PL * 32;
  int64_t chunk_groups = (3 << 20) / group_bytes;
  if (chunk_groups < 16) chunk_groups = 16;
  if (chunk_groups > 4096) chunk_groups = 4096;
  const size_t lds = 2 * (size_t)(CTG * NK * PL * 64) * 16 + 2 * GRr * 4 + TGT * 4 + 2 * GRr * 4 +
                     (size_t)(chunk_groups + 64) * 4;   // + the chunk's visit list
#define WCX_SCREEN_CASE(N, P, G)                                                               \
  {                                                                                            \
    k_screen_prep<N, P><<<gprep, NT, 0, st>>>(Xr, Bpad, S, Sp, cmean, perm, glob, F, info);     \
    rc = wcx_timer_end(ctx, "topk_prep");                                                      \
    if (rc) return rc;                                                                         \
    rc = wcx_timer_begin(ctx, "topk_screen");                                                  \
    if (rc) return rc;                                                                         \
    WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_screen<N, P, G>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
    for (int64_t g0 = 0; g0 < n_iter_groups; g0 += chunk_groups) {                             \
      const int64_t g1 = g0 + chunk_groups < n_iter_groups ? g0 + chunk_groups : n_iter_groups; \
      k_screen<N, P, G><<<(unsigned)blocks.size(), NT, lds, st>>>(                             \
          F, info, glob, B, Bpad, perm, rowpos, gmask, d_blocks, k, row_begin, sl, cnt_out,    \
          flags, g_state, g0, g1, g0 == 0, g1 == n_iter_groups, ctx->d_stats,                  \
          wcx_debug_value);                                                                    \
    }                                                                                          \
  }
  if (two_planes) {
    switch (NK) {
      case 1: WCX_SCREEN_CASE(1, 2, 2) break;
      case 2: WCX_SCREEN_CASE(2, 2, 2) break;
      case 3: WCX_SCREEN_CASE(3, 2, 2) break;
      case 4: WCX_SCREEN_CASE(4, 2, 2) break;
      case 5: WCX_SCREEN_CASE(5, 2, 2) break;
      case 6: WCX_SCREEN_CASE(6, 2, 2) break;
      case 7: WCX_SCREEN_CASE(7, 2, 2) break;
      default: WCX_SCREEN_CASE(8, 2, 2) break;
    }
  } else {
    switch (NK) {
      case 2: WCX_SCREEN_CASE(2, 1, 2) break;
      case 4: WCX_SCREEN_CASE(4, 1, 2) break;
      case 6: WCX_SCREEN_CASE(6, 1, 2) break;
      case 8: WCX_SCREEN_CASE(8, 1, 2) break;
      case 16: WCX_SCREEN_CASE(16, 1, 2) break;
      case 24: WCX_SCREEN_CASE(24, 1, 1) break;
      default: WCX_SCREEN_CASE(32, 1, 1) break;
    }
  }
#undef WCX_SCREEN_CASE
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "topk_screen");
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "topk_refine");
  if (rc) return rc;
  ChrTab tab;
  tab.n_chr = n_chr;
  for (int c = 0; c < 32; ++c) tab.cum[c] = c < n_chr ? chr_cum[c] : B;
  rc = wcx_refine_launch(ctx, Xr, S, Sp, tab, row_begin, n_rows, searched, sl, cnt_out, flags, k,
                         d_out_idx, d_out_dist, glob);
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
