// MFMA screen + exact fp64 refine for the reference-bin search (SURVEY.md §8a row a6;
// replaces newref_tools.py:255-278).  Results are identical to the exact kernel
// (newref_topk_exact.hip): the screen only decides WHICH candidates get the exact treatment.
//
// Pipeline (all on one stream):
//   k_col_stats    per-sample mean c_j and max |x - c_j|           (centring + fp16 scale)
//   k_transpose    Xs [S][B] -> Xr [B][S]                           (rows contiguous for refine)
//   k_transpose_norm / k_row_hist / k_scan_cells / k_scatter / k_group_mask
//                  best-first sweep order: counting sort by (norm bucket, chromosome)
//   k_screen_prep  a~ = fp16(2^p (x - c)) in MFMA-fragment order + four augmented k-columns that
//                  carry |a~|^2; per row: representation-error norm      (rigorous error budget)
//   k_screen       (screen_kernel.h) -2 a~.b~ Gram tiles on the matrix cores:
//                  v_mfma_f32_32x32x16_f16, fp32 accumulate; targets stay in registers as the B
//                  operands (their augmented columns carry the current threshold), candidates
//                  stream through LDS as the A operand; the screen test is the SIGN BIT of the
//                  accumulator; passing pairs are appended to the target's shortlist, which is cut
//                  back by a wave-level bitwise bisection select whenever it nears capacity.
//                  The sweep starts with a SAMPLED PRE-PASS (every SF-th candidate group): the
//                  r-th smallest value of the sample, r ~ k/SF + 6 sigma, is an estimate of the
//                  final threshold that is tight enough to cut the appends of the main pass to
//                  ~2 k per row and loose enough to hold the k nearest with overwhelming
//                  probability; the final cut PROVES it (k values below it, their filter bound not
//                  above it) or hands the row to the exact kernel -- results never depend on it.
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
#include <cmath>
#include <cstdlib>

#include "screen_kernel.h"
#include "screen_common.h"
#include "screen_hub1.h"

namespace {

// ------------------------------------------------------------------------------------------
constexpr int CSPLIT = 16;   // workgroups per sample column

// per-sample mean over finite entries: deterministic partial sums (no floating-point atomics)
__device__ __forceinline__ unsigned long long dord(double x) {   // order-preserving image
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dord_inv(unsigned long long k) {
  return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

// ONE sweep of X: per-sample sum, count, min and max over finite entries (max |x - mean| follows
// from min and max: it is attained at one of them, with the same rounding as fabs(x - mean)).
// The sums are DETERMINISTIC: every workgroup writes its partial (its waves added in wave order) to a
// table and k_col_stats adds the CSPLIT partials of a column in index order -- no floating-point atomics.
// The row-sharded symmetric sweep depends on it: every rank must derive bit-identical means, hence
// fragments, sweep order and thresholds (the records it exchanges name sweep positions).
__global__ __launch_bounds__(NT) void k_col_sum(const double *__restrict__ Xs, int64_t B,
                                                double *__restrict__ psum,      // [S][CSPLIT]
                                                double *__restrict__ pcnt,      // [S][CSPLIT]
                                                unsigned long long *__restrict__ cmin,
                                                unsigned long long *__restrict__ cmax) {
  __shared__ double ws[NT / 64], wc[NT / 64];
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
  if ((threadIdx.x & 63) == 0) {
    ws[threadIdx.x >> 6] = s;
    wc[threadIdx.x >> 6] = c;
    if (c > 0.0) {
      atomicMin(&cmin[blockIdx.x], dord(mn));       // (min / max: order-independent)
      atomicMax(&cmax[blockIdx.x], dord(mx));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tc = 0.0;
    for (int w = 0; w < NT / 64; ++w) { ts += ws[w]; tc += wc[w]; }
    psum[(int64_t)blockIdx.x * CSPLIT + blockIdx.y] = ts;
    pcnt[(int64_t)blockIdx.x * CSPLIT + blockIdx.y] = tc;
  }
}

// cmean[j] = sum / count (partials added in index order); global max |x - mean| over finite entries
__global__ void k_col_stats(int S, const double *__restrict__ psum, const double *__restrict__ pcnt,
                            const unsigned long long *__restrict__ cmin,
                            const unsigned long long *__restrict__ cmax,
                            double *__restrict__ cmean, ScreenGlobals *__restrict__ glob) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S) return;
  double cs = 0.0, cc = 0.0;
  for (int p = 0; p < CSPLIT; ++p) { cs += psum[(int64_t)j * CSPLIT + p]; cc += pcnt[(int64_t)j * CSPLIT + p]; }
  const double m = cc > 0 ? cs / cc : 0.0;
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

constexpr int HUB_BINS = 65536;          // histogram of (float bits of |a|^2) >> 16: 0.8 % steps in norm
// The row-major copy Xr and the rows' centred norms in one pass over Xs (round 6; before: k_transpose, then
// k_row_norm reading the matrix again): a workgroup owns 32 bins and walks ALL the sample
// blocks of 32, so that a bin's centred norm builds up in registers in a fixed order (thread ty takes the
// samples j = 8 i + ty of every block, the eight partial sums meet in LDS) while the tiles are transposed
// through a double-buffered LDS tile (one barrier per block).  Saves the second read of the matrix
// (0.25 / 0.2 ms per pass at 15 kb x 500 / 250); same outputs as the two kernels.
__global__ __launch_bounds__(256) void k_transpose_norm(const double *__restrict__ Xs, int64_t B, int S,
                                                        int Sp, double *__restrict__ Xr,
                                                        const double *__restrict__ cmean, ChrTab chr,
                                                        ScreenGlobals *__restrict__ glob,
                                                        unsigned int *__restrict__ rbits,
                                                        int *__restrict__ rchr,
                                                        unsigned int *__restrict__ rfine,
                                                        int *__restrict__ hubhist) {
  __shared__ double tile[2][32][33];
  __shared__ float part[8][32];
  const int64_t b0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  const int64_t bi = b0 + tx;
  float acc = 0.f;
  const int nblk = (Sp + 31) / 32;
  for (int jb = 0; jb < nblk; ++jb) {
    const int j0 = jb * 32, buf = jb & 1;
    double v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + ty + 8 * i;
      v[i] = (j < S && bi < B) ? Xs[(int64_t)j * B + bi] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + ty + 8 * i;
      if (j < S) { const float a = (float)(v[i] - cmean[j]); acc += a * a; }
      tile[buf][ty + 8 * i][tx] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t b = b0 + ty + 8 * i;
      const int j = j0 + tx;
      if (b < B && j < Sp) Xr[b * Sp + j] = tile[buf][tx][ty + 8 * i];
    }
  }
  part[ty][tx] = acc;
  __syncthreads();
  if (ty != 0 || bi >= B) return;
  const float s = ((part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx])) +
                  ((part[4][tx] + part[5][tx]) + (part[6][tx] + part[7][tx]));
  int c = 0;
  while (c < chr.n_chr - 1 && bi >= chr.cum[c]) ++c;
  rchr[bi] = c;
  unsigned int u = 0xffffffffu;                     // non-finite rows go last
  if (s < HUGE_VALF) {
    u = __float_as_uint(s) >> 20;
    atomicMax(&glob->uinv, 0xffffffffu - u);
  }
  rbits[bi] = u;
  if (rfine) {
    const unsigned int f = s < HUGE_VALF ? __float_as_uint(s) >> 16 : 0xffffffffu;
    rfine[bi] = f;
    if (f < (unsigned int)HUB_BINS) atomicAdd(&hubhist[f], 1);
  }
}

// Hub region = the rows below a norm quantile: the smallest key q with at least `want` rows at or below
// it (the whole last histogram bin is taken: a few rows more than asked for).
__global__ __launch_bounds__(1024) void k_hub_cut(const int *__restrict__ hubhist, int want,
                                                  ScreenGlobals *__restrict__ glob) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  constexpr int PER = HUB_BINS / 1024;
  int s = 0;
  for (int i = 0; i < PER; ++i) s += hubhist[t * PER + i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  const int before = part[t] - s;
  if (before < want && part[t] >= want) {           // the quantile lies in this thread's bins
    int run = before;
    for (int i = 0; i < PER; ++i) {
      run += hubhist[t * PER + i];
      if (run >= want) { glob->hub_key = (unsigned int)(t * PER + i); break; }
    }
  }
  if (t == 1023 && part[t] < want) glob->hub_key = (unsigned int)(HUB_BINS - 1);   // (fewer finite rows)
}

// Cell histogram with workgroup-private LDS counters (the rows pile into a few dozen cells: global
// atomics straight from every row serialise).
// Rows b with b % SF == 0 (SF > 0) form the SAMPLE region at the head of the sweep (cells
// [0, NCELL): one cell per chromosome), all other rows the main region (cells [NCELL, 2 NCELL),
// best-first order).
__global__ __launch_bounds__(NT) void k_row_hist(const unsigned int *__restrict__ rbits,
                                                 const int *__restrict__ rchr, int64_t B, int SF,
                                                 int fair_sample,
                                                 const ScreenGlobals *__restrict__ glob,
                                                 int *__restrict__ rkey, int *__restrict__ cellcnt,
                                                 const unsigned int *__restrict__ gate = nullptr,
                                                 const unsigned int *__restrict__ rfine = nullptr) {
  __shared__ int lh[2 * NCELL];
  if (gate && !*gate) return;
  for (int i = threadIdx.x; i < 2 * NCELL; i += NT) lh[i] = 0;
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b < B) {
    const unsigned int umin = 0xffffffffu - glob->uinv;
    const unsigned int u = rbits[b];
    unsigned int bucket = NBUCKET - 1;
    if (u != 0xffffffffu && u >= umin && u - umin < NBUCKET - 1) bucket = u - umin;
    // (rfine: the head region is the HUB region -- the rows at or below the norm quantile -- instead of a
    //  sample: screen_hub1.h)
    const bool sample = rfine ? rfine[b] <= glob->hub_key : (SF > 0 && (b % SF) == 0);
    // sample rows of a SEGMENTED sweep: by chromosome only (bucket 0), i.e. in random order with
    // respect to the norm, so that the split of the sample's groups over the candidate segments
    // gives every segment a fair subsample (best-first order would hand the few groups that hold
    // most of the nearest rows to one or two segments, whose estimates then come out too tight)
    const int key = sample ? (fair_sample ? 0 : (int)bucket * 32) + rchr[b]
                           : NCELL + (int)bucket * 32 + rchr[b];
    rkey[b] = key;
    atomicAdd(&lh[key], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * NCELL; i += NT)
    if (lh[i]) atomicAdd(&cellcnt[i], lh[i]);
}

// Exclusive scan of the cell counts -> first sweep position of every cell; the main region starts
// `pad` positions later so that the sample region ends on a group boundary.
// hub_glob != nullptr: the head region's size is only known here (the hub rows): pad is computed from it
// and the region's length in 32-row tiles goes to hub_glob->n_hub_tiles.
__global__ __launch_bounds__(1024) void k_scan_cells(const int *__restrict__ cellcnt,
                                                     int *__restrict__ cursor, int pad,
                                                     const unsigned int *__restrict__ gate = nullptr,
                                                     ScreenGlobals *__restrict__ hub_glob = nullptr) {
  __shared__ int part[1024];
  if (gate && !*gate) return;
  const int t = threadIdx.x;
  constexpr int PER = 2 * NCELL / 1024;
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
  if (hub_glob) {
    const int total0 = part[NCELL / PER - 1];            // rows of the head region
    pad = (CT - total0 % CT) % CT;
    if (t == 0) hub_glob->n_hub_tiles = (unsigned int)((total0 + pad) >> 5);
  }
  const int base = part[t] - s + (t * PER >= NCELL ? pad : 0);
#pragma unroll
  for (int i = 0; i < PER; ++i) cursor[t * PER + i] = base + loc[i];
}

// Rows -> sweep positions: each workgroup reserves one range per cell it touches.
__global__ __launch_bounds__(NT) void k_scatter(const int *__restrict__ rkey, int64_t B,
                                                int *__restrict__ cursor, int *__restrict__ perm,
                                                int *__restrict__ rowpos,
                                                const unsigned int *__restrict__ gate = nullptr) {
  __shared__ int lh[2 * NCELL];
  if (gate && !*gate) return;
  for (int i = threadIdx.x; i < 2 * NCELL; i += NT) lh[i] = 0;
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  int key = 0, local = 0;
  if (b < B) { key = rkey[b]; local = atomicAdd(&lh[key], 1); }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * NCELL; i += NT)
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
                                                   unsigned int *__restrict__ gmask,
                                                   const unsigned int *__restrict__ gate = nullptr) {
  const int64_t g = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (g >= n_groups || (gate && !*gate)) return;
  unsigned int m = 0;
  for (int i = 0; i < CT; ++i) {
    const int row = perm[g * CT + i];
    if (row >= 0) m |= 1u << rchr[row];
  }
  gmask[g] = m;
}

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
    ScreenGlobals *__restrict__ glob, half8 *__restrict__ F, RowInfo *__restrict__ info,
    const unsigned int *__restrict__ gate = nullptr) {
  // One workgroup per 32-row tile, a wave per row at a time: the 64 lanes read the row's 64 groups of
  // eight samples -- 4 KB contiguous -- convert them, and put the half8 pieces into the tile's fragment
  // block in LDS (NK KB), which then leaves in one contiguous stream.  (Round 6; before, a thread walked
  // its own row twice with 32-byte loads -- 64 scattered rows per wave, 1.5 TB/s.)  The row's |a~|^2 and
  // representation error are summed over the lanes by a butterfly: a fixed order, the same on every rank.
  __shared__ half8 frag[NK * 64];
  __shared__ unsigned int wmax[2][NT / 64];     // per wave: largest e / N of its rows (one atomic pair per tile)
  if (gate && !*gate) return;
  unsigned int my_e = 0u, my_n = 0u;
  const int64_t tile = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // scale 2^p so that the largest |a| lands in [1024, 2048): |a~|^2 <= 508 * 2^22 < 2^31.1 keeps
  // nb/65536 and any threshold/65536 inside the fp16 range; with more than 508 samples (NK > 32) one
  // binade lower: |a~|^2 <= 1020 * 2^20 < 2^30
  const double amax = __longlong_as_double((long long)glob->amax_bits);
  int ex = 0;
  double scale = 1.0;
  if (amax > 0.0) {
    frexp(amax, &ex);            // amax = m 2^ex, m in [0.5,1)
    scale = ldexp(1.0, (NK > 32 ? 10 : 11) - ex);
  }
  constexpr int NG = 2 * NK;                    // groups of eight k-columns per row
  constexpr int ITER = (NG + 63) / 64;
  // a lane keeps the same groups for every row: their column means stay in registers
  double cm[ITER][8];
#pragma unroll
  for (int it = 0; it < ITER; ++it)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = (it * 64 + lane) * 8 + e;
      cm[it][e] = j < S ? cmean[j] : 0.0;
    }
  for (int rl = wave; rl < 32; rl += NT / 64) {
    const int64_t b = tile * 32 + rl;
    if (b >= Bpad) continue;                     // wave-uniform
    const int64_t row = perm[b];
    // (rows are 32-byte aligned, Sp % 4 == 0, and the padding columns [S, Sp) are zero)
    const double4 *xrow = reinterpret_cast<const double4 *>(Xr + (row >= 0 ? row : 0) * Sp);
    double xv[ITER][8];
    bool bad = false;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int j0 = (it * 64 + lane) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[it][e] = 0.0;
      if (row >= 0 && it * 64 + lane < NG && j0 < Sp) {
        const double4 q0 = xrow[j0 / 4];
        xv[it][0] = q0.x; xv[it][1] = q0.y; xv[it][2] = q0.z; xv[it][3] = q0.w;
        if (j0 + 4 < Sp) {
          const double4 q1 = xrow[j0 / 4 + 1];
          xv[it][4] = q1.x; xv[it][5] = q1.y; xv[it][6] = q1.z; xv[it][7] = q1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) bad = bad || !(fabs(xv[it][e]) < HUGE_VAL);
      }
    }
    const bool zero = row < 0 || __any(bad);     // padding / NaN-inf rows: all-zero image
    double n2 = 0.0, e2 = 0.0;
    half8 last;                                   // the row's last half fragment (augmented columns)
#pragma unroll
    for (int e = 0; e < 8; ++e) last[e] = (_Float16)0;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int g = it * 64 + lane;
      if (g >= NG) continue;
      const int j0 = g * 8;
      half8 hi;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = j0 + e;
        double a = 0.0;
        if (!zero && j < S) a = (xv[it][e] - cm[it][e]) * scale;
        _Float16 hh = (_Float16)a;
        if (fabs((double)hh) < 6.103515625e-05) hh = (_Float16)0;   // no fp16 subnormals
        const double res = a - (double)hh;
        n2 += (double)hh * (double)hh;
        e2 += res * res;
        hi[e] = hh;
      }
      if (g == NG - 1) last = hi;
      else frag[(g >> 1) * 64 + (g & 1) * 32 + rl] = hi;
    }
    // fixed-order sums over the wave
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      n2 += __shfl_xor(n2, off);
      e2 += __shfl_xor(e2, off);
    }
    if (lane == ((NG - 1) & 63)) {
      // last half fragment: entries 4..7 are the augmented columns
      float nbf = (float)n2;
      if ((double)nbf > n2) nbf = __uint_as_float(__float_as_uint(nbf) - 1u);
      _Float16 u1, u2;
      float usum;
      split16(nbf * (1.f / 65536.f), false, u1, u2, usum);
      if (zero) { u1 = (_Float16)65504.f; u2 = (_Float16)65504.f; usum = 131008.f; }
      last[4] = -u1; last[5] = -u2; last[6] = (_Float16)AUG; last[7] = (_Float16)AUG;
      frag[(NK - 1) * 64 + 32 + rl] = last;
      RowInfo ri;
      ri.nb = usum * 65536.f;               // nb' exactly as the matrix pipe sees it
      ri.L = 0.f;
      if (zero) { ri.e = 0.f; ri.N = 0.f; }
      else {
        // representation error also covers the fp64 rounding of (x - c) * scale
        ri.e = up((float)(sqrt(e2) + 1e-15 * sqrt(n2)));
        ri.N = up((float)sqrt(n2));
        my_e = max(my_e, __float_as_uint(ri.e));      // (non-negative floats order like their bits)
        my_n = max(my_n, __float_as_uint(ri.N));
      }
      info[b] = ri;
    }
  }
  if (lane == ((NG - 1) & 63)) { wmax[0][wave] = my_e; wmax[1][wave] = my_n; }
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned int m = 0u;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) m = max(m, wmax[threadIdx.x][w]);
    if (m) atomicMax(threadIdx.x == 0 ? &glob->e_max : &glob->N_max, m);
  }
  half8 *dst = F + tile * (int64_t)(NK * 64);
  for (int e = threadIdx.x; e < NK * 64; e += NT)
    if (tile * 32 + (e & 31) < Bpad) dst[e] = frag[e];
}


// ------------------------------------------------------------------------------------------
__global__ void k_mark(unsigned char *searched, const ScreenBlock *__restrict__ blocks,
                       int64_t row_begin) {
  const ScreenBlock b = blocks[blockIdx.x];
  for (int i = threadIdx.x; i < b.nrows; i += blockDim.x) searched[b.row0 - row_begin + i] = 1;
}

// min over the candidate segments of a row's threshold, with the estimate bit of the winner
// (a tie is rigorous if either is)
__device__ __forceinline__ void min_threshold(float G0, int e0, float G1, int e1, float &G, int &e) {
  if (G0 < G1) { G = G0; e = e0; }
  else if (G1 < G0) { G = G1; e = e1; }
  else { G = G0; e = e0 & e1; }
}

// After the sampled pre-pass of a segmented sweep: every segment of a row continues with the
// tightest of the segments' estimates (each is an estimate of the same final threshold).
__global__ __launch_bounds__(NT) void k_share_thresholds(int64_t n_rows, int n_seg,
                                                         const unsigned char *__restrict__ searched,
                                                         float *__restrict__ g_state,
                                                         int *__restrict__ cnt) {
  const int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (r >= n_rows || !searched[r]) return;
  float G = g_state[r];
  int e = (cnt[r] >> 30) & 1;
  for (int s = 1; s < n_seg; ++s)
    min_threshold(G, e, g_state[(int64_t)s * n_rows + r], (cnt[(int64_t)s * n_rows + r] >> 30) & 1, G, e);
  for (int s = 0; s < n_seg; ++s) {
    const int64_t i = (int64_t)s * n_rows + r;
    g_state[i] = G;
    cnt[i] = (cnt[i] & CNT_MASK) | (e << 30);
  }
}

// Joins the shortlists of candidate segments s0 and s1 of every row into s0's: cut of the union
// at its k-th smallest screen value.  State (G, est) of the union: everything of both segments
// with t <= min(G0, G1) is in it.  is_root: the last merge -- a still unproven estimate hands the
// row to the exact kernel.  One wave per row.
__global__ __launch_bounds__(NT) void k_merge_segments(
    const RowInfo *__restrict__ info, const ScreenGlobals *__restrict__ glob,
    const int *__restrict__ rowpos, int64_t row_begin, int64_t n_rows,
    const unsigned char *__restrict__ searched, uint2 *__restrict__ sl, int *__restrict__ cnt,
    unsigned int *__restrict__ flags, float *__restrict__ g_state, int step, int n_seg, int k,
    float gamma, int is_root) {
  // one level of the merge tree per launch: blockIdx.y = the pair (segments s0, s0 + step)
  const int s0 = (int)blockIdx.y * 2 * step, s1 = s0 + step;
  if (s1 >= n_seg) return;
  const int lane = wcx::lane_id();
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  const float e_max = __uint_as_float(glob->e_max), N_max = __uint_as_float(glob->N_max);
  for (int64_t r = w0; r < n_rows; r += nw) {
    if (!searched[r]) continue;
    const int64_t i0 = (int64_t)s0 * n_rows + r, i1 = (int64_t)s1 * n_rows + r;
    const int c0 = cnt[i0], c1 = cnt[i1];
    const int n0 = c0 & CNT_MASK, n1 = c1 & CNT_MASK;
    if (flags[i0] | flags[i1] | (unsigned int)(n0 + n1 > CAP)) {   // -> exact fallback
      if (lane == 0) { flags[i0] = 1u; cnt[i0] = 0; }
      continue;
    }
    uint2 raw[CAP / 64];
#pragma unroll
    for (int q = 0; q < CAP / 64; ++q) {
      const int e = q * 64 + lane;
      raw[q] = e < n0 ? sl[i0 * CAP + e] : sl[i1 * CAP + (e - n0 < CAP ? e - n0 : 0)];
    }
    float na, E, Q, Gm;
    int em;
    row_budget(info[rowpos[row_begin + r]], e_max, N_max, gamma, na, E, Q);
    min_threshold(g_state[i0], (c0 >> 30) & 1, g_state[i1], (c1 >> 30) & 1, Gm, em);
    int n_new, e_new;
    const float Gn = compact_loaded<CAP / 64>(raw, n0 + n1, sl + i0 * CAP, k, na, E, Q, Gm, em, 2, is_root != 0,
                                    &flags[i0], n_new, e_new);
    if (lane == 0) { cnt[i0] = n_new | (e_new << 30); g_state[i0] = Gn; }
  }
}

// Rows the screen could not finish (shortlist overflow, unproven estimate) -> redo list for the
// exact kernels, built on the device: no host round trip.
//   ctr[0] = number of flagged rows, ctr[1] = number of tiles, ctr[2 + c] = flagged rows of
//   chromosome c, ctr[34 + c] = fill cursor of chromosome c in the row list.
__global__ __launch_bounds__(NT) void k_collect_redo(int64_t row_begin, int64_t n_rows,
                                                     const unsigned char *__restrict__ searched,
                                                     const unsigned int *__restrict__ flags,
                                                     ChrTab chr, TopkBlock *__restrict__ redo,
                                                     unsigned int *__restrict__ ctr,
                                                     unsigned long long *__restrict__ stats) {
  const int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (r >= n_rows || !searched[r] || !flags[r]) return;
  const int64_t row = row_begin + r;
  int64_t cs = 0, ce = chr.cum[0];
  int c = 0;
  for (int q = 1; q < chr.n_chr && row >= ce; ++q) { cs = ce; ce = chr.cum[q]; c = q; }
  const unsigned int slot = atomicAdd(&ctr[0], 1u);
  TopkBlock b;
  b.row0 = row; b.nrows = 1; b.pad = 0; b.cs = cs; b.ce = ce;
  redo[slot] = b;                       // one-row blocks: the device-wide path for a handful of rows
  atomicAdd(&ctr[2 + c], 1u);
  atomicAdd(&stats[3], 1ull);
}

// More than WCX_REDO_FAST flagged rows: tiles of <= 64 rows of one chromosome for the blocked exact
// kernel (tile.row0 = offset into the row list).  One thread; at most n_rows / 64 + n_chr tiles.
__global__ void k_redo_plan(ChrTab chr, unsigned int *__restrict__ ctr, TopkBlock *__restrict__ tiles) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (ctr[0] <= (unsigned int)WCX_REDO_FAST) { ctr[1] = 0; return; }
  unsigned int off = 0, t = 0;
  int64_t cs = 0;
  for (int c = 0; c < chr.n_chr; ++c) {
    const int64_t ce = chr.cum[c];
    const unsigned int cnt = ctr[2 + c];
    ctr[34 + c] = off;
    for (unsigned int q = 0; q < cnt; q += 64) {
      TopkBlock b;
      b.row0 = off + q; b.nrows = (int)(cnt - q < 64 ? cnt - q : 64); b.pad = 0; b.cs = cs; b.ce = ce;
      tiles[t++] = b;
    }
    off += cnt;
    cs = ce;
  }
  ctr[1] = t;
}

__global__ __launch_bounds__(NT) void k_redo_fill(int64_t row_begin, int64_t n_rows,
                                                  const unsigned char *__restrict__ searched,
                                                  const unsigned int *__restrict__ flags, ChrTab chr,
                                                  unsigned int *__restrict__ ctr,
                                                  int32_t *__restrict__ rowlist) {
  if (ctr[0] <= (unsigned int)WCX_REDO_FAST) return;
  const int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (r >= n_rows || !searched[r] || !flags[r]) return;
  const int64_t row = row_begin + r;
  int c = 0;
  for (int q = 1; q < chr.n_chr && row >= chr.cum[q - 1]; ++q) c = q;
  rowlist[atomicAdd(&ctr[34 + c], 1u)] = (int32_t)row;
}

// ------------------------------------------------------------------------------------------
// Symmetric sweep (screen_sym.h): sweep order, thresholds, final cut.
constexpr int SYM_NCLS = 32;               // coarse norm classes: float bits of |a|^2 >> 22 (two per octave)
constexpr int SYM_NCELL = SYM_NCLS * 32;   // (class, chromosome) cells, each padded to whole 32-row tiles
constexpr float SYM_DMAX = 4.0e9f;         // thresholds at or above this: the row goes to the exact kernel

constexpr int SYM_NSUB = 4;                // norm sub-steps inside a class (order only: no padding)

// rfine != nullptr: the rows at or below the hub key form the HUB region, the head of the order (their
// own (class, chromosome) cells, in front of everybody else's): cells [0, SYM_NCELL) hubs, the rest after.
__global__ __launch_bounds__(NT) void k_sym_hist(const unsigned int *__restrict__ rbits,
                                                 const int *__restrict__ rchr, int64_t B,
                                                 const ScreenGlobals *__restrict__ glob,
                                                 int *__restrict__ rkey, int *__restrict__ cellcnt,
                                                 int suborder, const unsigned int *__restrict__ rfine) {
  __shared__ int lh[2 * SYM_NCELL * SYM_NSUB];
  for (int i = threadIdx.x; i < 2 * SYM_NCELL * SYM_NSUB; i += NT) lh[i] = 0;
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b < B) {
    const unsigned int umin = 0xffffffffu - glob->uinv;
    const unsigned int u = rbits[b];
    unsigned int cls = SYM_NCLS - 1, sub = 0;           // non-finite rows go last
    if (u != 0xffffffffu && u >= umin) {
      cls = (u - umin) >> 2;
      sub = suborder ? ((u - umin) & 3u) : 0u;
      if (cls > SYM_NCLS - 2) { cls = SYM_NCLS - 2; sub = SYM_NSUB - 1; }
    }
    // inside a (class, chromosome) cell the rows are ordered by the finer norm steps: the rows of a
    // tile then have similar norms -- and similar thresholds, which is what the row-direction gate
    // of the symmetric sweep (one threshold per streamed tile) wants
    const bool hub = rfine && rfine[b] <= glob->hub_key;
    const int key = (((hub ? 0 : SYM_NCLS) + (int)cls) * 32 + rchr[b]) * SYM_NSUB + (int)sub;
    rkey[b] = key;
    atomicAdd(&lh[key], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * SYM_NCELL * SYM_NSUB; i += NT)
    if (lh[i]) atomicAdd(&cellcnt[i], lh[i]);
}

// First sweep position of every cell (cells padded to whole tiles), the chromosome of every tile,
// the number of tiles in use (rounded up to whole quads) and of the hub region (cells [0, SYM_NCELL)).
// tchr was preset to 255.  Thread t owns the cells 2 t and 2 t + 1.
__global__ __launch_bounds__(SYM_NCELL) void k_sym_scan(const int *__restrict__ cellcnt,
                                                        int *__restrict__ cursor,
                                                        unsigned char *__restrict__ tchr,
                                                        ScreenGlobals *__restrict__ glob) {
  __shared__ int part[SYM_NCELL];
  const int t = threadIdx.x;
  int sub[2][SYM_NSUB], padded[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    int total = 0;
#pragma unroll
    for (int j = 0; j < SYM_NSUB; ++j) { sub[c][j] = cellcnt[(2 * t + c) * SYM_NSUB + j]; total += sub[c][j]; }
    padded[c] = (total + 31) & ~31;
  }
  part[t] = padded[0] + padded[1];
  __syncthreads();
  for (int off = 1; off < SYM_NCELL; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int start = part[t] - padded[0] - padded[1];
  if (2 * t == SYM_NCELL) glob->n_hub_tiles = (unsigned int)(start >> 5);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    int run = start;
#pragma unroll
    for (int j = 0; j < SYM_NSUB; ++j) { cursor[(2 * t + c) * SYM_NSUB + j] = run; run += sub[c][j]; }
    for (int i = start >> 5; i < (start + padded[c]) >> 5; ++i) tchr[i] = (unsigned char)((2 * t + c) & 31);
    start += padded[c];
  }
  if (t == SYM_NCELL - 1) glob->n_tiles = (unsigned int)(((start >> 5) + 3) & ~3);
}

// ---- Deterministic order B (the row-sharded symmetric sweep: every rank must build the SAME tiles, the
// records they exchange name sweep positions).  k_scatter hands out the positions inside a cell in the
// order the workgroups' atomics arrive; here a row's position is  start[cell] + the number of rows
// before it with the same key  -- a stable counting sort:
//   k_det_local   per workgroup of 256 consecutive rows: rank among the workgroup's rows of the same key
//                 (compare loop in LDS) and, by the last such row, the workgroup's count -> table[key][wg]
//   k_det_scan    one wave per key: exclusive scan of its workgroup counts (+ the cell's start)
//   k_det_place   position = table[key][wg] + local rank
constexpr int DET_KEYS = 2 * SYM_NCELL * SYM_NSUB;
__global__ __launch_bounds__(NT) void k_det_local(const int *__restrict__ rkey, int64_t B, int n_wg,
                                                  int *__restrict__ table, unsigned short *__restrict__ lrank) {
  __shared__ int sk[NT];
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int key = b < B ? rkey[b] : -1;
  sk[threadIdx.x] = key;
  __syncthreads();
  if (key < 0) return;
  int before = 0, after = 0;
  for (int j = 0; j < NT; ++j) {
    const bool same = sk[j] == key;
    before += (same && j < (int)threadIdx.x) ? 1 : 0;
    after += (same && j > (int)threadIdx.x) ? 1 : 0;
  }
  lrank[b] = (unsigned short)before;
  if (after == 0) table[(int64_t)key * n_wg + blockIdx.x] = before + 1;
}
__global__ __launch_bounds__(NT) void k_det_scan(int *__restrict__ table, int n_wg,
                                                 const int *__restrict__ cursor) {
  const int lane = wcx::lane_id();
  const int key = (int)(((int64_t)blockIdx.x * NT + threadIdx.x) >> 6);
  if (key >= DET_KEYS) return;
  int *row = table + (int64_t)key * n_wg;
  int run = cursor[key];
  for (int w0 = 0; w0 < n_wg; w0 += 64) {
    const int w = w0 + lane;
    const int c = w < n_wg ? row[w] : 0;
    const int incl = wcx::wave_incl_scan_i(c);
    if (w < n_wg) row[w] = run + incl - c;
    run += __builtin_amdgcn_readlane(incl, 63);
  }
}
__global__ __launch_bounds__(NT) void k_det_place(const int *__restrict__ rkey, int64_t B, int n_wg,
                                                  const int *__restrict__ table,
                                                  const unsigned short *__restrict__ lrank,
                                                  int *__restrict__ perm, int *__restrict__ rowpos) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (b >= B) return;
  const int pos = table[(int64_t)rkey[b] * n_wg + blockIdx.x] + (int)lrank[b];
  perm[pos] = (int)b;
  rowpos[b] = pos;
}

// After the sampled pre-pass: the estimated threshold of every row, in screen-distance space
// (D = G + nb': the pre-pass admits t = nb'_c - 2 g~ <= G), as theta = -D/2 per sweep position;
// per-tile minimum; list counters reset (the pre-pass's entries are dropped: the symmetric sweep
// meets every pair again).  Rows without a usable estimate go to the exact kernel.
__global__ __launch_bounds__(NT) void k_sym_setup(const int *__restrict__ perm,
                                                  const RowInfo *__restrict__ info,
                                                  const ScreenGlobals *__restrict__ glob,
                                                  const float *__restrict__ g_state,
                                                  int *__restrict__ cnt, unsigned int *__restrict__ flags,
                                                  float *__restrict__ Dest,
                                                  unsigned int *__restrict__ tinfo,
                                                  float *__restrict__ tmin,
                                                  const unsigned int *__restrict__ gate = nullptr) {
  const int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int64_t tile = p >> 5;
  if (tile >= (int64_t)glob->n_tiles || (gate && !*gate)) return;
  const int l = (int)(p & 31);
  const int row = perm[p];
  float theta = HUGE_VALF;
  if (row >= 0) {
    const float G = g_state[row];
    const float D = G + info[p].nb;
    if (flags[row] || !(G < GMAX) || !(D < SYM_DMAX)) flags[row] = 1u;
    else theta = -0.5f * D;
    Dest[row] = D;
    cnt[row] = 0;
  }
  tinfo[tile * 64 + l] = __float_as_uint(theta);
  tinfo[tile * 64 + 32 + l] = (unsigned int)row;
  float m = theta;
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {
    const float o = __shfl_xor(m, off, 64);
    m = o < m ? o : m;
  }
  if (l == 0) tmin[tile] = m;
}

// Records of the symmetric sweep (row, partner position, d~ bits) -> the rows' lists.
__global__ __launch_bounds__(NT) void k_sym_regroup(const uint4 *__restrict__ pool,
                                                    const unsigned int *__restrict__ pool_head,
                                                    const unsigned int *__restrict__ pool_ovf,
                                                    unsigned int pool_cap, int64_t n_rows,
                                                    uint2 *__restrict__ sl, int *__restrict__ cnt,
                                                    unsigned int *__restrict__ flags, int cap2,
                                                    const unsigned int *__restrict__ gate = nullptr) {
  if (gate && !*gate) return;
  if (*pool_ovf) {                           // records were lost: every row goes to the exact kernel
    for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * NT)
      flags[r] = 1u;
    return;
  }
  unsigned int n = *pool_head;
  if (n > pool_cap) n = pool_cap;
  for (unsigned int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    const uint4 rec = pool[i];
    const int slot = atomicAdd(&cnt[rec.x], 1);
    if (slot < cap2) sl[(int64_t)rec.x * cap2 + slot] = make_uint2(rec.z, rec.y);
    else flags[rec.x] = 1u;
  }
}

// Final cut of the symmetric sweep, one wave per row: the k-th smallest screen distance of the
// list bounds the true k-th distance; its filter bound F must not exceed the estimate the list was
// collected under (everything with d~ <= D is in the list) -- then the entries with d~ <= F are
// exactly the ones the refine needs; else the row is flagged for the exact kernel.
// IPL = list slots per lane: the lists hold 64 IPL entries (SymArgs::cap2); max_keep = most entries the
// refine accepts.
// The cut of ONE row's list by one wave, NQ slices of 64 entries in registers (NQ >= the slices that hold
// entries).  Round 6: the kernel used to walk all 32 (64) slices of the list's CAPACITY in every one of the
// ~25 bisection steps -- 800 ballots per row whatever the list held, which is why halving the lists
// (890 -> 489 entries, round 5) left it at 0.96 ms; a row now pays for the slices it uses (8 at 489 entries).
template <int NQ>
__device__ __forceinline__ void sym_final_row(uint2 *__restrict__ row, int c, int k, const RowInfo &ri,
                                              float e_max, float N_max, float gamma, float dest, int max_keep,
                                              unsigned int *__restrict__ flag, int *__restrict__ cnt_r) {
  const int lane = wcx::lane_id();
  unsigned int key[NQ], idx[NQ];
  const int nq = (c + 63) >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    uint2 v = make_uint2(0u, 0u);
    if (q < nq) v = row[q * 64 + lane];
    key[q] = (q * 64 + lane < c) ? f32_key(__uint_as_float(v.x)) : 0xffffffffu;
    idx[q] = v.y;
  }
  const unsigned int kref = (unsigned int)__builtin_amdgcn_readfirstlane((int)key[0]);
  unsigned int x = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) x |= (q * 64 + lane < c) ? (key[q] ^ kref) : 0u;
  x = wcx::wave_or_u32(x);
  const int hb = 31 - __builtin_clz(x | 1u);
  unsigned int prefix = kref & ~((2u << hb) - 1u);
  for (int bit = hb; bit >= 0; --bit) {
    const unsigned int trial = prefix | (1u << bit);
    int n_lt = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) n_lt += __popcll(__ballot(key[q] < trial));
    if (n_lt < k) prefix = trial;
  }
  const float tk = key_f32(prefix);
  float na, E, Q;
  row_budget(ri, e_max, N_max, gamma, na, E, Q);
  const float dk = tk > 0.f ? tk : 0.f;
  const float rt = sqrtf(up(dk + Q)) * 1.0000005f + 2.f * E;
  const float Fb = up(up(rt * rt) + Q);
  if (!(Fb <= dest)) {                    // estimate unproven (or NaN): exact kernel
    if (lane == 0) { *flag = 1u; *cnt_r = 0; }
    return;
  }
  const unsigned int gkey = f32_key(Fb);
  int base = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const bool keep = key[q] <= gkey && (q * 64 + lane < c);
    const unsigned long long m = __ballot(keep);
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) row[pos] = make_uint2(__float_as_uint(key_f32(key[q])), idx[q]);
    base += __popcll(m);
  }
  if (lane == 0) {
    if (base > max_keep) { *flag = 1u; *cnt_r = 0; }
    else *cnt_r = base;
  }
}

template <int IPL>
__global__ __launch_bounds__(NT) void k_sym_final(const RowInfo *__restrict__ info,
                                                  const ScreenGlobals *__restrict__ glob,
                                                  const int *__restrict__ rowpos, int64_t n_rows,
                                                  uint2 *__restrict__ sl, int *__restrict__ cnt,
                                                  unsigned int *__restrict__ flags,
                                                  const float *__restrict__ Dest, int k, float gamma,
                                                  int max_keep, const unsigned int *__restrict__ gate) {
  if (gate && !*gate) return;
  const int lane = wcx::lane_id();
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  const float e_max = __uint_as_float(glob->e_max), N_max = __uint_as_float(glob->N_max);
  constexpr int CAPL = 64 * IPL;
  for (int64_t r = w0; r < n_rows; r += nw) {
    const int c = cnt[r];
    if (flags[r] || c > CAPL || c < k) {
      if (lane == 0) { flags[r] = 1u; cnt[r] = 0; }
      continue;
    }
    uint2 *row = sl + r * (int64_t)CAPL;
    const RowInfo ri = info[rowpos[r]];
    const float dest = Dest[r];
    // (wave-uniform: c is the row's count)
    if (c <= 64 * 8 && IPL >= 8)
      sym_final_row<8>(row, c, k, ri, e_max, N_max, gamma, dest, max_keep, &flags[r], &cnt[r]);
    else if (c <= 64 * 16 && IPL >= 16)
      sym_final_row<16>(row, c, k, ri, e_max, N_max, gamma, dest, max_keep, &flags[r], &cnt[r]);
    else
      sym_final_row<IPL>(row, c, k, ri, e_max, N_max, gamma, dest, max_keep, &flags[r], &cnt[r]);
  }
}

// After the first attempt (thresholds from the hub counts): rows the final cut could not finish.  More
// than the device-wide redo takes -- data whose neighbours are not its low-norm rows, lists overflowed --
// opens the gate of the second attempt (sampled pre-pass, distribution-free, + another sweep).
constexpr unsigned int HUB_FAIL_MAX = 96;
__global__ __launch_bounds__(NT) void k_hub_count_failed(const unsigned int *__restrict__ flags, int64_t n_rows,
                                                         unsigned int *__restrict__ n_failed) {
  int c = 0;
  for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * NT)
    c += flags[r] ? 1 : 0;
  c = wcx::wave_sum_i(c);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(n_failed, (unsigned int)c);
}
__global__ void k_hub_verdict(const unsigned int *__restrict__ n_failed, unsigned int *__restrict__ gate,
                              unsigned long long *__restrict__ stats) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned int bad = *n_failed > HUB_FAIL_MAX ? 1u : 0u;
    *gate = bad;
    if (stats) stats[18] = bad ? *n_failed : 0ull;       // (diagnostics: rows that sent the sweep round again)
  }
}
// Second attempt only: lists, flags, record pool, work queue and sequence counters back to empty.
__global__ __launch_bounds__(NT) void k_gate_reset(const unsigned int *__restrict__ gate, int64_t n_rows,
                                                   int *__restrict__ cnt, unsigned int *__restrict__ flags,
                                                   unsigned int *__restrict__ heads, int *__restrict__ seq,
                                                   int n_seq) {
  if (!*gate) return;
  for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * NT) {
    cnt[r] = 0;
    flags[r] = 0u;
    if (r < n_seq) seq[r] = 0;
    if (r < 3) heads[r] = 0u;                      // pool head | pool overflow | queue head
  }
}

// ---- Row-sharded symmetric sweep: the records of this rank's tile pairs, by the rank that owns the row.
struct RowBounds { int n; int64_t b[33]; };       // rank r owns rows [b[r], b[r + 1])
__device__ __forceinline__ int owner_of(const RowBounds &rb, unsigned int row) {
  int o = 0;
  while (o + 1 < rb.n && (int64_t)row >= rb.b[o + 1]) ++o;
  return o;
}
__global__ __launch_bounds__(NT) void k_rec_hist(const uint4 *__restrict__ pool,
                                                 const unsigned int *__restrict__ pool_head,
                                                 unsigned int pool_cap, RowBounds rb,
                                                 unsigned long long *__restrict__ counts) {
  __shared__ unsigned int h[32];
  if (threadIdx.x < 32) h[threadIdx.x] = 0u;
  __syncthreads();
  unsigned int n = *pool_head;
  if (n > pool_cap) n = pool_cap;
  for (unsigned int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT)
    atomicAdd(&h[owner_of(rb, pool[i].x)], 1u);
  __syncthreads();
  if (threadIdx.x < 32 && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
// cursor[r] = first slot of destination r in the send buffer (advanced by the records placed).
// A workgroup places 1024 records at a time: ranks within the chunk through LDS counters, ONE global
// atomic per (chunk, destination) -- a returning atomic per record on the same few addresses serialises
// at ~10 ns each (measured: 0.5 s for 44 M records).
__global__ __launch_bounds__(1024) void k_rec_bucket(const uint4 *__restrict__ pool,
                                                     const unsigned int *__restrict__ pool_head,
                                                     unsigned int pool_cap, RowBounds rb,
                                                     unsigned long long *__restrict__ cursor,
                                                     uint4 *__restrict__ send) {
  __shared__ unsigned int cntL[32];
  __shared__ unsigned long long baseL[32];
  unsigned int n = *pool_head;
  if (n > pool_cap) n = pool_cap;
  const int lane = wcx::lane_id();
  for (unsigned int c0 = blockIdx.x * 1024u; c0 < n; c0 += gridDim.x * 1024u) {
    if (threadIdx.x < 32) cntL[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned int i = c0 + threadIdx.x;
    const bool in = i < n;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    int dest = -1;
    if (in) { rec = pool[i]; dest = owner_of(rb, rec.x); }
    // per wave and destination: one LDS atomic, the lanes' ranks from the ballot
    unsigned int my = 0u;
    unsigned long long todo = __ballot(in);
    while (todo) {
      const int first = __builtin_ctzll(todo);
      const int d = __builtin_amdgcn_readlane(dest, first);
      const unsigned long long m = __ballot(dest == d);
      unsigned int b = 0u;
      if (lane == first) b = atomicAdd(&cntL[d], (unsigned int)__popcll(m));
      b = (unsigned int)__builtin_amdgcn_readlane((int)b, first);
      if (dest == d) my = b + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
      todo &= ~m;
    }
    __syncthreads();
    if (threadIdx.x < 32 && cntL[threadIdx.x])
      baseL[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)cntL[threadIdx.x]);
    __syncthreads();
    if (in) send[baseL[dest] + my] = rec;
    __syncthreads();
  }
}
// records received for the rows [row0, row0 + n_own): into the rows' lists (list index = row - row0)
__global__ __launch_bounds__(NT) void k_rec_regroup(const uint4 *__restrict__ recv, int64_t n_recv,
                                                    int64_t row0, int64_t n_own, uint2 *__restrict__ sl,
                                                    int *__restrict__ cnt, unsigned int *__restrict__ flags,
                                                    int cap2) {
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n_recv; i += (int64_t)gridDim.x * NT) {
    const uint4 rec = recv[i];
    const int64_t r = (int64_t)rec.x - row0;
    if (r < 0 || r >= n_own) continue;               // (not ours: a routing error would show as missing rows)
    const int slot = atomicAdd(&cnt[r], 1);
    if (slot < cap2) sl[r * cap2 + slot] = make_uint2(rec.z, rec.y);
    else flags[r] = 1u;
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

bool wcx_screen_supported(int64_t B, int S, int k) {
  return S <= SMAX_SCREEN && k <= KMAX_SCREEN && B >= 2048;
}

static int env_int(const char *name, int dflt);
// Row pitch of the row-major copy Xr the refine gathers from, in doubles: rows start on 128-byte
// lines (16 doubles), so that a 16-sample chunk of a candidate row is ONE aligned cache line -- with
// a 32-byte-aligned pitch (4 000-byte rows at S = 500) every chunk straddles two lines and the L2s
// see twice the requests.  WCX_ROW_ALIGN overrides (doubles; multiple of 4).
static int row_pitch(int S) {
  int a = env_int("WCX_ROW_ALIGN", 16);
  if (a < 4 || (a & 3)) a = 4;
  return (S + a - 1) / a * a;
}

static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

static int screen_dispatch(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds,
                           hipStream_t st) {
  int rc = wcx_screen_launch_k1(c, a, grid, lds, st);
  if (rc < 0) rc = wcx_screen_launch_k2(c, a, grid, lds, st);
  if (rc < 0) rc = wcx_screen_launch_k3(c, a, grid, lds, st);
  if (rc < 0) rc = wcx_screen_launch_k4(c, a, grid, lds, st);
  if (rc < 0) rc = wcx_screen_launch_k5(c, a, grid, lds, st);
  if (rc < 0) rc = wcx_screen_launch_k6(c, a, grid, lds, st);
  if (rc < 0) rc = wcx_screen_launch_k7(c, a, grid, lds, st);
  if (rc < 0) rc = wcx_screen_launch_k8(c, a, grid, lds, st);
  return rc;
}

static int sym_dispatch(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds,
                        hipStream_t st) {
  int rc = wcx_sym_launch_k1(nk, ctg, lb, ring, a, grid, lds, st);
  if (rc < 0) rc = wcx_sym_launch_k2(nk, ctg, lb, ring, a, grid, lds, st);
  if (rc < 0) rc = wcx_sym_launch_k3(nk, ctg, lb, ring, a, grid, lds, st);
  if (rc < 0) rc = wcx_sym_launch_k4(nk, ctg, lb, ring, a, grid, lds, st);
  if (rc < 0) rc = wcx_sym_launch_k5(nk, ctg, lb, ring, a, grid, lds, st);
  return rc;
}

// Row-sharded symmetric sweep: what phase 1 (wcx_newref_sym_sweep_dev) leaves for the record copy and
// for phase 2 (wcx_newref_sym_finish_dev); lives in the context (wcx_ctx::sym_state).
struct SymShardState {
  bool valid = false;
  int64_t B = 0, row_begin = 0, row_end = 0;
  int S = 0, Sp = 0, NK = 0, k = 0, cap2 = 0, n_parts = 0, part = 0;
  ChrTab tab;
  RowBounds rb;
  const double *dXs = nullptr;
  double *Xr = nullptr;
  RowInfo *info = nullptr;
  ScreenGlobals *glob = nullptr;
  int *rowpos = nullptr, *perm = nullptr, *cnt = nullptr;
  float *Dest = nullptr;
  unsigned int *flags = nullptr, *pool_head = nullptr, *d_nredo = nullptr;
  unsigned char *searched = nullptr;
  uint2 *sl = nullptr;
  uint4 *pool = nullptr;
  unsigned int pool_cap = 0;
  unsigned long long *d_counts = nullptr;       // [32] records per destination | [32] bucket cursors
  TopkBlock *d_redo = nullptr, *d_rtile = nullptr;
  int32_t *d_rlist = nullptr;
  void *rscr = nullptr;
  unsigned long long counts[32] = {0};
  bool overflow = false;                         // this rank's record pool overflowed: the exchange is void
};
struct SymShardCall { int part, n_parts; RowBounds rb; SymShardState *state; };

static int count_dispatch(int nk, int ctg, int lb, int ring, const CountArgs &a, unsigned grid, size_t lds,
                          hipStream_t st) {
  int rc = wcx_count_launch_k1(nk, ctg, lb, ring, a, grid, lds, st);
  if (rc < 0) rc = wcx_count_launch_k2(nk, ctg, lb, ring, a, grid, lds, st);
  if (rc < 0) rc = wcx_count_launch_k3(nk, ctg, lb, ring, a, grid, lds, st);
  if (rc < 0) rc = wcx_count_launch_k4(nk, ctg, lb, ring, a, grid, lds, st);
  return rc;
}

// The search of ALL rows against all rows with the symmetric sweep (screen_sym.h):
//   order B   all rows by (hub region first, norm class, chromosome), cells padded to tiles: fragments F
//   attempt 1 (K >= 256, B >= 32768): thresholds from the hub counts (screen_count.h: every row against
//             the low-norm 1/32 of the rows, no lists), k_screen_sym over all tile pairs {a < b} with
//             those FIXED thresholds, final cut: proves the threshold per row or flags it
//   verdict   more than HUB_FAIL_MAX rows flagged (data without hubs): the gate of attempt 2 opens
//   attempt 2 (always launched, returns at once behind a closed gate; the only attempt for small K):
//     order A   the sample rows (b = 0 mod SF) best-first, own small fragment array F_s
//     pre-pass  one-directional kernel, targets = all rows (fragments from F), candidates = F_s:
//               streaming top-r -> an estimated threshold per row (its list entries are dropped)
//     sweep + final cut as above
//   refine; exact redo of flagged rows
static int screen_sym_path(wcx_ctx *ctx, const double *dXs, int64_t B, int S, const int64_t *chr_cum,
                           int n_chr, const std::vector<ScreenBlock> &blocks, const ScreenCfg &cfg,
                           int SF, int cut_r, int raw_est, int slots, int k, int32_t *d_out_idx,
                           double *d_out_dist, SymShardCall *sh = nullptr) {
  const int NK = cfg.nk, CTG = cfg.ctg, GRr = CTG * 32;
  const int Sp = row_pitch(S);
  const int64_t n_rows = B;
  const int64_t n_own = sh ? sh->rb.b[sh->part + 1] - sh->rb.b[sh->part] : B;   // rows whose lists live here
  // list capacity per row: the estimates admit ~4 k entries at k = 300, ~2.7 k at k = 1000
  const int cap2 = k <= 448 ? CAP2 : CAP2_BIG;
  // hub-count estimates (attempt 1): WCX_SYM_HUB=0 turns them off; the region is 1 / WCX_HUB_FRAC of the
  // rows, at least 8 x the entries wanted (+ 512) below an estimate (1.18 k: the k-th neighbour's filter bound
  // ranks ~1.14 k)
  // (WCX_HUB_TEST_FAIL=1, tests: a count nobody reaches -- every row ends without an estimate, the verdict
  //  opens the gate and the second attempt must deliver the same bits)
  const int need = env_int("WCX_HUB_TEST_FAIL", 0) ? (1 << 28) : (int)(1.18 * k) + 8;
  // (few samples: the distances are noisier and the neighbours less concentrated on the low-norm rows --
  //  two thirds of them in the lowest 1/16 at S = 100 against 98 % at S = 500: a larger region)
  const int hub_frac = env_int("WCX_HUB_FRAC", NK >= 16 ? 32 : 12);
  int64_t hub_rows = hub_frac > 1 ? B / hub_frac : 0;
  // (at least 8 x the entries wanted + the 512 candidates of the moment phase: the loosest trial sits at
  //  4 x need among what is left after the row's own chromosome is taken out)
  const int64_t need_real = (int64_t)(1.18 * k) + 8;
  if (hub_rows < 8 * need_real + 512) hub_rows = 8 * need_real + 512;
  const bool use_hub = env_int("WCX_SYM_HUB", 1) != 0 && NK >= 5 && hub_frac > 1 && hub_rows * 6 <= B;
  if (sh && !use_hub) {
    wcx_set_error("the row-sharded symmetric sweep needs the hub-count thresholds (K >= 256, B >= %lld)",
                  (long long)(36 * (int64_t)need));
    return (int)WCX_ERR_UNSUPPORTED;
  }
  const int n1_tiles = env_int("WCX_HUB_N1", 16);
  const int64_t n_s = (B + SF - 1) / SF;
  const int64_t P_s = (n_s + CT - 1) / CT * CT;
  const int64_t Bpad2 = P_s + ((B - n_s) + CT - 1) / CT * CT;       // positions of the two-region order
  const int64_t NTb = (((B + 31) / 32 + (int64_t)2 * SYM_NCLS * n_chr + 4) + 7) / 8 * 8;   // tile bound
  const int64_t PB = NTb * 32;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_glob = carve(sizeof(ScreenGlobals));
  const size_t o_mean = carve((size_t)S * 8 * (3 + 2 * CSPLIT));   // mean | min | max | partial sums | partial counts
  const size_t o_xr = carve((size_t)B * Sp * 8 + 256);
  const size_t o_F = carve((size_t)PB * NK * 32);
  const size_t o_Fs = carve((size_t)P_s * NK * 32);
  const size_t o_info = carve((size_t)PB * sizeof(RowInfo));
  const size_t o_infs = carve((size_t)P_s * sizeof(RowInfo));
  const size_t o_perm = carve((size_t)PB * 4);
  const size_t o_perm2 = carve((size_t)Bpad2 * 4);
  const size_t o_rpos = carve((size_t)B * 4);
  const size_t o_rpos2 = carve((size_t)B * 4);
  const size_t o_rbit = carve((size_t)B * 4);
  const size_t o_rfin = carve((size_t)B * 4);
  const size_t o_hubh = carve((size_t)HUB_BINS * 4);
  const size_t o_rchr = carve((size_t)B * 4);
  const size_t o_rkey = carve((size_t)B * 4);
  const size_t o_cell = carve((size_t)2 * NCELL * 4);
  const size_t o_curs = carve((size_t)2 * NCELL * 4);
  const size_t o_gmsk = carve((size_t)(P_s / CT) * 4);
  const size_t o_tinf = carve((size_t)NTb * 64 * 4);
  const size_t o_tmin = carve((size_t)NTb * 4);
  const size_t o_tchr = carve((size_t)NTb);
  const size_t o_dest = carve((size_t)n_rows * 4);
  // records: row-direction hits (a few per cent of all) + everything from launches that split a
  // target quad over several work items (the high, hub-free tiles)
  // (a small matrix splits most of its chunks: nearly every hit is a record then -- half a list per row)
  const int64_t pool_want = sh ? n_rows * (int64_t)(cap2 / 2) / sh->n_parts * 2 : n_rows * (int64_t)(cap2 / 2);
  const unsigned int pool_cap = (unsigned int)(pool_want < 200000000ll ? pool_want : 200000000ll);
  const size_t o_pool = carve((size_t)pool_cap * 16);
  const size_t o_phead = carve(256);      // pool head | pool overflow | queue head | gate | failed rows
  const size_t o_desc = carve(256 * sizeof(SymDesc));
  const size_t o_seq = carve((size_t)(NTb / 4 + 1) * 4);
  const size_t o_sl = carve((size_t)n_own * cap2 * 8);
  const int n_wg_det = (int)((B + NT - 1) / NT);
  const size_t o_dett = carve(sh ? (size_t)DET_KEYS * n_wg_det * 4 : 0);
  const size_t o_detl = carve(sh ? (size_t)B * 2 : 0);
  const size_t o_cnts = carve(64 * 8);
  const size_t o_cnt = carve((size_t)n_rows * 4);
  const size_t o_gst = carve((size_t)n_rows * 4);
  const size_t o_flag = carve((size_t)n_rows * 4);
  const size_t o_srch = carve((size_t)n_rows);
  const size_t o_blk = carve(blocks.size() * sizeof(ScreenBlock));
  const size_t o_redo = carve((size_t)n_rows * sizeof(TopkBlock));
  const size_t o_nredo = carve(512);
  const size_t o_rtile = carve(((size_t)n_rows / 64 + 64) * sizeof(TopkBlock));
  const size_t o_rlist = carve((size_t)n_rows * 4);
  const size_t o_rscr = carve(wcx_topk_redo_scratch_bytes(k, B));
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, off, &scr);
  if (rc) return rc;
  char *base = reinterpret_cast<char *>(scr);
  ScreenGlobals *glob = reinterpret_cast<ScreenGlobals *>(base + o_glob);
  double *cmean = reinterpret_cast<double *>(base + o_mean);
  double *Xr = reinterpret_cast<double *>(base + o_xr);
  half8 *F = reinterpret_cast<half8 *>(base + o_F);
  half8 *Fs = reinterpret_cast<half8 *>(base + o_Fs);
  RowInfo *info = reinterpret_cast<RowInfo *>(base + o_info);
  RowInfo *infs = reinterpret_cast<RowInfo *>(base + o_infs);
  int *perm = reinterpret_cast<int *>(base + o_perm);
  int *perm2 = reinterpret_cast<int *>(base + o_perm2);
  int *rowpos = reinterpret_cast<int *>(base + o_rpos);
  int *rowpos2 = reinterpret_cast<int *>(base + o_rpos2);
  unsigned int *rbits = reinterpret_cast<unsigned int *>(base + o_rbit);
  unsigned int *rfine = reinterpret_cast<unsigned int *>(base + o_rfin);
  int *hubhist = reinterpret_cast<int *>(base + o_hubh);
  int *rchr = reinterpret_cast<int *>(base + o_rchr);
  int *rkey = reinterpret_cast<int *>(base + o_rkey);
  int *cellcnt = reinterpret_cast<int *>(base + o_cell);
  int *cursor = reinterpret_cast<int *>(base + o_curs);
  unsigned int *gmask = reinterpret_cast<unsigned int *>(base + o_gmsk);
  unsigned int *tinfo = reinterpret_cast<unsigned int *>(base + o_tinf);
  float *tmin = reinterpret_cast<float *>(base + o_tmin);
  unsigned char *tchr = reinterpret_cast<unsigned char *>(base + o_tchr);
  float *Dest = reinterpret_cast<float *>(base + o_dest);
  uint4 *pool = reinterpret_cast<uint4 *>(base + o_pool);
  unsigned int *pool_head = reinterpret_cast<unsigned int *>(base + o_phead);
  unsigned int *d_gate = pool_head + 3, *d_failed = pool_head + 4;
  SymDesc *d_desc = reinterpret_cast<SymDesc *>(base + o_desc);
  int *d_seq = reinterpret_cast<int *>(base + o_seq);
  uint2 *sl = reinterpret_cast<uint2 *>(base + o_sl);
  int *cnt_out = reinterpret_cast<int *>(base + o_cnt);
  float *g_state = reinterpret_cast<float *>(base + o_gst);
  unsigned int *flags = reinterpret_cast<unsigned int *>(base + o_flag);
  unsigned char *searched = reinterpret_cast<unsigned char *>(base + o_srch);
  ScreenBlock *d_blocks = reinterpret_cast<ScreenBlock *>(base + o_blk);
  TopkBlock *d_redo = reinterpret_cast<TopkBlock *>(base + o_redo);
  unsigned int *d_nredo = reinterpret_cast<unsigned int *>(base + o_nredo);
  TopkBlock *d_rtile = reinterpret_cast<TopkBlock *>(base + o_rtile);
  int32_t *d_rlist = reinterpret_cast<int32_t *>(base + o_rlist);
  // the gate of the second attempt: closed until the verdict opens it; without a first attempt the
  // second one is the only one and runs ungated
  const unsigned int *gate2 = use_hub ? d_gate : nullptr;

  hipStream_t st = ctx->stream;
  WCX_HIP(hipMemsetAsync(glob, 0, sizeof(ScreenGlobals), st));
  WCX_HIP(hipMemsetAsync(cnt_out, 0, (size_t)n_rows * 4, st));
  WCX_HIP(hipMemsetAsync(flags, 0, (size_t)n_rows * 4, st));
  WCX_HIP(hipMemsetAsync(searched, 1, (size_t)n_rows, st));          // every row is a target
  WCX_HIP(hipMemsetAsync(pool_head, 0, 256, st));
  WCX_HIP(hipMemsetAsync(d_nredo, 0, 512, st));
  WCX_HIP(hipMemsetAsync(ctx->d_stats, 0, 256, st));
  rc = wcx_upload_small(ctx, d_blocks, blocks.data(), blocks.size() * sizeof(ScreenBlock));
  if (rc) return rc;

  rc = wcx_timer_begin(ctx, "topk");
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "topk_prep");
  if (rc) return rc;
  unsigned long long *cmin = reinterpret_cast<unsigned long long *>(cmean + S);
  unsigned long long *cmax = cmin + S;
  double *psum = cmean + 3 * S, *pcnt = psum + (size_t)S * CSPLIT;
  WCX_HIP(hipMemsetAsync(cmin, 0xff, (size_t)S * 8, st));
  WCX_HIP(hipMemsetAsync(cmax, 0, (size_t)S * 8, st));
  k_col_sum<<<dim3((unsigned)S, CSPLIT), NT, 0, st>>>(dXs, B, psum, pcnt, cmin, cmax);
  k_col_stats<<<(unsigned)((S + 63) / 64), 64, 0, st>>>(S, psum, pcnt, cmin, cmax, cmean, glob);
  ChrTab tab;
  tab.n_chr = n_chr;
  for (int c = 0; c < 32; ++c) tab.cum[c] = c < n_chr ? chr_cum[c] : B;
  const unsigned gb = (unsigned)((B + NT - 1) / NT);
  if (use_hub) {
    WCX_HIP(hipMemsetAsync(hubhist, 0, (size_t)HUB_BINS * 4, st));
    k_transpose_norm<<<(unsigned)((B + 31) / 32), 256, 0, st>>>(dXs, B, S, Sp, Xr, cmean, tab, glob, rbits, rchr,
                                                                rfine, hubhist);
    k_hub_cut<<<1, 1024, 0, st>>>(hubhist, (int)hub_rows, glob);
  } else {
    k_transpose_norm<<<(unsigned)((B + 31) / 32), 256, 0, st>>>(dXs, B, S, Sp, Xr, cmean, tab, glob, rbits, rchr,
                                                                nullptr, nullptr);
  }
  // order B: (hub region | the rest) x (norm class, chromosome) cells padded to tiles
  WCX_HIP(hipMemsetAsync(cellcnt, 0, (size_t)2 * NCELL * 4, st));
  WCX_HIP(hipMemsetAsync(cursor, 0, (size_t)2 * NCELL * 4, st));
  WCX_HIP(hipMemsetAsync(perm, 0xff, (size_t)PB * 4, st));
  WCX_HIP(hipMemsetAsync(tchr, 0xff, (size_t)NTb, st));
  static_assert(2 * SYM_NCELL * SYM_NSUB <= 2 * NCELL, "cell tables");
  k_sym_hist<<<gb, NT, 0, st>>>(rbits, rchr, B, glob, rkey, cellcnt, env_int("WCX_SYM_SUBORDER", 1),
                                use_hub ? rfine : nullptr);
  k_sym_scan<<<1, SYM_NCELL, 0, st>>>(cellcnt, cursor, tchr, glob);
  if (sh) {      // the same order on every rank
    int *dett = reinterpret_cast<int *>(base + o_dett);
    unsigned short *detl = reinterpret_cast<unsigned short *>(base + o_detl);
    WCX_HIP(hipMemsetAsync(dett, 0, (size_t)DET_KEYS * n_wg_det * 4, st));
    k_det_local<<<gb, NT, 0, st>>>(rkey, B, n_wg_det, dett, detl);
    k_det_scan<<<(unsigned)((DET_KEYS * 64 + NT - 1) / NT), NT, 0, st>>>(dett, n_wg_det, cursor);
    k_det_place<<<gb, NT, 0, st>>>(rkey, B, n_wg_det, dett, detl, perm, rowpos);
  } else {
    k_scatter<<<gb, NT, 0, st>>>(rkey, B, cursor, perm, rowpos);
  }
  const unsigned gprep = (unsigned)((PB + NT - 1) / NT), gprep_s = (unsigned)((P_s + NT - 1) / NT);
  auto prep_frag = [&](int64_t n_pos, const int *pm, half8 *Fo, RowInfo *io, const unsigned int *gate) {
    const unsigned g = (unsigned)((n_pos + 31) / 32);            // one workgroup per 32-row tile
    switch (NK) {
#define WCX_PREP_CASE(N) case N: k_screen_prep<N><<<g, NT, 0, st>>>(Xr, n_pos, S, Sp, cmean, pm, glob, Fo, io, gate); break;
      WCX_PREP_CASE(1) WCX_PREP_CASE(2) WCX_PREP_CASE(3) WCX_PREP_CASE(4) WCX_PREP_CASE(5)
      WCX_PREP_CASE(6) WCX_PREP_CASE(7) WCX_PREP_CASE(8) WCX_PREP_CASE(10) WCX_PREP_CASE(12)
      WCX_PREP_CASE(14) WCX_PREP_CASE(16) WCX_PREP_CASE(20) WCX_PREP_CASE(24) WCX_PREP_CASE(28)
      WCX_PREP_CASE(40) WCX_PREP_CASE(48) WCX_PREP_CASE(56) WCX_PREP_CASE(64)
      default: k_screen_prep<32><<<g, NT, 0, st>>>(Xr, n_pos, S, Sp, cmean, pm, glob, Fo, io, gate); break;
#undef WCX_PREP_CASE
    }
  };
  (void)gprep_s;
  prep_frag(PB, perm, F, info, nullptr);
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "topk_prep");
  if (rc) return rc;
  const int kick_at = env_int("WCX_RANK_KICK", S >= 256 ? 1 : 0);
  if (kick_at == 0) {
    rc = wcx_aux_kick(ctx);
    if (rc) return rc;
  }
  rc = wcx_timer_begin(ctx, "topk_screen");
  if (rc) return rc;

  // ---- the symmetric sweep: ONE persistent launch; the workgroups pull (chunk, quad) work items from a
  // device-side queue in chunk-major order (k_screen_sym).  As long as a chunk has at least `fill`
  // quads above it, a quad is one work item per chunk and its column-direction hits go straight to
  // the lists (exclusive items, ordered per quad); the chunks high in the order, which few quads
  // still stream, split a quad over several items that write records instead.
  // (a chunk that fits an XCD's 4 MB L2 is fetched once per XCD and round of workgroups; an 8 MB chunk --
  //  the round-4 default at K = 512 -- cycles through it: 45.4 GB fetched per sweep against 16.6 GB at
  //  3 MB, 21.5 against 20.9 ms; 1.5 / 2 / 4 MB: 19.3 / 16.8 / 21.8 GB: scripts/sweep_sym_chunk_traffic.sh)
  int64_t Cz = ((int64_t)env_int("WCX_SYM_CHUNK_KB", 3072) << 10) / ((int64_t)NK * 1024);
  Cz = Cz / 8 * 8;
  if (Cz < 32) Cz = 32;
  if (Cz > 8192) Cz = 8192;
  while ((NTb + Cz - 1) / Cz > 256) Cz *= 2;                   // descriptor table: <= 256 chunks
  const int glist_cap = (int)((Cz / CTG + 64 + 3) / 4 * 4);
  const size_t lds_sym = (size_t)cfg.ring * (size_t)(CTG * NK * 64) * 16 + (size_t)cfg.ring * CTG * 256 +
                         (size_t)glist_cap * 4 + (size_t)4 * 64 * 16;   // visit list, staged records
  const int NQb = (int)(NTb / 4);
  SymArgs sa;
  int sym_grid = 0;
  {
    const int split_env = env_int("WCX_SYM_SPLIT", 0);
    const int fill = env_int("WCX_SYM_FILL", 1) * slots;        // work items a chunk should offer
    std::vector<SymDesc> descs;
    int total = 0;
    for (int64_t c0 = 0; c0 < NTb; c0 += Cz) {
      SymDesc d;
      d.c0 = (int)c0;
      d.c1 = (int)(c0 + Cz < NTb ? c0 + Cz : NTb);
      d.q_first = (int)(c0 / 4);
      d.n_q = NQb - d.q_first;
      int n_split = d.n_q >= fill ? 1 : (fill + d.n_q - 1) / d.n_q;
      const int max_split = (d.c1 - d.c0) / CTG / 8 > 1 ? (d.c1 - d.c0) / CTG / 8 : 1;
      if (n_split > max_split) n_split = max_split;
      if (split_env > 0) n_split = split_env < max_split ? split_env : max_split;
      if (n_split < 1) n_split = 1;
      if (!descs.empty() && n_split < descs.back().n_split) n_split = descs.back().n_split;   // (monotone:
      d.n_split = n_split;                          //  a quad's exclusive items are its chunks 0 .. L - 1)
      d.item_base = total;
      d.index = (int)descs.size();
      d.pad = 0;
      total += d.n_q * d.n_split;
      descs.push_back(d);
    }
    rc = wcx_upload_small(ctx, d_desc, descs.data(), descs.size() * sizeof(SymDesc));
    if (rc) return rc;
    WCX_HIP(hipMemsetAsync(d_seq, 0, (size_t)(NQb + 1) * 4, st));
    sa.F = F; sa.tinfo = tinfo; sa.tmin = tmin; sa.tchr = tchr; sa.glob = glob; sa.sl = sl; sa.cnt = cnt_out;
    sa.flags = flags; sa.stats = ctx->d_stats; sa.dbg = ctx->debug_flags; sa.cap2 = cap2;
    sa.pool = pool; sa.pool_head = pool_head; sa.pool_ovf = pool_head + 1; sa.pool_cap = pool_cap;
    sa.queue_head = pool_head + 2;
    sa.desc = d_desc; sa.n_desc = (int)descs.size(); sa.total_items = total; sa.seq = d_seq;
    sa.glist_cap = glist_cap;
    sa.gate = nullptr;
    if (sh) { sa.force_records = 1; sa.part = sh->part; sa.n_parts = sh->n_parts; }
    sym_grid = total < slots ? total : slots;
  }
  const float gamma = (float)(16 * NK + 12) * 1.1920929e-7f;
  const unsigned gf = (unsigned)((n_rows + 3) / 4 < 65536 ? (n_rows + 3) / 4 : 65536);
  auto sweep_and_cut = [&](const unsigned int *gate) -> int {
    sa.gate = gate;
    const int e = sym_dispatch(NK, CTG, cfg.lb, cfg.ring, sa, (unsigned)sym_grid, lds_sym, st);
    if (e < 0) {
      wcx_set_error("symmetric screen kernel nk=%d ctg=%d lb=%d ring=%d is not instantiated", NK, CTG, cfg.lb,
                    cfg.ring);
      return (int)WCX_ERR_UNSUPPORTED;
    }
    if (e != 0) {
      wcx_set_error("symmetric screen kernel launch failed: %s", hipGetErrorString((hipError_t)e));
      return (int)WCX_ERR_HIP;
    }
    k_sym_regroup<<<2048, NT, 0, st>>>(pool, pool_head, pool_head + 1, pool_cap, n_rows, sl, cnt_out, flags,
                                       cap2, gate);
    if (cap2 == CAP2)
      k_sym_final<CAP2 / 64><<<gf, NT, 0, st>>>(info, glob, rowpos, n_rows, sl, cnt_out, flags, Dest, k, gamma,
                                                CAP, gate);
    else
      k_sym_final<CAP2_BIG / 64><<<gf, NT, 0, st>>>(info, glob, rowpos, n_rows, sl, cnt_out, flags, Dest, k,
                                                    gamma, REFINE_MAX, gate);
    WCX_HIP(hipGetLastError());
    return (int)WCX_OK;
  };

  // ---- attempt 1: thresholds from the hub counts
  if (use_hub) {
    rc = wcx_timer_begin(ctx, "topk_pre");
    if (rc) return rc;
    CountArgs ca;
    ca.F = F; ca.tchr = tchr; ca.glob = glob; ca.perm = perm; ca.tinfo = tinfo; ca.tmin = tmin;
    ca.Dest = Dest; ca.cnt = cnt_out; ca.flags = flags; ca.stats = ctx->d_stats;
    ca.need = need; ca.n1 = n1_tiles; ca.gate = nullptr;
    const bool hub_pass = env_int("WCX_HUB_APPEND", 0) != 0 && !(ctx->debug_flags & 123) && !sh;
    ca.sl = hub_pass ? sl : nullptr; ca.cap2 = cap2;
    sa.hub_appended = hub_pass ? 1 : 0;
    // the visit list holds the hub groups only: room for twice the rows asked for (the quantile takes a
    // whole histogram bin) + one padding tile per cell; a bigger region is cut off there by the kernel
    const int64_t hub_t = (hub_rows * 2 + 31) / 32 + (int64_t)SYM_NCLS * n_chr + 8;
    const int64_t cap_t = hub_t < NTb ? hub_t : NTb;
    ca.glist_cap = (int)(cap_t / CTG + 64);
    const size_t lds_cnt = (size_t)cfg.ring * (size_t)(CTG * NK * 64) * 16 + (size_t)ca.glist_cap * 4;
    int e = count_dispatch(NK, CTG, cfg.lb, cfg.ring, ca, (unsigned)NQb, lds_cnt, st);
    if (e == 0 && hub_pass) {
      ca.append_pass = 1;
      e = count_dispatch(NK, CTG, cfg.lb, cfg.ring, ca, (unsigned)NQb, lds_cnt, st);
    }
    if (e < 0) {
      wcx_set_error("hub-count kernel nk=%d ctg=%d lb=%d ring=%d is not instantiated", NK, CTG, cfg.lb, cfg.ring);
      return (int)WCX_ERR_UNSUPPORTED;
    }
    if (e != 0) {
      wcx_set_error("hub-count kernel launch failed: %s", hipGetErrorString((hipError_t)e));
      return (int)WCX_ERR_HIP;
    }
    rc = wcx_timer_end(ctx, "topk_pre");
    if (rc) return rc;
    if (sh) {
      // this rank's tile pairs -> records; how many go to which rank (the call synchronises here)
      const int e2 = sym_dispatch(NK, CTG, cfg.lb, cfg.ring, sa, (unsigned)sym_grid, lds_sym, st);
      if (e2 != 0) {
        wcx_set_error("symmetric screen kernel nk=%d ctg=%d: launch failed (%d)", NK, CTG, e2);
        return e2 < 0 ? (int)WCX_ERR_UNSUPPORTED : (int)WCX_ERR_HIP;
      }
      unsigned long long *d_counts = reinterpret_cast<unsigned long long *>(base + o_cnts);
      WCX_HIP(hipMemsetAsync(d_counts, 0, 64 * 8, st));
      k_rec_hist<<<1024, NT, 0, st>>>(pool, pool_head, pool_cap, sh->rb, d_counts);
      WCX_HIP(hipGetLastError());
      rc = wcx_timer_end(ctx, "topk_screen");
      if (rc) return rc;
      SymShardState &Z = *sh->state;
      unsigned int head2[2] = {0, 0};
      WCX_HIP(hipMemcpyAsync(Z.counts, d_counts, 32 * 8, hipMemcpyDeviceToHost, st));
      WCX_HIP(hipMemcpyAsync(head2, pool_head, 8, hipMemcpyDeviceToHost, st));
      WCX_HIP(hipStreamSynchronize(st));
      // Record pool overflow (loose thresholds on data without hubs; data-dependent, per rank): a lost record
      // would be a wrong neighbour list somewhere, so the whole exchange is declared void -- counts of -1
      // reach every peer with the counts' all-to-all, nobody sends records, and every rank redoes its own
      // rows with the exact kernel (wcx_newref_sym_finish_dev, n_recv < 0): the unsharded path's answer to
      // the same overflow.  No rank raises alone, nobody is left waiting in a collective.
      // (WCX_SYM_TEST_POOL_OVF=1, tests: as if the pool had overflowed)
      Z.overflow = head2[1] || head2[0] > pool_cap || env_int("WCX_SYM_TEST_POOL_OVF", 0) != 0;
      if (Z.overflow)
        for (int r = 0; r < 32; ++r) Z.counts[r] = 0;
      Z.valid = true;
      Z.B = B; Z.row_begin = sh->rb.b[sh->part]; Z.row_end = sh->rb.b[sh->part + 1];
      Z.S = S; Z.Sp = Sp; Z.NK = NK; Z.k = k; Z.cap2 = cap2; Z.n_parts = sh->n_parts; Z.part = sh->part;
      Z.tab = tab; Z.rb = sh->rb; Z.dXs = dXs; Z.Xr = Xr; Z.info = info; Z.glob = glob; Z.rowpos = rowpos;
      Z.perm = perm; Z.cnt = cnt_out; Z.Dest = Dest; Z.flags = flags; Z.pool_head = pool_head;
      Z.d_nredo = d_nredo; Z.searched = searched; Z.sl = sl; Z.pool = pool; Z.pool_cap = pool_cap;
      Z.d_counts = d_counts; Z.d_redo = d_redo; Z.d_rtile = d_rtile; Z.d_rlist = d_rlist;
      Z.rscr = base + o_rscr;
      return (int)WCX_OK;
    }
    rc = sweep_and_cut(nullptr);
    if (rc) return rc;
    if (!(ctx->debug_flags & 123)) {      // (the ablations leave every row unfinished: one sweep is what they time)
      k_hub_count_failed<<<512, NT, 0, st>>>(flags, n_rows, d_failed);
      k_hub_verdict<<<1, 64, 0, st>>>(d_failed, d_gate, ctx->d_stats);
    }
    k_gate_reset<<<1024, NT, 0, st>>>(d_gate, n_rows, cnt_out, flags, pool_head, d_seq, NQb + 1);
  } else {
    rc = wcx_timer_begin(ctx, "topk_pre");
    if (rc) return rc;
  }

  // ---- attempt 2 (behind the gate when attempt 1 ran): sampled pre-pass + sweep
  {
    // order A: the two-region best-first order; only its sample region [0, P_s) is used
    WCX_HIP(hipMemsetAsync(cellcnt, 0, (size_t)2 * NCELL * 4, st));
    WCX_HIP(hipMemsetAsync(perm2, 0xff, (size_t)Bpad2 * 4, st));
    k_row_hist<<<gb, NT, 0, st>>>(rbits, rchr, B, SF, 0, glob, rkey, cellcnt, gate2);
    k_scan_cells<<<1, 1024, 0, st>>>(cellcnt, cursor, (int)(P_s - n_s), gate2);
    k_scatter<<<gb, NT, 0, st>>>(rkey, B, cursor, perm2, rowpos2, gate2);
    k_group_mask<<<(unsigned)((P_s / CT + NT - 1) / NT), NT, 0, st>>>(perm2, rchr, P_s / CT, gmask, gate2);
    prep_frag(P_s, perm2, Fs, infs, gate2);
    WCX_HIP(hipGetLastError());
    const int64_t group_bytes = (int64_t)GRr * NK * 32;
    int64_t chunk_groups = ((int64_t)env_int("WCX_SCREEN_CHUNK_KB", NK > 16 ? 16384 : 3072) << 10) / group_bytes;
    if (chunk_groups < 16) chunk_groups = 16;
    if (chunk_groups > 4096) chunk_groups = 4096;
    const size_t lds = (size_t)(cfg.ring >= 2 ? cfg.ring : 2) * (size_t)(CTG * NK * 64) * 16 +
                       (size_t)(chunk_groups + 64) * 4;
    ScreenArgs a;
    a.F = Fs; a.Ft = F; a.info = info; a.glob = glob; a.perm = perm2; a.rowpos = rowpos; a.gmask = gmask;
    a.blocks = d_blocks; a.sl = sl; a.cnt = cnt_out; a.flags = flags; a.g_state = g_state;
    a.stats = ctx->d_stats; a.row_begin = 0; a.n_rows_all = n_rows;
    a.k = k; a.dbg = (ctx->debug_flags & ~3) | env_int("WCX_PRE_DBG", 0); a.n_seg = 1; a.n_blocks = (int)blocks.size();
    a.raw_est = raw_est;
    a.gate = gate2;
    // (lists of at most 5 x 64 entries -- trigger + one group's appends -- take the short cut path of
    //  k_screen: five slices to load, select and write back instead of sixteen)
    int trig_a = 4 * cut_r + 64;
    if (trig_a > 256 && 2 * cut_r + 32 <= 256) trig_a = 256;
    if (trig_a > LIM) trig_a = LIM;
    const int64_t g_end = P_s / GRr;
    bool first = true;
    for (int64_t g0 = 0; g0 < g_end; g0 += chunk_groups) {
      const int64_t g1 = g0 + chunk_groups < g_end ? g0 + chunk_groups : g_end;
      a.g_start = g0; a.g_count = (int)(g1 - g0);
      a.cut_k = cut_r; a.cut_mode = 1; a.trig = trig_a; a.end_cut = g1 == g_end ? 1 : 0;
      a.first = first ? 1 : 0;
      first = false;
      const int e = screen_dispatch(cfg, a, (unsigned)blocks.size(), lds, st);
      if (e < 0) {
        wcx_set_error("screen kernel configuration nk=%d ctg=%d is not instantiated", cfg.nk, cfg.ctg);
        return (int)WCX_ERR_UNSUPPORTED;
      }
      if (e != 0) {
        wcx_set_error("screen kernel launch failed: %s", hipGetErrorString((hipError_t)e));
        return (int)WCX_ERR_HIP;
      }
    }
    k_sym_setup<<<gprep, NT, 0, st>>>(perm, info, glob, g_state, cnt_out, flags, Dest, tinfo, tmin, gate2);
    WCX_HIP(hipGetLastError());
    if (!use_hub) {
      rc = wcx_timer_end(ctx, "topk_pre");
      if (rc) return rc;
    }
    sa.hub_appended = 0;
    rc = sweep_and_cut(gate2);
    if (rc) return rc;
  }
  rc = wcx_timer_end(ctx, "topk_screen");
  if (rc) return rc;
  if (ctx->ev_after_sweep) WCX_HIP(hipEventRecord(ctx->ev_after_sweep, st));   // (wcx_sweep_event)
  if (kick_at >= 1) {
    const bool pending = ctx->rank_pending;
    rc = wcx_aux_kick(ctx);
    if (rc) return rc;
    // (2: the refine waits for the ranking instead of running beside it -- an experiment switch)
    if (kick_at == 2 && pending) WCX_HIP(hipStreamWaitEvent(st, ctx->ev_rank, 0));
  }
  rc = wcx_timer_begin(ctx, "topk_refine");
  if (rc) return rc;
  rc = wcx_refine_launch(ctx, Xr, S, Sp, tab, 0, n_rows, searched, sl, cnt_out, flags, perm, k, d_out_idx,
                         d_out_dist, glob, cap2);
  if (rc) return rc;
  rc = wcx_timer_end(ctx, "topk_refine");
  if (rc) return rc;
  k_collect_redo<<<(unsigned)((n_rows + NT - 1) / NT), NT, 0, st>>>(0, n_rows, searched, flags, tab, d_redo,
                                                                   d_nredo, ctx->d_stats);
  k_redo_plan<<<1, 64, 0, st>>>(tab, d_nredo, d_rtile);
  k_redo_fill<<<(unsigned)((n_rows + NT - 1) / NT), NT, 0, st>>>(0, n_rows, searched, flags, tab, d_nredo,
                                                                d_rlist);
  WCX_HIP(hipGetLastError());
  rc = wcx_topk_exact_redo_launch(ctx, dXs, B, S, d_redo, d_nredo, d_rtile, d_nredo + 1, d_rlist,
                                  base + o_rscr, 0, k, d_out_idx, d_out_dist);
  if (rc) return rc;
  return wcx_timer_end(ctx, "topk");
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
  static const int nk_list[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64};
  int NK = 64;
  for (int v : nk_list)
    if (16 * v >= S + 4) { NK = v; break; }
  // Kernel configuration: small K keeps TWO target tiles per wave in registers (every candidate
  // fragment read from LDS feeds two MFMAs, one barrier per 4 NK MFMAs); large K uses eight
  // waves per workgroup so that 256 targets share every staged candidate group.
  ScreenCfg cfg;
  cfg.nk = NK;
  cfg.prof = (ctx->debug_flags & 4) && (NK == 7 || NK == 32) ? 1 : 0;
  // Kernel configuration (measured on MI355X): 128 targets per workgroup; small K:
  // 64 candidates per iteration, 3 waves per SIMD, LDS-DMA ring of 3; large K: 2 waves per SIMD
  // (the targets' fragments alone take 4 NK registers), LDS-DMA double buffer.
  // More than 508 samples (NK = 40 .. 64): the fragments of a wave's 32 targets alone take 160 .. 256
  // registers -- one wave per SIMD on the unified VGPR + AGPR file, one workgroup per CU, a double
  // buffer of 40 .. 64 KB groups in LDS.
  if (NK <= 8) { cfg.ctg = 2; cfg.tt = 1; cfg.wpb = 4; cfg.lb = 3; cfg.ring = 3; }
  else if (NK <= 32) { cfg.ctg = NK <= 16 ? 2 : 1; cfg.tt = 1; cfg.wpb = 4; cfg.lb = 2; cfg.ring = 2; }
  else { cfg.ctg = 1; cfg.tt = 1; cfg.wpb = 4; cfg.lb = 1; cfg.ring = 2; }
  if (const char *e = getenv("WCX_SCREEN_TILE")) {     // testing / tuning: "ctg,tt,wpb,lb,ring"
    int a = 0, b = 0, c = 0, d = 0, r = 0;
    if (sscanf(e, "%d,%d,%d,%d,%d", &a, &b, &c, &d, &r) == 5) {
      cfg.ctg = a; cfg.tt = b; cfg.wpb = c; cfg.lb = d; cfg.ring = r;
    }
  }
  const int CTG = cfg.ctg;
  const int TGT_WG = 32 * cfg.tt * cfg.wpb;
  const int GRr = CTG * 32;
  // Sampled pre-pass: the rows b = 0 (mod SF) are swept first (own region of the sweep order).
  // (small problems with few samples -- the reference's default 100 kb bins: 27 k rows -- are bound by
  //  their appends like every K < 256 sweep; without estimates the lists fill to the cut trigger first:
  //  100 kb x 100 samples, sampling 0 / 4 / 8 / 16: appends per row 1 540 / 1 230 / 1 080 / 1 075, sweep
  //  1.61 / 1.49 / 1.44 / 1.56 ms.  K >= 256 below 32 768 rows stays without: the symmetric path it would
  //  open is sized and tested for the larger problems.)
  int SF = (B >= 32768) ? 16 : ((B >= 8192 && NK < 16) ? 8 : 0);
  SF = env_int("WCX_SCREEN_SAMPLE", SF);
  if (SF < 2 || SF > 64) SF = 0;
  int64_t n_s = SF ? (B + SF - 1) / SF : 0;                   // rows in the sample
  int64_t P_s = (n_s + CT - 1) / CT * CT;                     // positions of the sample region
  int64_t Bpad = P_s + ((B - n_s) + CT - 1) / CT * CT;
  // regroup the searched row ranges into workgroups of <= TGT_WG rows (same chromosome)
  auto make_blocks = [&](int max_rows) {
    std::vector<ScreenBlock> out;
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
             exact_blocks[j].row0 == sb.row0 + sb.nrows && sb.nrows + exact_blocks[j].nrows <= max_rows) {
        sb.nrows += exact_blocks[j].nrows;
        ++j;
      }
      out.push_back(sb);
      i = j;
    }
    return out;
  };
  const std::vector<ScreenBlock> blocks = make_blocks(TGT_WG);
  int64_t n_iter_groups = Bpad / GRr;
  // Hub-count thresholds (decided further down, needed here for the segment rule): see use_hub1
  const int need1 = (int)(1.18 * k) + 8;
  const int hub_frac1 = env_int("WCX_HUB_FRAC", NK >= 16 ? 32 : 12);
  int64_t hub_rows1 = hub_frac1 > 1 ? B / hub_frac1 : 0;
  if (hub_rows1 < 8 * (int64_t)need1 + 512) hub_rows1 = 8 * (int64_t)need1 + 512;
  auto hub1_possible = [&]() {
    // (an explicit sampling rate / sample rank asks for the sampled pre-pass: tests of that path)
    const int hub1_dflt = (getenv("WCX_SCREEN_SAMPLE") || getenv("WCX_SCREEN_CUT_R")) ? 0 : 1;
    return env_int("WCX_SCREEN_HUB", hub1_dflt) != 0 && NK >= 5 && hub_frac1 > 1 && hub_rows1 * 6 <= B &&
           cfg.wpb == 4 && cfg.ring >= 2;
  };
  // Candidate segments fill the chip when a row shard has few target blocks (multi-GPU builds)
  // and even out the last round of workgroups: work items = blocks x segments.
  int hw_cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
      hw_cus = prop.multiProcessorCount;
  }
  const int wg_per_cu = cfg.lb * 4 / cfg.wpb > 0 ? cfg.lb * 4 / cfg.wpb : 1;
  const int slots = hw_cus * wg_per_cu;
  // (measured on row shards of 1/2 .. 1/8 of the 15 kb problem, profiles/r02/shard_search_*.json: with
  // the sampled pre-pass and 2-3 resident workgroups per CU the split no longer pays -- 1 segment
  // is as fast or faster down to 178 blocks -- so it is off unless asked for)
  int n_seg = 1;
  // ... except for very small shards: the gonosomal passes search ~80-100 blocks of chrX / chrY
  // rows on 512 slots; 1 / 2 / 4 segments: F pass 7.7 / 6.8 / 6.4 ms, M pass 8.8 / 7.6 / 7.4 ms
  // (15 kb, 250 samples each, profiles/r03)
  // Round 6 (hub-count thresholds: the segments' lists are short, the merge cheap): as many segments as fill
  // ONE round of workgroup slots -- F pass, 77 blocks: 4 / 5 / 6 / 8 segments = screen 1.77 / 1.65 / 1.60 /
  // 1.95 ms; M pass, 97 blocks: 4 / 5 / 6 = 2.13 / 1.95 / 2.50 (6 x 97 > 512 slots: a second round)
  // (with hub-count thresholds already from half a round on: 100 kb x 100, 213 blocks on 768 slots, 1 / 2 / 3
  //  segments = sweep 1.27 / 1.02 / 0.94 ms; with the sampled pre-pass segments made that shape slower, round 5)
  const bool hub1_ahead = hub1_possible();
  if ((int)blocks.size() * (hub1_ahead ? 2 : 4) <= slots) {
    int fill = slots / (int)blocks.size();
    if (fill > 8) fill = 8;
    n_seg = env_int("WCX_SCREEN_SEGMENTS_SMALL", fill);
  }
  n_seg = env_int("WCX_SCREEN_SEGMENTS", n_seg);
  if (n_seg < 1) n_seg = 1;
  if (n_seg > 8) n_seg = 8;
  // r = a rank in the sample that the k-th nearest of all candidates stays below with
  // overwhelming probability (mean k/SF of the k nearest fall into the sample; + 10 % for the
  // uneven share of the own chromosome, Poisson tail 1e-6 per row): the estimate admits ~r SF
  // candidates; a row whose estimate fails costs ~0.1 ms in the device-wide redo.
  // WCX_EST_MARGIN=1: the estimate carries the filter margin like a proven threshold (round 2-4).
  // Default: the raw r-th sample value, r chosen for the rank the final cut needs below the estimate --
  // that of the k-th neighbour's filter bound: 1.135 k entries survive the final cut at 15 kb (S = 100
  // and 500), allowed for with 1.18 k.
  const int raw_est = env_int("WCX_EST_MARGIN", 0) ? 0 : 1;
  auto sample_rank = [&](int nseg) {
    if (!SF) return 0;
    // smallest r with P(Poisson(lambda) >= r) <= 1e-6 / n_seg,  lambda = 1.1 k / (SF n_seg)
    const double lambda = 1.1 * (raw_est ? 1.18 : 1.0) * (double)k / ((double)SF * nseg);
    const double target = 1e-6 / nseg;
    double term = exp(-lambda), cdf = 0.0;   // term = P(X = i)
    int i = 0;
    for (; i < 4 * k; ++i) {
      if (1.0 - cdf <= target && (double)i > lambda) break;
      cdf += term;
      term *= lambda / (double)(i + 1);
    }
    int r = i + 1;
    // the sample must hold several times r candidates and r must be well below k
    if (r * 2 > k || P_s / nseg < 16 * (int64_t)r) r = 0;
    // testing: a deliberately unsafe rank makes estimates fail, which the final cut must detect
    // (rows go to the exact kernel; results stay identical)
    const int forced = env_int("WCX_SCREEN_CUT_R", 0);
    if (forced > 0 && forced < k) r = forced;
    return r;
  };
  int cut_r = sample_rank(n_seg);
  // All rows searched against all rows with a sampled pre-pass available: the symmetric sweep
  // (half the matrix work; screen_sym.h).  WCX_SCREEN_SYM=0 keeps the one-directional sweep.
  {
    int64_t covered = 0;
    for (const ScreenBlock &sb : blocks) covered += sb.nrows;
    const int cut_r1 = sample_rank(1);     // (the symmetric sweep has no candidate segments)
    // WCX_SCREEN_SYM: 0 = never, 1 = where it pays (default), 2 = whenever possible (tests).  Small K
    // is bound by the appends, not by the matrix pipe, and the symmetric sweep's hit path is the dearer
    // one (15 kb, symmetric against one-directional: K = 112: 13.9 / 11.3 ms; K = 192: 17.3 / 15.6; K = 256: 17.4 /
    // 20.0; K = 512: 23.4 / 31.5)
    const int sym_mode = env_int("WCX_SCREEN_SYM", 1);
    if (cut_r1 && row_begin == 0 && n_rows == B && covered == B && cfg.tt == 1 && cfg.wpb == 4 &&
        cfg.ring >= 2 && (sym_mode == 2 || (sym_mode == 1 && NK >= 16)))
      return screen_sym_path(ctx, dXs, B, S, chr_cum, n_chr, blocks, cfg, SF, cut_r1, raw_est, slots, k,
                             d_out_idx, d_out_dist);
  }
  // the one-directional sweep keeps k + its filter margin inside shortlists of CAP entries: a larger
  // refsize of a row shard / gonosomal pass goes to the exact kernel
  if (k > KMAX_ONE_DIR)
    return wcx_topk_exact_launch(ctx, dXs, B, S, exact_blocks, row_begin, n_rows, k, d_out_idx, d_out_dist);
  // Thresholds from COUNTS over the low-norm rows instead of the sampled pre-pass (screen_hub1.h; round 6:
  // what round 5 gave the symmetric sweep, for row shards, gonosomal passes and K < 256).  The head region
  // of the sweep order is then the hub region -- 1 / WCX_HUB_FRAC of the rows, at least 8 x the entries
  // wanted below an estimate + the 512 candidates of the moment phase -- whose size only the device knows:
  // the host sizes everything for the bound (one more group of padding at most).  WCX_SCREEN_HUB=0: off.
  const bool use_hub1 = hub1_ahead && k <= KMAX_ONE_DIR;
  // (the count pass always runs 128-row blocks in its own configuration; a sweep with two target tiles per
  //  wave -- WCX_SCREEN_TILE, experiments -- gets its own block list)
  std::vector<ScreenBlock> hub_blocks;
  if (use_hub1 && TGT_WG != 128) hub_blocks = make_blocks(128);
  if (use_hub1) {
    SF = 0;
    n_s = 0;
    P_s = 0;
    Bpad = (B + CT - 1) / CT * CT + CT;
    n_iter_groups = Bpad / GRr;
    cut_r = 0;
  }
  // scratch layout
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_glob = carve(sizeof(ScreenGlobals));
  const size_t o_mean = carve((size_t)S * 8 * (3 + 2 * CSPLIT));   // mean | min | max | partial sums | partial counts
  const int Sp = row_pitch(S);
  const size_t o_xr = carve((size_t)B * Sp * 8 + 256);   // + slack: refine loads whole 128-B chunks
  const size_t o_F = carve((size_t)Bpad * NK * 32);  // Bpad/32 tiles * NK * 1 KiB
  const size_t o_info = carve((size_t)Bpad * sizeof(RowInfo));
  const int64_t n_groups = Bpad / CT;
  const size_t o_perm = carve((size_t)Bpad * 4);
  const size_t o_rpos = carve((size_t)B * 4);
  const size_t o_rbit = carve((size_t)B * 4);
  const size_t o_rfin = carve(use_hub1 ? (size_t)B * 4 : 0);
  const size_t o_hubh = carve(use_hub1 ? (size_t)HUB_BINS * 4 : 0);
  const size_t o_rchr = carve((size_t)B * 4);
  const size_t o_rkey = carve((size_t)B * 4);
  const size_t o_cell = carve((size_t)2 * NCELL * 4);
  const size_t o_curs = carve((size_t)2 * NCELL * 4);
  const size_t o_gmsk = carve((size_t)n_groups * 4);
  const size_t o_sl = carve((size_t)n_seg * n_rows * CAP * 8);
  const size_t o_cnt = carve((size_t)n_seg * n_rows * 4);
  const size_t o_gst = carve((size_t)n_seg * n_rows * 4);
  const size_t o_flag = carve((size_t)n_seg * n_rows * 4);
  const size_t o_srch = carve((size_t)n_rows);
  const size_t o_blk = carve(blocks.size() * sizeof(ScreenBlock));
  const size_t o_hblk = carve(hub_blocks.size() * sizeof(ScreenBlock));
  const size_t o_redo = carve((size_t)n_rows * sizeof(TopkBlock));
  const size_t o_nredo = carve(512);                                   // counters, see k_collect_redo
  const size_t o_rtile = carve(((size_t)n_rows / 64 + 64) * sizeof(TopkBlock));
  const size_t o_rlist = carve((size_t)n_rows * 4);
  const size_t o_rscr = carve(wcx_topk_redo_scratch_bytes(k, B));
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
  TopkBlock *d_redo = reinterpret_cast<TopkBlock *>(base + o_redo);
  unsigned int *d_nredo = reinterpret_cast<unsigned int *>(base + o_nredo);
  TopkBlock *d_rtile = reinterpret_cast<TopkBlock *>(base + o_rtile);
  int32_t *d_rlist = reinterpret_cast<int32_t *>(base + o_rlist);

  hipStream_t st = ctx->stream;
  WCX_HIP(hipMemsetAsync(glob, 0, sizeof(ScreenGlobals), st));
  WCX_HIP(hipMemsetAsync(cnt_out, 0, (size_t)n_seg * n_rows * 4, st));
  WCX_HIP(hipMemsetAsync(flags, 0, (size_t)n_seg * n_rows * 4, st));
  WCX_HIP(hipMemsetAsync(searched, 0, (size_t)n_rows, st));
  WCX_HIP(hipMemsetAsync(d_nredo, 0, 512, st));
  WCX_HIP(hipMemsetAsync(ctx->d_stats, 0, 256, st));
  rc = wcx_upload_small(ctx, d_blocks, blocks.data(), blocks.size() * sizeof(ScreenBlock));
  if (rc) return rc;
  k_mark<<<(unsigned)blocks.size(), 256, 0, st>>>(searched, d_blocks, row_begin);

  rc = wcx_timer_begin(ctx, "topk");
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "topk_prep");
  if (rc) return rc;
  unsigned long long *cmin = reinterpret_cast<unsigned long long *>(cmean + S);
  unsigned long long *cmax = cmin + S;
  double *psum = cmean + 3 * S, *pcnt = psum + (size_t)S * CSPLIT;
  WCX_HIP(hipMemsetAsync(cmin, 0xff, (size_t)S * 8, st));
  WCX_HIP(hipMemsetAsync(cmax, 0, (size_t)S * 8, st));
  k_col_sum<<<dim3((unsigned)S, CSPLIT), NT, 0, st>>>(dXs, B, psum, pcnt, cmin, cmax);
  k_col_stats<<<(unsigned)((S + 63) / 64), 64, 0, st>>>(S, psum, pcnt, cmin, cmax, cmean, glob);
  ChrTab tab;
  tab.n_chr = n_chr;
  for (int c = 0; c < 32; ++c) tab.cum[c] = c < n_chr ? chr_cum[c] : B;
  {
    const unsigned gb = (unsigned)((B + NT - 1) / NT);
    const unsigned gtn = (unsigned)((B + 31) / 32);
    WCX_HIP(hipMemsetAsync(cellcnt, 0, (size_t)2 * NCELL * 4, st));
    WCX_HIP(hipMemsetAsync(perm, 0xff, (size_t)Bpad * 4, st));
    if (use_hub1) {
      unsigned int *rfine = reinterpret_cast<unsigned int *>(base + o_rfin);
      int *hubhist = reinterpret_cast<int *>(base + o_hubh);
      WCX_HIP(hipMemsetAsync(hubhist, 0, (size_t)HUB_BINS * 4, st));
      k_transpose_norm<<<gtn, 256, 0, st>>>(dXs, B, S, Sp, Xr, cmean, tab, glob, rbits, rchr, rfine, hubhist);
      k_hub_cut<<<1, 1024, 0, st>>>(hubhist, (int)hub_rows1, glob);
      k_row_hist<<<gb, NT, 0, st>>>(rbits, rchr, B, 0, 0, glob, rkey, cellcnt, nullptr, rfine);
      k_scan_cells<<<1, 1024, 0, st>>>(cellcnt, cursor, 0, nullptr, glob);
    } else {
      k_transpose_norm<<<gtn, 256, 0, st>>>(dXs, B, S, Sp, Xr, cmean, tab, glob, rbits, rchr, nullptr, nullptr);
      k_row_hist<<<gb, NT, 0, st>>>(rbits, rchr, B, SF, n_seg > 1 ? 1 : 0, glob, rkey, cellcnt);
      k_scan_cells<<<1, 1024, 0, st>>>(cellcnt, cursor, (int)(P_s - n_s));
    }
    k_scatter<<<gb, NT, 0, st>>>(rkey, B, cursor, perm, rowpos);
    k_group_mask<<<(unsigned)((n_groups + NT - 1) / NT), NT, 0, st>>>(perm, rchr, n_groups, gmask);
  }
  const unsigned gprep = (unsigned)((Bpad + 31) / 32);              // one workgroup per 32-row tile
  switch (NK) {
#define WCX_PREP_CASE(N) case N: k_screen_prep<N><<<gprep, NT, 0, st>>>(Xr, Bpad, S, Sp, cmean, perm, glob, F, info); break;
    WCX_PREP_CASE(1) WCX_PREP_CASE(2) WCX_PREP_CASE(3) WCX_PREP_CASE(4) WCX_PREP_CASE(5)
    WCX_PREP_CASE(6) WCX_PREP_CASE(7) WCX_PREP_CASE(8) WCX_PREP_CASE(10) WCX_PREP_CASE(12)
    WCX_PREP_CASE(14) WCX_PREP_CASE(16) WCX_PREP_CASE(20) WCX_PREP_CASE(24) WCX_PREP_CASE(28)
    WCX_PREP_CASE(40) WCX_PREP_CASE(48) WCX_PREP_CASE(56) WCX_PREP_CASE(64)
    default: k_screen_prep<32><<<gprep, NT, 0, st>>>(Xr, Bpad, S, Sp, cmean, perm, glob, F, info); break;
#undef WCX_PREP_CASE
  }
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "topk_prep");
  if (rc) return rc;
  // pending null-sample ranking (auxiliary stream): 0 = start beside the sweep, 1 = beside the
  // refine.  Measured (15 kb, S = 500 / 100): the step takes the same time either way (73.2 / 73.1 ms;
  // serial: 73.6), but beside the sweep the sort's HBM streaming costs the power-limited MFMA loop
  // 6 % (33.4 vs 31.6 ms), beside the refine it costs the refine 2 ms -- so it goes there.
  // (a short refine -- few samples -- cannot hide the 2.4 ms sort: S = 100 is 0.8 ms faster with
  // the sort beside the sweep)
  const int kick_at = env_int("WCX_RANK_KICK", S >= 256 ? 1 : 0);
  if (kick_at == 0) {
    rc = wcx_aux_kick(ctx);
    if (rc) return rc;
  }
  rc = wcx_timer_begin(ctx, "topk_screen");
  if (rc) return rc;

  // candidate chunk per launch: a few MB of fragments (3 MB fits the 4 MB XCD L2)
  const int64_t group_bytes = (int64_t)GRr * NK * 32;
  // (every launch costs ~30 us of prologue per round of workgroups -- target fragments, visit
  // list --, so large K, whose groups are big, takes bigger chunks: L2 misses go to the MALL.
  // Every launch also re-reads the 128 KB of target fragments of each of its workgroups, which is
  // most of what reaches the fabric; fetched bytes per sweep at K = 512 against the chunk size
  // (rocprofv3 FETCH_SIZE x 2, scripts/sweep_chunk_traffic.sh): 2 / 3 / 4 / 8 / 16 / 32 / 64 / 128 MB
  // -> 59 / 44 / 36 / 24.5 / 19.8 / 25 / 45 / 63 GB, sweep 32.7 / 31.8 / 31.5 / 31.3 / 31.2 / 32.5 /
  // 33.7 / 42.7 ms: beyond 16 MB the workgroups of a launch drift apart and stop sharing lines)
  int64_t chunk_groups = ((int64_t)env_int("WCX_SCREEN_CHUNK_KB", NK > 16 ? 16384 : 3072) << 10) / group_bytes;
  // (small shards swept in candidate segments -- the gonosomal passes: ~300 workgroups, less than one
  // round -- have no tail to hide and no L2 to share in step; their launches only cost: 3 / 6 / 12 /
  // 24 MB chunks: F pass screen 4.56 / 3.73 / 3.43 / 3.22 ms, M pass 5.33 / 4.87 / 4.34 / 4.16 ms)
  if (n_seg > 1) chunk_groups = ((int64_t)env_int("WCX_SCREEN_CHUNK_KB_SMALL", 24576) << 10) / group_bytes;
  if (chunk_groups < 16) chunk_groups = 16;
  if (chunk_groups > 4096) chunk_groups = 4096;
  const size_t lds = (size_t)(cfg.ring >= 2 ? cfg.ring : 2) * (size_t)(CTG * NK * 64) * 16 +
                     (size_t)(chunk_groups + 64) * 4;   // + the chunk's visit list
  ScreenArgs a;
  a.F = F; a.Ft = F; a.info = info; a.glob = glob; a.perm = perm; a.rowpos = rowpos; a.gmask = gmask;
  a.blocks = d_blocks; a.sl = sl; a.cnt = cnt_out; a.flags = flags; a.g_state = g_state;
  a.stats = ctx->d_stats; a.row_begin = row_begin; a.n_rows_all = n_rows;
  a.k = k; a.dbg = ctx->debug_flags; a.n_seg = n_seg; a.n_blocks = (int)blocks.size();
  a.raw_est = raw_est;
  // Every chunk launch ends with a partly filled last round of workgroups (1423 blocks on 512
  // slots = 2.78 rounds at 15 kb).  With more than one round of blocks the target blocks are split
  // in two halves that sweep the same chunks on two streams: when one half's launch drains, the
  // other half's workgroups fill the freed slots -- the per-launch tails overlap instead of adding
  // up.  (Blocks are independent: all per-target state is addressed through ScreenBlock::row0.)
  if (use_hub1) {
    rc = wcx_timer_begin(ctx, "topk_pre");
    if (rc) return rc;
    // its own configuration (the default rule for this K), whatever tile the sweep was asked to use
    ScreenCfg hc;
    hc.nk = NK; hc.tt = 1; hc.wpb = 4; hc.prof = 0;
    if (NK <= 8) { hc.ctg = 2; hc.lb = 3; hc.ring = 3; }
    else if (NK <= 32) { hc.ctg = NK <= 16 ? 2 : 1; hc.lb = 2; hc.ring = 2; }
    else { hc.ctg = 1; hc.lb = 1; hc.ring = 2; }
    const ScreenBlock *d_hblocks = d_blocks;
    size_t n_hblocks = blocks.size();
    if (!hub_blocks.empty()) {
      ScreenBlock *dh = reinterpret_cast<ScreenBlock *>(base + o_hblk);
      rc = wcx_upload_small(ctx, dh, hub_blocks.data(), hub_blocks.size() * sizeof(ScreenBlock));
      if (rc) return rc;
      d_hblocks = dh;
      n_hblocks = hub_blocks.size();
    }
    Hub1Args ha;
    ha.F = F; ha.glob = glob; ha.perm = perm; ha.rowpos = rowpos; ha.gmask = gmask; ha.blocks = d_hblocks;
    ha.g_state = g_state; ha.cnt = cnt_out; ha.stats = ctx->d_stats; ha.row_begin = row_begin;
    ha.n_rows_all = n_rows; ha.n_seg = n_seg; ha.need = env_int("WCX_HUB_TEST_FAIL", 0) ? (1 << 28) : need1;
    ha.n1 = env_int("WCX_HUB_N1", 16);
    // the visit list holds the hub groups: room for twice the rows asked for (the quantile takes a whole
    // histogram bin); a bigger region is cut off there by the kernel
    const int hGR = hc.ctg * 32;
    int64_t cap_g = (hub_rows1 * 2 + hGR - 1) / hGR + 2;
    if (cap_g > Bpad / hGR) cap_g = Bpad / hGR;
    if (cap_g > 8192) cap_g = 8192;
    ha.glist_cap = (int)cap_g + 64;
    const size_t lds_h = (size_t)hc.ring * (size_t)(hc.ctg * NK * 64) * 16 + (size_t)ha.glist_cap * 4;
    // (trial thresholds per row: eight are a finer ladder -- fewer rows whose estimate overshoots into an
    //  in-sweep cut -- but cost the count pass 2 x 8 vector instructions per output, which at K <= 128 and
    //  more than a round of workgroups is what bounds it: 15 kb x 100: 4 / 8 trials = pass 1.24 / 1.91 ms,
    //  sweep total 11.59 / 11.89; 100 kb x 100, 213 workgroups: 1.63 / 1.36 ms)
    const int trials = env_int("WCX_HUB1_TRIALS", (NK >= 16 || (int)n_hblocks <= hw_cus * hc.lb) ? 8 : 4);
    int e = wcx_hub1_launch_k1(NK, hc.ctg, hc.lb, hc.ring, trials, ha, (unsigned)n_hblocks, lds_h, st);
    if (e < 0) e = wcx_hub1_launch_k2(NK, hc.ctg, hc.lb, hc.ring, trials, ha, (unsigned)n_hblocks, lds_h, st);
    if (e < 0) e = wcx_hub1_launch_k3(NK, hc.ctg, hc.lb, hc.ring, trials, ha, (unsigned)n_hblocks, lds_h, st);
    if (e < 0) e = wcx_hub1_launch_k4(NK, hc.ctg, hc.lb, hc.ring, trials, ha, (unsigned)n_hblocks, lds_h, st);
    if (e < 0) {
      wcx_set_error("hub-count kernel (one-directional) nk=%d ctg=%d lb=%d ring=%d trials=%d is not instantiated",
                    NK, hc.ctg, hc.lb, hc.ring, trials);
      return (int)WCX_ERR_UNSUPPORTED;
    }
    if (e != 0) {
      wcx_set_error("hub-count kernel launch failed: %s", hipGetErrorString((hipError_t)e));
      return (int)WCX_ERR_HIP;
    }
    rc = wcx_timer_end(ctx, "topk_pre");
    if (rc) return rc;
  }
  int n_streams = ((int)blocks.size() > slots && n_seg == 1) ? 2 : 1;
  n_streams = env_int("WCX_SCREEN_STREAMS", n_streams);
  if (n_streams != 2 || n_seg != 1 || blocks.size() < 2) n_streams = 1;
  hipStream_t st2 = st;
  if (n_streams == 2) {
    if (!ctx->sweep_stream) {
      WCX_HIP(hipStreamCreateWithFlags(&ctx->sweep_stream, hipStreamNonBlocking));
      WCX_HIP(hipEventCreateWithFlags(&ctx->ev_sweep0, hipEventDisableTiming));
      WCX_HIP(hipEventCreateWithFlags(&ctx->ev_sweep1, hipEventDisableTiming));
    }
    st2 = ctx->sweep_stream;
    WCX_HIP(hipEventRecord(ctx->ev_sweep0, st));          // prep and state resets are done
    WCX_HIP(hipStreamWaitEvent(st2, ctx->ev_sweep0, 0));
  }
  const int half0 = n_streams == 2 ? (int)(blocks.size() + 1) / 2 : (int)blocks.size();
  bool first = !use_hub1;          // (hub counts: the per-row state -- threshold, estimate bit -- is there already)
  auto launch = [&](int64_t g0, int64_t g1, int cut_k, int cut_mode, int trig, int end_cut) {
    a.g_start = g0; a.g_count = (int)(g1 - g0);
    a.cut_k = cut_k; a.cut_mode = cut_mode; a.trig = trig; a.end_cut = end_cut;
    a.first = first ? 1 : 0;
    first = false;
    for (int h = 0; h < n_streams; ++h) {
      a.blocks = d_blocks + (h ? half0 : 0);
      a.n_blocks = h ? (int)blocks.size() - half0 : half0;
      const int e = screen_dispatch(cfg, a, (unsigned)(a.n_blocks * n_seg), lds, h ? st2 : st);
      if (e < 0) {
        wcx_set_error("screen kernel configuration nk=%d ctg=%d tt=%d wpb=%d lb=%d ring=%d is not instantiated",
                      cfg.nk, cfg.ctg, cfg.tt, cfg.wpb, cfg.lb, cfg.ring);
        return (int)WCX_ERR_UNSUPPORTED;
      }
      if (e != 0) {
        wcx_set_error("screen kernel launch failed: %s", hipGetErrorString((hipError_t)e));
        return (int)WCX_ERR_HIP;
      }
    }
    return (int)WCX_OK;
  };
  const int trig_main = (ctx->debug_flags >> 8) ? (ctx->debug_flags >> 8) : LIM;   // (diagnostics)
  int64_t g_main = 0;
  if (cut_r) {   // sampled pre-pass: a streaming top-r over the sample region
    g_main = P_s / GRr;
    int trig_a = n_seg > 1 ? 2 * cut_r + 32 : 4 * cut_r + 64;   // (random order: cut more often)
    if (trig_a > LIM) trig_a = LIM;
    for (int64_t g0 = 0; g0 < g_main; g0 += chunk_groups) {
      const int64_t g1 = g0 + chunk_groups < g_main ? g0 + chunk_groups : g_main;
      rc = launch(g0, g1, cut_r, 1, trig_a, g1 == g_main ? 1 : 0);
      if (rc) return rc;
    }
    if (n_seg > 1)
      k_share_thresholds<<<(unsigned)((n_rows + NT - 1) / NT), NT, 0, st>>>(n_rows, n_seg, searched,
                                                                            g_state, cnt_out);
  }
  for (int64_t g0 = g_main; g0 < n_iter_groups; g0 += chunk_groups) {
    const int64_t g1 = g0 + chunk_groups < n_iter_groups ? g0 + chunk_groups : n_iter_groups;
    rc = launch(g0, g1, k, 0, trig_main, g1 == n_iter_groups ? 2 : 0);
    if (rc) return rc;
  }
  if (n_seg > 1) {
    const float gamma = (float)(16 * NK + 12) * 1.1920929e-7f;
    const unsigned gm = (unsigned)((n_rows + 3) / 4 < 65536 ? (n_rows + 3) / 4 : 65536);
    for (int step = 1; step < n_seg; step *= 2) {
      const unsigned pairs = (unsigned)((n_seg - step + 2 * step - 1) / (2 * step));
      k_merge_segments<<<dim3(gm, pairs), NT, 0, st>>>(info, glob, rowpos, row_begin, n_rows, searched, sl,
                                                       cnt_out, flags, g_state, step, n_seg, k, gamma,
                                                       step * 2 >= n_seg ? 1 : 0);
    }
  }
  if (n_streams == 2) {                                   // the second half joins the main stream
    WCX_HIP(hipEventRecord(ctx->ev_sweep1, st2));
    WCX_HIP(hipStreamWaitEvent(st, ctx->ev_sweep1, 0));
  }
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "topk_screen");
  if (rc) return rc;
  if (ctx->ev_after_sweep) WCX_HIP(hipEventRecord(ctx->ev_after_sweep, st));   // (wcx_sweep_event)
  if (kick_at >= 1) {
    const bool pending = ctx->rank_pending;
    rc = wcx_aux_kick(ctx);
    if (rc) return rc;
    // (2: the refine waits for the ranking instead of running beside it -- an experiment switch)
    if (kick_at == 2 && pending) WCX_HIP(hipStreamWaitEvent(st, ctx->ev_rank, 0));
  }
  rc = wcx_timer_begin(ctx, "topk_refine");
  if (rc) return rc;
  rc = wcx_refine_launch(ctx, Xr, S, Sp, tab, row_begin, n_rows, searched, sl, cnt_out, flags, perm,
                         k, d_out_idx, d_out_dist, glob);
  if (rc) return rc;
  rc = wcx_timer_end(ctx, "topk_refine");
  if (rc) return rc;
  // rows the screen could not finish (none on all data seen) are redone exactly, device-driven:
  // redo list and its length never leave the device
  k_collect_redo<<<(unsigned)((n_rows + NT - 1) / NT), NT, 0, st>>>(row_begin, n_rows, searched, flags,
                                                                   tab, d_redo, d_nredo, ctx->d_stats);
  k_redo_plan<<<1, 64, 0, st>>>(tab, d_nredo, d_rtile);
  k_redo_fill<<<(unsigned)((n_rows + NT - 1) / NT), NT, 0, st>>>(row_begin, n_rows, searched, flags, tab,
                                                                d_nredo, d_rlist);
  WCX_HIP(hipGetLastError());
  rc = wcx_topk_exact_redo_launch(ctx, dXs, B, S, d_redo, d_nredo, d_rtile, d_nredo + 1, d_rlist,
                                  base + o_rscr, row_begin, k, d_out_idx, d_out_dist);
  if (rc) return rc;
  return wcx_timer_end(ctx, "topk");
}

// ------------------------------------------------------------------------------------------
// Row-sharded reference-bin search WITH the symmetric sweep (multi-GPU).  Every rank holds all
// of X (one all-gather) and builds the SAME sweep order (deterministic counting sort) and the same
// thresholds (hub counts of all rows: 1.4 ms, not sharded); the tile PAIRS of the sweep are dealt out to
// the ranks item by item, each computed once for both directions, and every hit travels as a 16-byte
// record (row, partner position, screen distance) to the rank that owns the row's list: ONE all-to-all.
void wcx_sym_state_free(wcx_ctx *ctx) {
  delete reinterpret_cast<SymShardState *>(ctx->sym_state);
  ctx->sym_state = nullptr;
}

int wcx_sym_shard_sweep(wcx_ctx *ctx, const double *dXs, int64_t B, int S, const int64_t *chr_cum, int n_chr,
                        int k, int part, int n_parts, const int64_t *row_bounds, int64_t *counts_out) {
  if (!wcx_screen_supported(B, S, k) || n_parts < 1 || n_parts > 32 || B < 32768) {
    wcx_set_error("sharded symmetric sweep: unsupported shape (B=%lld S=%d k=%d parts=%d)", (long long)B, S, k,
                  n_parts);
    return WCX_ERR_UNSUPPORTED;
  }
  static const int nk_list[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64};
  int NK = 64;
  for (int v : nk_list)
    if (16 * v >= S + 4) { NK = v; break; }
  ScreenCfg cfg;
  cfg.nk = NK;
  cfg.prof = 0;
  if (NK <= 8) { cfg.ctg = 2; cfg.tt = 1; cfg.wpb = 4; cfg.lb = 3; cfg.ring = 3; }
  else if (NK <= 32) { cfg.ctg = NK <= 16 ? 2 : 1; cfg.tt = 1; cfg.wpb = 4; cfg.lb = 2; cfg.ring = 2; }
  else { cfg.ctg = 1; cfg.tt = 1; cfg.wpb = 4; cfg.lb = 1; cfg.ring = 2; }
  int hw_cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
      hw_cus = prop.multiProcessorCount;
  }
  const int slots = hw_cus * (cfg.lb * 4 / cfg.wpb > 0 ? cfg.lb * 4 / cfg.wpb : 1);
  if (!ctx->sym_state) ctx->sym_state = new SymShardState();
  SymShardState *Z = reinterpret_cast<SymShardState *>(ctx->sym_state);
  Z->valid = false;
  SymShardCall call;
  call.part = part; call.n_parts = n_parts; call.state = Z;
  call.rb.n = n_parts;
  for (int r = 0; r <= n_parts; ++r) call.rb.b[r] = row_bounds[r];
  for (int r = n_parts + 1; r < 33; ++r) call.rb.b[r] = B;
  if (row_bounds[0] != 0 || row_bounds[n_parts] != B) {
    wcx_set_error("row_bounds must run from 0 to B");
    return WCX_ERR_ARG;
  }
  std::vector<ScreenBlock> none;
  const int rc = screen_sym_path(ctx, dXs, B, S, chr_cum, n_chr, none, cfg, 16, 0, 1, slots, k, nullptr, nullptr,
                                 &call);
  if (rc) return rc;
  for (int r = 0; r < n_parts; ++r) counts_out[r] = Z->overflow ? -1 : (int64_t)Z->counts[r];
  return WCX_OK;
}

int wcx_sym_shard_records(wcx_ctx *ctx, void *d_send) {
  SymShardState *Z = reinterpret_cast<SymShardState *>(ctx->sym_state);
  if (!Z || !Z->valid) {
    wcx_set_error("wcx_newref_sym_records_dev without a sweep");
    return WCX_ERR_ARG;
  }
  unsigned long long cur[32];
  unsigned long long run = 0;
  for (int r = 0; r < 32; ++r) { cur[r] = run; run += r < Z->n_parts ? Z->counts[r] : 0; }
  int rc = wcx_upload_small(ctx, Z->d_counts + 32, cur, sizeof(cur));
  if (rc) return rc;
  if (run)
    k_rec_bucket<<<2048, 1024, 0, ctx->stream>>>(Z->pool, Z->pool_head, Z->pool_cap, Z->rb, Z->d_counts + 32,
                                               reinterpret_cast<uint4 *>(d_send));
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

int wcx_sym_shard_finish(wcx_ctx *ctx, const void *d_recv, int64_t n_recv, int32_t *d_out_idx,
                         double *d_out_dist) {
  SymShardState *Z = reinterpret_cast<SymShardState *>(ctx->sym_state);
  if (!Z || !Z->valid) {
    wcx_set_error("wcx_newref_sym_finish_dev without a sweep");
    return WCX_ERR_ARG;
  }
  Z->valid = false;
  hipStream_t st = ctx->stream;
  const int64_t r0 = Z->row_begin, n_own = Z->row_end - Z->row_begin;
  if (n_own <= 0) return wcx_timer_end(ctx, "topk");
  int rc = wcx_timer_begin(ctx, "topk_cut");
  if (rc) return rc;
  int *cnt = Z->cnt + r0;
  unsigned int *flags = Z->flags + r0;
  WCX_HIP(hipMemsetAsync(cnt, 0, (size_t)n_own * 4, st));
  // n_recv < 0: some rank's record pool overflowed, the exchange is void -- every row of this rank goes to
  // the exact kernel (flags != 0), like the unsharded path's overflow
  if (n_recv < 0) WCX_HIP(hipMemsetAsync(flags, 1, (size_t)n_own * 4, st));
  if (n_recv > 0)
    k_rec_regroup<<<2048, NT, 0, st>>>(reinterpret_cast<const uint4 *>(d_recv), n_recv, r0, n_own, Z->sl, cnt, flags,
                                       Z->cap2);
  const float gamma = (float)(16 * Z->NK + 12) * 1.1920929e-7f;
  const unsigned gf = (unsigned)((n_own + 3) / 4 < 65536 ? (n_own + 3) / 4 : 65536);
  if (Z->cap2 == CAP2)
    k_sym_final<CAP2 / 64><<<gf, NT, 0, st>>>(Z->info, Z->glob, Z->rowpos + r0, n_own, Z->sl, cnt, flags,
                                              Z->Dest + r0, Z->k, gamma, CAP, nullptr);
  else
    k_sym_final<CAP2_BIG / 64><<<gf, NT, 0, st>>>(Z->info, Z->glob, Z->rowpos + r0, n_own, Z->sl, cnt, flags,
                                                  Z->Dest + r0, Z->k, gamma, REFINE_MAX, nullptr);
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "topk_cut");
  if (rc) return rc;
  if (ctx->ev_after_sweep) WCX_HIP(hipEventRecord(ctx->ev_after_sweep, st));
  rc = wcx_aux_kick(ctx);
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "topk_refine");
  if (rc) return rc;
  rc = wcx_refine_launch(ctx, Z->Xr, Z->S, Z->Sp, Z->tab, r0, n_own, Z->searched, Z->sl, cnt, flags, Z->perm,
                         Z->k, d_out_idx, d_out_dist, Z->glob, Z->cap2);
  if (rc) return rc;
  rc = wcx_timer_end(ctx, "topk_refine");
  if (rc) return rc;
  k_collect_redo<<<(unsigned)((n_own + NT - 1) / NT), NT, 0, st>>>(r0, n_own, Z->searched, flags, Z->tab, Z->d_redo,
                                                                  Z->d_nredo, ctx->d_stats);
  k_redo_plan<<<1, 64, 0, st>>>(Z->tab, Z->d_nredo, Z->d_rtile);
  k_redo_fill<<<(unsigned)((n_own + NT - 1) / NT), NT, 0, st>>>(r0, n_own, Z->searched, flags, Z->tab, Z->d_nredo,
                                                               Z->d_rlist);
  WCX_HIP(hipGetLastError());
  rc = wcx_topk_exact_redo_launch(ctx, Z->dXs, Z->B, Z->S, Z->d_redo, Z->d_nredo, Z->d_rtile, Z->d_nredo + 1,
                                  Z->d_rlist, Z->rscr, r0, Z->k, d_out_idx, d_out_dist);
  if (rc) return rc;
  return wcx_timer_end(ctx, "topk");
}
