// Exact fp64 refine of the screened shortlists (see newref_topk_screen.hip for the pipeline).
#include "wave_sort.h"
#include "wcx_common.h"
#include "screen_common.h"

#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------
// Pair (distance, index) wave bitonic sort, lane-minor layout (see wave_sort.h).
template <int IPL>
__device__ __forceinline__ void wave_sort_pairs(double (&d)[IPL], int (&ix)[IPL]) {
  constexpr int N = 64 * IPL;
  const int lane = wcx::lane_id();
#pragma unroll
  for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride >= 1; stride >>= 1) {
      if (stride >= 64) {
        const int rs = stride >> 6;
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          if ((r & rs) == 0) {
            const bool asc = (((r * 64) & size) == 0);
            const double a = d[r], b = d[r | rs];
            const int ia = ix[r], ib = ix[r | rs];
            const bool b_less = (b < a) || (b == a && ib < ia);
            if (b_less == asc) { d[r] = b; d[r | rs] = a; ix[r] = ib; ix[r | rs] = ia; }
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const bool asc = (((r * 64 + lane) & size) == 0);
          const bool lower = ((lane & stride) == 0);
          const double pd = wcx::shfl_xor_f64(d[r], stride);
          const int pi = __shfl_xor(ix[r], stride, 64);
          const bool p_less = (pd < d[r]) || (pd == d[r] && pi < ix[r]);
          const bool want_min = (lower == asc);
          const bool take = want_min ? p_less : !p_less;
          if (take) { d[r] = pd; ix[r] = pi; }
        }
      }
    }
  }
}

// One wave per target row: exact distances of the shortlisted candidates, sort, emit top k.
//
// Lane l owns shortlist entries e = q*64 + l.  Candidate rows live in the row-major copy Xr
// (stride Sp doubles).  A lane walking its own row would touch a different cache line than every
// other lane on every load (64 tag look-ups per instruction, ~3 visits per line); instead the wave
// loads 16-double chunks COALESCED -- 8 lanes per candidate, 128 contiguous bytes -- into a
// padded per-wave LDS tile and each lane then reads its own candidate's 16 values back
// (conflict-free: 144-byte row pitch).  The sums still advance strictly left to right
// (newref_tools.py:260 arithmetic, unfused).
constexpr int RCH = 16;                 // doubles per chunk
constexpr int RPITCH = RCH + 2;         // LDS row pitch in doubles (144 B)

// Exact distances of 64 candidate rows (row ids in g_s, one per lane) to the target row held in
// xt_s, each summed strictly left to right; returns the lane's candidate's distance.
// NL = how many of the eight 8-row load instructions per chunk carry a real candidate: a list's last
// pass is partly padding (339 entries on average = five full passes + 19 lanes), and a padding group's
// loads -- the target row again, L1 hits -- still cost the texture addresser its 16 cycles each; they
// are left out (the lanes of an unloaded group compute on stale LDS values and are discarded).
template <int NL>
__device__ __forceinline__ double pass_distances(const double *__restrict__ Xr, int S, int Sp,
                                                 double *__restrict__ tile,
                                                 const double *__restrict__ xt_s,
                                                 const int *__restrict__ g_s) {
  const int lane = wcx::lane_id();
  const int sub = lane >> 3, part = lane & 7;               // load role: candidate sub, 16-B piece
  int64_t base[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) base[i] = (int64_t)g_s[i * 8 + sub] * Sp + part * 2;
  double acc = 0.0;
  // the chunk loop is a chain of L2 round trips: the next chunk's loads are issued before the
  // current one is consumed (three chunks = 24 KB per wave in flight)
  // (eight named registers per buffer: arrays carried around the loop end up in scratch)
  const double2 z2 = make_double2(0.0, 0.0);
  double2 va0 = z2, va1 = z2, va2 = z2, va3 = z2, va4 = z2, va5 = z2, va6 = z2, va7 = z2, vb0 = z2, vb1 = z2,
          vb2 = z2, vb3 = z2, vb4 = z2, vb5 = z2, vb6 = z2, vb7 = z2, vc0 = z2, vc1 = z2, vc2 = z2, vc3 = z2,
          vc4 = z2, vc5 = z2, vc6 = z2, vc7 = z2;
#define WCX_LD(i, C0) (*reinterpret_cast<const double2 *>(Xr + base[i] + (C0)))
#define WCX_LDI(P, i, C0) if constexpr (NL > i) P##i = WCX_LD(i, C0);
#define WCX_FETCH(P, C0)                                                                         \
  WCX_LDI(P, 0, C0) WCX_LDI(P, 1, C0) WCX_LDI(P, 2, C0) WCX_LDI(P, 3, C0)                        \
  WCX_LDI(P, 4, C0) WCX_LDI(P, 5, C0) WCX_LDI(P, 6, C0) WCX_LDI(P, 7, C0)
#define WCX_ST(i, V)                                                                             \
  if constexpr (NL > i) *reinterpret_cast<double2 *>(&tile[((i) * 8 + sub) * RPITCH + part * 2]) = V
#define WCX_PUT(P)                                                                               \
  WCX_ST(0, P##0); WCX_ST(1, P##1); WCX_ST(2, P##2); WCX_ST(3, P##3);                            \
  WCX_ST(4, P##4); WCX_ST(5, P##5); WCX_ST(6, P##6); WCX_ST(7, P##7);
  auto dot = [&](int c0) __attribute__((always_inline)) {
    __builtin_amdgcn_wave_barrier();
    const int jn = (S - c0) < RCH ? (S - c0) : RCH;
    const double *mine = &tile[lane * RPITCH];
    if (jn == RCH) {
#pragma unroll
      for (int jj = 0; jj < RCH; jj += 2) {
        const double2 cv = *reinterpret_cast<const double2 *>(mine + jj);
        const double2 tv = *reinterpret_cast<const double2 *>(xt_s + c0 + jj);
        double diff = cv.x - tv.x;
        double sq = diff * diff;
        acc = acc + sq;
        diff = cv.y - tv.y; sq = diff * diff; acc = acc + sq;
      }
    } else {
      for (int jj = 0; jj < jn; ++jj) {
        const double diff = mine[jj] - xt_s[c0 + jj];
        const double sq = diff * diff;
        acc = acc + sq;
      }
    }
    __builtin_amdgcn_wave_barrier();                       // chunk fully consumed
  };
  // coalesced loads of chunk [c0, c0+16) of 64 candidate rows: 8 instructions x 8 rows x 128 B
  WCX_FETCH(va, 0)
  if (RCH < S) { WCX_FETCH(vb, RCH) }
  for (int c0 = 0; c0 < S; c0 += 3 * RCH) {
    if (c0 + 2 * RCH < S) { WCX_FETCH(vc, c0 + 2 * RCH) }
    WCX_PUT(va)
    dot(c0);
    if (c0 + RCH < S) {
      if (c0 + 3 * RCH < S) { WCX_FETCH(va, c0 + 3 * RCH) }
      WCX_PUT(vb)
      dot(c0 + RCH);
      if (c0 + 2 * RCH < S) {
        if (c0 + 4 * RCH < S) { WCX_FETCH(vb, c0 + 4 * RCH) }
        WCX_PUT(vc)
        dot(c0 + 2 * RCH);
      }
    }
  }
#undef WCX_LD
#undef WCX_LDI
#undef WCX_FETCH
#undef WCX_ST
#undef WCX_PUT
  return acc;
}

// Sort the wave's (distance, index) pairs and emit the k best (NaN / >= 1e10 never admitted).
template <int IPL>
__device__ __forceinline__ void sort_emit(double (&d)[IPL], int (&ix)[IPL], int k,
                                          int32_t *__restrict__ oi, double *__restrict__ od) {
  const int lane = wcx::lane_id();
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const bool ok = (ix[q] != 0x7fffffff) && (d[q] < 1e10);   // NaN / >= 1e10 never admitted
    if (!ok) { d[q] = HUGE_VAL; ix[q] = 0x7fffffff; }
  }
  wave_sort_pairs<IPL>(d, ix);
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int e = q * 64 + lane;
    if (e < k) {
      const bool ok = d[q] < 1e10;
      oi[e] = ok ? ix[q] : -1;
      od[e] = ok ? d[q] : 1e10;
    }
  }
}


template <int IPL>
__device__ __forceinline__ void refine_row(const double *__restrict__ Xr, int S, int Sp,
                                           int64_t row, int64_t cs, int64_t own,
                                           const uint2 *__restrict__ sl_row,
                                           const int *__restrict__ perm, int n, int k,
                                           int32_t *__restrict__ oi, double *__restrict__ od,
                                           double *__restrict__ tile, double *__restrict__ xt_s,
                                           int *__restrict__ g_s) {
  const int lane = wcx::lane_id();
  double d[IPL];
  int ix[IPL];
  const double *xt = Xr + row * (int64_t)Sp;
  for (int j = lane; j < Sp; j += 64) xt_s[j] = xt[j];     // target row -> LDS (broadcast reads)
  // full passes first (64 real candidates each), then the list's last, partly filled one with only the
  // load groups that carry candidates
  const int n_full = __builtin_amdgcn_readfirstlane(n >> 6);
  const int nv = __builtin_amdgcn_readfirstlane(n & 63);   // real candidates of the last pass (0: none)
  // every pass's candidate rows are looked up before the first pass (list entry -> sweep position ->
  // row: two dependent gathers, paid once per row instead of once per pass; at S = 100 a pass is only
  // seven chunks long and these round trips were a third of it)
  constexpr bool AHEAD = IPL <= 16;                          // (IPL = 32 has no registers to spare)
  int gq[AHEAD ? IPL : 1];
  auto lookup = [&](int q, int &id) __attribute__((always_inline)) {
    const int e = q * 64 + lane;
    int g = (int)row;                                        // padding lanes read the target row
    id = 0x7fffffff;
    if (e < n) {
      g = perm[sl_row[e].y];                                // shortlists hold sweep positions
      id = g < cs ? g : g - (int)own;                        // own-chromosome-excluded index
    }
    return g;
  };
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    d[q] = 0.0;
    ix[q] = 0x7fffffff;
    if constexpr (AHEAD) {
      gq[q] = (int)row;
      if (q * 64 < n) gq[q] = lookup(q, ix[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    if (q >= n_full) continue;                               // wave-uniform
    if constexpr (AHEAD) g_s[lane] = gq[q];
    else g_s[lane] = lookup(q, ix[q]);
    __builtin_amdgcn_wave_barrier();
    d[q] = pass_distances<8>(Xr, S, Sp, tile, xt_s, g_s);
    __builtin_amdgcn_wave_barrier();
  }
  if (nv) {
    int id = 0x7fffffff, g = (int)row;
    if constexpr (AHEAD) {
#pragma unroll
      for (int q = 0; q < IPL; ++q)
        if (q == n_full) { g = gq[q]; id = ix[q]; }
    } else {
      g = lookup(n_full, id);
    }
    g_s[lane] = g;
    __builtin_amdgcn_wave_barrier();
    double acc;
    switch ((nv + 7) >> 3) {
      case 1: acc = pass_distances<1>(Xr, S, Sp, tile, xt_s, g_s); break;
      case 2: acc = pass_distances<2>(Xr, S, Sp, tile, xt_s, g_s); break;
      case 3: acc = pass_distances<3>(Xr, S, Sp, tile, xt_s, g_s); break;
      case 4: acc = pass_distances<4>(Xr, S, Sp, tile, xt_s, g_s); break;
      case 5: acc = pass_distances<5>(Xr, S, Sp, tile, xt_s, g_s); break;
      case 6: acc = pass_distances<6>(Xr, S, Sp, tile, xt_s, g_s); break;
      case 7: acc = pass_distances<7>(Xr, S, Sp, tile, xt_s, g_s); break;
      default: acc = pass_distances<8>(Xr, S, Sp, tile, xt_s, g_s); break;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < IPL; ++q)
      if (q == n_full) { d[q] = acc; ix[q] = id; }
  }
  sort_emit<IPL>(d, ix, k, oi, od);
}

// Rows whose shortlist fits 64 IPL entries (IPL per lane; 8 -> 512: the common case, refsize <= 448;
// 16 / 32 for larger refsizes).
template <int IPL>
__global__ __launch_bounds__(NT) void k_refine(const double *__restrict__ Xr, int S, int Sp,
                                               ChrTab chr, int64_t row_begin, int64_t n_rows,
                                               const unsigned char *__restrict__ searched,
                                               const uint2 *__restrict__ sl,
                                               const int *__restrict__ cnt_out,
                                               const unsigned int *__restrict__ flags,
                                               const int *__restrict__ perm, int k,
                                               int32_t *__restrict__ out_idx,
                                               double *__restrict__ out_dist,
                                               ScreenGlobals *__restrict__ glob,
                                               unsigned long long *__restrict__ stats, int sls) {
  extern __shared__ __align__(16) unsigned char rsm[];
  const int wave = threadIdx.x >> 6;
  const size_t per_wave = (size_t)(64 * RPITCH + Sp) * 8 + 64 * 4;
  double *tile = reinterpret_cast<double *>(rsm + wave * per_wave);
  double *xt_s = tile + 64 * RPITCH;
  int *g_s = reinterpret_cast<int *>(xt_s + Sp);
  const int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * NT) >> 6;
  for (int64_t r = w0; r < n_rows; r += nw) {
    if (!searched[r]) continue;
    if (flags[r]) {
      if (wcx::lane_id() == 0) atomicAdd(&glob->n_overflow, 1u);
      continue;
    }
    const int n = cnt_out[r] & 0x3fffffff;
    if (wcx::lane_id() == 0) atomicAdd(&stats[5], (unsigned long long)n);   // pairs re-evaluated exactly
    if (n > 64 * IPL) continue;     // k_refine_big
    const int64_t row = row_begin + r;
    int64_t cs = 0, ce = chr.cum[0];
    for (int c = 1; c < chr.n_chr && row >= ce; ++c) { cs = ce; ce = chr.cum[c]; }
    refine_row<IPL>(Xr, S, Sp, row, cs, ce - cs, sl + r * (int64_t)sls, perm, n, k,
                  out_idx + r * (int64_t)k, out_dist + r * (int64_t)k, tile, xt_s, g_s);
  }
}

// Rare rows with more entries than the wave kernel takes (n_small < n <= REFINE_MAX): one workgroup per
// row, LDS bitonic sort.
__global__ __launch_bounds__(NT) void k_refine_big(const double *__restrict__ Xr, int S, int Sp,
                                                   ChrTab chr, int64_t row_begin, int64_t n_rows,
                                                   const unsigned char *__restrict__ searched,
                                                   const uint2 *__restrict__ sl,
                                                   const int *__restrict__ cnt_out,
                                                   const unsigned int *__restrict__ flags,
                                                   const int *__restrict__ perm, int k,
                                                   int32_t *__restrict__ out_idx,
                                                   double *__restrict__ out_dist, int sls, int n_small) {
  constexpr int CAP = REFINE_MAX;
  __shared__ double sd[CAP];
  __shared__ int si[CAP];
  for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    if (!searched[r] || flags[r]) continue;
    const int n = cnt_out[r] & 0x3fffffff;
    if (n <= n_small) continue;
    const int64_t row = row_begin + r;
    int64_t cs = 0, ce = chr.cum[0];
    for (int c = 1; c < chr.n_chr && row >= ce; ++c) { cs = ce; ce = chr.cum[c]; }
    const int64_t own = ce - cs;
    const double *xt = Xr + row * (int64_t)Sp;
    const uint2 *sl_row = sl + r * (int64_t)sls;
    __syncthreads();
    for (int e = threadIdx.x; e < CAP; e += NT) {
      double acc = HUGE_VAL;
      int ci = 0x7fffffff;
      if (e < n) {
        const int64_t g = perm[sl_row[e].y];
        ci = (int)(g < cs ? g : g - own);
        const double *xc = Xr + g * (int64_t)Sp;
        acc = 0.0;
        for (int j = 0; j < S; ++j) {
          const double diff = xc[j] - xt[j];
          const double sq = diff * diff;
          acc = acc + sq;
        }
        if (!(acc < 1e10)) { acc = HUGE_VAL; ci = 0x7fffffff; }
      }
      sd[e] = acc;
      si[e] = ci;
    }
    // workgroup bitonic sort by (distance, index)
    for (int size = 2; size <= CAP; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        __syncthreads();
        for (int t = threadIdx.x; t < CAP / 2; t += NT) {
          const int lo = 2 * t - (t & (stride - 1));
          const int hi = lo + stride;
          const bool asc = ((lo & size) == 0);
          const double dl = sd[lo], dh = sd[hi];
          const int il = si[lo], ih = si[hi];
          const bool hi_less = (dh < dl) || (dh == dl && ih < il);
          if (hi_less == asc) { sd[lo] = dh; sd[hi] = dl; si[lo] = ih; si[hi] = il; }
        }
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < k; e += NT) {
      const bool ok = e < CAP && sd[e] < 1e10;
      out_idx[r * (int64_t)k + e] = ok ? si[e] : -1;
      out_dist[r * (int64_t)k + e] = ok ? sd[e] : 1e10;
    }
  }
}

}  // namespace

int wcx_refine_launch(wcx_ctx *ctx, const double *Xr, int S, int Sp, const ChrTab &tab,
                      int64_t row_begin, int64_t n_rows, const unsigned char *searched,
                      const uint2 *sl, const int *cnt_out, const unsigned int *flags,
                      const int *perm, int k, int32_t *d_out_idx, double *d_out_dist,
                      ScreenGlobals *glob, int sl_stride) {
  const unsigned gref = (unsigned)((n_rows + 3) / 4 < 65536 ? (n_rows + 3) / 4 : 65536);
  const size_t rlds = (NT / 64) * ((size_t)(64 * RPITCH + Sp) * 8 + 64 * 4);
  // entries per lane of the wave kernel: a final shortlist holds ~1.13 k entries
  const int ipl = k <= 448 ? 8 : (k <= 900 ? 16 : 32);
#define WCX_REFINE(I)                                                                              \
  {                                                                                                \
    WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_refine<I>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));           \
    k_refine<I><<<gref, NT, rlds, ctx->stream>>>(Xr, S, Sp, tab, row_begin, n_rows, searched, sl,  \
                                                 cnt_out, flags, perm, k, d_out_idx, d_out_dist,   \
                                                 glob, ctx->d_stats, sl_stride);                   \
  }
  if (ipl == 8) WCX_REFINE(8) else if (ipl == 16) WCX_REFINE(16) else WCX_REFINE(32)
#undef WCX_REFINE
  const unsigned gbig = (unsigned)(n_rows < 2048 ? n_rows : 2048);
  k_refine_big<<<gbig, NT, 0, ctx->stream>>>(Xr, S, Sp, tab, row_begin, n_rows, searched, sl,
                                             cnt_out, flags, perm, k, d_out_idx, d_out_dist, sl_stride,
                                             64 * ipl);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}
