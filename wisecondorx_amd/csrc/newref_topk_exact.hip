// Exact fp64 reference-bin search (replaces newref_tools.py:255-278).
//
// One workgroup owns TM=64 target rows of ONE chromosome and streams every candidate row
// outside that chromosome in tiles of TN=64.  A 256-thread workgroup computes the 64x64
// squared distances of a tile in registers (4x4 per thread) with the reference's exact
// arithmetic: acc = acc + (c - t)*(c - t) for j = 0..S-1, every operation separately rounded
// (fp contraction is OFF for this file: an FMA would change the last bit and with it the
// neighbour order).  Distances not above the row's current k-th-smallest bound `tau` are
// appended to a per-row shortlist in HBM scratch (capacity C); when a shortlist might
// overflow on the next tile it is sorted by (distance, index) in LDS and cut back to k,
// which tightens tau.  The final sort writes the k neighbours in the reference's order.
//
// Roofline: fp64 VALU bound -- 3 flop per (pair, sample); HBM traffic is one read of X per
// XCD wave of workgroups (L2/MALL resident) plus the 12*k bytes of output per row.
#include "wcx_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int TM = 64;   // target rows per workgroup
constexpr int TN = 64;   // candidate rows per tile
constexpr int NT = 256;  // threads per workgroup
constexpr int JC = 32;   // samples staged per chunk
constexpr int HDR = 1024;  // bytes of persistent LDS header (tau, cnt, flag)

__device__ __forceinline__ bool key_less(double da, int ia, double db, int ib) {
  return (da < db) || (da == db && ia < ib);
}

// Workgroup-wide bitonic sort of n (power of two) (distance,index) pairs held in LDS.
__device__ void bitonic_sort_lds(double *sd, int *si, int n) {
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (n >> 1); t += NT) {
        int lo = 2 * t - (t & (stride - 1));
        int hi = lo + stride;
        bool asc = ((lo & size) == 0);
        double dl = sd[lo], dh = sd[hi];
        int il = si[lo], ih = si[hi];
        bool hi_less = key_less(dh, ih, dl, il);
        if (hi_less == asc) {
          sd[lo] = dh; sd[hi] = dl;
          si[lo] = ih; si[hi] = il;
        }
      }
    }
  }
  __syncthreads();
}

// n_blocks_dev == nullptr: one workgroup per entry of `blocks` (grid = number of blocks), shortlist
// scratch row = output row.  Otherwise (device-driven redo of rows the screen flagged): the
// number of blocks is read from the device, the fixed grid strides over them and the scratch rows
// are per workgroup (TM rows each).
__global__ __launch_bounds__(NT) void k_topk_exact(
    const double *__restrict__ Xs, int64_t B, int S, const TopkBlock *__restrict__ blocks,
    const unsigned int *__restrict__ n_blocks_dev, unsigned int first_blk,
    const int32_t *__restrict__ rowlist,
    int k, int C, double *__restrict__ scr_d, int *__restrict__ scr_i, int64_t row_begin,
    int32_t *__restrict__ out_idx, double *__restrict__ out_dist,
    unsigned long long *__restrict__ stats) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *tau = reinterpret_cast<double *>(smem);             // [TM]
  int *cnt = reinterpret_cast<int *>(smem + TM * 8);          // [TM]
  int *flag = reinterpret_cast<int *>(smem + TM * 12);        // [1]
  double *st = reinterpret_cast<double *>(smem + HDR);        // [JC][TM]
  double *sc = st + JC * TM;                                  // [JC][TN]
  double *sd = reinterpret_cast<double *>(smem + HDR);        // [C]   (aliases the tiles)
  int *si = reinterpret_cast<int *>(smem + HDR + (size_t)C * 8);  // [C]

  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const unsigned int n_blk = n_blocks_dev ? *n_blocks_dev : gridDim.x;
  const int lim = C - TN;  // a row may receive at most TN appends per tile
  unsigned long long n_compact = 0;
  for (unsigned int bi = first_blk + blockIdx.x; bi < n_blk; bi += gridDim.x) {
  const TopkBlock blk = blocks[bi];
  // rowlist != nullptr (redo of many flagged rows): blk.row0 is an offset into `rowlist`, whose
  // entries are the (scattered) target rows of this tile -- all of one chromosome
  auto trow = [&](int n) -> int64_t { return rowlist ? (int64_t)rowlist[blk.row0 + n] : blk.row0 + n; };
  const int64_t orow0 = blk.row0 - row_begin;                       // output row of local row 0
  const int64_t srow0 = n_blocks_dev ? (int64_t)blockIdx.x * TM : orow0;   // scratch row

  __syncthreads();
  if (tid < TM) { tau[tid] = 1e10; cnt[tid] = 0; }
  if (tid == 0) *flag = 0;
  __syncthreads();

  auto compact_row = [&](int r, bool final_pass) {
    const int n = cnt[r];
    const int64_t base = (srow0 + r) * (int64_t)C;
    for (int t = tid; t < C; t += NT) {
      if (t < n) { sd[t] = scr_d[base + t]; si[t] = scr_i[base + t]; }
      else { sd[t] = HUGE_VAL; si[t] = 0x7fffffff; }
    }
    bitonic_sort_lds(sd, si, C);
    const int keep = n < k ? n : k;
    if (!final_pass) {
      for (int t = tid; t < keep; t += NT) { scr_d[base + t] = sd[t]; scr_i[base + t] = si[t]; }
      if (tid == 0) { cnt[r] = keep; tau[r] = (n >= k) ? sd[k - 1] : 1e10; }
    } else {
      const int64_t ob = (rowlist ? trow(r) - row_begin : orow0 + r) * (int64_t)k;
      for (int t = tid; t < k; t += NT) {
        out_idx[ob + t] = t < keep ? si[t] : -1;
        out_dist[ob + t] = t < keep ? sd[t] : 1e10;
      }
    }
    __syncthreads();
  };

  const int64_t own = blk.ce - blk.cs;
  for (int range = 0; range < 2; ++range) {
    const int64_t lo = range == 0 ? 0 : blk.ce;
    const int64_t hi = range == 0 ? blk.cs : B;
    const int64_t shift = range == 0 ? 0 : own;  // candidate row -> stored index
    for (int64_t g0 = lo; g0 < hi; g0 += TN) {
      double acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;

      for (int j0 = 0; j0 < S; j0 += JC) {
        const int jn = (S - j0) < JC ? (S - j0) : JC;
        __syncthreads();  // previous chunk (or a compaction) is done with the LDS tiles
        {
          const int n = tid & 63;
          const bool tval = n < blk.nrows;
          const bool cval = (g0 + n) < hi;
          for (int jj = tid >> 6; jj < jn; jj += 4) {
            const double *row = Xs + (int64_t)(j0 + jj) * B;
            st[jj * TM + n] = tval ? row[trow(n)] : 0.0;
            sc[jj * TN + n] = cval ? row[g0 + n] : 0.0;
          }
        }
        __syncthreads();
#pragma unroll 2
        for (int jj = 0; jj < jn; ++jj) {
          const double2 t01 = *reinterpret_cast<const double2 *>(&st[jj * TM + ty * 2]);
          const double2 t23 = *reinterpret_cast<const double2 *>(&st[jj * TM + 32 + ty * 2]);
          const double2 c01 = *reinterpret_cast<const double2 *>(&sc[jj * TN + tx * 2]);
          const double2 c23 = *reinterpret_cast<const double2 *>(&sc[jj * TN + 32 + tx * 2]);
          const double t[4] = {t01.x, t01.y, t23.x, t23.y};
          const double c[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              const double diff = c[b] - t[a];   // chr_data - target (newref_tools.py:260)
              const double sq = diff * diff;
              acc[a][b] = acc[a][b] + sq;
            }
        }
      }

      // threshold filter + append
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int m = (a >> 1) * 32 + ty * 2 + (a & 1);
        if (m < blk.nrows) {
          const double tr = tau[m];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int64_t g = g0 + (b >> 1) * 32 + tx * 2 + (b & 1);
            const double d = acc[a][b];
            if (g < hi && d <= tr && d < 1e10) {
              const int pos = atomicAdd(&cnt[m], 1);
              const int64_t base = (srow0 + m) * (int64_t)C;
              scr_d[base + pos] = d;
              scr_i[base + pos] = (int)(g - shift);
              if (pos + 1 > lim) *flag = 1;
            }
          }
        }
      }
      __syncthreads();
      if (*flag) {
        for (int r = 0; r < blk.nrows; ++r) {
          if (cnt[r] > lim) { compact_row(r, false); ++n_compact; }
        }
        if (tid == 0) *flag = 0;
        __syncthreads();
      }
    }
  }
  for (int r = 0; r < blk.nrows; ++r) compact_row(r, true);
  }
  if (tid == 0 && stats && n_compact) atomicAdd(&stats[2], n_compact);
}

// Fast redo of a HANDFUL of flagged rows (the normal case when the screen's sampled estimate fails
// for a row): the candidate sweep of one row is spread over the whole device instead of one
// workgroup.  k_redo_dist writes the row's exact distances to all candidates outside its chromosome
// (same arithmetic as k_topk_exact: sequential, separately rounded) as order-preserving 64-bit keys;
// k_redo_select finds the k-th smallest key by an 8 x 8-bit radix select, collects everything below
// it plus the lowest-index ties, and rank-sorts the k pairs by (distance, index).
constexpr int RS_NT = 1024;

__global__ __launch_bounds__(NT) void k_redo_dist(const double *__restrict__ Xs, int64_t B, int S,
                                                  const TopkBlock *__restrict__ blocks,
                                                  const unsigned int *__restrict__ n_blocks_dev,
                                                  unsigned long long *__restrict__ keys) {
  const unsigned int n = *n_blocks_dev;
  if (n > (unsigned)WCX_REDO_FAST) return;          // many rows: the tiled redo takes all of them
  const int64_t nchunk = (B + NT - 1) / NT;
  for (int64_t w = blockIdx.x; w < (int64_t)n * nchunk; w += gridDim.x) {
    const unsigned int slot = (unsigned int)(w / nchunk);
    const int64_t c = (w % nchunk) * NT + threadIdx.x;
    const TopkBlock blk = blocks[slot];
    if (c >= B || (c >= blk.cs && c < blk.ce)) continue;
    double acc = 0.0;
    for (int j = 0; j < S; ++j) {
      const double *row = Xs + (int64_t)j * B;
      const double diff = row[c] - row[blk.row0];
      const double sq = diff * diff;
      acc = acc + sq;
    }
    const int64_t stored = c < blk.cs ? c : c - (blk.ce - blk.cs);
    keys[(int64_t)slot * B + stored] = acc < 1e10 ? (unsigned long long)__double_as_longlong(acc) : ~0ull;
  }
}

__global__ __launch_bounds__(RS_NT) void k_redo_select(int64_t B, const TopkBlock *__restrict__ blocks,
                                                       const unsigned int *__restrict__ n_blocks_dev,
                                                       const unsigned long long *__restrict__ keys,
                                                       int k, int64_t row_begin,
                                                       int32_t *__restrict__ out_idx,
                                                       double *__restrict__ out_dist) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *sd = reinterpret_cast<double *>(smem);                  // [k]
  int *si = reinterpret_cast<int *>(smem + (size_t)k * 8);        // [k]
  __shared__ unsigned int hist[256];
  __shared__ unsigned int wtot[RS_NT / 64];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned int s_rem, s_cnt;
  const unsigned int n = *n_blocks_dev;
  if (n > (unsigned)WCX_REDO_FAST) return;
  const int tid = threadIdx.x;
  for (unsigned int slot = blockIdx.x; slot < n; slot += gridDim.x) {
    const TopkBlock blk = blocks[slot];
    const int64_t Bc = B - (blk.ce - blk.cs);
    const unsigned long long *kb = keys + (int64_t)slot * B;
    __syncthreads();
    int mine = 0;
    for (int64_t i = tid; i < Bc; i += RS_NT) mine += kb[i] != ~0ull;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (mine) atomicAdd(&s_cnt, (unsigned)mine);
    __syncthreads();
    const int kk = (int)((unsigned)k < s_cnt ? (unsigned)k : s_cnt);
    __syncthreads();
    if (tid == 0) { s_prefix = 0; s_rem = (unsigned)kk; s_cnt = 0; }
    if (kk > 0) {
      for (int pass = 7; pass >= 0; --pass) {
        const int shift = pass * 8;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for (int64_t i0 = tid; i0 < Bc; i0 += 4 * RS_NT) {
          unsigned long long kv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + (int64_t)u * RS_NT;
            kv[u] = i < Bc ? kb[i] : ~0ull;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool in = kv[u] != ~0ull && (pass == 7 || (kv[u] >> (shift + 8)) == (prefix >> (shift + 8)));
            if (in) atomicAdd(&hist[(unsigned)(kv[u] >> shift) & 255u], 1u);
          }
        }
        __syncthreads();
        if (tid == 0) {
          unsigned int cum = 0, rem = s_rem;
          int b = 0;
          for (; b < 255; ++b) {
            if (cum + hist[b] >= rem) break;
            cum += hist[b];
          }
          s_rem = rem - cum;
          s_prefix = prefix | ((unsigned long long)b << shift);
        }
        __syncthreads();
      }
    }
    __syncthreads();
    const unsigned long long T = s_prefix;
    const int m = (int)s_rem;          // ties at T to take (lowest indices first)
    const int n_lt = kk - m;
    int tie_base = 0;
    if (kk > 0) {
      for (int64_t base = 0; base < Bc; base += RS_NT) {
        const int64_t i = base + tid;
        const unsigned long long key = i < Bc ? kb[i] : ~0ull;
        if (key < T) {
          const unsigned int pos = atomicAdd(&s_cnt, 1u);
          sd[pos] = __longlong_as_double((long long)key);
          si[pos] = (int)i;
        }
        const bool tie = key == T;
        const int nt = __syncthreads_count(tie);
        if (nt && tie_base < m) {
          const unsigned long long bal = __ballot(tie);
          const int lane = tid & 63, wv = tid >> 6;
          if (lane == 0) wtot[wv] = (unsigned)__popcll(bal);
          __syncthreads();
          int p = __popcll(bal & ((1ull << lane) - 1ull));
          for (int w2 = 0; w2 < wv; ++w2) p += (int)wtot[w2];
          if (tie && tie_base + p < m) {
            sd[n_lt + tie_base + p] = __longlong_as_double((long long)key);
            si[n_lt + tie_base + p] = (int)i;
          }
          __syncthreads();
        }
        tie_base += nt;
      }
    }
    __syncthreads();
    const int64_t ob = (blk.row0 - row_begin) * (int64_t)k;
    for (int t = tid; t < kk; t += RS_NT) {
      const double d = sd[t];
      const int ix = si[t];
      int rank = 0;
      for (int j = 0; j < kk; ++j) rank += key_less(sd[j], si[j], d, ix);
      out_idx[ob + rank] = ix;
      out_dist[ob + rank] = d;
    }
    for (int t = kk + tid; t < k; t += RS_NT) { out_idx[ob + t] = -1; out_dist[ob + t] = 1e10; }
  }
}

__global__ void k_fill_dummy(int32_t *idx, double *dist, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { idx[i] = 0; dist[i] = 1.0; }
}

}  // namespace

int wcx_fill_dummy_rows(wcx_ctx *ctx, int32_t *d_idx, double *d_dist, int64_t row_lo,
                        int64_t row_hi, int k) {
  int64_t n = (row_hi - row_lo) * (int64_t)k;
  if (n <= 0) return WCX_OK;
  k_fill_dummy<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(
      d_idx + row_lo * (int64_t)k, d_dist + row_lo * (int64_t)k, n);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

int wcx_topk_exact_launch(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                          const std::vector<TopkBlock> &blocks, int64_t row_begin,
                          int64_t n_rows, int k, int32_t *d_out_idx, double *d_out_dist) {
  if (blocks.empty()) return WCX_OK;
  int C = 1024;
  while (C < k + TN) C <<= 1;
  if (C > 8192) {
    wcx_set_error("refsize %d too large for the exact top-k kernel (max %d)", k, 8192 - TN);
    return WCX_ERR_UNSUPPORTED;
  }
  const size_t tile_bytes = (size_t)JC * (TM + TN) * 8;
  const size_t sort_bytes = (size_t)C * 12;
  const size_t lds = HDR + (tile_bytes > sort_bytes ? tile_bytes : sort_bytes);

  // scratch: [n_rows][C] distances + [n_rows][C] indices + block table + stats
  const size_t sd_bytes = (size_t)n_rows * C * 8;
  const size_t si_bytes = (size_t)n_rows * C * 4;
  const size_t blk_bytes = blocks.size() * sizeof(TopkBlock);
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, sd_bytes + si_bytes + blk_bytes + 64, &scr);
  if (rc) return rc;
  double *scr_d = reinterpret_cast<double *>(scr);
  int *scr_i = reinterpret_cast<int *>(reinterpret_cast<char *>(scr) + sd_bytes);
  TopkBlock *d_blocks =
      reinterpret_cast<TopkBlock *>(reinterpret_cast<char *>(scr) + sd_bytes + si_bytes);
  unsigned long long *d_stats = ctx->d_stats;
  rc = wcx_upload_small(ctx, d_blocks, blocks.data(), blk_bytes);
  if (rc) return rc;
  WCX_HIP(hipMemsetAsync(d_stats, 0, 256, ctx->stream));   // (all counters: the screen's too -- this call did not screen)
  WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_topk_exact),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  rc = wcx_timer_begin(ctx, "topk");
  if (rc) return rc;
  k_topk_exact<<<(unsigned)blocks.size(), NT, lds, ctx->stream>>>(
      dXs, B, S, d_blocks, nullptr, 0u, nullptr, k, C, scr_d, scr_i, row_begin, d_out_idx, d_out_dist, d_stats);
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "topk");
  if (rc) return rc;
  return WCX_OK;
}

// Device-driven redo (no host round trip): `d_blocks[0 .. *d_count)` (one-row blocks) was written
// by the screen's k_collect_redo (normally *d_count = 0 and every launch returns at once).  Up to
// WCX_REDO_FAST rows take the device-wide path (k_redo_dist + k_redo_select, ~0.1 ms per row).
// More rows (data the fp16 screen cannot resolve, e.g. a tenth of the rows being high-variance
// outliers) were grouped by chromosome into tiles of <= 64 rows (`d_tiles`, `d_rowlist`, see
// k_redo_plan): a fixed grid of WCX_REDO_GRID workgroups of the blocked exact kernel strides over
// the tiles, each tile sweeping the candidates once for its 64 gathered rows.
// `scratch` must hold wcx_topk_redo_scratch_bytes(k, B) bytes.
static size_t redo_blocked_bytes(int k) {
  int C = 1024;
  while (C < k + TN) C <<= 1;
  return ((size_t)WCX_REDO_GRID * TM * C * 12 + 255) / 256 * 256;
}

size_t wcx_topk_redo_scratch_bytes(int k, int64_t B) {
  return redo_blocked_bytes(k) + (size_t)WCX_REDO_FAST * (size_t)B * 8;
}

int wcx_topk_exact_redo_launch(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                               const TopkBlock *d_blocks, const unsigned int *d_count,
                               const TopkBlock *d_tiles, const unsigned int *d_ntiles,
                               const int32_t *d_rowlist, void *scratch, int64_t row_begin, int k,
                               int32_t *d_out_idx, double *d_out_dist) {
  int C = 1024;
  while (C < k + TN) C <<= 1;
  if (C > 8192) {
    wcx_set_error("refsize %d too large for the exact top-k kernel (max %d)", k, 8192 - TN);
    return WCX_ERR_UNSUPPORTED;
  }
  unsigned long long *keys =
      reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(scratch) + redo_blocked_bytes(k));
  k_redo_dist<<<1024, NT, 0, ctx->stream>>>(dXs, B, S, d_blocks, d_count, keys);
  WCX_HIP(hipGetLastError());
  const size_t sel_lds = (size_t)k * 12;
  WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_redo_select),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
  k_redo_select<<<WCX_REDO_FAST, RS_NT, sel_lds, ctx->stream>>>(B, d_blocks, d_count, keys, k, row_begin,
                                                               d_out_idx, d_out_dist);
  WCX_HIP(hipGetLastError());

  const size_t tile_bytes = (size_t)JC * (TM + TN) * 8;
  const size_t sort_bytes = (size_t)C * 12;
  const size_t lds = HDR + (tile_bytes > sort_bytes ? tile_bytes : sort_bytes);
  double *scr_d = reinterpret_cast<double *>(scratch);
  int *scr_i = reinterpret_cast<int *>(reinterpret_cast<char *>(scratch) +
                                       (size_t)WCX_REDO_GRID * TM * C * 8);
  WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_topk_exact),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  k_topk_exact<<<WCX_REDO_GRID, NT, lds, ctx->stream>>>(dXs, B, S, d_tiles, d_ntiles, 0u, d_rowlist, k, C,
                                                        scr_d, scr_i, row_begin, d_out_idx, d_out_dist,
                                                        nullptr);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}
