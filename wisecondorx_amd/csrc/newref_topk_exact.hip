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
    const unsigned int *__restrict__ n_blocks_dev,
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
  for (unsigned int bi = blockIdx.x; bi < n_blk; bi += gridDim.x) {
  const TopkBlock blk = blocks[bi];
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
      const int64_t ob = (orow0 + r) * (int64_t)k;
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
            st[jj * TM + n] = tval ? row[blk.row0 + n] : 0.0;
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
  WCX_HIP(hipMemsetAsync(d_stats, 0, 32, ctx->stream));
  WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_topk_exact),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  rc = wcx_timer_begin(ctx, "topk");
  if (rc) return rc;
  k_topk_exact<<<(unsigned)blocks.size(), NT, lds, ctx->stream>>>(
      dXs, B, S, d_blocks, nullptr, k, C, scr_d, scr_i, row_begin, d_out_idx, d_out_dist, d_stats);
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "topk");
  if (rc) return rc;
  return WCX_OK;
}

// Device-driven redo (no host round trip): `d_blocks[0 .. *d_count)` was written by the screen's
// k_collect_redo; a fixed grid of WCX_REDO_GRID workgroups strides over it (normally *d_count = 0
// and the launch returns at once).  `scratch` must hold wcx_topk_redo_scratch_bytes(k) bytes.
size_t wcx_topk_redo_scratch_bytes(int k) {
  int C = 1024;
  while (C < k + TN) C <<= 1;
  return (size_t)WCX_REDO_GRID * TM * C * 12;
}

int wcx_topk_exact_redo_launch(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                               const TopkBlock *d_blocks, const unsigned int *d_count, void *scratch,
                               int64_t row_begin, int k, int32_t *d_out_idx, double *d_out_dist) {
  int C = 1024;
  while (C < k + TN) C <<= 1;
  const size_t tile_bytes = (size_t)JC * (TM + TN) * 8;
  const size_t sort_bytes = (size_t)C * 12;
  const size_t lds = HDR + (tile_bytes > sort_bytes ? tile_bytes : sort_bytes);
  double *scr_d = reinterpret_cast<double *>(scratch);
  int *scr_i = reinterpret_cast<int *>(reinterpret_cast<char *>(scratch) +
                                       (size_t)WCX_REDO_GRID * TM * C * 8);
  WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_topk_exact),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  k_topk_exact<<<WCX_REDO_GRID, NT, lds, ctx->stream>>>(dXs, B, S, d_blocks, d_count, k, C, scr_d,
                                                        scr_i, row_begin, d_out_idx, d_out_dist,
                                                        nullptr);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}
