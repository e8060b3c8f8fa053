// Device side of the multi-GPU plumbing: the RCCL all-gather delivers `world`
// equally padded row shards; these kernels turn that buffer into the layouts the hot path takes
// WITHOUT a second full copy on the host side of torch (no torch.cat):
//   wcx_gather_transpose_dev  padded row shards of X (row-major [B][S]) -> sample-major [S][B]
//                             (the bytes of the reference's F-ordered matrix, newref_tools.py:147)
//   wcx_compact_rows_dev      padded row shards of a row-major table -> dense [B][row_bytes]
// Shard r holds rows [start[r], start[r+1]) (the reference's _get_part boundaries,
// newref_tools.py:244-247) at padded position r * pad_rows.
#include "wcx_common.h"

namespace {

struct ShardTab {
  int world;
  int64_t pad_rows;
  int64_t start[65];
};

__device__ __forceinline__ int64_t padded_row(const ShardTab &t, int64_t b) {
  int r = 0;
  while (r + 1 < t.world && b >= t.start[r + 1]) ++r;
  return (int64_t)r * t.pad_rows + (b - t.start[r]);
}

// src rows are bins (padded layout), S doubles each; dst[j][b] = src[padded(b)][j]
__global__ __launch_bounds__(256) void k_gather_transpose(const double *__restrict__ src, ShardTab t,
                                                          int64_t B, int S, double *__restrict__ dst) {
  __shared__ double tile[32][33];
  const int64_t b0 = (int64_t)blockIdx.x * 32;
  const int j0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
  for (int r = ty; r < 32; r += 8) {
    const int64_t b = b0 + r;
    const int j = j0 + tx;
    tile[r][tx] = (b < B && j < S) ? src[padded_row(t, b) * S + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = j0 + r;
    const int64_t b = b0 + tx;
    if (j < S && b < B) dst[(int64_t)j * B + b] = tile[tx][r];
  }
}

// 16-byte granules: row_bytes must be a multiple of 4; generic 4-byte copy
__global__ __launch_bounds__(256) void k_compact_rows(const uint32_t *__restrict__ src, ShardTab t,
                                                      int64_t B, int64_t row_words,
                                                      uint32_t *__restrict__ dst) {
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint32_t *s = src + padded_row(t, b) * row_words;
    uint32_t *d = dst + b * row_words;
    for (int64_t i = threadIdx.x; i < row_words; i += 256) d[i] = s[i];
  }
}

int make_tab(int world, int64_t pad_rows, int64_t B, ShardTab &t) {
  WCX_ARG(world >= 1 && world <= 64, "world must be 1..64");
  t.world = world;
  t.pad_rows = pad_rows;
  for (int r = 0; r <= world; ++r)   // newref_tools.py:245-246: int(bincount / float(outof) * partnum)
    t.start[r] = (int64_t)((double)B / (double)world * (double)r);
  for (int r = 0; r < world; ++r)
    WCX_ARG(t.start[r + 1] - t.start[r] <= pad_rows, "pad_rows smaller than a shard");
  return WCX_OK;
}

}  // namespace

extern "C" {

int wcx_gather_transpose_dev(wcx_ctx *ctx, const double *d_src, int world, int64_t pad_rows,
                             int64_t B, int S, double *d_dst) {
  WCX_ARG(ctx && d_src && d_dst && B > 0 && S > 0, "bad argument");
  WCX_ARG((S + 31) / 32 < 65536, "S too large");
  WCX_HIP(hipSetDevice(ctx->device));
  ShardTab t;
  int rc = make_tab(world, pad_rows, B, t);
  if (rc) return rc;
  k_gather_transpose<<<dim3((unsigned)((B + 31) / 32), (unsigned)((S + 31) / 32)), 256, 0,
                       ctx->stream>>>(d_src, t, B, S, d_dst);
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

int wcx_compact_rows_dev(wcx_ctx *ctx, const void *d_src, int world, int64_t pad_rows, int64_t B,
                         int64_t row_bytes, void *d_dst) {
  WCX_ARG(ctx && d_src && d_dst && B > 0 && row_bytes > 0 && row_bytes % 4 == 0, "bad argument");
  WCX_HIP(hipSetDevice(ctx->device));
  ShardTab t;
  int rc = make_tab(world, pad_rows, B, t);
  if (rc) return rc;
  const unsigned grid = (unsigned)(B < 65536 ? B : 65536);
  k_compact_rows<<<grid, 256, 0, ctx->stream>>>(reinterpret_cast<const uint32_t *>(d_src), t, B,
                                                row_bytes / 4, reinterpret_cast<uint32_t *>(d_dst));
  WCX_HIP(hipGetLastError());
  return WCX_OK;
}

}  // extern "C"
