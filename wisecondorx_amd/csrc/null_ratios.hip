// Null ratios (replaces newref_tools.py:210-223).
//
//   out[r][m] = log2( x_m[row_begin+r] / median_t x_m[idx[r][t]] ),  x_m = sample sid[m]
//
// One wave owns one (row, sample) pair at a time: 64 lanes gather the k reference values
// (the sample vector is a contiguous double[B] slice of the sample-major matrix, L2
// resident), sort them in registers (wave_sort.h) and take the median.  The index row is
// applied to the FULL bin vector exactly like the reference does (newref_tools.py:219-221);
// index -1 (padding) wraps to the last bin as NumPy's negative indexing does.
//
// Roofline: HBM/L2-gather bound in principle (k*4 bytes of indices per row are read once per
// workgroup and reused for all samples); in practice the register sort (VALU + ds_bpermute)
// dominates -- see DESIGN.md.
#include "wave_sort.h"
#include "wcx_common.h"

namespace {

constexpr int NT = 256;  // 4 waves per workgroup

// samples per wave-visit: the index row is loaded once and reused for MS sample vectors
constexpr int MS = 4;

template <int IPL>
__global__ __launch_bounds__(NT) void k_null_ratios(
    const double *__restrict__ Xs, int64_t B, const int32_t *__restrict__ idx,
    int64_t row_begin, int64_t n_rows, int k, const int32_t *__restrict__ sids, int n_ids,
    double *__restrict__ out) {
  const int lane = wcx::lane_id();
  const int wave = threadIdx.x >> 6;
  // blockIdx.x (fastest in dispatch order) walks the rows, blockIdx.y the sample groups: at any
  // moment the whole chip gathers from the same few sample vectors (1.5 MB each at 15 kb),
  // which therefore stay L2-resident.
  const int64_t r = (int64_t)blockIdx.x * (NT / 64) + wave;
  if (r >= n_rows) return;
  int64_t g[IPL];
  unsigned int act = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int t = q * 64 + lane;
    const bool valid = t < k;
    int64_t c = valid ? (int64_t)idx[r * (int64_t)k + t] : 0;
    if (c < 0) c += B;  // NumPy negative index
    g[q] = c;
    act |= valid ? (1u << q) : 0u;
  }
  const int m0 = blockIdx.y * MS;
  for (int m = m0; m < m0 + MS && m < n_ids; ++m) {
    const double *x = Xs + (int64_t)sids[m] * B;
    double v[IPL];
    bool has_nan = false;
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
      const double val = ((act >> q) & 1u) ? x[g[q]] : 0.0;
      has_nan |= (val != val);
      v[q] = val;
    }
    double med;
    if (__any(has_nan)) med = __builtin_nan("");  // np.median propagates NaN
    else med = wcx::wave_median_select<IPL>(v, act, k);
    if (lane == 0) out[r * (int64_t)n_ids + m] = log2(x[row_begin + r] / med);
  }
}

}  // namespace

extern "C" {

int wcx_null_ratios_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                        const int32_t *d_idx, int64_t row_begin, int64_t row_end, int k,
                        const int32_t *sample_ids, int n_ids, double *d_out) {
  WCX_ARG(ctx && dXs && d_idx && sample_ids && d_out, "NULL argument");
  WCX_ARG(B > 0 && S > 0 && k > 0 && n_ids >= 0, "bad sizes");
  WCX_ARG(0 <= row_begin && row_begin <= row_end && row_end <= B, "bad row range");
  for (int i = 0; i < n_ids; ++i)
    WCX_ARG(sample_ids[i] >= 0 && sample_ids[i] < S, "sample id out of range");
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t n_rows = row_end - row_begin;
  if (n_rows == 0 || n_ids == 0) return WCX_OK;
  if (k > 64 * 32) {
    wcx_set_error("refsize %d too large for the null-ratio kernel (max 2048)", k);
    return WCX_ERR_UNSUPPORTED;
  }
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, (size_t)n_ids * 4, &scr);
  if (rc) return rc;
  int32_t *d_sids = reinterpret_cast<int32_t *>(scr);
  rc = wcx_upload_small(ctx, d_sids, sample_ids, (size_t)n_ids * 4);
  if (rc) return rc;
  const dim3 grid((unsigned)((n_rows + NT / 64 - 1) / (NT / 64)), (unsigned)((n_ids + MS - 1) / MS));
  rc = wcx_timer_begin(ctx, "null_ratios");
  if (rc) return rc;
#define WCX_NR_LAUNCH(IPL)                                                              \
  k_null_ratios<IPL><<<grid, NT, 0, ctx->stream>>>(dXs, B, d_idx, row_begin, n_rows, k, \
                                                   d_sids, n_ids, d_out)
  const int ipl = (k + 63) / 64;
  if (ipl <= 1) WCX_NR_LAUNCH(1);
  else if (ipl <= 2) WCX_NR_LAUNCH(2);
  else if (ipl <= 3) WCX_NR_LAUNCH(3);
  else if (ipl <= 4) WCX_NR_LAUNCH(4);
  else if (ipl <= 5) WCX_NR_LAUNCH(5);
  else if (ipl <= 6) WCX_NR_LAUNCH(6);
  else if (ipl <= 8) WCX_NR_LAUNCH(8);
  else if (ipl <= 16) WCX_NR_LAUNCH(16);
  else WCX_NR_LAUNCH(32);
#undef WCX_NR_LAUNCH
  WCX_HIP(hipGetLastError());
  return wcx_timer_end(ctx, "null_ratios");
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
