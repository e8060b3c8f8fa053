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

// The null samples are first packed 8 to a 64-byte line:  Xg[sg][b][8] = X[b][sid[8 sg + 0..7]]
// so that ONE gathered cache line serves 8 medians (the gather traffic, not the selection, is
// what bounds this kernel).
__global__ __launch_bounds__(NT) void k_nr_pack(const double *__restrict__ Xs, int64_t B,
                                                const int32_t *__restrict__ sids, int n_ids,
                                                double *__restrict__ Xg) {
  const int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int sg = blockIdx.y;
  if (b >= B) return;
  double v[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int m = sg * 8 + s;
    v[s] = m < n_ids ? Xs[(int64_t)sids[m] * B + b] : 1.0;
  }
  double2 *dst = reinterpret_cast<double2 *>(Xg + ((int64_t)sg * B + b) * 8);
  dst[0] = make_double2(v[0], v[1]);
  dst[1] = make_double2(v[2], v[3]);
  dst[2] = make_double2(v[4], v[5]);
  dst[3] = make_double2(v[6], v[7]);
}

// One wave per (row, group of 8 samples).  blockIdx.x (fastest in dispatch order) walks the rows,
// blockIdx.y the sample groups: at any moment the whole chip gathers from ONE 64*B-byte slab.
template <int IPL>
__global__ __launch_bounds__(NT) void k_null_ratios(
    const double *__restrict__ Xg, int64_t B, const int32_t *__restrict__ idx,
    int64_t row_begin, int64_t n_rows, int k, int n_ids, double *__restrict__ out) {
  const int lane = wcx::lane_id();
  const int wave = threadIdx.x >> 6;
  __shared__ int s_hist[NT / 64][64];
  __shared__ double s_slots[NT / 64][64];
  const int64_t r = (int64_t)blockIdx.x * (NT / 64) + wave;
  if (r >= n_rows) return;
  const int sg = blockIdx.y;
  const double *slab = Xg + (int64_t)sg * B * 8;
  double v[8][IPL];
  unsigned int act = 0;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int t = q * 64 + lane;
    const bool valid = t < k;
    int64_t c = valid ? (int64_t)idx[r * (int64_t)k + t] : 0;
    if (c < 0) c += B;  // NumPy negative index
    act |= valid ? (1u << q) : 0u;
    const double2 *src = reinterpret_cast<const double2 *>(slab + c * 8);
    const double2 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
    v[0][q] = a0.x; v[1][q] = a0.y; v[2][q] = a1.x; v[3][q] = a1.y;
    v[4][q] = a2.x; v[5][q] = a2.y; v[6][q] = a3.x; v[7][q] = a3.y;
  }
  double my_med = 0.0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    bool has_nan = false;
#pragma unroll
    for (int q = 0; q < IPL; ++q) has_nan |= ((act >> q) & 1u) && (v[s][q] != v[s][q]);
    double med;
    if (__any(has_nan)) med = __builtin_nan("");  // np.median propagates NaN
    else med = wcx::wave_median_bucket<IPL>(v[s], act, k, s_hist[wave], s_slots[wave]);
    if (lane == s) my_med = med;
  }
  const int m = sg * 8 + lane;
  if (lane < 8 && m < n_ids) {
    const double xr = slab[(row_begin + r) * 8 + lane];
    out[r * (int64_t)n_ids + m] = log2(xr / my_med);
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
  const int n_sg = (n_ids + 7) / 8;
  const size_t xg_bytes = (size_t)n_sg * B * 64;
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, xg_bytes + (size_t)n_ids * 4 + 256, &scr);
  if (rc) return rc;
  double *Xg = reinterpret_cast<double *>(scr);
  int32_t *d_sids = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(scr) + xg_bytes);
  rc = wcx_upload_small(ctx, d_sids, sample_ids, (size_t)n_ids * 4);
  if (rc) return rc;
  rc = wcx_timer_begin(ctx, "null_ratios");
  if (rc) return rc;
  k_nr_pack<<<dim3((unsigned)((B + NT - 1) / NT), (unsigned)n_sg), NT, 0, ctx->stream>>>(
      dXs, B, d_sids, n_ids, Xg);
  const dim3 grid((unsigned)((n_rows + NT / 64 - 1) / (NT / 64)), (unsigned)n_sg);
#define WCX_NR_LAUNCH(IPL)                                                                    \
  k_null_ratios<IPL><<<grid, NT, 0, ctx->stream>>>(Xg, B, d_idx, row_begin, n_rows, k, n_ids, \
                                                   d_out)
  const int ipl = (k + 63) / 64;
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
