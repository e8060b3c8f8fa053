// Segment z-scores and the null-ratio matrix they are scored against.
// a17 replaces overall_tools.get_z_score (overall_tools.py:88-119): HBM-bound column sums.
// (The segmentation itself -- row a16 -- lives in cbs_seg.hip.)
#include <algorithm>
#include <cmath>
#include <cstring>

#include "wave_sort.h"
#include "wcx_common.h"

namespace {

// ------------------------------------------------------------------ a17 segment z
// Segment z in two steps so that long segments (a whole chromosome = 16 k bins at 15 kb) do not
// serialise on one workgroup: per-chunk partial weighted sums of every null column (chunks of
// <= SZ_CHUNK bins, accumulated in bin order), then one workgroup per segment adds its chunks in
// order -- deterministic, the same association for every launch.
constexpr int SZ_CHUNK = 256;

__global__ __launch_bounds__(128) void k_segz_partial(const double *__restrict__ r,
                                                      const double *__restrict__ w,
                                                      const double *__restrict__ nr, int m,
                                                      int64_t nb,   // rows of nr: bin = b mod nb (batches)
                                                      const int64_t *__restrict__ cb0,
                                                      const int64_t *__restrict__ cb1,
                                                      double *__restrict__ pnum,
                                                      double *__restrict__ pden,
                                                      int *__restrict__ pany) {
  const int c = blockIdx.x;
  const int j = threadIdx.x;
  if (j >= m) return;
  double num = 0.0, den = 0.0;
  int any = 0;
  const int64_t b0 = cb0[c], b1 = cb1[c];
  int64_t row = b0 % nb;                            // (kept incrementally: no 64-bit modulo per bin)
  for (int64_t b = b0; b < b1; b += 4) {
    // four bins' loads in flight; the sums still run in bin order
    double v[4], wb[4];
    bool on[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      on[u] = b + u < b1 && r[b + u] != 0.0;        // overall_tools.py:98-100
      int64_t rw = row + u;
      if (rw >= nb) rw %= nb;
      v[u] = on[u] ? nr[rw * m + j] : 0.0;
      wb[u] = on[u] ? w[b + u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (on[u] && fabs(v[u]) < HUGE_VAL) { num += v[u] * wb[u]; den += wb[u]; any = 1; }   // :101-110
    row += 4;
    if (row >= nb) row %= nb;
  }
  pnum[(int64_t)c * m + j] = num;
  pden[(int64_t)c * m + j] = den;
  pany[(int64_t)c * m + j] = any;
}

__global__ __launch_bounds__(128) void k_segment_z(const double *__restrict__ pnum,
                                                   const double *__restrict__ pden,
                                                   const int *__restrict__ pany, int m,
                                                   const int *__restrict__ chunk0,
                                                   const double *__restrict__ seg_r, int n_seg,
                                                   double *__restrict__ out_z,
                                                   double *__restrict__ out_nnull) {
  const int s = blockIdx.x;
  if (s >= n_seg) return;
  __shared__ double avg[128];
  const int j = threadIdx.x;
  double a = __builtin_nan("");
  if (j < m) {
    double num = 0.0, den = 0.0;
    bool any = false;
    for (int c = chunk0[s]; c < chunk0[s + 1]; ++c) {
      num += pnum[(int64_t)c * m + j];
      den += pden[(int64_t)c * m + j];
      any |= pany[(int64_t)c * m + j] != 0;
    }
    if (any) a = num / den;
  }
  avg[j] = a;
  __syncthreads();
  if (j == 0) {
    double sum = 0.0;
    int cnt = 0;
    for (int q = 0; q < m; ++q) if (fabs(avg[q]) < HUGE_VAL) { sum += avg[q]; ++cnt; }   // :111
    double z = __builtin_nan("");
    if (cnt > 0) {
      const double mean = sum / cnt;
      double ss = 0.0;
      for (int q = 0; q < m; ++q) if (fabs(avg[q]) < HUGE_VAL) { const double e = avg[q] - mean; ss += e * e; }
      const double sd = sqrt(ss / cnt);                // np.ma.std: population
      z = (seg_r[s] - mean) / sd;                      // :113
      if (z == z) { z = z < 1000.0 ? z : 1000.0; z = z > -1000.0 ? z : -1000.0; }   // :114-115
    }
    out_z[s] = z;
    if (out_nnull) out_nnull[s] = (double)cnt;
  }
}

__global__ __launch_bounds__(128) void k_inflate_rows(const double *__restrict__ src,
                                                      const int32_t *__restrict__ row_of,
                                                      int64_t n_bins, int m, double *__restrict__ dst) {
  for (int64_t b = blockIdx.x; b < n_bins; b += gridDim.x) {
    const int32_t r = row_of[b];
    for (int j = threadIdx.x; j < m; j += 128) dst[b * m + j] = r >= 0 ? src[(int64_t)r * m + j] : 0.0;
  }
}

}  // namespace

extern "C" {

int wcx_set_null_matrix(wcx_ctx *ctx, const double *nr, int64_t n_bins, int m) {
  WCX_ARG(ctx, "ctx is NULL");
  WCX_HIP(hipSetDevice(ctx->device));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->d_nullm) { WCX_HIP(hipFree(ctx->d_nullm)); ctx->d_nullm = nullptr; }
  ctx->nullm_bins = 0;
  ctx->nullm_m = 0;
  if (!nr) return WCX_OK;
  WCX_ARG(n_bins > 0 && m > 0 && m <= 128, "bad sizes (m <= 128)");
  const size_t bytes = (size_t)n_bins * m * 8;
  if (hipMalloc(reinterpret_cast<void **>(&ctx->d_nullm), bytes) != hipSuccess) {
    wcx_set_error("hipMalloc(%zu) for the null-ratio matrix failed", bytes);
    return WCX_ERR_NOMEM;
  }
  WCX_HIP(hipMemcpyAsync(ctx->d_nullm, nr, bytes, hipMemcpyHostToDevice, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  ctx->nullm_bins = n_bins;
  ctx->nullm_m = m;
  return WCX_OK;
}

// Device variant: d_nr holds the rows of the MASKED bins (B x m, as the reference build leaves
// them in HBM); mask[n_bins] (host, 0/1) says which unmasked bins they belong to.  The matrix is
// inflated on the device (masked-out rows = 0, like predict_tools.py:163-170) -- no host trip.
int wcx_set_null_matrix_dev(wcx_ctx *ctx, const double *d_nr, int64_t B, int m,
                            const unsigned char *mask, int64_t n_bins) {
  WCX_ARG(ctx && d_nr && mask, "NULL argument");
  WCX_ARG(B > 0 && n_bins >= B && m > 0 && m <= 128, "bad sizes (m <= 128)");
  WCX_HIP(hipSetDevice(ctx->device));
  // the bin -> row map depends on the mask only: built and uploaded once per mask (a predict batch,
  // or the bench's step, attaches a table per call), so that the call is one kernel and no host wait
  unsigned long long h = 1469598103934665603ull ^ (unsigned long long)n_bins;
  int64_t sel = 0;
  for (int64_t i = 0; i < n_bins;) {
    if (i + 8 <= n_bins) {                      // eight bytes per step (the per-byte chain costs 0.3 ms at 15 kb)
      unsigned long long wd;
      memcpy(&wd, mask + i, 8);
      h = (h ^ wd) * 1099511628211ull;
      if ((wd & ~0x0101010101010101ull) == 0) sel += __builtin_popcountll(wd);
      else for (int u = 0; u < 8; ++u) sel += mask[i + u] != 0;
      i += 8;
    } else {
      h = (h ^ mask[i]) * 1099511628211ull;
      sel += mask[i] != 0;
      ++i;
    }
  }
  WCX_ARG(sel == B, "mask does not select B bins");
  const size_t bytes = (size_t)n_bins * m * 8;
  if (!(ctx->d_nullm && ctx->nullm_bins == n_bins && ctx->nullm_m == m)) {
    WCX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_nullm) { WCX_HIP(hipFree(ctx->d_nullm)); ctx->d_nullm = nullptr; }
    ctx->nullm_bins = 0;
    if (hipMalloc(reinterpret_cast<void **>(&ctx->d_nullm), bytes) != hipSuccess) {
      wcx_set_error("hipMalloc(%zu) for the null-ratio matrix failed", bytes);
      return WCX_ERR_NOMEM;
    }
  }
  // (a hash hit is confirmed on the mask itself: n_bins bytes of memcmp against the kept copy -- a
  //  collision would misplace null rows silently)
  const bool same_mask = ctx->d_nullsrc && ctx->nullsrc_bins == n_bins && ctx->nullsrc_hash == h &&
                         ctx->nullsrc_mask.size() == (size_t)n_bins &&
                         memcmp(ctx->nullsrc_mask.data(), mask, (size_t)n_bins) == 0;
  if (!same_mask) {
    std::vector<int32_t> src((size_t)n_bins, -1);
    int64_t j = 0;
    for (int64_t i = 0; i < n_bins; ++i)
      if (mask[i]) src[(size_t)i] = (int32_t)j++;
    WCX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->nullsrc_bins != n_bins) {
      if (ctx->d_nullsrc) { WCX_HIP(hipFree(ctx->d_nullsrc)); ctx->d_nullsrc = nullptr; }
      ctx->nullsrc_bins = 0;
      if (hipMalloc(reinterpret_cast<void **>(&ctx->d_nullsrc), (size_t)n_bins * 4) != hipSuccess) {
        wcx_set_error("hipMalloc(%zu) for the null-ratio row map failed", (size_t)n_bins * 4);
        return WCX_ERR_NOMEM;
      }
    }
    WCX_HIP(hipMemcpyAsync(ctx->d_nullsrc, src.data(), (size_t)n_bins * 4, hipMemcpyHostToDevice, ctx->stream));
    WCX_HIP(hipStreamSynchronize(ctx->stream));   // (src is a host temporary)
    ctx->nullsrc_bins = n_bins;
    ctx->nullsrc_hash = h;
    ctx->nullsrc_mask.assign(mask, mask + n_bins);
  }
  k_inflate_rows<<<(unsigned)(n_bins < 65536 ? n_bins : 65536), 128, 0, ctx->stream>>>(
      d_nr, ctx->d_nullsrc, n_bins, m, ctx->d_nullm);
  WCX_HIP(hipGetLastError());
  ctx->nullm_bins = n_bins;
  ctx->nullm_m = m;
  return WCX_OK;
}

static int segment_z_impl(wcx_ctx *ctx, const double *r, const double *w, bool rw_on_device,
                          const double *nr, int m, const int64_t *chr_off, int n_chr,
                          const double *seg, int n_seg, double *out_z, double *out_nnull,
                          int n_samples = 1, const int *seg_count = nullptr) {
  WCX_ARG(ctx && r && w && chr_off && seg && out_z, "NULL argument");
  WCX_ARG(n_chr > 0 && n_seg >= 0, "bad sizes");
  if (n_seg == 0) return WCX_OK;
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t nb = chr_off[n_chr];
  const bool attached = (nr == nullptr);
  if (attached) {
    WCX_ARG(ctx->d_nullm && ctx->nullm_bins == nb, "no attached null matrix of matching size");
    m = ctx->nullm_m;
  }
  WCX_ARG(m > 0 && m <= 128, "bad sizes (m <= 128)");
  std::vector<int64_t> b0(n_seg), b1(n_seg);
  std::vector<double> sr(n_seg);
  int smp = 0, left = seg_count ? seg_count[0] : n_seg;      // batches: segments listed sample by sample
  for (int s = 0; s < n_seg; ++s) {
    while (seg_count && left == 0 && smp + 1 < n_samples) left = seg_count[++smp];
    --left;
    const int c = (int)seg[s * 4];
    WCX_ARG(c >= 0 && c < n_chr, "segment chromosome out of range");
    b0[s] = chr_off[c] + (int64_t)seg[s * 4 + 1];
    b1[s] = chr_off[c] + (int64_t)seg[s * 4 + 2];
    WCX_ARG(b0[s] >= chr_off[c] && b1[s] <= chr_off[c + 1] && b0[s] <= b1[s], "segment out of range");
    b0[s] += (int64_t)smp * nb;
    b1[s] += (int64_t)smp * nb;
    sr[s] = seg[s * 4 + 3];
  }
  // chunk table: segment s owns chunks [chunk0[s], chunk0[s+1])
  std::vector<int64_t> cb0, cb1;
  std::vector<int> chunk0(n_seg + 1);
  for (int s = 0; s < n_seg; ++s) {
    chunk0[s] = (int)cb0.size();
    for (int64_t b = b0[s]; b < b1[s]; b += SZ_CHUNK) {
      cb0.push_back(b);
      cb1.push_back(std::min<int64_t>(b + SZ_CHUNK, b1[s]));
    }
  }
  chunk0[n_seg] = (int)cb0.size();
  const size_t n_chunks = cb0.size();
  const size_t vb = rw_on_device ? 8 : (size_t)nb * 8, nrb = attached ? 0 : (size_t)nb * m * 8, sb = (size_t)n_seg * 8;
  const size_t cb = (n_chunks + 1) * 8, pb = (n_chunks + 1) * (size_t)m * 8;
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, 2 * vb + nrb + 4 * sb + 2 * cb + 3 * pb + 4096, &scr);
  if (rc) return rc;
  char *p = reinterpret_cast<char *>(scr);
  double *dr = (double *)p; p += vb;
  double *dw = (double *)p; p += vb;
  double *dnr = (double *)p; p += nrb;
  double *dsr = (double *)p; p += sb;
  double *dz = (double *)p; p += sb;
  double *dn = (double *)p; p += sb;
  int *dchunk0 = (int *)p; p += sb + 8;
  int64_t *dcb0 = (int64_t *)p; p += cb;
  int64_t *dcb1 = (int64_t *)p; p += cb;
  double *dpnum = (double *)p; p += pb;
  double *dpden = (double *)p; p += pb;
  int *dpany = (int *)p;
  hipStream_t st = ctx->stream;
  if (rw_on_device) {
    dr = const_cast<double *>(r);
    dw = const_cast<double *>(w);
  } else {
    WCX_HIP(hipMemcpyAsync(dr, r, vb, hipMemcpyHostToDevice, st));
    WCX_HIP(hipMemcpyAsync(dw, w, vb, hipMemcpyHostToDevice, st));
  }
  if (!attached) WCX_HIP(hipMemcpyAsync(dnr, nr, nrb, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(dsr, sr.data(), sb, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(dchunk0, chunk0.data(), (size_t)(n_seg + 1) * 4, hipMemcpyHostToDevice, st));
  if (n_chunks) {
    WCX_HIP(hipMemcpyAsync(dcb0, cb0.data(), n_chunks * 8, hipMemcpyHostToDevice, st));
    WCX_HIP(hipMemcpyAsync(dcb1, cb1.data(), n_chunks * 8, hipMemcpyHostToDevice, st));
  }
  rc = wcx_timer_begin(ctx, "segment_z");
  if (rc) return rc;
  if (n_chunks)
    k_segz_partial<<<(unsigned)n_chunks, 128, 0, st>>>(dr, dw, attached ? ctx->d_nullm : dnr, m, nb, dcb0,
                                                       dcb1, dpnum, dpden, dpany);
  k_segment_z<<<n_seg, 128, 0, st>>>(dpnum, dpden, dpany, m, dchunk0, dsr, n_seg, dz, dn);
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "segment_z");
  if (rc) return rc;
  WCX_HIP(hipMemcpyAsync(out_z, dz, sb, hipMemcpyDeviceToHost, st));
  if (out_nnull) WCX_HIP(hipMemcpyAsync(out_nnull, dn, sb, hipMemcpyDeviceToHost, st));
  WCX_HIP(hipStreamSynchronize(st));
  return WCX_OK;
}

int wcx_segment_z(wcx_ctx *ctx, const double *r, const double *w, const double *nr, int m,
                  const int64_t *chr_off, int n_chr, const double *seg, int n_seg, double *out_z,
                  double *out_nnull) {
  return segment_z_impl(ctx, r, w, false, nr, m, chr_off, n_chr, seg, n_seg, out_z, out_nnull);
}

int wcx_segment_z_dev(wcx_ctx *ctx, const double *d_r, const double *d_w, const int64_t *chr_off,
                      int n_chr, const double *seg, int n_seg, double *out_z, double *out_nnull) {
  return segment_z_impl(ctx, d_r, d_w, true, nullptr, 0, chr_off, n_chr, seg, n_seg, out_z, out_nnull);
}

int wcx_segment_z_batch_dev(wcx_ctx *ctx, const double *d_r, const double *d_w, int n_samples,
                            const int64_t *chr_off, int n_chr, const double *seg,
                            const int *seg_count, double *out_z, double *out_nnull) {
  WCX_ARG(n_samples > 0 && seg_count, "bad parameters");
  int n_seg = 0;
  for (int s = 0; s < n_samples; ++s) n_seg += seg_count[s];
  return segment_z_impl(ctx, d_r, d_w, true, nullptr, 0, chr_off, n_chr, seg, n_seg, out_z, out_nnull,
                        n_samples, seg_count);
}

}  // extern "C"
