// Input producers of newref on the device: the bin mask
// (newref_tools.get_mask, newref_tools.py:77-102) and the depth normalisation + masking
// (newref_tools.normalize_and_mask, newref_tools.py:110-129) straight from the cohort's bin counts,
// resident in HBM as int32 [S_all][n_bins] (every sample's chromosomes laid out over the longest
// sample's bins, zero padded -- what the reference's np.zeros + copy loops build).  The normalised,
// masked matrix is written directly into the PCA stage's buffer (wcx_pca_begin_counts_dev), so the
// 729 MB matrix of a 15 kb x 500 cohort never exists on the host.  HBM-streaming kernels.
#include "wave_sort.h"
#include "wcx_common.h"

int wcx_nanmedian_rows_launch(wcx_ctx *ctx, const double *d_a, int64_t n, int64_t stride, int count,
                              double *d_out);   // predict.hip

namespace {

// total read count of the selected samples over bins [0, n_sum)   (integers: exact in fp64)
__global__ __launch_bounds__(1024) void k_counts_total(const int32_t *__restrict__ counts, int64_t n_bins,
                                                       const int32_t *__restrict__ sel, int64_t n_sum,
                                                       double *__restrict__ total) {
  __shared__ double red[16];
  const int32_t *c = counts + (int64_t)sel[blockIdx.x] * n_bins;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n_sum; i += 1024) s += (double)c[i];
  s = wcx::wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += red[q];
    total[blockIdx.x] = t;
  }
}

// NumPy's pairwise summation of n terms term(lo) .. term(lo + n - 1), as np.sum runs it along a
// contiguous axis (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum): fewer than 8 terms in
// sequence; up to 128 terms in 8 interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+
// (r6+r7)) plus the remainder in sequence; longer ranges split at n/2 rounded down to a multiple of 8.
// The recursion is unrolled with a small explicit stack (depth <= log2(n / 128) + 1).
template <typename F>
__device__ double numpy_pairwise_sum(F term, int n) {
  int lo_stack[16], n_stack[16];
  double acc_stack[16];                 // value of the left half while the right one is summed
  int state[16];                        // 0 = to do, 1 = left done (right pending), 2 = both done
  int sp = 0;
  lo_stack[0] = 0; n_stack[0] = n; state[0] = 0; acc_stack[0] = 0.0;
  double ret = 0.0;
  while (sp >= 0) {
    const int lo = lo_stack[sp], m = n_stack[sp];
    if (m <= 128) {
      double res;
      if (m < 8) {
        res = 0.0;
        for (int i = 0; i < m; ++i) res += term(lo + i);
      } else {
        double r0 = term(lo), r1 = term(lo + 1), r2 = term(lo + 2), r3 = term(lo + 3), r4 = term(lo + 4),
               r5 = term(lo + 5), r6 = term(lo + 6), r7 = term(lo + 7);
        int i = 8;
        for (; i < m - (m % 8); i += 8) {
          r0 += term(lo + i); r1 += term(lo + i + 1); r2 += term(lo + i + 2); r3 += term(lo + i + 3);
          r4 += term(lo + i + 4); r5 += term(lo + i + 5); r6 += term(lo + i + 6); r7 += term(lo + i + 7);
        }
        res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < m; ++i) res += term(lo + i);
      }
      ret = res;
      --sp;
    } else if (state[sp] == 0) {
      int n2 = m / 2;
      n2 -= n2 % 8;
      state[sp] = 1;
      ++sp;
      lo_stack[sp] = lo; n_stack[sp] = n2; state[sp] = 0;
      continue;
    }
    // a child has just returned `ret` to the frame now on top
    while (sp >= 0 && n_stack[sp] > 128) {
      const int lo2 = lo_stack[sp], m2 = n_stack[sp];
      int n2 = m2 / 2;
      n2 -= n2 % 8;
      if (state[sp] == 1) {             // left half done: start the right half
        acc_stack[sp] = ret;
        state[sp] = 2;
        ++sp;
        lo_stack[sp] = lo2 + n2; n_stack[sp] = m2 - n2; state[sp] = 0;
        break;
      }
      ret = acc_stack[sp] + ret;        // both halves done
      --sp;
    }
  }
  return ret;
}

// sum_per_bin[b] = np.sum over the selected samples of counts[s][b] / total[s] -- in NumPy's own
// summation order (the reduction runs along the contiguous sample axis of the (bins x samples) matrix,
// newref_tools.py:97), starting from +0.0 like np.add.reduce; pos[b] = the same where positive, else
// NaN (for the median of the covered bins)
__global__ __launch_bounds__(256) void k_mask_colsum(const int32_t *__restrict__ counts, int64_t n_bins,
                                                     const int32_t *__restrict__ sel, int ns,
                                                     const double *__restrict__ total,
                                                     double *__restrict__ colsum, double *__restrict__ pos) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= n_bins) return;
  const double s = 0.0 + numpy_pairwise_sum(
      [&](int q) { return (double)counts[(int64_t)sel[q] * n_bins + b] / total[q]; }, ns);
  colsum[b] = s;
  pos[b] = s > 0.0 ? s : __builtin_nan("");
}

__global__ __launch_bounds__(256) void k_mask_apply(const double *__restrict__ colsum, int64_t n_bins,
                                                    const double *__restrict__ med,
                                                    unsigned char *__restrict__ mask) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b < n_bins) mask[b] = colsum[b] > 0.05 * med[0] ? 1 : 0;          // newref_tools.py:101
}

// out[q][i] = counts[sel[q]][pos[i]] / total[q]     (newref_tools.py:124-128, sample-major)
__global__ __launch_bounds__(256) void k_counts_normalize(const int32_t *__restrict__ counts, int64_t n_bins,
                                                          const int32_t *__restrict__ sel,
                                                          const double *__restrict__ total,
                                                          const int32_t *__restrict__ pos, int64_t B,
                                                          double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int q = blockIdx.y;
  if (i >= B) return;
  out[(int64_t)q * B + i] = (double)counts[(int64_t)sel[q] * n_bins + pos[i]] / total[q];
}

}  // namespace

// pca.hip: allocates the PCA stage's buffers for (B, S) and returns the device matrix t [S][B];
// then the Gram step on whatever was written there
int wcx_pca_alloc(wcx_ctx *ctx, int64_t B, int S, double **dt_out);
int wcx_pca_gram_from_dt(wcx_ctx *ctx, double *mean_out, double *gram_out);

extern "C" {

int wcx_prep_mask_dev(wcx_ctx *ctx, const int32_t *d_counts, int64_t n_bins, const int32_t *sel, int ns,
                      unsigned char *mask_out, double *sum_per_bin_out) {
  WCX_ARG(ctx && d_counts && sel && mask_out && n_bins > 0 && ns > 0, "bad parameters");
  WCX_HIP(hipSetDevice(ctx->device));
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, (size_t)ns * 16 + (size_t)n_bins * 17 + 1024, &scr);
  if (rc) return rc;
  char *p = reinterpret_cast<char *>(scr);
  double *d_total = reinterpret_cast<double *>(p); p += (size_t)ns * 8;
  double *d_col = reinterpret_cast<double *>(p); p += (size_t)n_bins * 8;
  double *d_pos = reinterpret_cast<double *>(p); p += (size_t)n_bins * 8;
  double *d_med = reinterpret_cast<double *>(p); p += 64;
  int32_t *d_sel = reinterpret_cast<int32_t *>(p); p += ((size_t)ns * 4 + 63) / 64 * 64;
  unsigned char *d_mask = reinterpret_cast<unsigned char *>(p);
  hipStream_t st = ctx->stream;
  WCX_HIP(hipMemcpyAsync(d_sel, sel, (size_t)ns * 4, hipMemcpyHostToDevice, st));
  k_counts_total<<<(unsigned)ns, 1024, 0, st>>>(d_counts, n_bins, d_sel, n_bins, d_total);
  const unsigned gb = (unsigned)((n_bins + 255) / 256);
  k_mask_colsum<<<gb, 256, 0, st>>>(d_counts, n_bins, d_sel, ns, d_total, d_col, d_pos);
  rc = wcx_nanmedian_rows_launch(ctx, d_pos, n_bins, n_bins, 1, d_med);
  if (rc) return rc;
  k_mask_apply<<<gb, 256, 0, st>>>(d_col, n_bins, d_med, d_mask);
  WCX_HIP(hipGetLastError());
  WCX_HIP(hipMemcpyAsync(mask_out, d_mask, (size_t)n_bins, hipMemcpyDeviceToHost, st));
  if (sum_per_bin_out)
    WCX_HIP(hipMemcpyAsync(sum_per_bin_out, d_col, (size_t)n_bins * 8, hipMemcpyDeviceToHost, st));
  WCX_HIP(hipStreamSynchronize(st));
  return WCX_OK;
}

int wcx_pca_begin_counts_dev(wcx_ctx *ctx, const int32_t *d_counts, int64_t n_bins, const int32_t *sel,
                             int ns, int64_t n_bins_pass, const int32_t *pos, int64_t B, double *mean_out,
                             double *gram_out) {
  WCX_ARG(ctx && d_counts && sel && pos && mean_out && gram_out, "NULL argument");
  WCX_ARG(n_bins > 0 && ns > 1 && ns <= 4096 && n_bins_pass > 0 && n_bins_pass <= n_bins && B > 0 && B <= n_bins_pass,
          "bad sizes");
  for (int64_t i = 0; i < B; ++i)
    WCX_ARG(pos[i] >= 0 && pos[i] < n_bins_pass, "kept-bin position outside the pass");
  for (int q = 0; q < ns; ++q) WCX_ARG(sel[q] >= 0, "negative sample index");
  WCX_HIP(hipSetDevice(ctx->device));
  double *dt = nullptr;
  int rc = wcx_pca_alloc(ctx, B, ns, &dt);
  if (rc) return rc;
  void *scr = nullptr;
  rc = wcx_scratch2(ctx, (size_t)ns * 16 + (size_t)B * 4 + 1024, &scr);
  if (rc) return rc;
  char *p = reinterpret_cast<char *>(scr);
  double *d_total = reinterpret_cast<double *>(p); p += (size_t)ns * 8;
  int32_t *d_sel = reinterpret_cast<int32_t *>(p); p += ((size_t)ns * 4 + 63) / 64 * 64;
  int32_t *d_pos = reinterpret_cast<int32_t *>(p);
  hipStream_t st = ctx->stream;
  WCX_HIP(hipMemcpyAsync(d_sel, sel, (size_t)ns * 4, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(d_pos, pos, (size_t)B * 4, hipMemcpyHostToDevice, st));
  k_counts_total<<<(unsigned)ns, 1024, 0, st>>>(d_counts, n_bins, d_sel, n_bins_pass, d_total);
  k_counts_normalize<<<dim3((unsigned)((B + 255) / 256), (unsigned)ns), 256, 0, st>>>(d_counts, n_bins, d_sel,
                                                                                     d_total, d_pos, B, dt);
  WCX_HIP(hipGetLastError());
  WCX_HIP(hipStreamSynchronize(st));          // (sel / pos are the caller's host buffers)
  return wcx_pca_gram_from_dt(ctx, mean_out, gram_out);
}

}  // extern "C"
