// PCA correction of the reference samples on the GPU (the input producers a2 /
// a3 of the search): replaces the host NumPy of newref_tools.train_pca (newref_tools.py:138-147)
// and the distance-to-median profile of the PCA-distance bin filter (newref_control.py:38-47).
//
// The reference fits scikit-learn's PCA(5) with an UNSEEDED randomized SVD; here the decomposition
// is exact and deterministic: thin SVD through the S x S Gram matrix of the centred data.  The
// O(S^2 B) Gram product, the (5 x B) components, the reconstruction and the ratio
// X = t / reconstruction run here in fp64; only the S x S symmetric eigenproblem (S <= a few
// hundred) stays with the caller's LAPACK (numpy.linalg.eigh) between the two entry points.
// Every sum is accumulated in a fixed order: two runs give identical bits.
//
// Data layout: t_data sample-major double[S][B] (bin b of sample s at s*B + b) -- also the layout of
// the result X, which is what the search takes (wcx.h: "Xs").
//
// Rooflines: k_pca_gram is fp64-FMA bound (2 S^2 B / 2 flop for the upper triangle, 78.6 TFLOP/s
// peak); everything else is one or two HBM sweeps of the S x B matrix.
#include "wave_sort.h"
#include "wcx_common.h"

#pragma clang fp contract(fast)

namespace {

constexpr int GT = 64;       // Gram tile (samples x samples)
constexpr int GK = 16;       // bins per staged chunk
constexpr int GSLICES = 64;  // K slices (bins are cut into GSLICES ranges, reduced in order)

__global__ __launch_bounds__(256) void k_pca_mean(const double *__restrict__ t, int64_t B, int S,
                                                  double *__restrict__ mean) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int j = 0; j < S; ++j) s += t[(int64_t)j * B + b];
  mean[b] = s / S;
}

// partial[slice][i][j] for the upper-triangle tile pairs (ti <= tj)
__global__ __launch_bounds__(256) void k_pca_gram(const double *__restrict__ t,
                                                  const double *__restrict__ mean, int64_t B, int S,
                                                  int ntile, double *__restrict__ partial) {
  // blockIdx.x = tile pair index, blockIdx.y = slice
  int ti = 0, rem = blockIdx.x;
  while (rem >= ntile - ti) { rem -= ntile - ti; ++ti; }
  const int tj = ti + rem;
  const int64_t per = (B + gridDim.y - 1) / gridDim.y;
  const int64_t b_lo = (int64_t)blockIdx.y * per, b_hi = b_lo + per < B ? b_lo + per : B;
  __shared__ double sa[GK][GT + 1], sb[GK][GT + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = 0.0;
  for (int64_t b0 = b_lo; b0 < b_hi; b0 += GK) {
    __syncthreads();
    // stage 64 samples x 16 bins of both tiles, centred; thread -> (sample = tid / 4 + 0|.., bins)
    for (int e = threadIdx.x; e < GT * GK; e += 256) {
      const int smp = e / GK, kk = e % GK;
      const int64_t b = b0 + kk;
      const int sA = ti * GT + smp, sB = tj * GT + smp;
      const double m = b < b_hi ? mean[b] : 0.0;
      sa[kk][smp] = (b < b_hi && sA < S) ? t[(int64_t)sA * B + b] - m : 0.0;
      sb[kk][smp] = (b < b_hi && sB < S) ? t[(int64_t)sB * B + b] - m : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; ++kk) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) { av[a] = sa[kk][ty * 4 + a]; bv[a] = sb[kk][tx * 4 + a]; }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] += av[a] * bv[c];
    }
  }
  double *out = partial + (int64_t)blockIdx.y * S * S;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = ti * GT + ty * 4 + a, j = tj * GT + tx * 4 + c;
      if (i < S && j < S) out[(int64_t)i * S + j] = acc[a][c];
    }
}

// gram[i][j] = sum over slices (in order) of the upper-triangle partials; mirrored
__global__ __launch_bounds__(256) void k_pca_gram_reduce(const double *__restrict__ partial, int S,
                                                         int slices, double *__restrict__ gram) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)S * S) return;
  const int i = (int)(e / S), j = (int)(e % S);
  const int a = i <= j ? i : j, b = i <= j ? j : i;
  // tiles with ti <= tj only were written: (a, b) with a <= b lies in such a tile unless both fall
  // into the same tile with a > b -- impossible here
  double s = 0.0;
  for (int q = 0; q < slices; ++q) s += partial[(int64_t)q * S * S + (int64_t)a * S + b];
  gram[e] = s;
}

// comps[c][b] = sum_s u[s][c] (t[s][b] - mean[b]) / sv[c]
template <int NC>
__global__ __launch_bounds__(256) void k_pca_comps(const double *__restrict__ t,
                                                   const double *__restrict__ mean,
                                                   const double *__restrict__ u,   // [S][NC]
                                                   const double *__restrict__ sv, int64_t B, int S,
                                                   double *__restrict__ comps) {
  extern __shared__ double su[];   // [S][NC]
  for (int e = threadIdx.x; e < S * NC; e += 256) su[e] = u[e];
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const double m = mean[b];
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  for (int s = 0; s < S; ++s) {
    const double v = t[(int64_t)s * B + b] - m;
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] += su[s * NC + c] * v;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) comps[(int64_t)c * B + b] = acc[c] / (sv[c] > 0.0 ? sv[c] : 1.0);
}

// tr[s][c] = sum_b (t[s][b] - mean[b]) comps[c][b]: one workgroup per sample, fixed reduction order
template <int NC>
__global__ __launch_bounds__(1024) void k_pca_transform(const double *__restrict__ t,
                                                        const double *__restrict__ mean,
                                                        const double *__restrict__ comps, int64_t B,
                                                        double *__restrict__ tr) {
  const int s = blockIdx.x, tid = threadIdx.x;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  for (int64_t b = tid; b < B; b += 1024) {
    const double v = t[(int64_t)s * B + b] - mean[b];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] += v * comps[(int64_t)c * B + b];
  }
  __shared__ double red[NC][16];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const double w = wcx::wave_sum(acc[c]);
    if ((tid & 63) == 0) red[c][tid >> 6] = w;
  }
  __syncthreads();
  if (tid < NC) {
    double a = 0.0;
    for (int q = 0; q < 16; ++q) a += red[tid][q];
    tr[s * NC + tid] = a;
  }
}

// X[s][b] = t[s][b] / (sum_c tr[s][c] comps[c][b] + mean[b])     (newref_tools.py:143-147)
template <int NC>
__global__ __launch_bounds__(256) void k_pca_correct(const double *__restrict__ t,
                                                     const double *__restrict__ mean,
                                                     const double *__restrict__ comps,
                                                     const double *__restrict__ tr, int64_t B,
                                                     double *__restrict__ X) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  if (b >= B) return;
  double rec = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) rec += tr[s * NC + c] * comps[(int64_t)c * B + b];
  rec += mean[b];
  X[(int64_t)s * B + b] = t[(int64_t)s * B + b] / rec;
}

// dist_to_med[b] = sum_s (X[s][b] - med[s])^2                      (newref_control.py:40-41)
__global__ __launch_bounds__(256) void k_pca_dist2med(const double *__restrict__ X,
                                                      const double *__restrict__ med, int64_t B,
                                                      int S, double *__restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int j = 0; j < S; ++j) { const double e = X[(int64_t)j * B + b] - med[j]; s += e * e; }
  out[b] = s;
}

}  // namespace

int wcx_nanmedian_rows_launch(wcx_ctx *ctx, const double *d_a, int64_t n, int64_t stride, int count,
                              double *d_out);   // predict.hip

// persistent buffers of the context: t | X | mean | comps | dist (freed by wcx_pca_end / ctx destroy)
int wcx_pca_alloc(wcx_ctx *ctx, int64_t B, int S, double **dt_out) {
  WCX_ARG(B > 0 && S > 1 && S <= 4096, "bad sizes (2 <= S <= 4096)");
  hipStream_t st = ctx->stream;
  const size_t tb = (size_t)S * B * 8;
  if (ctx->pca_bytes < 2 * tb + (size_t)B * 8 * 18) {
    WCX_HIP(hipStreamSynchronize(st));
    if (ctx->d_pca) { WCX_HIP(hipFree(ctx->d_pca)); ctx->d_pca = nullptr; ctx->pca_bytes = 0; }
    const size_t need = 2 * tb + (size_t)B * 8 * 18;
    if (hipMalloc(&ctx->d_pca, need) != hipSuccess) {
      wcx_set_error("hipMalloc(%zu) for the PCA buffers failed", need);
      return WCX_ERR_NOMEM;
    }
    ctx->pca_bytes = need;
  }
  ctx->pca_B = B;
  ctx->pca_S = S;
  *dt_out = reinterpret_cast<double *>(ctx->d_pca);
  return WCX_OK;
}

// per-bin mean and the S x S Gram matrix of the centred data in the context's t buffer
int wcx_pca_gram_from_dt(wcx_ctx *ctx, double *mean_out, double *gram_out) {
  hipStream_t st = ctx->stream;
  const int64_t B = ctx->pca_B;
  const int S = ctx->pca_S;
  double *dt = reinterpret_cast<double *>(ctx->d_pca);
  double *dmean = dt + (size_t)S * B * 2;
  const int ntile = (S + GT - 1) / GT;
  // slices of the bin range reduced in a fixed order (bit-reproducible for a given S): 64 for the
  // cohorts the reference is built from; fewer for very large cohorts, whose S x S partials would
  // otherwise take GSLICES * S^2 * 8 bytes (8.6 GB at S = 4096)
  const int slices = S <= 1024 ? GSLICES : S <= 2048 ? 16 : 4;
  const size_t part_b = (size_t)slices * S * S * 8;
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, part_b + (size_t)S * S * 8 + 256, &scr);
  if (rc) return rc;
  double *dpart = reinterpret_cast<double *>(scr);
  double *dgram = dpart + (size_t)slices * S * S;
  rc = wcx_timer_begin(ctx, "pca_gram");
  if (rc) return rc;
  k_pca_mean<<<(unsigned)((B + 255) / 256), 256, 0, st>>>(dt, B, S, dmean);
  k_pca_gram<<<dim3((unsigned)(ntile * (ntile + 1) / 2), (unsigned)slices), 256, 0, st>>>(dt, dmean, B, S, ntile,
                                                                                          dpart);
  k_pca_gram_reduce<<<(unsigned)(((int64_t)S * S + 255) / 256), 256, 0, st>>>(dpart, S, slices, dgram);
  WCX_HIP(hipGetLastError());
  rc = wcx_timer_end(ctx, "pca_gram");
  if (rc) return rc;
  WCX_HIP(hipMemcpyAsync(mean_out, dmean, (size_t)B * 8, hipMemcpyDeviceToHost, st));
  WCX_HIP(hipMemcpyAsync(gram_out, dgram, (size_t)S * S * 8, hipMemcpyDeviceToHost, st));
  WCX_HIP(hipStreamSynchronize(st));
  return WCX_OK;
}

extern "C" {

int wcx_pca_begin(wcx_ctx *ctx, const double *t_data, int64_t B, int S, double *mean_out,
                  double *gram_out) {
  WCX_ARG(ctx && t_data && mean_out && gram_out, "NULL argument");
  WCX_ARG(B > 0 && S > 1 && S <= 4096, "bad sizes (2 <= S <= 4096)");
  WCX_HIP(hipSetDevice(ctx->device));
  double *dt = nullptr;
  int rc = wcx_pca_alloc(ctx, B, S, &dt);
  if (rc) return rc;
  WCX_HIP(hipMemcpyAsync(dt, t_data, (size_t)S * B * 8, hipMemcpyHostToDevice, ctx->stream));
  return wcx_pca_gram_from_dt(ctx, mean_out, gram_out);
}

int wcx_pca_corrected_dev(wcx_ctx *ctx, double **dX_out) {
  WCX_ARG(ctx && dX_out, "NULL argument");
  WCX_ARG(ctx->d_pca && ctx->pca_B > 0, "wcx_pca_begin / wcx_pca_finish must come first");
  *dX_out = reinterpret_cast<double *>(ctx->d_pca) + (size_t)ctx->pca_S * ctx->pca_B;
  return WCX_OK;
}

int wcx_pca_finish(wcx_ctx *ctx, const double *u, const double *sv, int ncomp, double *comps_out,
                   double *X_out, double *dist_to_med_out) {
  WCX_ARG(ctx && u && sv && comps_out, "NULL argument");
  WCX_ARG(ctx->d_pca && ctx->pca_B > 0, "wcx_pca_begin must come first");
  WCX_ARG(ncomp == 5, "this build instantiates 5 components (newref_tools.py:138: pcacomp=5)");
  WCX_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t B = ctx->pca_B;
  const int S = ctx->pca_S;
  double *dt = reinterpret_cast<double *>(ctx->d_pca);
  double *dX = dt + (size_t)S * B;
  double *dmean = dt + (size_t)S * B * 2;
  double *dcomps = dmean + B;            // [5][B]
  double *dd2m = dcomps + (size_t)5 * B; // [B]
  void *scr = nullptr;
  int rc = wcx_scratch(ctx, (size_t)S * 5 * 8 * 2 + 64 * 8 + (size_t)S * 8 + 1024, &scr);
  if (rc) return rc;
  double *du = reinterpret_cast<double *>(scr);
  double *dsv = du + (size_t)S * 5;
  double *dtr = dsv + 8;
  double *dmed = dtr + (size_t)S * 5;
  WCX_HIP(hipMemcpyAsync(du, u, (size_t)S * 5 * 8, hipMemcpyHostToDevice, st));
  WCX_HIP(hipMemcpyAsync(dsv, sv, 5 * 8, hipMemcpyHostToDevice, st));
  rc = wcx_timer_begin(ctx, "pca_apply");
  if (rc) return rc;
  const unsigned gb = (unsigned)((B + 255) / 256);
  // the eigenvectors sit in LDS: S * 5 * 8 bytes, beyond the default 64 KB from S = 1639 (160 KB
  // = the CU's LDS at the S = 4096 limit of wcx_pca_begin)
  WCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pca_comps<5>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)S * 5 * 8)));
  k_pca_comps<5><<<gb, 256, (size_t)S * 5 * 8, st>>>(dt, dmean, du, dsv, B, S, dcomps);
  k_pca_transform<5><<<S, 1024, 0, st>>>(dt, dmean, dcomps, B, dtr);
  k_pca_correct<5><<<dim3(gb, (unsigned)S), 256, 0, st>>>(dt, dmean, dcomps, dtr, B, dX);
  WCX_HIP(hipGetLastError());
  if (dist_to_med_out) {
    rc = wcx_nanmedian_rows_launch(ctx, dX, B, B, S, dmed);     // np.median(X, axis=0): per sample
    if (rc) return rc;
    k_pca_dist2med<<<gb, 256, 0, st>>>(dX, dmed, B, S, dd2m);
    WCX_HIP(hipGetLastError());
  }
  rc = wcx_timer_end(ctx, "pca_apply");
  if (rc) return rc;
  WCX_HIP(hipMemcpyAsync(comps_out, dcomps, (size_t)5 * B * 8, hipMemcpyDeviceToHost, st));
  if (X_out) WCX_HIP(hipMemcpyAsync(X_out, dX, (size_t)S * B * 8, hipMemcpyDeviceToHost, st));
  if (dist_to_med_out)
    WCX_HIP(hipMemcpyAsync(dist_to_med_out, dd2m, (size_t)B * 8, hipMemcpyDeviceToHost, st));
  WCX_HIP(hipStreamSynchronize(st));
  if (dist_to_med_out) {
    // np.median (newref_control.py:40), not nanmedian: one NaN in a sample's column makes that
    // sample's median NaN and with it EVERY bin's distance (the filter then drops nothing).  A NaN
    // in X shows up as a NaN distance of its own bin.
    bool any_nan = false;
    for (int64_t b = 0; b < B && !any_nan; ++b) any_nan = dist_to_med_out[b] != dist_to_med_out[b];
    if (any_nan)
      for (int64_t b = 0; b < B; ++b) dist_to_med_out[b] = __builtin_nan("");
  }
  return WCX_OK;
}

int wcx_pca_end(wcx_ctx *ctx) {
  WCX_ARG(ctx, "ctx is NULL");
  if (ctx->d_pca) {
    WCX_HIP(hipStreamSynchronize(ctx->stream));
    WCX_HIP(hipFree(ctx->d_pca));
    ctx->d_pca = nullptr;
    ctx->pca_bytes = 0;
    ctx->pca_B = 0;
  }
  return WCX_OK;
}

}  // extern "C"
