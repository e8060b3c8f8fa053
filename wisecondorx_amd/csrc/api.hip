// C-ABI entry points of libwcx_hip.so (see include/wcx.h): context, memory, and the host-side
// orchestration of the newref search.  Kernels live in the sibling .hip files.
#include "wcx_common.h"

#include <cstring>
#include <thread>

static thread_local char g_err[1024] = "";

void wcx_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int wcx_scratch(wcx_ctx *ctx, size_t bytes, void **out) {
  if (bytes > ctx->scratch_bytes) {
    if (ctx->scratch) {
      WCX_HIP(hipStreamSynchronize(ctx->stream));
      WCX_HIP(hipFree(ctx->scratch));
      ctx->scratch = nullptr;
      ctx->scratch_bytes = 0;
    }
    hipError_t e = hipMalloc(&ctx->scratch, bytes);
    if (e != hipSuccess) {
      wcx_set_error("hipMalloc(%zu bytes scratch) failed: %s", bytes, hipGetErrorString(e));
      return WCX_ERR_NOMEM;
    }
    ctx->scratch_bytes = bytes;
  }
  *out = ctx->scratch;
  return WCX_OK;
}

// Pinned host staging area (grown on demand, kept for the life of the context): callers that
// assemble upload buffers on the host fill it directly, so the H2D copy is a true async DMA and
// no call pays for fresh, zero-filled pages.
int wcx_host_scratch(wcx_ctx *ctx, size_t bytes, void **out) {
  if (bytes > ctx->host_scratch_bytes) {
    if (ctx->host_scratch) {
      WCX_HIP(hipStreamSynchronize(ctx->stream));
      WCX_HIP(hipHostFree(ctx->host_scratch));
      ctx->host_scratch = nullptr;
      ctx->host_scratch_bytes = 0;
    }
    const size_t grown = bytes + bytes / 4;
    hipError_t e = hipHostMalloc(&ctx->host_scratch, grown, hipHostMallocDefault);
    if (e != hipSuccess) {
      wcx_set_error("hipHostMalloc(%zu bytes) failed: %s", grown, hipGetErrorString(e));
      return WCX_ERR_NOMEM;
    }
    ctx->host_scratch_bytes = grown;
  }
  *out = ctx->host_scratch;
  return WCX_OK;
}

int wcx_scratch2(wcx_ctx *ctx, size_t bytes, void **out) {
  if (bytes > ctx->scratch2_bytes) {
    if (ctx->scratch2) {
      WCX_HIP(hipStreamSynchronize(ctx->stream));
      WCX_HIP(hipFree(ctx->scratch2));
      ctx->scratch2 = nullptr;
      ctx->scratch2_bytes = 0;
    }
    hipError_t e = hipMalloc(&ctx->scratch2, bytes);
    if (e != hipSuccess) {
      wcx_set_error("hipMalloc(%zu bytes scratch2) failed: %s", bytes, hipGetErrorString(e));
      return WCX_ERR_NOMEM;
    }
    ctx->scratch2_bytes = bytes;
  }
  *out = ctx->scratch2;
  return WCX_OK;
}

int wcx_upload_small(wcx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
  if (ctx->stage.size() > 64) {
    WCX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stage.clear();
  }
  ctx->stage.emplace_back((const unsigned char *)src_host, (const unsigned char *)src_host + bytes);
  WCX_HIP(hipMemcpyAsync(dst_dev, ctx->stage.back().data(), bytes, hipMemcpyHostToDevice,
                         ctx->stream));
  return WCX_OK;
}

int wcx_timer_begin(wcx_ctx *ctx, const char *name) {
  KernelTimer &t = ctx->timers[ctx->timer_tag + name];
  if (!t.start) {
    WCX_HIP(hipEventCreate(&t.start));
    WCX_HIP(hipEventCreate(&t.stop));
  }
  WCX_HIP(hipEventRecord(t.start, ctx->stream));
  return WCX_OK;
}

int wcx_timer_end(wcx_ctx *ctx, const char *name) {
  KernelTimer &t = ctx->timers[ctx->timer_tag + name];
  WCX_HIP(hipEventRecord(t.stop, ctx->stream));
  t.used = true;
  return WCX_OK;
}

// Large copies between PAGEABLE host memory and the device.  The runtime stages those through its
// own pinned buffer with a single-threaded host memcpy (8 GB/s measured on the 0.8 GB result tables
// of a 15 kb reference); here the DMA runs between HBM and two pinned halves of the context's
// staging area while COPY_THREADS host threads move the other half to / from the caller's pages
// (which, when fresh, they also fault in -- in parallel).
namespace {
constexpr size_t STAGED_MIN = (size_t)32 << 20;
constexpr size_t STAGED_CHUNK = (size_t)32 << 20;
constexpr int COPY_THREADS = 8;

void host_copy_mt(char *dst, const char *src, size_t n) {
  const size_t per = ((n + COPY_THREADS - 1) / COPY_THREADS + 4095) & ~(size_t)4095;
  std::thread th[COPY_THREADS];
  int used = 0;
  for (size_t o = per; o < n; o += per)
    th[used++] = std::thread([=] { memcpy(dst + o, src + o, o + per <= n ? per : n - o); });
  memcpy(dst, src, per <= n ? per : n);
  for (int i = 0; i < used; ++i) th[i].join();
}

int staged_copy(wcx_ctx *ctx, void *dst, const void *src, size_t bytes, bool to_host) {
  void *pin = nullptr;
  int rc = wcx_host_scratch(ctx, 2 * STAGED_CHUNK, &pin);
  if (rc) return rc;
  char *half[2] = {reinterpret_cast<char *>(pin), reinterpret_cast<char *>(pin) + STAGED_CHUNK};
  char *d = reinterpret_cast<char *>(dst);
  const char *s = reinterpret_cast<const char *>(src);
  hipStream_t st = ctx->stream;
  const size_t n_chunks = (bytes + STAGED_CHUNK - 1) / STAGED_CHUNK;
  auto len = [&](size_t i) { return i + 1 < n_chunks ? STAGED_CHUNK : bytes - i * STAGED_CHUNK; };
  if (to_host) {
    WCX_HIP(hipMemcpyAsync(half[0], s, len(0), hipMemcpyDeviceToHost, st));
    WCX_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < n_chunks; ++i) {
      if (i + 1 < n_chunks)
        WCX_HIP(hipMemcpyAsync(half[(i + 1) & 1], s + (i + 1) * STAGED_CHUNK, len(i + 1),
                               hipMemcpyDeviceToHost, st));
      host_copy_mt(d + i * STAGED_CHUNK, half[i & 1], len(i));
      WCX_HIP(hipStreamSynchronize(st));
    }
  } else {
    host_copy_mt(half[0], s, len(0));
    for (size_t i = 0; i < n_chunks; ++i) {
      WCX_HIP(hipMemcpyAsync(d + i * STAGED_CHUNK, half[i & 1], len(i), hipMemcpyHostToDevice, st));
      if (i + 1 < n_chunks) host_copy_mt(half[(i + 1) & 1], s + (i + 1) * STAGED_CHUNK, len(i + 1));
      WCX_HIP(hipStreamSynchronize(st));
    }
  }
  return WCX_OK;
}
}  // namespace

extern "C" {

int wcx_version(void) { return 100; }

int wcx_sweep_event(wcx_ctx *ctx, void **out_event) {
  WCX_ARG(ctx && out_event, "NULL argument");
  WCX_HIP(hipSetDevice(ctx->device));
  if (!ctx->ev_after_sweep) WCX_HIP(hipEventCreateWithFlags(&ctx->ev_after_sweep, hipEventDisableTiming));
  *out_event = ctx->ev_after_sweep;
  return WCX_OK;
}

int wcx_wait_event(wcx_ctx *ctx, void *event) {
  WCX_ARG(ctx && event, "NULL argument");
  WCX_HIP(hipSetDevice(ctx->device));
  WCX_HIP(hipStreamWaitEvent(ctx->stream, reinterpret_cast<hipEvent_t>(event), 0));
  return WCX_OK;
}

int wcx_debug_flags(wcx_ctx *ctx, int flags) {
  if (!ctx) return 0;
  const int old = ctx->debug_flags;
  ctx->debug_flags = flags;
  return old;
}

const char *wcx_last_error(void) { return g_err; }

int wcx_timer_tag(wcx_ctx *ctx, const char *tag) {
  WCX_ARG(ctx, "ctx is NULL");
  ctx->timer_tag = tag ? tag : "";
  return WCX_OK;
}

int wcx_ctx_create(int device, void *stream, wcx_ctx **out) {
  WCX_ARG(out != nullptr, "out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    wcx_set_error("no HIP device available (%s); libwcx_hip.so has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    return WCX_ERR_HIP;
  }
  WCX_ARG(device >= 0 && device < n, "device index out of range");
  WCX_HIP(hipSetDevice(device));
  wcx_ctx *ctx = new wcx_ctx();
  ctx->device = device;
  if (stream == WCX_STREAM_DEFAULT) {          // the device's default (null) stream
    ctx->stream = nullptr;
    ctx->own_stream = false;
  } else if (stream) {
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    ctx->own_stream = false;
  } else {
    WCX_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  WCX_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->d_stats), 256));
  WCX_HIP(hipMemsetAsync(ctx->d_stats, 0, 256, ctx->stream));
  WCX_HIP(hipMalloc(&ctx->d_small, 8192));
  *out = ctx;
  return WCX_OK;
}

int wcx_ctx_destroy(wcx_ctx *ctx) {
  if (!ctx) return WCX_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto &kv : ctx->timers) {
    if (kv.second.start) hipEventDestroy(kv.second.start);
    if (kv.second.stop) hipEventDestroy(kv.second.stop);
  }
  wcx_sym_state_free(ctx);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->host_scratch) hipHostFree(ctx->host_scratch);
  for (auto &sl : ctx->sel_pool)
    if (sl.p) hipFree(sl.p);
  if (ctx->host_scratch2) hipHostFree(ctx->host_scratch2);
  if (ctx->scratch2) hipFree(ctx->scratch2);
  if (ctx->d_stats) hipFree(ctx->d_stats);
  if (ctx->d_small) hipFree(ctx->d_small);
  if (ctx->d_nullm) hipFree(ctx->d_nullm);
  if (ctx->d_nullsrc) hipFree(ctx->d_nullsrc);
  if (ctx->d_pca) hipFree(ctx->d_pca);
  if (ctx->sweep_stream) {
    hipStreamSynchronize(ctx->sweep_stream);
    hipStreamDestroy(ctx->sweep_stream);
    hipEventDestroy(ctx->ev_sweep0);
    hipEventDestroy(ctx->ev_sweep1);
  }
  if (ctx->ev_after_sweep) hipEventDestroy(ctx->ev_after_sweep);
  if (ctx->copy_stream) {
    hipStreamSynchronize(ctx->copy_stream);
    hipStreamDestroy(ctx->copy_stream);
    hipEventDestroy(ctx->ev_cbs_fill);
    hipEventDestroy(ctx->ev_cbs_xw);
  }
  if (ctx->aux_stream) {
    hipStreamSynchronize(ctx->aux_stream);
    hipStreamDestroy(ctx->aux_stream);
    hipEventDestroy(ctx->ev_main);
    hipEventDestroy(ctx->ev_rank);
  }
  if (ctx->d_rank) hipFree(ctx->d_rank);
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return WCX_OK;
}

int wcx_sync(wcx_ctx *ctx) {
  WCX_ARG(ctx, "ctx is NULL");
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  return WCX_OK;
}

int wcx_malloc(wcx_ctx *ctx, size_t bytes, void **dptr) {
  WCX_ARG(ctx && dptr, "NULL argument");
  WCX_HIP(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
  if (e != hipSuccess) {
    wcx_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return WCX_ERR_NOMEM;
  }
  return WCX_OK;
}

int wcx_free(wcx_ctx *ctx, void *dptr) {
  WCX_ARG(ctx, "ctx is NULL");
  if (dptr) {
    WCX_HIP(hipStreamSynchronize(ctx->stream));
    WCX_HIP(hipFree(dptr));
  }
  return WCX_OK;
}

int wcx_memcpy_h2d(wcx_ctx *ctx, void *dst, const void *src, size_t bytes) {
  WCX_ARG(ctx && dst && src, "NULL argument");
  WCX_HIP(hipSetDevice(ctx->device));
  if (bytes >= STAGED_MIN) return staged_copy(ctx, dst, src, bytes, false);
  WCX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  return WCX_OK;
}

int wcx_memcpy_d2h(wcx_ctx *ctx, void *dst, const void *src, size_t bytes) {
  WCX_ARG(ctx && dst && src, "NULL argument");
  WCX_HIP(hipSetDevice(ctx->device));
  if (bytes >= STAGED_MIN) return staged_copy(ctx, dst, src, bytes, true);
  WCX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  return WCX_OK;
}

double wcx_last_kernel_ms(wcx_ctx *ctx, const char *name) {
  if (!ctx || !name) return -1.0;
  auto it = ctx->timers.find(name);
  if (it == ctx->timers.end() || !it->second.used) return -1.0;
  if (hipEventSynchronize(it->second.stop) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, it->second.start, it->second.stop) != hipSuccess) return -1.0;
  return (double)ms;
}

int wcx_transpose_dev(wcx_ctx *ctx, const double *d_src, int64_t rows, int64_t cols,
                      double *d_dst) {
  WCX_ARG(ctx && d_src && d_dst, "NULL argument");
  WCX_ARG(rows > 0 && cols > 0 && rows < (1ll << 31) && (rows + 31) / 32 < 65536, "bad sizes");
  WCX_HIP(hipSetDevice(ctx->device));
  return wcx_transpose_launch(ctx, d_src, rows, cols, d_dst);
}

int wcx_last_topk_stats(wcx_ctx *ctx, int64_t out[24]) {
  WCX_ARG(ctx && out, "NULL argument");
  unsigned long long h[24] = {0};
  WCX_HIP(hipMemcpyAsync(h, ctx->d_stats, 24 * 8, hipMemcpyDeviceToHost, ctx->stream));
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  out[0] = ctx->topk_stats[0];
  out[1] = ctx->topk_stats[1];
  out[2] = (int64_t)h[2];
  out[3] = (int64_t)h[3];
  for (int i = 4; i < 24; ++i) out[i] = (int64_t)h[i];
  return WCX_OK;
}

// ------------------------------------------------------------------ row-sharded symmetric search
int wcx_newref_sym_sweep_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S, const int64_t *chr_cum,
                             int n_chr, int k, int part, int n_parts, const int64_t *row_bounds,
                             int64_t *counts_out) {
  WCX_ARG(ctx && dXs && chr_cum && row_bounds && counts_out, "NULL argument");
  WCX_ARG(B > 0 && S > 0 && n_chr > 0 && k > 0, "B, S, n_chr, k must be positive");
  WCX_ARG(chr_cum[n_chr - 1] == B, "chr_cum[n_chr-1] must equal B");
  WCX_ARG(n_parts >= 1 && part >= 0 && part < n_parts, "bad part");
  WCX_ARG(B < (int64_t)0x7fffffff, "B must fit in int32 (indices are int32)");
  if (n_chr > 22) {
    wcx_set_error("the sharded symmetric sweep searches every row: not a gonosomal pass (n_chr = %d)", n_chr);
    return WCX_ERR_UNSUPPORTED;
  }
  WCX_HIP(hipSetDevice(ctx->device));
  if (ctx->rank_X && !ctx->rank_pending) ctx->rank_X = nullptr;
  const int64_t r0 = row_bounds[part], r1 = row_bounds[part + 1];
  WCX_ARG(0 <= r0 && r0 <= r1 && r1 <= B, "bad row bounds");
  int64_t pairs = 0;
  for (int c = 0; c < n_chr; ++c) {
    const int64_t cs = c ? chr_cum[c - 1] : 0, ce = chr_cum[c];
    const int64_t lo = cs > r0 ? cs : r0, hi = ce < r1 ? ce : r1;
    if (lo < hi) pairs += (hi - lo) * (B - (ce - cs));
  }
  ctx->topk_stats[0] = r1 - r0;
  ctx->topk_stats[1] = pairs;
  wcx_aux_cancel_if_few_rows(ctx, B, r1 - r0);
  return wcx_sym_shard_sweep(ctx, dXs, B, S, chr_cum, n_chr, k, part, n_parts, row_bounds, counts_out);
}

int wcx_newref_sym_records_dev(wcx_ctx *ctx, void *d_send) {
  WCX_ARG(ctx, "NULL argument");
  WCX_HIP(hipSetDevice(ctx->device));
  return wcx_sym_shard_records(ctx, d_send);
}

int wcx_newref_sym_finish_dev(wcx_ctx *ctx, const void *d_recv, int64_t n_recv, int32_t *d_out_idx,
                              double *d_out_dist) {
  WCX_ARG(ctx && d_out_idx && d_out_dist && (d_recv || n_recv <= 0), "bad argument");   // (n_recv < 0: void exchange)
  WCX_HIP(hipSetDevice(ctx->device));
  return wcx_sym_shard_finish(ctx, d_recv, n_recv, d_out_idx, d_out_dist);
}

// ------------------------------------------------------------------ newref search (a4-a6)
int wcx_newref_topk_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                        const int64_t *chr_cum, int n_chr, int64_t row_begin, int64_t row_end,
                        int k, int mode, int32_t *d_out_idx, double *d_out_dist) {
  WCX_ARG(ctx && dXs && chr_cum && d_out_idx && d_out_dist, "NULL argument");
  WCX_ARG(B > 0 && S > 0 && n_chr > 0 && k > 0, "B, S, n_chr, k must be positive");
  WCX_ARG(chr_cum[n_chr - 1] == B, "chr_cum[n_chr-1] must equal B");
  WCX_ARG(0 <= row_begin && row_begin <= row_end && row_end <= B, "bad row range");
  WCX_ARG(B < (int64_t)0x7fffffff, "B must fit in int32 (indices are int32)");
  WCX_ARG(mode >= 0 && mode <= 2, "mode must be 0, 1 or 2");
  WCX_HIP(hipSetDevice(ctx->device));
  // A null-sample ranking that was started inside an EARLIER search and never consumed (that search
  // or its caller failed before wcx_null_ratios_dev) must not be matched by device address later:
  // a prepared ranking lives through one search at most.
  if (ctx->rank_X && !ctx->rank_pending) ctx->rank_X = nullptr;
  const int64_t n_rows = row_end - row_begin;
  if (n_rows == 0) return WCX_OK;

  // Split the row range into per-chromosome blocks of <= 64 target rows
  // (newref_tools._split_by_chr + the clamps at newref_tools.py:181-184).
  std::vector<TopkBlock> blocks;
  int64_t pairs = 0, searched = 0;
  int64_t dummy_lo = row_begin, dummy_hi = row_begin;      // pending run of dummy rows
  for (int c = 0; c < n_chr; ++c) {
    const int64_t cs = c ? chr_cum[c - 1] : 0, ce = chr_cum[c];
    WCX_ARG(ce >= cs, "chr_cum must be non-decreasing");
    const int64_t lo = cs > row_begin ? cs : row_begin;
    const int64_t hi = ce < row_end ? ce : row_end;
    if (lo >= hi) continue;
    if (n_chr > 22 && c != 22 && c != 23) {  // newref_tools.py:186-191
      // (the autosomes are adjacent: their dummy rows are filled by ONE launch, not 22)
      if (dummy_hi == lo) dummy_hi = hi;
      else {
        int rc = wcx_fill_dummy_rows(ctx, d_out_idx, d_out_dist, dummy_lo - row_begin, dummy_hi - row_begin, k);
        if (rc) return rc;
        dummy_lo = lo; dummy_hi = hi;
      }
      continue;
    }
    for (int64_t r = lo; r < hi; r += 64) {
      TopkBlock b;
      b.row0 = r;
      b.nrows = (int32_t)((hi - r) < 64 ? (hi - r) : 64);
      b.pad = 0;
      b.cs = cs;
      b.ce = ce;
      blocks.push_back(b);
    }
    searched += hi - lo;
    pairs += (hi - lo) * (B - (ce - cs));
  }
  {
    int rc = wcx_fill_dummy_rows(ctx, d_out_idx, d_out_dist, dummy_lo - row_begin, dummy_hi - row_begin, k);
    if (rc) return rc;
  }
  ctx->topk_stats[0] = searched;
  ctx->topk_stats[1] = pairs;
  wcx_aux_cancel_if_few_rows(ctx, B, searched);   // few target rows: no ranking of the null samples
  const bool can_screen = wcx_screen_supported(B, S, k);
  if (mode == 2 && !can_screen) {
    wcx_set_error("mode 2 (MFMA screen) needs S <= 1020, refsize <= 1024, B >= 2048 "
                  "(got B=%lld S=%d k=%d)", (long long)B, S, k);
    return WCX_ERR_UNSUPPORTED;
  }
  if (mode == 2 || (mode == 0 && can_screen))
    return wcx_topk_screen_launch(ctx, dXs, B, S, chr_cum, n_chr, blocks, row_begin, n_rows, k,
                                  d_out_idx, d_out_dist);
  return wcx_topk_exact_launch(ctx, dXs, B, S, blocks, row_begin, n_rows, k, d_out_idx,
                               d_out_dist);
}

int wcx_newref_topk(wcx_ctx *ctx, const double *Xs, int64_t B, int S, const int64_t *chr_cum,
                    int n_chr, int64_t row_begin, int64_t row_end, int k, int mode,
                    int32_t *out_idx, double *out_dist) {
  WCX_ARG(ctx && Xs && out_idx && out_dist, "NULL argument");
  WCX_ARG(B > 0 && S > 0 && k > 0 && row_end >= row_begin, "bad sizes");
  WCX_HIP(hipSetDevice(ctx->device));
  const int64_t n_rows = row_end - row_begin;
  const size_t xb = (size_t)B * S * 8, ib = (size_t)n_rows * k * 4, db = (size_t)n_rows * k * 8;
  void *buf = nullptr;
  int rc = wcx_scratch2(ctx, xb + ib + db + 64, &buf);
  if (rc) return rc;
  double *dX = reinterpret_cast<double *>(buf);
  double *dD = reinterpret_cast<double *>(reinterpret_cast<char *>(buf) + xb);
  int32_t *dI = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(buf) + xb + db);
  WCX_HIP(hipMemcpyAsync(dX, Xs, xb, hipMemcpyHostToDevice, ctx->stream));
  rc = wcx_newref_topk_dev(ctx, dX, B, S, chr_cum, n_chr, row_begin, row_end, k, mode, dI, dD);
  if (rc) return rc;
  if (n_rows) {
    WCX_HIP(hipMemcpyAsync(out_idx, dI, ib, hipMemcpyDeviceToHost, ctx->stream));
    WCX_HIP(hipMemcpyAsync(out_dist, dD, db, hipMemcpyDeviceToHost, ctx->stream));
  }
  WCX_HIP(hipStreamSynchronize(ctx->stream));
  return WCX_OK;
}

}  // extern "C"
