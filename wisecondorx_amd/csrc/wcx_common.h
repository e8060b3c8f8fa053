// Internal shared declarations for libwcx_hip.so (gfx950 only; no CUDA/HIP dual path).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/wcx.h"

void wcx_set_error(const char *fmt, ...);

#define WCX_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      wcx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return WCX_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

#define WCX_ARG(cond, msg)                          \
  do {                                              \
    if (!(cond)) {                                  \
      wcx_set_error("bad argument: %s", msg);       \
      return WCX_ERR_ARG;                           \
    }                                               \
  } while (0)

struct KernelTimer {
  hipEvent_t start = nullptr, stop = nullptr;
  bool used = false;
};

struct wcx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::map<std::string, KernelTimer> timers;
  std::string timer_tag;                  // prefix of the timer names (wcx_timer_tag)
  int debug_flags = 0;                    // diagnostics (wcx_debug_flags): per context
  int64_t topk_stats[4] = {0, 0, 0, 0};
  unsigned long long *d_stats = nullptr;  // 32 device counters (wcx_last_topk_stats reports the first 24)
  void *d_small = nullptr;                // 8 KB of persistent device workspace (radix-select state)
  // growable device scratch owned by the context
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  std::vector<double> cbs_trace;     // per-test records of the last wcx_cbs_batch (wcx_debug_flags & 128)
  long long cbs_pairs_listed = 0, cbs_pairs_total = 0;   // block pairs evaluated / existing (arc search)
  long long cbs_shortcuts = 0;       // hybrid CBS tests decided by the short-arc bound (no permutations)
  void *host_scratch = nullptr;      // pinned host staging (wcx_host_scratch)
  size_t host_scratch_bytes = 0;
  void *host_scratch2 = nullptr;     // pinned copy of r | w for wcx_cbs_batch_dev
  size_t host_scratch2_bytes = 0;
  void *scratch2 = nullptr;
  size_t scratch2_bytes = 0;
  // null-ratio matrix attached for wcx_segment_z (wcx_set_null_matrix)
  double *d_nullm = nullptr;
  int64_t nullm_bins = 0;
  int nullm_m = 0;
  int32_t *d_nullsrc = nullptr;     // bin -> row of the masked table (-1: masked out), kept per mask
  int64_t nullsrc_bins = 0;
  unsigned long long nullsrc_hash = 0;
  std::vector<unsigned char> nullsrc_mask;   // the mask the map was built from (compared on a hash hit)
  // null-sample ranking done ahead on an auxiliary stream (wcx_null_rank_prepare_dev)
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_main = nullptr, ev_rank = nullptr;
  hipStream_t sweep_stream = nullptr;        // second stream of the screen sweep (see wcx_topk_screen_launch)
  hipEvent_t ev_sweep0 = nullptr, ev_sweep1 = nullptr;
  hipEvent_t ev_after_sweep = nullptr;       // recorded by every search after its sweep (wcx_sweep_event)
  hipStream_t copy_stream = nullptr;         // device -> host copies of wcx_cbs_batch_dev beside its kernels
  hipEvent_t ev_cbs_fill = nullptr, ev_cbs_xw = nullptr;
  void *d_rank = nullptr;
  size_t rank_bytes = 0;
  const double *rank_X = nullptr;
  int64_t rank_B = 0;
  std::vector<int32_t> rank_ids;
  // selection masks of reference handles (predict): buffers are lent to a handle and taken back by
  // wcx_ref_free WITHOUT a hipFree -- a free synchronises the device in the middle of every predict
  struct SelSlot { void *p; size_t bytes; bool used; };
  std::vector<SelSlot> sel_pool;
  bool rank_pending = false;   // ranking requested, not yet started (wcx_aux_kick)
  // PCA stage (wcx_pca_begin .. wcx_pca_end): t | X | mean | components | dist_to_med
  void *d_pca = nullptr;
  size_t pca_bytes = 0;
  int64_t pca_B = 0;
  int pca_S = 0;
  // host staging for small async uploads (kept alive until the next stream sync)
  std::vector<std::vector<unsigned char>> stage;
  void *sym_state = nullptr;   // row-sharded symmetric sweep between its phases (newref_topk_screen.hip)
};
void wcx_sym_state_free(wcx_ctx *ctx);
int wcx_sym_shard_sweep(wcx_ctx *ctx, const double *dXs, int64_t B, int S, const int64_t *chr_cum, int n_chr,
                        int k, int part, int n_parts, const int64_t *row_bounds, int64_t *counts_out);
int wcx_sym_shard_records(wcx_ctx *ctx, void *d_send);
int wcx_sym_shard_finish(wcx_ctx *ctx, const void *d_recv, int64_t n_recv, int32_t *d_out_idx,
                         double *d_out_dist);

struct wcx_ref {
  const int32_t *d_idx = nullptr;
  const double *d_dist = nullptr;
  bool owned = false;
  int64_t B = 0;
  int k = 0;
  int64_t row0 = 0, nrows = 0;          // rows held by d_idx / d_dist (all B unless row-sharded)
  unsigned long long *d_sel = nullptr;   // selection mask of those rows (dist < cutoff)
  size_t sel_bytes = 0;
  std::vector<int64_t> chr_cum;
  int64_t *d_chr_cum = nullptr;
};

int wcx_scratch(wcx_ctx *ctx, size_t bytes, void **out);
int wcx_scratch2(wcx_ctx *ctx, size_t bytes, void **out);
int wcx_upload_small(wcx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int wcx_transpose_launch(wcx_ctx *ctx, const double *src, int64_t rows, int64_t cols, double *dst);
int wcx_timer_begin(wcx_ctx *ctx, const char *name);
int wcx_timer_end(wcx_ctx *ctx, const char *name);

// Own-chromosome row range of a block of target rows, built on the host.
struct TopkBlock {
  int64_t row0;  // first target row (global row id)
  int32_t nrows; // <= TM
  int32_t pad;
  int64_t cs, ce; // own chromosome [cs,ce): excluded from the candidates
};

int wcx_topk_exact_launch(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                          const std::vector<TopkBlock> &blocks, int64_t row_begin,
                          int64_t n_rows, int k, int32_t *d_out_idx, double *d_out_dist);
constexpr int WCX_REDO_GRID = 128;  // workgroups of the device-driven tiled exact redo
constexpr int WCX_REDO_FAST = 128;  // flagged rows that take the device-wide redo path
size_t wcx_topk_redo_scratch_bytes(int k, int64_t B);
int wcx_host_scratch(wcx_ctx *ctx, size_t bytes, void **out);
int wcx_aux_kick(wcx_ctx *ctx);   // null_ratios.hip: start pending auxiliary-stream work
// null_ratios.hip, for the batched normalise (predict.hip): the rows of a row-major [n][B] matrix ranked
// (n <= 128; pieces Rg[group of 8 rows][bin][8], values by rank V[n][B]) into `base` (wcx_rank_bytes), and
// the medians of the LAST normalisation pass selected on those ranks.
struct WcxRankView { unsigned int *Rg; const double *V; int n_sg; };
struct WcxChrCum { int n_chr; int64_t cum[32]; };
size_t wcx_rank_bytes(int64_t B, int n);
int wcx_rank_rows_launch(const double *d_x, int64_t B, int n, char *base, hipStream_t st, WcxRankView *out);
int wcx_norm_rank_mark_launch(const WcxRankView &rk, const double *copyT, int64_t B, int NS, int n_samples,
                              hipStream_t st);
int wcx_norm_median_rank_launch(const WcxRankView &rk, const int32_t *d_idx, const unsigned long long *d_sel,
                                const double *xT, int64_t B, int k, int ipl, int NS, int n_samples, int64_t ct,
                                const WcxChrCum &chr, double *rT, double *lrT, hipStream_t st);
bool wcx_null_ratios_direct_pays(int64_t B, int64_t n_rows);
void wcx_aux_cancel_if_few_rows(wcx_ctx *ctx, int64_t B, int64_t n_rows);
int wcx_topk_exact_redo_launch(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                               const TopkBlock *d_blocks, const unsigned int *d_count,
                               const TopkBlock *d_tiles, const unsigned int *d_ntiles,
                               const int32_t *d_rowlist, void *scratch, int64_t row_begin, int k,
                               int32_t *d_out_idx, double *d_out_dist);
bool wcx_screen_supported(int64_t B, int S, int k);
int wcx_topk_screen_launch(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                           const int64_t *chr_cum, int n_chr,
                           const std::vector<TopkBlock> &exact_blocks, int64_t row_begin,
                           int64_t n_rows, int k, int32_t *d_out_idx, double *d_out_dist);
int wcx_fill_dummy_rows(wcx_ctx *ctx, int32_t *d_idx, double *d_dist, int64_t row_lo,
                        int64_t row_hi, int k);
