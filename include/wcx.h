/* wcx.h -- C-ABI of libwcx_hip.so: the MI355X (gfx950) implementation of the WisecondorX
 * newref / predict hot path.  Plain pointers and sizes only; every entry point returns an
 * int status (0 = WCX_OK) and records a message retrievable with wcx_last_error().
 *
 * The reference (CenterForMedicalGeneticsGhent/WisecondorX v1.2.10) has no FFI: its seams
 * are Python functions.  Each entry point below names the reference call site it replaces
 * (paths relative to src/wisecondorx/).  INTEGRATION.md shows the ctypes binding a
 * maintainer would add at each of those call sites.
 *
 * Conventions
 *  - "Xs" is the PCA-corrected bin-feature matrix in SAMPLE-MAJOR layout double[S][B]; these
 *    are exactly the bytes of the Fortran-ordered (B,S) array newref_tools.train_pca returns
 *    (newref_tools.py:147) and newref_control saves (newref_control.py:68), so a NumPy
 *    F-ordered array is passed zero-copy.
 *  - chr_cum[n_chr] = masked_bins_per_chr_cum (newref_control.py:64-66); chromosome c owns
 *    rows [chr_cum[c-1], chr_cum[c]).
 *  - Reference-bin indices are in the reference's "own chromosome removed" index space
 *    (newref_tools.py:192-202): candidate row g of a target in chromosome [cs,ce) is stored
 *    as g if g < cs, else g-(ce-cs).
 *  - *_dev entry points take DEVICE pointers and are asynchronous on the context's stream;
 *    the others take HOST pointers and are synchronous.
 */
#ifndef WCX_H
#define WCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WCX_OK 0
#define WCX_ERR_ARG 1
#define WCX_ERR_HIP 2
#define WCX_ERR_NOMEM 3
#define WCX_ERR_UNSUPPORTED 4

typedef struct wcx_ctx wcx_ctx; /* one per (process, GPU): device id, stream, scratch */
typedef struct wcx_ref wcx_ref; /* a reference (indexes/distances) resident in HBM    */

int wcx_version(void);
const char *wcx_last_error(void); /* thread-local, never NULL */

/* ---- context / memory ------------------------------------------------------------- */
/* stream == NULL: the context creates (and owns) its own non-blocking HIP stream; otherwise the
 * caller's hipStream_t is used.  The default (null) stream's handle IS 0, so it is named by
 * WCX_STREAM_DEFAULT: a caller that shares PyTorch's current stream passes
 * torch.cuda.current_stream().cuda_stream, or WCX_STREAM_DEFAULT when that value is 0. */
#define WCX_STREAM_DEFAULT ((void *)1)
int wcx_ctx_create(int device, void *stream, wcx_ctx **out);
int wcx_ctx_destroy(wcx_ctx *ctx);
/* Diagnostics only, per context: switches used by the profiling scripts (0 = normal operation).
 * 4 = per-phase cycle accounting of the screen kernel (results stay valid, ~20 % slower); 1 (no
 * shortlist appends), 16 / 64 (null ratios without gathers / without selection) are ablations whose
 * results are INVALID; bits 8.. = compaction trigger level.  Returns the previous value. */
int wcx_debug_flags(wcx_ctx *ctx, int flags);
int wcx_sync(wcx_ctx *ctx);
/* Overlap of INDEPENDENT searches on several contexts of one device (the A, F and M passes of newref,
 * newref_control.py:90-109 runs them one after another): wcx_sweep_event returns the event this context's
 * searches record when their MFMA sweep is done and the L2-bound exact refine begins; another context
 * told to wcx_wait_event on it starts its own (MFMA-bound) sweep beside that refine.  Must be called
 * AFTER the search that records the event has been enqueued. */
int wcx_sweep_event(wcx_ctx *ctx, void **out_event);
int wcx_wait_event(wcx_ctx *ctx, void *event);
int wcx_malloc(wcx_ctx *ctx, size_t bytes, void **dptr);
int wcx_free(wcx_ctx *ctx, void *dptr);
int wcx_memcpy_h2d(wcx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int wcx_memcpy_d2h(wcx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* Duration in ms of the last launch of the named kernel group on this context, measured
 * with hipEvents on the context's stream ("topk", "null_ratios", "normalize", "cutoff",
 * "weights", "cbs", "segment_z"); synchronises the stream.  <0 if never launched. */
double wcx_last_kernel_ms(wcx_ctx *ctx, const char *name);
/* Prefix for the timer names of the calls that follow ("A:" -> "A:topk", ...; NULL or "" = none), so
 * that a caller running several passes per step (newref's A / F / M) can read each pass's timers at
 * the end without synchronising in between. */
int wcx_timer_tag(wcx_ctx *ctx, const char *tag);
/* Counters of the last wcx_newref_topk*: [0] rows searched, [1] candidate pairs evaluated,
 * [2] shortlist compactions, [3] rows that fell back to the exact brute-force path,
 * [4] pairs that passed the MFMA screen (shortlist appends), [5] pairs refined exactly,
 * [6] tile pairs of the symmetric sweep that took the per-output path, [7] its row-direction appends,
 * [8..13] one-directional sweep with wcx_debug_flags(4): per-phase wave cycles; symmetric sweep:
 * [8..11] column / row gates opened and column / row events, [14..15] 0,
 * [16] sum of the trial indices the hub-count estimators chose, [17] rows they left without an estimate,
 * [18] rows that sent the symmetric sweep into its second attempt (0: the first one stood), [19..23] 0. */
int wcx_last_topk_stats(wcx_ctx *ctx, int64_t out[24]);

/* Device transpose of a row-major double matrix: d_dst[c][r] = d_src[r][c] (rows < 2^21).  A
 * multi-GPU build all-gathers row shards of the (bins x samples) matrix; this turns the result
 * into the sample-major layout the search takes (the bytes of the reference's F-ordered array). */
int wcx_transpose_dev(wcx_ctx *ctx, const double *d_src, int64_t rows, int64_t cols, double *d_dst);

/* Multi-GPU plumbing: an RCCL all-gather delivers `world` equally padded row shards
 * (shard r = rows [start_r, start_r+1) with the reference's _get_part boundaries,
 * newref_tools.py:244-247, at padded row r * pad_rows).  wcx_gather_transpose_dev turns the padded
 * shards of the row-major (bins x samples) matrix straight into the sample-major layout the search
 * takes; wcx_compact_rows_dev turns padded shards of a row-major table (indexes, distances, null
 * ratios; row_bytes a multiple of 4) into the dense table.  world <= 64. */
int wcx_gather_transpose_dev(wcx_ctx *ctx, const double *d_src, int world, int64_t pad_rows,
                             int64_t B, int S, double *d_dst);
int wcx_compact_rows_dev(wcx_ctx *ctx, const void *d_src, int world, int64_t pad_rows, int64_t B,
                         int64_t row_bytes, void *d_dst);

/* ---- newref: PCA correction of the reference samples (input producer of the search) ---- */
/* Replaces the numerical part of newref_tools.train_pca (newref_tools.py:138-147: PCA with 5
 * components, X = t / inverse_transform(transform(t))) and the distance-to-median profile of the
 * PCA-distance bin filter (newref_control.py:38-47).  Exact, deterministic thin SVD through the
 * S x S Gram matrix instead of the reference's unseeded randomized SVD.  Two steps around the
 * caller's symmetric eigensolver (numpy.linalg.eigh):
 *   wcx_pca_begin   t_data double[S][B] sample-major (host) -> per-bin mean[B], Gram matrix
 *                   double[S][S] of the centred data
 *   wcx_pca_finish  u double[S][ncomp] (top eigenvectors), sv[ncomp] (singular values)
 *                   -> components double[ncomp][B] (sign as computed: the caller applies
 *                   scikit-learn's svd_flip), X double[S][B] sample-major (may be NULL),
 *                   dist_to_med double[B] (may be NULL).  ncomp must be 5.
 *   wcx_pca_end     frees the device buffers kept between the two. */
int wcx_pca_begin(wcx_ctx *ctx, const double *t_data, int64_t B, int S, double *mean_out,
                  double *gram_out);
int wcx_pca_finish(wcx_ctx *ctx, const double *u, const double *sv, int ncomp, double *comps_out,
                   double *X_out, double *dist_to_med_out);
int wcx_pca_end(wcx_ctx *ctx);

/* ---- newref: bin mask and depth normalisation from device-resident bin counts ---------- */
/* d_counts int32 [S_all][n_bins]: every sample's per-chromosome bin counts (after the gender
 * correction) laid out over the longest sample's bins per chromosome, zero padded -- the matrix
 * the reference's np.zeros + copy loops build (newref_tools.py:80-91, :113-122), before any
 * division.  sel int32[ns] (host): the samples of the call (a pass's gender subset), in order.
 *   wcx_prep_mask_dev         replaces newref_tools.get_mask (newref_tools.py:77-102): counts /
 *                             per-sample total, summed per bin over the selected samples, mask =
 *                             sum > 5 % of the median of the positive sums -> mask_out uint8[n_bins]
 *                             (host); the per-bin sums are formed in NumPy's pairwise order, so the
 *                             threshold decision sees the bits the reference's np.sum produces;
 *                             sum_per_bin_out double[n_bins] (host, may be NULL) returns them.
 *   wcx_pca_begin_counts_dev  replaces newref_tools.normalize_and_mask (newref_tools.py:110-129) +
 *                             wcx_pca_begin: counts of bins pos[0..B) (host int32: the kept bins of
 *                             the pass, all < n_bins_pass) divided by the sample's total over bins
 *                             [0, n_bins_pass) (the chromosomes of the pass) are written straight
 *                             into the PCA stage's device matrix; then as wcx_pca_begin.
 *   wcx_pca_corrected_dev             device pointer of the corrected matrix X double[S][B] (sample-major)
 *                             wcx_pca_finish left in HBM: the input of wcx_newref_topk_dev /
 *                             wcx_null_ratios_dev without a host round trip.  Valid until the next
 *                             wcx_pca_begin* / wcx_pca_end on this context. */
int wcx_prep_mask_dev(wcx_ctx *ctx, const int32_t *d_counts, int64_t n_bins, const int32_t *sel, int ns,
                      unsigned char *mask_out, double *sum_per_bin_out);
int wcx_pca_begin_counts_dev(wcx_ctx *ctx, const int32_t *d_counts, int64_t n_bins, const int32_t *sel,
                             int ns, int64_t n_bins_pass, const int32_t *pos, int64_t B, double *mean_out,
                             double *gram_out);
int wcx_pca_corrected_dev(wcx_ctx *ctx, double **dX_out);

/* ---- newref: reference-bin search ------------------------------------------------- */
/* Replaces newref_tools.get_ref_for_bins (newref_tools.py:255-278) as driven by
 * newref_tools.get_reference (newref_tools.py:176-206) for target rows
 * [row_begin,row_end): for every target row, the k candidate rows outside its own
 * chromosome with the smallest squared Euclidean distance
 *   d = sum_{j=0..S-1} (X[c][j]-X[t][j])^2   (sequential fp64, separately rounded -- the
 *   value NumPy produces at newref_tools.py:260 on the F-ordered matrix),
 * ordered by (d, candidate index) ascending; candidates with d >= 1e10 or NaN are never
 * admitted; short rows are padded with index -1 / distance 1e10.
 * As in newref_tools.py:186-191, when n_chr > 22 (a gonosomal pass) only rows of chromosome
 * index 22/23 (X/Y) are searched; other rows get index 0 / distance 1.
 * mode: 0 = auto, 1 = exact fp64 brute force, 2 = MFMA screen + exact fp64 refine.
 * out_idx int32[row_end-row_begin][k], out_dist double[row_end-row_begin][k]. */
int wcx_newref_topk(wcx_ctx *ctx, const double *Xs, int64_t B, int S, const int64_t *chr_cum,
                    int n_chr, int64_t row_begin, int64_t row_end, int k, int mode,
                    int32_t *out_idx, double *out_dist);
int wcx_newref_topk_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                        const int64_t *chr_cum /*host*/, int n_chr, int64_t row_begin,
                        int64_t row_end, int k, int mode, int32_t *d_out_idx,
                        double *d_out_dist);

/* Row-sharded search of an autosomal pass WITH the symmetric sweep (multi-GPU; the parts of
 * newref_control.py:90-109 / newref_tools.py:244-247, one per rank).  Every rank holds all of X (one
 * all-gather) and builds the same sweep order and thresholds; the tile PAIRS are dealt out to the ranks,
 * each pair computed once for both directions, and every hit becomes a 16-byte record
 * (row, partner sweep position, screen distance bits, 0) for the rank that owns the row:
 *   wcx_newref_sym_sweep_dev    rank `part` of `n_parts` (<= 32); row_bounds int64[n_parts + 1]: rank r owns
 *                               rows [row_bounds[r], row_bounds[r + 1]).  counts_out int64[n_parts] (host):
 *                               records for each destination rank; all -1 when this rank's record pool
 *                               overflowed: the exchange is then VOID on every rank (the counts reach all
 *                               peers first) and each rank finishes with n_recv = -1.  Synchronises.
 *                               WCX_ERR_UNSUPPORTED where the symmetric sweep does not apply (K < 256,
 *                               B < 32768, a gonosomal pass): use wcx_newref_topk_dev on the row range.
 *   wcx_newref_sym_records_dev  the records, grouped by destination rank in rank order, into d_send
 *                               (device, sum(counts) x 16 bytes)
 *   -- the caller exchanges them: ONE all-to-all --
 *   wcx_newref_sym_finish_dev   d_recv: the n_recv records received for this rank's rows: lists, final
 *                               cut, exact fp64 refine, exact redo -> out_idx int32[own rows][k],
 *                               out_dist double[own rows][k], identical to wcx_newref_topk_dev's.
 *                               n_recv < 0 (void exchange): every own row is redone by the exact kernel. */
int wcx_newref_sym_sweep_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S, const int64_t *chr_cum,
                             int n_chr, int k, int part, int n_parts, const int64_t *row_bounds,
                             int64_t *counts_out);
int wcx_newref_sym_records_dev(wcx_ctx *ctx, void *d_send);
int wcx_newref_sym_finish_dev(wcx_ctx *ctx, const void *d_recv, int64_t n_recv, int32_t *d_out_idx,
                              double *d_out_dist);

/* Replaces the null-ratio loop newref_tools.py:210-223: out[r][m] =
 * log2(X[row_begin+r][sid[m]] / median_k X[idx[r][k]][sid[m]]), the index row applied to the
 * FULL bin vector without re-offsetting (reference quirk, newref_tools.py:219-221; index -1
 * wraps to the last bin as in NumPy).  sample_ids are chosen by the host (random.sample,
 * newref_tools.py:214-217).  idx int32[n][k]; out double[n][n_ids].  Limits: n_ids <= 128
 * (the reference uses min(S, 100)), B < 2^25, n_ids * B < 2^31, k <= 2048. */
int wcx_null_ratios(wcx_ctx *ctx, const double *Xs, int64_t B, int S, const int32_t *idx,
                    int64_t row_begin, int64_t row_end, int k, const int32_t *sample_ids,
                    int n_ids, double *out);
/* Optional head start for wcx_null_ratios_dev: ranks the null samples (the part of the null-ratio
 * work that depends on X only) on an auxiliary stream of the context, concurrently with whatever
 * follows on the main stream (the search); the next wcx_null_ratios_dev with the same dXs and
 * sample_ids uses it.  Results are identical with or without this call.  When the search that
 * follows covers few rows (a gonosomal pass, a rank's shard of a 4- / 8-GPU build: B / rows >= 4)
 * the announced ranking is dropped and wcx_null_ratios_dev selects without one (on the high halves
 * of the values' keys, settled on the doubles) -- same results. */
int wcx_null_rank_prepare_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                              const int32_t *sample_ids /*host*/, int n_ids);
/* The same for rows whose reference-bin row is the dummy of a gonosomal pass (all indices 0,
 * newref_tools.py:186-191): the median of k copies of x[0] is x[0], so out[r][m] =
 * log2(X[row][sid[m]] / X[0][sid[m]]) exactly as newref_tools.py:219-221 computes it, without gathers. */
int wcx_null_ratios_dummy_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S, int64_t row_begin,
                              int64_t row_end, const int32_t *sample_ids /*host*/, int n_ids,
                              double *d_out);
int wcx_null_ratios_dev(wcx_ctx *ctx, const double *dXs, int64_t B, int S,
                        const int32_t *d_idx, int64_t row_begin, int64_t row_end, int k,
                        const int32_t *sample_ids /*host*/, int n_ids, double *d_out);

/* ---- predict ---------------------------------------------------------------------- */
/* Upload a reference's indexes/distances (reference .npz keys "indexes{ap}",
 * "distances{ap}", "masked_bins_per_chr_cum{ap}") once per batch. */
int wcx_ref_upload(wcx_ctx *ctx, const int32_t *idx, const double *dist, int64_t B, int k,
                   const int64_t *chr_cum, int n_chr, wcx_ref **out);
int wcx_ref_wrap_dev(wcx_ctx *ctx, const int32_t *d_idx, const double *d_dist, int64_t B,
                     int k, const int64_t *chr_cum /*host*/, int n_chr, wcx_ref **out);
int wcx_ref_free(wcx_ctx *ctx, wcx_ref *ref);

/* Replaces predict_tools.get_optimal_cutoff (predict_tools.py:74-82). */
int wcx_cutoff(wcx_ctx *ctx, const wcx_ref *ref, int repeats, double *cutoff);
/* Replaces predict_tools.get_weights (predict_tools.py:152-155): out[B]. */
int wcx_weights(wcx_ctx *ctx, const wcx_ref *ref, double *out);
/* The same with the result left on the device (d_out double[B]), asynchronous on the stream. */
int wcx_weights_dev(wcx_ctx *ctx, const wcx_ref *ref, double *d_out);
/* Replaces predict_tools.normalize_repeat (predict_tools.py:94-142) for a batch of
 * n_samples projected sample vectors x double[n_samples][B]: three masked passes, rows from
 * `ct` (first row of chromosome index `cp`).  Outputs per sample: z,r,n double[B-ct];
 * m_lr, m_z double[n_samples]. */
int wcx_predict_normalize(wcx_ctx *ctx, const wcx_ref *ref, const double *x, int n_samples,
                          double cutoff, int64_t ct, int cp, double *out_z, double *out_r,
                          double *out_n, double *out_mlr, double *out_mz);
int wcx_predict_normalize_dev(wcx_ctx *ctx, const wcx_ref *ref, const double *d_x,
                              int n_samples, double cutoff, int64_t ct, int cp,
                              double *d_out_z, double *d_out_r, double *d_out_n,
                              double *d_out_mlr, double *d_out_mz);

/* Replaces, for the device-resident outputs of wcx_predict_normalize_dev of ONE sample without a
 * gonosomal pass, main.py:246-250 (z - m_z, w / nanmean(w)), predict_control.get_post_processed_result
 * for r, z, w (predict_control.py:49-63: bins with fewer than minrefbins reference bins -> 0, inflate
 * to the unmasked length) and predict_tools.log_trans (predict_tools.py:180-193).  d_pos int32[B] =
 * position of masked bin i in the unmasked vector.  out_r (log2 ratios), out_z, out_w: HOST
 * double[n_bins] (pinned memory makes the copies asynchronous DMA); synchronises the stream. */
int wcx_post_process_dev(wcx_ctx *ctx, const double *d_z, const double *d_r, const double *d_n,
                         const double *d_w, int64_t B, const double *d_m_lr, const double *d_m_z,
                         double minrefbins, const int32_t *d_pos, int64_t n_bins, double *out_r,
                         double *out_z, double *out_w);

/* Sample preparation of a batch on the device: predict_tools.coverage_normalize_and_mask
 * (predict_tools.py:32-48: counts / total read count over the bins of this pass, masked bins dropped)
 * followed by predict_tools.project_pc (predict_tools.py:56-65 with the scikit-learn <= 1.4 transform
 * the reference pins: x / (((x - mean) . C^T) . C + mean)).  d_counts int32[n_samples][n_bins]: the
 * samples' bin counts laid out over the reference's bins (each chromosome truncated or zero-padded to
 * bins_per_chr{ap}: a host memcpy per chromosome); d_pos int32[B] = unmasked position of masked bin i;
 * d_mean double[B], d_comps double[n_comp][B] = pca_mean{ap}, pca_components{ap} (n_comp must be 5).
 * d_x double[n_samples][B] receives the projected vectors wcx_predict_normalize_dev takes. */
int wcx_predict_prep_dev(wcx_ctx *ctx, const int32_t *d_counts, int n_samples, int64_t n_bins,
                         const int32_t *d_pos, int64_t B, const double *d_mean, const double *d_comps,
                         int n_comp, double *d_x);

/* The general, batched form: main.py:242-257 (autosomal + gonosomal results appended, z - m_z of the
 * autosomal pass, w = append(wA * nanmean(wG), wG * nanmean(wA)) / nanmean(...), all weights 1 if any
 * is NaN / inf), predict_control.get_post_processed_result for r, z, w (predict_control.py:49-63)
 * and predict_tools.log_trans (predict_tools.py:180-193) for n_samples samples whose normalisation
 * outputs are device-resident:
 *   autosomal  d_zA, d_rA, d_nA double[n_samples][BA] (wcx_predict_normalize_dev, ct = 0), d_wA double[BA]
 *   gonosomal  d_zG, d_rG, d_nG double[n_samples][BG] (rows from ct of the .F / .M reference), d_wG
 *              double[BG] (wcx_weights_dev of that reference, rows from ct); BG = 0: autosomes only
 *   d_m_lr, d_m_z double[n_samples]: the autosomal pass's medians; d_pos int32[BA + BG]: position of
 *   merged masked bin i in the unmasked vector (mask{ap}).
 * Outputs stay on the DEVICE: d_out_r (log2 ratios), d_out_z, d_out_w double[n_samples][n_bins]
 * (masked-out bins 0) -- the inputs of wcx_cbs_batch_dev / wcx_segment_z_dev.  *weights_fallback
 * (may be NULL; non-NULL synchronises) = 1 when the all-ones rule fired (main.py:252-256 logs a warning). */
int wcx_post_process_merge_dev(wcx_ctx *ctx, const double *d_zA, const double *d_rA,
                               const double *d_nA, const double *d_wA, int64_t BA,
                               const double *d_zG, const double *d_rG, const double *d_nG,
                               const double *d_wG, int64_t BG, int n_samples, const double *d_m_lr,
                               const double *d_m_z, double minrefbins, const int32_t *d_pos,
                               int64_t n_bins, double *d_out_r, double *d_out_z, double *d_out_w,
                               int *weights_fallback);

/* ---- row-sharded predict (multi-GPU) --------------------------------------
 * A handle made by wcx_ref_wrap_rows_dev holds only rows [row0,row0+nrows) of indexes/distances
 * (the block this rank built).  The host drives normalize_repeat (predict_tools.py:94-108)
 * pass by pass and exchanges the masked copy between passes:
 *   wcx_cutoff_moments_dev  phase 0: out2 = {sum, count} of local dist < cutoff;
 *                           phase 1: out2 = {sum (dist-mean)^2, .}   -> all-reduce on the host
 *   wcx_predict_pass_dev    one _normalize_once pass (predict_tools.py:111-142) over the local
 *                           rows >= ct: reads the full x / copy_in vectors [B], writes z, r, n, log2 r
 *                           at position (row - ct) and copy_out[row]; build_mask != 0 on the
 *                           first pass (turns dist < cutoff into the per-row selection mask);
 *                           r and log2 r are only computed when last != 0 (the earlier passes of
 *                           normalize_repeat feed nothing but the z-mask, predict_tools.py:99-108)
 *   wcx_nanmedian2_dev      np.nanmedian of two arrays (m_lr, m_z; predict_tools.py:105-106) */
int wcx_ref_wrap_rows_dev(wcx_ctx *ctx, const int32_t *d_idx, const double *d_dist, int64_t B,
                          int k, const int64_t *chr_cum /*host*/, int n_chr, int64_t row0,
                          int64_t nrows, wcx_ref **out);
int wcx_cutoff_moments_dev(wcx_ctx *ctx, const wcx_ref *ref, double cutoff, double mean, int phase,
                           double *out2 /*host*/);
int wcx_predict_pass_dev(wcx_ctx *ctx, wcx_ref *ref, const double *d_x, const double *d_copy_in,
                         double *d_copy_out, double cutoff, int64_t ct, int build_mask, int last,
                         double *d_z, double *d_r, double *d_n, double *d_lr);
int wcx_nanmedian2_dev(wcx_ctx *ctx, const double *d_a0, const double *d_a1, int64_t n,
                       double *d_out0, double *d_out1);

/* Replaces predict_tools.exec_cbs -> overall_tools.exec_R -> include/CBS.R (main.py:279,
 * predict_tools.py:242-257, CBS.R:21-132): weighted circular binary segmentation of the
 * per-chromosome log2 ratios.  r,w double[n_bins] (0 = missing), chr_off int64[n_chr+1].
 * out_seg: rows of 4 doubles {chr (0-based), start (0-based), end (exclusive), ratio};
 * *out_count receives the number of segments (<= cap). */
int wcx_cbs(wcx_ctx *ctx, const double *r, const double *w, const int64_t *chr_off, int n_chr,
            double alpha, int64_t binsize, uint64_t seed, double *out_seg, int cap,
            int *out_count);
/* The same for a batch of n_samples samples that share the chromosome layout: r, w
 * double[n_samples][n_bins] (sample-major; chr_off[n_chr] <= n_bins), out_seg
 * double[n_samples][cap][4], out_count int[n_samples].  All chromosomes of all samples advance
 * together (level-synchronous recursion): this is the throughput path of a predict batch. */
int wcx_cbs_batch(wcx_ctx *ctx, const double *r, const double *w, int n_samples, int64_t n_bins,
                  const int64_t *chr_off, int n_chr, double alpha, int64_t binsize, uint64_t seed,
                  double *out_seg, int cap, int *out_count);
/* wcx_cbs_batch on DEVICE-resident r, w (wcx_post_process_merge_dev's outputs).  The NA-free series
 * (CBS.R:41-63) are compacted on the device and stay there: the host's decisions between the rounds
 * fetch a series' x | w when one of its segments first needs them (a few per cent of the series), the
 * segments' post-processing (CBS.R:84-129) reads the bin positions (exported into pinned memory beside
 * the rounds' kernels) and takes its weighted means from a device kernel.  r and w themselves only
 * come down when they hold +-inf (dropped from the series, but not "NA" to CBS.R:84-113).  Same
 * results, bit for bit, as wcx_cbs_batch on host copies of r and w. */
int wcx_cbs_batch_dev(wcx_ctx *ctx, const double *d_r, const double *d_w, int n_samples,
                      int64_t n_bins, const int64_t *chr_off, int n_chr, double alpha, int64_t binsize,
                      uint64_t seed, double *out_seg, int cap, int *out_count);
/* Diagnostics of the CBS calls on this context since its creation: out[0] = hybrid tests decided
 * by the short-arc bound (their permutations were not run), out[1] / out[2] = block pairs of the arc
 * search evaluated arc by arc / existing (block-bound pruning), out[3] reserved. */
int wcx_cbs_stats(wcx_ctx *ctx, int64_t out[4]);
/* Per-test records of the LAST wcx_cbs / wcx_cbs_batch call on this context, kept when
 * wcx_debug_flags(ctx, 128) was set before it (what tests/test_gpu_cbs_oracle.py compares with the
 * NumPy oracle).  One record = 20 doubles: sample, chromosome, lo, hi (segment in the chromosome's
 * NA-free series), n, best arc bi, bj, t^2, tail p (NaN if n <= 200; NEGATIVE = "at least |value|": a
 * proven lower bound that already exceeds alpha, the series was not evaluated), delta, why (1 constant /
 * invalid, 2 t <= 0.1, 3 t >= 7, 4 tail p > alpha, 5 permutations, 6 short-arc bound), budget nrejc,
 * exceedances nrej and permutations np at the stop (-1 without permutations), significant,
 * change-points kept, then (kept, nrej; -1 = t^2 > 25 rule, -2 = no test) of the two edge tests.
 * *count receives the number of records available; at most cap_records are copied. */
int wcx_cbs_trace(wcx_ctx *ctx, double *out, int cap_records, int *count);
/* The sequential stopping boundary of the permutation tests (Venkatraman & Olshen 2007; DNAcopy's
 * getbdry(eta, nperm, max.ones), called from segment() -- the DNAcopy call of CBS.R:73 -- with
 * eta = 0.05, nperm = 10000, max.ones = floor(nperm * alpha) + 1): out int32[max_ones (max_ones + 1) / 2],
 * block j (length j, offset j (j - 1) / 2) = stopping points of a test that tolerates j - 1
 * exceedances.  Host-only, no device needed. */
int wcx_cbs_getbdry(double eta, int nperm, int max_ones, int32_t *out);
/* Replaces overall_tools.get_z_score (overall_tools.py:88-119).  nr double[n_bins][m]
 * (rows of masked bins ignored; pad ragged rows with NaN), seg as produced by wcx_cbs;
 * out_z double[n_seg] (NaN where undefined), out_nnull double[n_seg] (may be NULL) = number of
 * finite null-segment averages: 0 is where the reference returns the string "nan". */
/* Attach (upload once) the dense null-ratio matrix of a reference to the context; later
 * wcx_segment_z calls may pass nr = NULL to use it (batches: 165 MB at 15 kb stays in HBM).
 * nr = NULL detaches. */
int wcx_set_null_matrix(wcx_ctx *ctx, const double *nr, int64_t n_bins, int m);
/* The same from a DEVICE table of the masked bins' rows (d_nr double[B][m], e.g. straight out of
 * wcx_null_ratios_dev): inflated to n_bins rows on the device with the reference's mask
 * (mask[n_bins] host bytes, B of them non-zero; masked-out rows = 0, predict_tools.py:163-170).
 * Asynchronous on the context's stream: the bin -> row map is kept per mask, so only the first
 * call with a mask uploads anything and waits; `mask` is read before the call returns; d_nr is read BY
 * THE STREAM: it must stay valid (and unmodified by other streams) until the context's stream has passed
 * this call. */
int wcx_set_null_matrix_dev(wcx_ctx *ctx, const double *d_nr, int64_t B, int m,
                            const unsigned char *mask, int64_t n_bins);
int wcx_segment_z(wcx_ctx *ctx, const double *r, const double *w, const double *nr, int m,
                  const int64_t *chr_off, int n_chr, const double *seg, int n_seg,
                  double *out_z, double *out_nnull);
/* The same with DEVICE-resident r, w (one sample's rows of wcx_post_process_merge_dev's outputs)
 * and the attached null matrix. */
int wcx_segment_z_dev(wcx_ctx *ctx, const double *d_r, const double *d_w, const int64_t *chr_off,
                      int n_chr, const double *seg, int n_seg, double *out_z, double *out_nnull);
/* ... and for a batch in ONE call: d_r, d_w double[n_samples][chr_off[n_chr]], the segments of all
 * samples listed sample by sample (seg_count[i] of them for sample i, as wcx_cbs_batch_dev leaves them
 * after compaction), out_z / out_nnull in the same order. */
int wcx_segment_z_batch_dev(wcx_ctx *ctx, const double *d_r, const double *d_w, int n_samples,
                            const int64_t *chr_off, int n_chr, const double *seg,
                            const int *seg_count, double *out_z, double *out_nnull);

/* ---- output tables (host code, no device work) --------------------------------
 * The rows of ID_bins.bed for one chromosome (predict_output.py:51-75 of the reference):
 *   "{chr}\t{start}\t{end}\t{chr}:{start}-{end}\t{ratio}\t{zscore}\n", start = i binsize + 1, end =
 * (i + 1) binsize, a value of 0 printed as "nan", every other as Python's str(float) (shortest
 * round-trip digits, exponent form below 1e-4 and from 1e16, ".0" on integers).  Returns the number of
 * bytes written to out, or -1 if cap is too small (n * (2 strlen(chr) + 140) always suffices). */
int64_t wcx_format_bins_bed(const char *chr_name, int64_t n, int64_t binsize, const double *r,
                            const double *z, char *out, int64_t cap);
/* str(float) of n doubles, each followed by `sep`; cap >= 26 n.  Returns the bytes written or -1. */
int64_t wcx_format_floats(const double *v, int64_t n, char sep, char *out, int64_t cap);

/* Bin counts of a batch of samples laid out over the reference's bins (predict_tools.py:36-44 per sample:
 * every chromosome truncated or zero-padded to bins_per_chr): src[s * n_chr + c] -> the int32 counts of
 * chromosome c of sample s, len[s * n_chr + c] of them; out int32[n_samples][sum(bins_per_chr)] (host; pinned
 * memory makes the upload that follows a true DMA).  Host code, n_threads threads over the samples. */
int wcx_layout_counts(const int32_t *const *src, const int64_t *len, int n_samples, int n_chr,
                      const int64_t *bins_per_chr, int32_t *out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* WCX_H */
