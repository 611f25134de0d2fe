/*
 * icnv.h -- C ABI of libicnv_hip.so: inferCNV's expression-smoothing chain,
 * i6/i3 HMM Viterbi and 2-D median denoise as hand-written HIP kernels for
 * AMD MI355X (gfx950).
 *
 * The reference (broadinstitute/infercnv, pure R, NeedsCompilation: no) has no
 * FFI layer; the boundary this library sits behind is the step-function
 * contract of infercnv::run()  f(infercnv_obj, scalars) -> infercnv_obj  on
 * infercnv_obj@expr.data (R/inferCNV_ops.R:771,817,865,911,952,1031,
 * 1255-1304,1469-1471,1573-1588; scripts/inferCNV.R:1116).  Each entry point
 * below names the reference function(s) it replaces.  INTEGRATION.md shows
 * the R-side .Call shim and the Python ctypes binding.
 *
 * Conventions
 *   - Matrices are column-major genes x cells, element (g, c) at x[g + G*c]:
 *     exactly R's layout of expr.data (R/inferCNV.R:18) and at the same time
 *     the "cell-major" HBM layout (one cell's genes are contiguous).
 *   - All indices are 0-based int32.  Genes of one chromosome are contiguous
 *     (.order_reduce, R/inferCNV.R:407): chr_start[] holds n_chr+1 offsets,
 *     chr_start[0] = 0, chr_start[n_chr] = G.
 *   - Group lists are "packed": idx[] concatenates the member cell indices of
 *     all groups, off[] holds n_grp+1 offsets into idx[].
 *   - Every function returns ICNV_OK (0) or an error code; the message is
 *     available from icnv_last_error() (thread-local).  The library never
 *     longjmps/aborts: an R shim turns codes into stop() after cleanup.
 *   - *_dev entry points take DEVICE pointers for matrices/outputs and enqueue
 *     all work on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) without synchronising the host.  Small descriptor arrays
 *     (chr_start, group lists, HMM parameters) are HOST pointers in both
 *     flavours and are copied to the device by the library.
 *   - The host-buffer flavours upload, run the *_dev path and download.
 *   - Caller owns every buffer it passes; inputs are never modified.
 *   - Threading: every call acts on the calling thread's current HIP device (icnv_init selects it).  Library state
 *     (workspace pool, emission-table cache, Viterbi statistics) is kept per device and guarded by mutexes, so
 *     threads driving DIFFERENT devices may call concurrently.  On one device use ONE stream at a time: scratch
 *     buffers return to the device's pool when a *_dev call returns while its kernels may still be queued, and the
 *     next call's work must be ordered behind them (same stream, or a stream the caller has made wait).
 */
#ifndef ICNV_H
#define ICNV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICNV_OK 0
#define ICNV_ERR_ARG 1        /* invalid argument                                   */
#define ICNV_ERR_HIP 2        /* HIP runtime error (message has the hipError string) */
#define ICNV_ERR_UNSUPPORTED 3 /* size/config outside what the kernels support        */
#define ICNV_ERR_UNDERFLOW 4  /* "Problems With Underflow" (R/inferCNV_HMM.R:1165)   */
#define ICNV_ERR_NOMEM 5

/* Stages of the smoothing chain, named by run()'s step numbers
 * (R/inferCNV_ops.R:771-1589).  Stages always execute in this order. */
#define ICNV_ST_SUBTRACT_REF_1 0x01u /* step  8 subtract_ref_expr_from_obs       :1678-1786 */
#define ICNV_ST_MAX_THRESH     0x02u /* step  9 apply_max_threshold_bounds       :2970-2983 */
#define ICNV_ST_SMOOTH         0x04u /* step 10 smooth_by_chromosome             :2406-2532 */
#define ICNV_ST_CENTER         0x08u /* step 11 center_cell_expr_across_chromosome (median) :2074-2109 */
#define ICNV_ST_SUBTRACT_REF_2 0x10u /* step 12 subtract_ref_expr_from_obs (again)         */
#define ICNV_ST_INVERT_LOG2    0x20u /* step 14 invert_log2                      :2814-2826 */
#define ICNV_ST_DENOISE        0x40u /* step 22 clear_noise_via_ref_mean_sd / clear_noise :2232-2346 */
#define ICNV_ST_ALL            0x7Fu
#define ICNV_ST_CENTER_MEAN    0x80u /* modifier: step 11 subtracts the mean instead of the median */
#define ICNV_ST_NA_AWARE       0x100u /* modifier: the matrix may hold NA / NaN.  The cells that do are recomputed with the reference's
                                         NA semantics (csrc/chain_na.hip: step 8 / 12 with bounds turn an NA into 0, the smoothing strips
                                         and re-inserts NAs per chromosome, the centre is taken over the values present,
                                         R/inferCNV_ops.R:1757-1768, 2098, 2487-2489, 2529); costs one extra pass over the input.  Without
                                         it a NaN is not looked for (run()'s chain input, log2(x + 1) of counts, has none).  Any gene count
                                         (fused, two-pass and three-pass chain); a call that runs in place keeps the flagged cells'
                                         input columns aside first (fused chain; ICNV_ERR_UNSUPPORTED in place beyond the LDS-resident
                                         limit); ICNV_ERR_UNSUPPORTED together with inv_log or noise_logistic (no silent default) */

/* ---- library state ------------------------------------------------------ */
int icnv_version(void);
const char *icnv_last_error(void);
/* Selects the HIP device for the calling thread's subsequent calls (-1 = keep
 * current).  Fails with ICNV_ERR_HIP when no gfx950-class GPU is usable. */
int icnv_init(int device);
/* Frees cached device workspaces. */
void icnv_shutdown(void);

/* Devices of the HOST-BUFFER entry points (the ones an R process reaches through the .Call shim).  n_devices = 0: every
 * visible device, n: devices 0 .. n-1, 1 (the default): the calling thread's current device.  With more than one device
 * icnv_smooth_chain and icnv_viterbi_cells split the cells into one contiguous block per device (one host thread, one
 * stream per device); the chain's reference statistics (per-gene sums of the reference groups) are added
 * on the host in device order.  icnv_viterbi_groups and icnv_median_filter deal WHOLE groups / tiles to the devices
 * (longest first onto the least loaded one; every worker packs the columns of its groups' cells, no exchange between
 * devices).  The remaining host-buffer entry points run on the current device.  This is what lets a
 * single R process (infercnv::run() is single-threaded) use the 8 GPUs of a node, each over its own PCIe link.
 * The *_dev entry points are not affected: a device-resident caller (one process per GPU, infercnv_amd/sharded.py)
 * shards by itself. */
int icnv_set_devices(int n_devices);
int icnv_get_devices(void);

/* Residency of the host-buffer entry points.  run() hands every step the matrix the previous step returned
 * (R/inferCNV_ops.R:771-1031, 1237-1309).  With icnv_residency(1) the library keeps the matrices it uploaded or
 * produced on the device(s) -- up to 6 per device, ICNV_RESIDENT_MAX_GB (default 64), given back under memory
 * pressure -- and recognises a host matrix BY CONTENT: its length, a strided sample of ~16 000 values as the quick
 * reject, then a 64-bit hash of every value (the host's cores hash the incoming matrix at memory speed, several times
 * faster than the upload it saves; a device reduction hashes what the library produced).  A recognised matrix is not
 * uploaded again.  Addresses play no part: the caller may free, reuse or edit host memory at any time -- one changed
 * element changes the hash and the matrix is uploaded (the hash is linear inside a 64-byte block: an edit of ONE word is
 * always seen; edits of several words of one block cancel only if sum_j delta_j K_j = 0 mod 2^64, odds 2^-64 for unrelated
 * data).  Whether it pays depends on the host: the hash is one pass over the matrix by the cores the process may use, the
 * upload it saves is one pass by the DMA engines at PCIe speed -- on a 16-core quota the two are about even (the bench
 * line's `host_path` shows both, with the phase times of icnv_host_path_stats).  Off by default, also in the R glue.
 *   icnv_residency_stats  out4 = {matrices recognised, matrices uploaded, resident bytes, resident matrices} */
int icnv_residency(int on);
void icnv_residency_drop(void);
int icnv_residency_stats(int64_t *out4);

/* Where a host-buffer call spends its time (wall-clock milliseconds accumulated since the last reset, per process):
 *   out[0] calls            host-buffer entry points that moved a matrix
 *   out[1] fingerprint_ms   residency: strided samples of incoming matrices
 *   out[2] hash_ms          residency: full content hashes on the host's cores        out[3] hash_threads (of the last hash)
 *   out[4] h2d_ms           time the uploading thread was busy                         out[5] h2d_bytes
 *   out[6] d2h_ms           time the downloading thread was busy                       out[7] d2h_bytes
 *   out[8] device_ms        waiting for the kernels alone (not overlapped with a copy)
 *   out[9] pipelined_calls  calls that ran the three-thread pipeline (upload | kernels | download of column blocks)
 *   out[10] wall_ms         whole calls, entry to return
 *   out[11] alloc_ms        hipMalloc inside the calls (the workspace pool is grow-only: zero in the steady state; with
 *                           residency on, the first calls after a change of shape allocate what the residents hold)
 * n = number of doubles the caller's buffer holds (<= 12 are written).  icnv_host_path_stats_reset() zeroes them.
 * The smoothing chain and the per-cell Viterbi on ONE device pipeline their transfers over column blocks when
 * residency is off (the reference cells' blocks first: the chain's statistics need them before any block can be
 * finished): uploads, kernels and downloads overlap on three streams driven by three host threads, because a copy from
 * or to pageable memory -- what R hands over -- occupies the thread that issues it.  ICNV_HOST_PIPELINE=0 switches it off. */
int icnv_host_path_stats(double *out, int32_t n);
void icnv_host_path_stats_reset(void);

/* ---- smoothing chain ---------------------------------------------------- */
typedef struct icnv_chain_cfg {
    int64_t G;               /* genes                                              */
    int64_t C;               /* cells in this (local) matrix                       */
    const int32_t *chr_start; /* HOST, n_chr+1 offsets                              */
    int32_t n_chr;
    int32_t window_length;   /* odd; < 2 = no smoothing (R/inferCNV_ops.R:2444)    */
    double max_thresh;       /* step 9 threshold; NaN = skip                       */
    int32_t use_bounds;      /* steps 8/12: 1 = min/max-of-group-means bounds      */
    int32_t inv_log;         /* subtract_ref_expr_from_obs(inv_log = TRUE): group means as log2(mean(2^x - 1) + 1)
                                (R/inferCNV_ops.R:1714-1717).  run() never sets it (:771, :952), so it is the
                                stand-alone step only: stage_mask must be ICNV_ST_SUBTRACT_REF_1 alone.  Sits in
                                the padding after use_bounds: zero-initialised configurations keep their meaning */
    double sd_amplifier;     /* step 22 (clear_noise_via_ref_mean_sd)              */
    double noise_filter;     /* step 22: NaN = sd-based; else clear_noise(threshold) */
    uint32_t stage_mask;     /* ICNV_ST_* bits                                     */
    int32_t noise_logistic;  /* step 22 with noise_logistic = TRUE (R/inferCNV_ops.R:2249-2252, 2326-2330;
                                .apply_logistic_val_adj, R/inferCNV_heatmap.R:2791-2810): instead of the select,
                                x <- m +- p |x - m|, p = 1 / (1 + exp(-20 (|x - m| - s))), m and s = the select's centre
                                and half width.  Sits in the padding after stage_mask: zero-initialised
                                configurations keep their meaning */
    const int32_t *ref_idx;  /* HOST packed LOCAL reference cell indices (or, with */
    const int32_t *ref_off;  /* no references, one group of all observation cells, */
    int32_t n_ref_grp;       /* R/inferCNV_ops.R:1686-1688); HOST n_ref_grp+1      */
} icnv_chain_cfg;

/* One-call form, host buffers.  Replaces the R functions listed at the
 * ICNV_ST_* bits; with stage_mask = a single bit it is the stand-alone step
 * (so run(up_to_step=), resume files and the .hspike mirror keep working).
 * pre_denoise (nullable) receives the matrix before step 22 (the HMM's input,
 * R/inferCNV_ops.R:1237-1309 reads the step-14..16 object). */
int icnv_smooth_chain(const double *expr_in, double *expr_out, double *pre_denoise,
                      const icnv_chain_cfg *cfg);
/* Same with device-resident matrices; expr_out may alias expr_in. */
int icnv_smooth_chain_dev(const double *expr_in, double *expr_out, double *pre_denoise,
                          const icnv_chain_cfg *cfg, void *stream);

/* Split-phase form for cell-sharded multi-GPU runs (one process per GPU).
 * The chain has one "reference round" per reference-dependent stage present
 * in stage_mask (steps 8, 12, 22, in that order).  For round r the caller
 *   1. icnv_chain_round_partial_dev(): enqueues this rank's partial statistic
 *      over its LOCAL reference cells into a device buffer of *n doubles --
 *      subtract rounds: [G*n_ref_grp gene sums | n_ref_grp cell counts],
 *      denoise round:   [sum x, sum_c sd_c, n_ref_cells, n_ref_values];
 *   2. all-reduces (sum) that buffer across ranks (RCCL; nothing to do on 1 GPU);
 *   3. icnv_chain_round_finish_dev(): turns the reduced buffer into the
 *      stage's parameters (bounds / mu,s) on the device.
 * Then icnv_chain_apply_dev() streams every local cell through the fused pass.
 * The rounds and the apply must be given the SAME matrix (expr_in, unchanged in between): the round that
 * first smooths the reference cells keeps its output (one column per reference cell), and the later rounds and
 * the apply continue from it instead of smoothing those cells again.  The kept columns are tied to the expr_in
 * pointer and consumed by the apply; an apply without fresh rounds recomputes every cell from expr_in with the
 * parameters of the last rounds. */
typedef struct icnv_chain icnv_chain_t;
int icnv_chain_begin(icnv_chain_t **chain, const icnv_chain_cfg *cfg);
int icnv_chain_num_rounds(const icnv_chain_t *chain);
int icnv_chain_round_partial_dev(icnv_chain_t *chain, int round, const double *expr_in,
                                 double **partial_dev, int64_t *n, void *stream);
int icnv_chain_round_finish_dev(icnv_chain_t *chain, int round, void *stream);
int icnv_chain_apply_dev(icnv_chain_t *chain, const double *expr_in, double *expr_out,
                         double *pre_denoise, void *stream);
/* The same with a LEADING DIMENSION for the HMM input: column c of pre_denoise starts at pre_denoise + c * ld_pre (ld_pre >= G;
 * 0 or G: contiguous columns, the R layout).  A multiple of 16 puts every column on a cache line of its own, which is what
 * icnv_viterbi_cells_ld_dev reads fastest when G is not a multiple of 16 (every real, filtered gene set).  Device-resident
 * pipelines only -- an R matrix is contiguous; the host-buffer entry points pad on upload by themselves.  Fused chain with
 * step 22 only (ICNV_ERR_UNSUPPORTED otherwise).  Replaces nothing in the reference: a layout option of this library. */
int icnv_chain_apply_ld_dev(icnv_chain_t *chain, const double *expr_in, double *expr_out,
                            double *pre_denoise, int64_t ld_pre, void *stream);
/* Copies {mu, s} of the denoise stage to the host (synchronises the stream). */
int icnv_chain_get_denoise(icnv_chain_t *chain, double *mu_s, void *stream);
void icnv_chain_end(icnv_chain_t *chain);

/* get_average_bounds (R/inferCNV_ops.R:2723-2742): out2 = {mean_c min_g x,
 * mean_c max_g x}; threshold "auto" of step 9 is mean(abs(out2)). */
int icnv_average_bounds(const double *expr, int64_t G, int64_t C, double *out2);
int icnv_average_bounds_dev(const double *expr, int64_t G, int64_t C, double *out2_host, void *stream);

/* scale_infercnv_expr (step 5 of run(), scale_data, off by default; R/inferCNV_ops.R:3174-3185): t(scale(t(x))) -- every
 * gene minus its mean over the cells, divided by sqrt(sum(centred^2) / max(1, C - 1)) (a constant gene becomes NaN, as in R).
 * expr_out may alias expr_in in the _dev form. */
int icnv_scale_genes(const double *expr_in, double *expr_out, int64_t G, int64_t C);
int icnv_scale_genes_dev(const double *expr_in, double *expr_out, int64_t G, int64_t C, void *stream);

/* remove_outliers_norm (step 16 of run(), R/inferCNV_ops.R:1969-2054; between the chain and the HMM when prune_outliers
 * is set): values below / above the bounds are set to the bounds.  Both bounds given (not NaN) = hard thresholds
 * (:2017-2022); otherwise out_method = "average_bound", the bounds of icnv_average_bounds over the input (:2029-2033).
 * bounds_used2 (nullable, host) receives {lower, upper}.  expr_out may alias expr_in in the _dev form. */
int icnv_remove_outliers(const double *expr_in, double *expr_out, int64_t G, int64_t C, double lower_bound, double upper_bound,
                         double *bounds_used2);
int icnv_remove_outliers_dev(const double *expr_in, double *expr_out, int64_t G, int64_t C, double lower_bound,
                             double upper_bound, double *bounds_used2, void *stream);

/* ---- ingest from the raw COUNT matrix: steps 2, 3, 4 of run() in one call ------------------
 * Replaces require_above_min_mean_expr_cutoff + require_above_min_cells_ref (R/inferCNV_ops.R:2128-2213; run() :560-566),
 * normalize_counts_by_seq_depth (:3064-3111) and log2xplus1 (:2756-2769).  The counts cross PCIe once as integers --
 * dense int32 (G x C column-major) or CSC (colptr [C + 1], rowidx / vals [nnz]; 0-based, any order inside a column) --
 * and the f64 matrix of the KEPT genes (G_out x C) is formed on the device.  Integer sums are exact: the result is bit
 * for bit what icnv_gene_stats + icnv_select_genes + icnv_normalize_log2 give on the f64 copy of the counts.
 *   min_mean_expr_cutoff  NaN: no filter; a gene with rowMeans(counts) < cutoff is removed
 *   min_cells_per_gene    <= 0: no filter; a gene needs counts > 0 in at least that many cells
 *   normalize_factor      NaN: median(colSums) over the kept genes (the reference's default, normalize_factor = NA)
 *   keep_idx [G], G_out   the kept genes (ascending, 0-based) and their number; "All genes removed" is an error (:2194)
 *   expr_out              capacity G x C doubles, filled G_out x C; h2d_bytes (nullable): bytes uploaded */
typedef struct icnv_counts {
    const int32_t *dense;    /* dense form, or NULL */
    const int64_t *colptr;   /* CSC form, or NULL */
    const int32_t *rowidx;
    const int32_t *vals;
    int64_t nnz;             /* CSC: stored entries */
} icnv_counts;
int icnv_ingest_counts(const icnv_counts *cnt, int64_t G, int64_t C, double min_mean_expr_cutoff, int32_t min_cells_per_gene,
                       double normalize_factor, int32_t *keep_idx, int64_t *G_out, double *expr_out, double *factor_used,
                       int64_t *h2d_bytes);
/* The same with the count arrays and the output already on the device (keep_idx stays a host array). */
int icnv_ingest_counts_dev(const icnv_counts *cnt, int64_t G, int64_t C, double min_mean_expr_cutoff, int32_t min_cells_per_gene,
                           double normalize_factor, int32_t *keep_idx_host, int64_t *G_out, double *expr_out_dev,
                           double *factor_used, void *stream);
/* Split phases for a cell-sharded caller (infercnv_amd/sharded.py: ShardedIngest):
 *   gene_stats   stats2G_dev = [G sums of the counts | G numbers of cells with count > 0] as doubles -> all-reduce(sum);
 *                a negative entry (R's NA_integer_ is INT_MIN) is ICNV_ERR_ARG: counts must be >= 0
 *   select       the filter decision from the all-reduced statistics (host arithmetic, the same on every rank)
 *   col_sums     colSums over the kept genes (keep_mask_dev: G bytes, 1 = kept)     -> all-gather, median = the factor
 *   apply        expr_out[j, c] = log2(count[keep[j], c] / col_sum[c] * factor + 1), the reference's operation order */
int icnv_ingest_gene_stats_dev(const icnv_counts *cnt, int64_t G, int64_t C, double *stats2G_dev, void *stream);
int icnv_ingest_select(const double *stats2G_host, int64_t G, int64_t C_total, double min_mean_expr_cutoff, int32_t min_cells_per_gene,
                       int32_t *keep_idx, int64_t *G_out);
int icnv_ingest_col_sums_dev(const icnv_counts *cnt, int64_t G, int64_t C, const uint8_t *keep_mask_dev, double *col_sums_dev, void *stream);
int icnv_ingest_apply_dev(const icnv_counts *cnt, int64_t G, int64_t C, const int32_t *keep_idx_dev, int64_t G_out,
                          const double *col_sums_dev, double factor, int32_t do_normalize, int32_t do_log2, double *expr_out,
                          void *stream);

/* ---- ingest: steps 3 and 4 of run() (first "next" row) -------- */
/* colSums(expr.data) per cell (R/inferCNV_ops.R:3089), device pointers. */
int icnv_col_sums_dev(const double *expr, int64_t G, int64_t C, double *sums_dev, void *stream);
/* normalize_counts_by_seq_depth (R/inferCNV_ops.R:3064-3111): x / colSum * factor, and
 * log2xplus1 (:2756-2769): log2(x + 1); either part can be switched off. */
int icnv_normalize_log2_dev(const double *expr_in, double *expr_out, int64_t G, int64_t C,
                            const double *col_sums_dev, double normalize_factor, int32_t do_normalize,
                            int32_t do_log2, void *stream);
/* Host buffers; normalize_factor NaN = median(colSums) like the reference's default
 * (normalize_factor=NA); the factor used is returned through factor_used (nullable). */
int icnv_normalize_log2(const double *expr_in, double *expr_out, int64_t G, int64_t C,
                        double normalize_factor, int32_t do_normalize, int32_t do_log2, double *factor_used);

/* ---- cell-cell distances ------------------------------------ */
/* parallelDist(t(expr.data[, cells])), method "euclidean", as the reference calls it before hclust
 * (R/inferCNV_tumor_subclusters.R:191; R/inferCNV_ops.R:1930, 3242; R/inferCNV_heatmap.R:719-1079):
 * dist_out [n x n] (symmetric, zero diagonal; the R `dist` object is its lower triangle in column order).
 * cell_idx [n] HOST, 0-based.  fp64 Gram matrix of the group-centred cells on the matrix cores
 * (v_mfma_f64_16x16x4_f64), D_ij = sqrt(max(0, S_ii + S_jj - 2 S_ij)). */
int icnv_cell_distances(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx, int64_t n, double *dist_out);
int icnv_cell_distances_dev(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx, int64_t n, double *dist_out,
                            void *stream);

/* ---- HMM ---------------------------------------------------------------- */
/* Viterbi.dthmm.adj (R/inferCNV_HMM.R:1101-1176) for every (cell, chromosome):
 * predict_CNV_via_HMM_on_indiv_cells (R/inferCNV_HMM.R:284-324) with K = 6 and
 * i3HMM_predict_CNV_via_HMM_on_indiv_cells (R/inferCNV_i3HMM.R:180-225) with
 * K = 3.  Host-prepared parameters, exactly as the reference prepares them in
 * R: mean[K]; sd_shared = median(pm$sd) (:1122); logPi = log(Pi) K x K
 * column-major (logPi[j + K*k] = log Pi[j,k]); logDelta = log(delta).
 * states[g + G*c] in 1..K (uint8); chromosomes with < 2 genes get 3 (:1104).
 * n_underflow (nullable, HOST) receives the number of sequences for which the
 * reference would stop("Problems With Underflow"); the host flavour returns
 * ICNV_ERR_UNDERFLOW when it is non-zero. */
int icnv_viterbi_cells(const double *expr, uint8_t *states, int64_t G, int64_t C,
                       const int32_t *chr_start, int32_t n_chr, int32_t K, const double *mean,
                       double sd_shared, const double *logPi, const double *logDelta);
int icnv_viterbi_cells_dev(const double *expr, uint8_t *states, int64_t G, int64_t C,
                           const int32_t *chr_start, int32_t n_chr, int32_t K, const double *mean,
                           double sd_shared, const double *logPi, const double *logDelta,
                           int32_t *n_underflow_dev, void *stream);
/* The same on matrices with LEADING DIMENSIONS: expr[g + ld_expr * c], states[g + ld_states * c] (both >= G).  The per-cell
 * kernels read one cache line per cell and request and write the states in 16-byte words: with columns that start on line /
 * word boundaries (ld a multiple of 16) a gene count that is not a multiple of 16 runs as fast as one that is (+27 % otherwise,
 * profiles/r05_sweep.json).  icnv_viterbi_cells (host buffers -- what R hands over) uploads into such a layout by itself. */
int icnv_viterbi_cells_ld_dev(const double *expr, int64_t ld_expr, uint8_t *states, int64_t ld_states, int64_t G, int64_t C,
                              const int32_t *chr_start, int32_t n_chr, int32_t K, const double *mean,
                              double sd_shared, const double *logPi, const double *logDelta,
                              int32_t *n_underflow_dev, void *stream);

/* Certified fast path of the per-cell Viterbi (DESIGN.md "Certified fast Viterbi").  With a shared sd and the
 * transition structure of .get_HMM / .i3HMM_get_HMM (R/inferCNV_HMM.R:230-265, R/inferCNV_i3HMM.R:99-156: one
 * off-diagonal and one diagonal probability) icnv_viterbi_cells[_dev] computes the emission scores from a
 * verified polynomial table, tests every arg-max decision against a certified error band and recomputes the
 * flagged sequences with the exact kernel: the states are those of the exact kernel, bit for bit.
 * A column batch with more than 2 % of its sequences flagged is recomputed as a whole by the exact kernel instead
 * (decided on the device from that batch's own flag count: no state carries over from one call to the next).
 * Two kernels serve the fast path: the staged one (observations requested by whole cache lines through LDS-DMA, a short
 * table: tails of >= 4 sd beyond the outer state means) is tried first; a column batch whose data leave its table is redone
 * by the register kernel with the full table (tails up to 19 sd), and only a batch that still has > 2 % flagged goes to the
 * exact kernel -- all decided on the device.
 *   icnv_viterbi_set_mode   0 = auto (default), 1 = exact kernel only, 2 = auto without the staged kernel (developer A/B);
 *                           process-wide
 *   icnv_viterbi_last_stats out4 = {path of the calling thread's device's last call (0 exact / 1 fast, register kernel /
 *                           2 fast, last column batch recomputed by the exact kernel / 3 fast, staged kernel / 4 staged
 *                           kernel, last column batch redone with the full table), sequences, flagged sequences of
 *                           the last column batch, table intervals}; synchronises with that call
 *   icnv_hmm_emission_table host-only: the table for (K, mean, sd); meta8 = {n_records, x_lo, x_hi, eps_tab,
 *                           s_max, degree, 1, eps_spec}; seg_out [4] = the uniform grid {origin, 1/width, 0, grid
 *                           intervals - 1} (n_records = grid intervals + K: an interval with a state mean has two);
 *                           coef_out [n_records*K*(degree+1)] (nullable; polynomials of s_k - s_1, row k = 0 zero);
 *                           eps_tab bounds |table - (s_k - s_1)|, s_max bounds |s_k| and |s_k - s_1|
 *   icnv_hmm_emission_scores host-only: which = 0 the exact scores of R/inferCNV_HMM.R:1129-1133 in 80-bit
 *                           arithmetic, which = 1 the table's values through the kernel's double operations: the
 *                           scores RELATIVE TO STATE 1, s_k - s_1 (column 0 is 0; a term common to all states
 *                           changes no decision of the recurrence, so the table does not carry it)
 *                           (ok_out[i] = 0 and NaN where x[i] is outside the table's domain); out [n*K] */
int icnv_viterbi_set_mode(int mode);
int icnv_viterbi_last_stats(int64_t *out4);
int icnv_hmm_emission_table(int32_t K, const double *mean, double sd, double *meta8, double *seg_out, double *coef_out,
                            int64_t coef_cap);
int icnv_hmm_emission_scores(int32_t K, const double *mean, double sd, const double *x, int64_t n, int32_t which,
                             double *out, uint8_t *ok_out);

/* predict_CNV_via_HMM_on_tumor_subclusters / _whole_tumor_samples
 * (R/inferCNV_HMM.R:345-408, 509-567) and the i3 variants
 * (R/inferCNV_i3HMM.R:249-389): Viterbi on rowMeans over each group's cells
 * with that group's shared sd, trace broadcast to all member cells.  Cells in
 * no group get 0xFF (the reference leaves -1).  For the per-chromosome
 * grouping of ..._tumor_subclusters_per_chr (:412-487) call once per
 * chromosome with n_chr = 1 windows. */
int icnv_viterbi_groups(const double *expr, uint8_t *states, int64_t G, int64_t C,
                        const int32_t *chr_start, int32_t n_chr, const int32_t *grp_idx,
                        const int32_t *grp_off, int32_t n_grp, int32_t K, const double *mean,
                        const double *sd_shared_per_grp, const double *logPi,
                        const double *logDelta);
int icnv_viterbi_groups_dev(const double *expr, uint8_t *states, int64_t G, int64_t C,
                            const int32_t *chr_start, int32_t n_chr, const int32_t *grp_idx,
                            const int32_t *grp_off, int32_t n_grp, int32_t K, const double *mean,
                            const double *sd_shared_per_grp, const double *logPi,
                            const double *logDelta, int32_t *n_underflow_dev, void *stream);

/* rowMeans(expr.data[, group_cells]) per group (R/inferCNV_HMM.R:383):
 * out[g + G*q], device pointers. */
int icnv_group_means_dev(const double *expr, int64_t G, int64_t C, const int32_t *grp_idx,
                         const int32_t *grp_off, int32_t n_grp, double *out, void *stream);

/* Gene filters of the ingest (first "next" row).  icnv_gene_stats: per gene the sum over all cells
 * and the number of cells with expr > 0 -- what require_above_min_mean_expr_cutoff (rowMeans(expr) < cutoff,
 * R/inferCNV_ops.R:2128-2163) and require_above_min_cells_ref (sum(x > 0 & !is.na(x)) >= min_cells, :2182-2213)
 * decide on; a cell-sharded caller all-reduces both vectors.  icnv_select_genes: remove_genes on the matrix,
 * expr_out[j + G_out*c] = expr_in[keep_idx[j] + G_in*c] (0-based, any order). */
int icnv_gene_stats(const double *expr, int64_t G, int64_t C, double *gene_sums, int32_t *gene_nnz);
int icnv_gene_stats_dev(const double *expr, int64_t G, int64_t C, double *gene_sums, int32_t *gene_nnz, void *stream);
int icnv_select_genes(const double *expr_in, int64_t G_in, int64_t C, const int32_t *keep_idx, int64_t G_out,
                      double *expr_out);
int icnv_select_genes_dev(const double *expr_in, int64_t G_in, int64_t C, const int32_t *keep_idx, int64_t G_out,
                          double *expr_out, void *stream);

/* mean() and sd() over the block expr[gene_idx, cell_idx] (gene_idx NULL = all genes): the per-CNV-level emission
 * statistics of get_spike_dists (R/inferCNV_HMM.R:15-99; SURVEY.md 8f, third "next" row).  out2 = {mean, sd} on the
 * host; index lists are host arrays, 0-based. */
int icnv_block_mean_sd(const double *expr, int64_t G, int64_t C, const int32_t *gene_idx, int64_t n_genes,
                       const int32_t *cell_idx, int64_t n_cells, double *out2);
int icnv_block_mean_sd_dev(const double *expr, int64_t G, int64_t C, const int32_t *gene_idx, int64_t n_genes,
                           const int32_t *cell_idx, int64_t n_cells, double *out2_host, void *stream);

/* .get_state_consensus (R/inferCNV_HMM.R:977-987): per gene the most frequent state among each group's
 * cells (ties -> smallest state, -1/0xFF first, as table()+order() do).  consensus (nullable): uint8
 * [g + G*q].  states_out (nullable, may alias states): every member cell receives its group's consensus
 * -- the overwrite that predict_CNV_via_HMM_on_tumor_subclusters_per_chr (R/inferCNV_HMM.R:473-483) and
 * get_predicted_CNV_regions (:706-764) are built on (second "next" row). */
int icnv_state_consensus(const uint8_t *states, int64_t G, int64_t C, const int32_t *grp_idx,
                         const int32_t *grp_off, int32_t n_grp, uint8_t *consensus, uint8_t *states_out);
int icnv_state_consensus_dev(const uint8_t *states, int64_t G, int64_t C, const int32_t *grp_idx,
                             const int32_t *grp_off, int32_t n_grp, uint8_t *consensus,
                             uint8_t *states_out, void *stream);

/* assign_HMM_states_to_proxy_expr_vals (R/inferCNV_HMM.R:1191-1206; K = 6:
 * {0,0.5,1,1.5,2,3}) and i3HMM_assign_... (R/inferCNV_i3HMM.R:405-417; K = 3:
 * {0.5,1,1.5}).  n = G*C elements. */
int icnv_states_to_proxy(const uint8_t *states, double *out, int64_t n, int32_t K);
int icnv_states_to_proxy_dev(const uint8_t *states, double *out, int64_t n, int32_t K, void *stream);

/* Mean and sd over ALL values of the listed cells (i3 parameters,
 * R/inferCNV_i3HMM.R:17-80; also clear_noise's centre).  out2_host = {mu, sigma}. */
int icnv_cells_mean_sd_dev(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx,
                           int64_t n_cells, double *out2_host, void *stream);
int icnv_cells_mean_sd(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx, int64_t n_cells, double *out2);
/* The same statistic split in two for a cell-sharded caller ("i3: [sum x, sum x^2, n] over reference
 * values"; in the two-pass form R's sd() uses):
 *   phase 0: out3_host = {sum of the values, number of values, 0}                 -> all-reduce(sum) -> mean = sum / n
 *   phase 1: out3_host = {sum of (x - mean)^2 over the values, number, 0}         -> all-reduce(sum) -> sd = sqrt(ss / (n - 1))
 * A rank that holds none of the cells passes n_cells = 0 (expr may then be NULL) and contributes zeros.
 * Replaces the mean(ref values) / sd(ref values) of .i3HMM_get_sd_trend_by_num_cells_fit, R/inferCNV_i3HMM.R:38-52. */
int icnv_cells_moments_partial_dev(const double *expr, int64_t G, int64_t C, const int32_t *cell_idx, int64_t n_cells,
                                   int32_t phase, double mean, double *out3_host, void *stream);

/* The i3 HMM at group level as a PLAN with device-resident parameters (round 6).  i3HMM_predict_CNV_via_HMM_on_tumor_subclusters /
 * _whole_tumor_samples (R/inferCNV_i3HMM.R:249-389) compute mean(ref) and sd(ref) of the reference cells' values
 * (.i3HMM_get_sd_trend_by_num_cells_fit, :38-52), place the three state means at mu and mu +- delta (:435-445) and run the
 * Viterbi on every group's mean profile.  The plan uploads the group structure ONCE; a step is
 *   icnv_group_hmm_i3_partial_dev(): group means + the reference cells' shifted moments {sum (x - 1), sum (x - 1)^2, n} in ONE pass
 *       over the matrix; *moments_dev points at the three doubles on the device -- all-reduce(sum) them in a cell-sharded run;
 *   icnv_group_hmm_i3_finish_dev(): mu, sigma, delta from the moments ON THE DEVICE (delta = delta_abs when it is a number -- the
 *       KS-based value of use_KS = TRUE, computed by the caller from sigma --, sigma * z_abs otherwise: |qnorm(p, 0, sigma)| =
 *       sigma |qnorm(p)|), Viterbi per group with its parameters read from device memory, broadcast to the cells.
 * No upload, download or stream synchronisation inside a step.  Every cell in at most one group, every reference cell in exactly
 * one, at most 8192 (group, chromosome) sequences (ICNV_ERR_UNSUPPORTED otherwise: icnv_viterbi_groups_dev serves those).  The
 * moments are shifted around 1, the level the chain's output is centred at: mu and sigma agree with the two-pass long-double values
 * of icnv_cells_moments_partial_dev to ~1e-15 relative. */
typedef struct icnv_group_hmm icnv_group_hmm_t;
int icnv_group_hmm_begin(icnv_group_hmm_t **plan, int64_t G, int64_t C, const int32_t *chr_start, int32_t n_chr,
                         const int32_t *grp_idx, const int32_t *grp_off, int32_t n_grp, const int32_t *ref_idx, int64_t n_ref);
int icnv_group_hmm_i3_partial_dev(icnv_group_hmm_t *plan, const double *expr, double **moments_dev, void *stream);
int icnv_group_hmm_i3_finish_dev(icnv_group_hmm_t *plan, uint8_t *states, const double *logPi /* 3 x 3, column-major */,
                                 const double *logDelta /* [3] */, double z_abs, double delta_abs,
                                 int32_t *n_underflow_dev, void *stream);
/* {mu, sigma, delta} of the last finish (synchronises the stream). */
int icnv_group_hmm_get_i3_params(icnv_group_hmm_t *plan, double *mu_sigma_delta, void *stream);
void icnv_group_hmm_end(icnv_group_hmm_t *plan);

/* Values of the matrix at element offsets g + G c (host list in, host values out): the draws of
 * sample(expr_vals, size = ncells, replace = TRUE) in get_hspike_cnv_mean_sd_trend_by_num_cells_fit
 * (R/inferCNV_HMM.R:164) taken from the resident hidden-spike matrix; the index stream itself is R's RNG, restated on the
 * host (infercnv_amd/r_rng.py). */
int icnv_gather_values_dev(const double *expr, int64_t n_elements, const int64_t *offsets_host, int64_t n, double *out_host, void *stream);
int icnv_gather_values(const double *expr, int64_t G, int64_t C, const int64_t *offsets, int64_t n, double *out);

/* ---- 2-D median denoise -------------------------------------------------- */
/* apply_median_filtering / .median_filter (R/noise_reduction.R:43-113): for
 * every (tile, chromosome) block -- tile = one tumour subcluster or one whole
 * reference group, cells in stored order -- out[p,q] = median over the clamped
 * (window_size+2)^2 neighbourhood.  Cells in no tile are copied through. */
int icnv_median_filter(const double *expr_in, double *expr_out, int64_t G, int64_t C,
                       const int32_t *chr_start, int32_t n_chr, const int32_t *tile_idx,
                       const int32_t *tile_off, int32_t n_tiles, int32_t window_size);
int icnv_median_filter_dev(const double *expr_in, double *expr_out, int64_t G, int64_t C,
                           const int32_t *chr_start, int32_t n_chr, const int32_t *tile_idx,
                           const int32_t *tile_off, int32_t n_tiles, int32_t window_size,
                           void *stream);

/* ---- profiling hooks (used by bench.py) --------------------------------- */
/* When enabled, every kernel launch is bracketed by hipEvents recorded on the
 * launch stream; icnv_timing_get() synchronises those events and returns the
 * accumulated milliseconds and launch count of one kernel family:
 * "chain_apply", "chain_gene_sums", "chain_cell_stats", "viterbi",
 * "group_means", "broadcast_states", "median_filter", ...
 * on = 1: every kernel family; on = 2: only "chain_apply" and "viterbi" (an event pair costs a few microseconds of
 * stream time: fourteen pairs per bench step are 2.5 % of it, two pairs are not); on = 0: off. */
void icnv_timing_enable(int on);
void icnv_timing_reset(void);
int icnv_timing_get(const char *kernel, double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* ICNV_H */
