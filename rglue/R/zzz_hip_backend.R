## zzz_hip_backend.R -- opt-in MI355X backend for infercnv's hot path.
##
## WRITTEN BLIND (R is not installed in the build image).  Drop this file into
## the package's R/ directory next to src/icnv_shim.c (INTEGRATION.md).  When
## options(infercnv.backend = "hip") is set, the step functions below replace
## the package's own definitions (same names, same signatures, same return
## value: an infercnv object), so run(), up_to_step, resume files, the .hspike
## mirror and scripts/inferCNV.R keep working unchanged.
##
## No arithmetic happens here: index preparation (1-based -> 0-based, packing
## of group lists) and parameter preparation exactly as the reference does it
## (get_spike_dists, .get_HMM, median(sd), log(Pi), log(delta)).

.icnv_ST <- c(sub1 = 1L, thresh = 2L, smooth = 4L, center = 8L, sub2 = 16L, exp2 = 32L, denoise = 64L, mean = 128L)

.icnv_use_hip <- function() identical(getOption("infercnv.backend", "R"), "hip")

.icnv_chr_layout <- function(infercnv_obj) {
    chr <- as.character(infercnv_obj@gene_order[[C_CHR]])
    ord <- match(chr, unique(chr))                       # order of first appearance
    perm <- order(ord)                                   # stable; identity after .order_reduce
    list(perm = perm, chr_start = as.integer(c(0L, cumsum(tabulate(ord)))))
}

.icnv_pack <- function(groups) {                         # list of 1-based index vectors -> 0-based packed
    list(idx = as.integer(unlist(groups, use.names = FALSE)) - 1L,
         off = as.integer(c(0L, cumsum(vapply(groups, length, integer(1))))))
}

.icnv_ref_groups <- function(infercnv_obj) {             # R/inferCNV_ops.R:1683-1688
    if (has_reference_cells(infercnv_obj)) infercnv_obj@reference_grouped_cell_indices
    else list(proxyNormal = unlist(infercnv_obj@observation_grouped_cell_indices))
}

.icnv_chain <- function(infercnv_obj, mask, window_length = 101L, max_thresh = NA_real_, use_bounds = TRUE,
                        sd_amplifier = 1.5, noise_filter = NA_real_, want_pre = FALSE, inv_log = FALSE) {
    lay <- .icnv_chr_layout(infercnv_obj)
    ref <- .icnv_pack(.icnv_ref_groups(infercnv_obj))
    x <- as.matrix(infercnv_obj@expr.data)[lay$perm, , drop = FALSE]
    storage.mode(x) <- "double"
    res <- .Call("icnv_R_smooth_chain", x, lay$chr_start, ref$idx, ref$off, as.integer(window_length),
                 as.numeric(max_thresh), as.logical(use_bounds), as.numeric(sd_amplifier),
                 as.numeric(noise_filter), as.integer(sum(mask)), as.logical(want_pre), as.logical(inv_log))
    inv <- order(lay$perm)
    lapply(res, function(m) if (is.null(m)) NULL else m[inv, , drop = FALSE])
}

hip_subtract_ref_expr_from_obs <- function(infercnv_obj, inv_log = FALSE, use_bounds = TRUE) {
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, .icnv_ST["sub1"], use_bounds = use_bounds, inv_log = inv_log)[[1]]
    if (!is.null(infercnv_obj@.hspike))
        infercnv_obj@.hspike <- hip_subtract_ref_expr_from_obs(infercnv_obj@.hspike, inv_log, use_bounds)
    infercnv_obj
}

hip_apply_max_threshold_bounds <- function(infercnv_obj, threshold) {
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, .icnv_ST["thresh"], max_thresh = threshold)[[1]]
    if (!is.null(infercnv_obj@.hspike))
        infercnv_obj@.hspike <- hip_apply_max_threshold_bounds(infercnv_obj@.hspike, threshold)
    infercnv_obj
}

hip_smooth_by_chromosome <- function(infercnv_obj, window_length, smooth_ends = TRUE) {
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, .icnv_ST["smooth"], window_length = window_length)[[1]]
    if (!is.null(infercnv_obj@.hspike))
        infercnv_obj@.hspike <- hip_smooth_by_chromosome(infercnv_obj@.hspike, window_length, smooth_ends)
    infercnv_obj
}

hip_center_cell_expr_across_chromosome <- function(infercnv_obj, method = "mean") {
    mask <- if (method == "median") .icnv_ST["center"] else .icnv_ST[c("center", "mean")]
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, mask)[[1]]
    if (!is.null(infercnv_obj@.hspike))
        infercnv_obj@.hspike <- hip_center_cell_expr_across_chromosome(infercnv_obj@.hspike, method)
    infercnv_obj
}

hip_invert_log2 <- function(infercnv_obj) {
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, .icnv_ST["exp2"])[[1]]
    if (!is.null(infercnv_obj@.hspike)) infercnv_obj@.hspike <- hip_invert_log2(infercnv_obj@.hspike)
    infercnv_obj
}

hip_clear_noise_via_ref_mean_sd <- function(infercnv_obj, sd_amplifier = 1.5, noise_logistic = FALSE) {
    if (noise_logistic) stop("noise_logistic=TRUE is not offered by the hip backend")
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, .icnv_ST["denoise"], sd_amplifier = sd_amplifier)[[1]]
    infercnv_obj
}

hip_clear_noise <- function(infercnv_obj, threshold, noise_logistic = FALSE) {
    if (noise_logistic) stop("noise_logistic=TRUE is not offered by the hip backend")
    if (threshold == 0) return(infercnv_obj)
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, .icnv_ST["denoise"], noise_filter = threshold)[[1]]
    infercnv_obj
}

## steps 8..14(+22) back to back in one fused device pass
hip_smooth_chain <- function(infercnv_obj, window_length = 101, max_centered_threshold = 3, sd_amplifier = 1.5,
                             denoise = TRUE) {
    mask <- .icnv_ST[c("sub1", "thresh", "smooth", "center", "sub2", "exp2", if (denoise) "denoise")]
    res <- .icnv_chain(infercnv_obj, mask, window_length, max_centered_threshold, TRUE, sd_amplifier,
                       want_pre = TRUE)
    hmm_input <- infercnv_obj
    hmm_input@expr.data <- if (denoise) res[[2]] else res[[1]]
    infercnv_obj@expr.data <- res[[1]]
    list(infercnv_obj = infercnv_obj, hmm_input = hmm_input)
}

.icnv_hmm_states <- function(infercnv_obj, HMM_info, sd, groups = NULL) {
    lay <- .icnv_chr_layout(infercnv_obj)
    x <- as.matrix(infercnv_obj@expr.data)[lay$perm, , drop = FALSE]
    storage.mode(x) <- "double"
    pm <- HMM_info[["state_emission_params"]]
    logPi <- log(HMM_info[["state_transitions"]]); logDelta <- log(HMM_info[["delta"]])
    st <- if (is.null(groups)) {
        .Call("icnv_R_viterbi_cells", x, lay$chr_start, as.numeric(pm$mean), as.numeric(sd), logPi, logDelta)
    } else {
        g <- .icnv_pack(groups)
        .Call("icnv_R_viterbi_groups", x, lay$chr_start, g$idx, g$off, as.numeric(pm$mean), as.numeric(sd),
              logPi, logDelta)
    }
    infercnv_obj@expr.data <- st[order(lay$perm), , drop = FALSE]
    infercnv_obj
}

hip_predict_CNV_via_HMM_on_indiv_cells <- function(infercnv_obj,
        cnv_mean_sd = get_spike_dists(infercnv_obj@.hspike), t = 1e-6) {
    HMM_info <- .get_HMM(cnv_mean_sd, t)
    .icnv_hmm_states(infercnv_obj, HMM_info, median(HMM_info[["state_emission_params"]]$sd))   # :1122
}

hip_predict_CNV_via_HMM_on_tumor_subclusters <- function(infercnv_obj,
        cnv_mean_sd = get_spike_dists(infercnv_obj@.hspike),
        cnv_level_to_mean_sd_fit = get_hspike_cnv_mean_sd_trend_by_num_cells_fit(infercnv_obj@.hspike), t = 1e-6) {
    HMM_info <- .get_HMM(cnv_mean_sd, t)
    groups <- unlist(infercnv_obj@tumor_subclusters[["subclusters"]], recursive = FALSE)
    sd <- vapply(groups, function(g) median(.get_state_emission_params(length(g), cnv_mean_sd,
                                                                        cnv_level_to_mean_sd_fit)$sd), numeric(1))
    .icnv_hmm_states(infercnv_obj, HMM_info, sd, groups)
}

hip_apply_median_filtering <- function(infercnv_obj, window_size = 7, on_observations = TRUE, on_references = TRUE) {
    tiles <- list()
    if (on_observations) for (tt in names(infercnv_obj@observation_grouped_cell_indices))
        tiles <- c(tiles, infercnv_obj@tumor_subclusters[["subclusters"]][[tt]])
    if (on_references) tiles <- c(tiles, infercnv_obj@reference_grouped_cell_indices)
    lay <- .icnv_chr_layout(infercnv_obj)
    tl <- .icnv_pack(tiles)
    x <- as.matrix(infercnv_obj@expr.data)[lay$perm, , drop = FALSE]
    storage.mode(x) <- "double"
    out <- .Call("icnv_R_median_filter", x, lay$chr_start, tl$idx, tl$off, as.integer(window_size))
    infercnv_obj@expr.data <- out[order(lay$perm), , drop = FALSE]
    infercnv_obj
}

## parallelDist(t(tumor_expr_data)) of .single_tumor_subclustering (R/inferCNV_tumor_subclusters.R:191) as a `dist`
## object: hclust(hip_cell_dist(tumor_expr_data), method = hclust_method)
hip_cell_dist <- function(tumor_expr_data) {
    x <- as.matrix(tumor_expr_data)
    storage.mode(x) <- "double"
    d <- .Call("icnv_R_cell_distances", x, seq_len(ncol(x)) - 1L)
    dimnames(d) <- list(colnames(x), colnames(x))
    stats::as.dist(d)
}

## Swap the package's step functions for the hip ones (called from .onLoad when the option is set).
.icnv_enable_hip_backend <- function(device = -1L) {
    .Call("icnv_R_init", as.integer(device))
    ns <- asNamespace("infercnv")
    swap <- c(subtract_ref_expr_from_obs = "hip_subtract_ref_expr_from_obs",
              apply_max_threshold_bounds = "hip_apply_max_threshold_bounds",
              smooth_by_chromosome = "hip_smooth_by_chromosome",
              center_cell_expr_across_chromosome = "hip_center_cell_expr_across_chromosome",
              invert_log2 = "hip_invert_log2",
              clear_noise_via_ref_mean_sd = "hip_clear_noise_via_ref_mean_sd",
              clear_noise = "hip_clear_noise",
              predict_CNV_via_HMM_on_indiv_cells = "hip_predict_CNV_via_HMM_on_indiv_cells",
              predict_CNV_via_HMM_on_tumor_subclusters = "hip_predict_CNV_via_HMM_on_tumor_subclusters",
              apply_median_filtering = "hip_apply_median_filtering")
    for (nm in names(swap)) {
        unlockBinding(nm, ns)
        assign(nm, get(swap[[nm]], envir = ns), envir = ns)
        lockBinding(nm, ns)
    }
    invisible(TRUE)
}

.onLoad <- function(libname, pkgname) {
    if (.icnv_use_hip()) .icnv_enable_hip_backend()
}
