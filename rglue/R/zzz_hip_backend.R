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

## expr.data in per-chromosome contiguous gene order.  After .order_reduce (R/inferCNV.R:407) the permutation is the
## identity: the matrix is then handed over AS IS (no copy).  The library recognises the matrix it returned from the
## previous step by its content and skips the upload (icnv_residency, include/icnv.h).
.icnv_matrix <- function(infercnv_obj, lay) {
    x <- infercnv_obj@expr.data
    if (!is.matrix(x)) x <- as.matrix(x)                 # dgCMatrix -> dense, like R/inferCNV_ops.R:1924-1926
    if (is.unsorted(lay$perm)) x <- x[lay$perm, , drop = FALSE]
    if (storage.mode(x) != "double") storage.mode(x) <- "double"
    x
}
## 0x100 = ICNV_ST_NA_AWARE: the matrix holds NAs -- the library recomputes the cells that do with the reference's NA semantics
## (.smooth_helper strips and re-inserts them, median(na.rm = TRUE), which() never selects one: csrc/chain_na.hip).  The
## steps that are not the chain (HMM, median filter, step 5 / 16) have no NA semantics in the reference either: they stop.
.icnv_na_flag <- function(x) if (anyNA(x)) 256L else 0L
.icnv_no_na <- function(x, what) if (anyNA(x)) stop(sprintf("%s: the matrix holds NA / NaN values", what)) else x
.icnv_unpermute <- function(m, lay) if (is.null(m) || !is.unsorted(lay$perm)) m else m[order(lay$perm), , drop = FALSE]

.icnv_chain <- function(infercnv_obj, mask, window_length = 101L, max_thresh = NA_real_, use_bounds = TRUE,
                        sd_amplifier = 1.5, noise_filter = NA_real_, want_pre = FALSE, inv_log = FALSE,
                        noise_logistic = FALSE) {
    lay <- .icnv_chr_layout(infercnv_obj)
    ref <- .icnv_pack(.icnv_ref_groups(infercnv_obj))
    x <- .icnv_matrix(infercnv_obj, lay)
    res <- .Call("icnv_R_smooth_chain", x, lay$chr_start, ref$idx, ref$off, as.integer(window_length),
                 as.numeric(max_thresh), as.logical(use_bounds), as.numeric(sd_amplifier),
                 as.numeric(noise_filter), as.integer(sum(mask) + .icnv_na_flag(x)), as.logical(want_pre), as.logical(inv_log),
                 as.logical(noise_logistic))
    lapply(res, .icnv_unpermute, lay = lay)
}

hip_subtract_ref_expr_from_obs <- function(infercnv_obj, inv_log = FALSE, use_bounds = TRUE) {
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, .icnv_ST["sub1"], use_bounds = use_bounds, inv_log = inv_log)[[1]]
    if (!is.null(infercnv_obj@.hspike))
        infercnv_obj@.hspike <- hip_subtract_ref_expr_from_obs(infercnv_obj@.hspike, inv_log, use_bounds)
    infercnv_obj
}

## get_average_bounds (R/inferCNV_ops.R:2723-2742): c(mean over cells of the per-cell minimum, ... maximum); run()
## takes mean(abs(.)) of it for max_centered_threshold = "auto" (:802-806)
hip_get_average_bounds <- function(infercnv_obj) {
    x <- infercnv_obj@expr.data
    if (!is.matrix(x)) x <- as.matrix(x)
    if (storage.mode(x) != "double") storage.mode(x) <- "double"
    .Call("icnv_R_average_bounds", x)
}

## step 5 (scale_data): R/inferCNV_ops.R:3174-3185
hip_scale_infercnv_expr <- function(infercnv_obj) {
    x <- infercnv_obj@expr.data
    if (!is.matrix(x)) x <- as.matrix(x)
    if (storage.mode(x) != "double") storage.mode(x) <- "double"
    infercnv_obj@expr.data <- .Call("icnv_R_scale_genes", x)
    if (!is.null(infercnv_obj@.hspike)) infercnv_obj@.hspike <- hip_scale_infercnv_expr(infercnv_obj@.hspike)
    infercnv_obj
}

## step 16 (prune_outliers): R/inferCNV_ops.R:1969-2054; NA bounds = out_method "average_bound"
hip_remove_outliers_norm <- function(infercnv_obj, out_method = "average_bound", lower_bound = NA, upper_bound = NA) {
    if (is.na(lower_bound) || is.na(upper_bound)) {
        if (is.na(out_method)) stop(992)
        if (out_method != "average_bound") stop(991)
        lower_bound <- upper_bound <- NA_real_
    }
    x <- infercnv_obj@expr.data
    if (!is.matrix(x)) x <- as.matrix(x)
    if (storage.mode(x) != "double") storage.mode(x) <- "double"
    infercnv_obj@expr.data <- .Call("icnv_R_remove_outliers", x, as.numeric(lower_bound), as.numeric(upper_bound))
    if (!is.null(infercnv_obj@.hspike))
        infercnv_obj@.hspike <- hip_remove_outliers_norm(infercnv_obj@.hspike, out_method, lower_bound, upper_bound)
    infercnv_obj
}

hip_apply_max_threshold_bounds <- function(infercnv_obj, threshold) {
    if (is.character(threshold) && threshold == "auto")       # run() resolves "auto" itself; accepted here as well
        threshold <- mean(abs(hip_get_average_bounds(infercnv_obj)))
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
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, .icnv_ST["denoise"], sd_amplifier = sd_amplifier,
                                          noise_logistic = noise_logistic)[[1]]
    infercnv_obj
}

hip_clear_noise <- function(infercnv_obj, threshold, noise_logistic = FALSE) {
    if (threshold == 0) return(infercnv_obj)
    infercnv_obj@expr.data <- .icnv_chain(infercnv_obj, .icnv_ST["denoise"], noise_filter = threshold,
                                          noise_logistic = noise_logistic)[[1]]
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
    x <- .icnv_no_na(.icnv_matrix(infercnv_obj, lay), "HMM (the reference's Viterbi has no NA handling either)")
    pm <- HMM_info[["state_emission_params"]]
    logPi <- log(HMM_info[["state_transitions"]]); logDelta <- log(HMM_info[["delta"]])
    st <- if (is.null(groups)) {
        .Call("icnv_R_viterbi_cells", x, lay$chr_start, as.numeric(pm$mean), as.numeric(sd), logPi, logDelta)
    } else {
        g <- .icnv_pack(groups)
        .Call("icnv_R_viterbi_groups", x, lay$chr_start, g$idx, g$off, as.numeric(pm$mean), as.numeric(sd),
              logPi, logDelta)
    }
    infercnv_obj@expr.data <- .icnv_unpermute(st, lay)
    infercnv_obj
}

hip_predict_CNV_via_HMM_on_indiv_cells <- function(infercnv_obj,
        cnv_mean_sd = get_spike_dists(infercnv_obj@.hspike), t = 1e-6) {
    HMM_info <- .get_HMM(cnv_mean_sd, t)
    .icnv_hmm_states(infercnv_obj, HMM_info, median(HMM_info[["state_emission_params"]]$sd))   # :1122
}

.icnv_i6_group_sd <- function(groups, cnv_mean_sd, cnv_level_to_mean_sd_fit)       # .get_state_emission_params + median(sd), :586-614, :1122
    vapply(groups, function(g) median(.get_state_emission_params(length(g), cnv_mean_sd, cnv_level_to_mean_sd_fit)$sd),
           numeric(1))

## tumor_samples of predict_CNV_via_HMM_on_whole_tumor_samples (R/inferCNV_HMM.R:528-533), the reference's own two
## expressions.  Note what the second one does: c() of an integer VECTOR with a list gives a list with one element per
## vector entry, i.e. with cluster_by_groups = FALSE every observation cell is a "sample" of its own (num_cells = 1)
## next to the reference groups.  Reference behaviour, kept.
.icnv_whole_sample_groups <- function(infercnv_obj, cluster_by_groups) {
    if (cluster_by_groups == TRUE) c(infercnv_obj@observation_grouped_cell_indices, infercnv_obj@reference_grouped_cell_indices)
    else c(all_observations = unlist(infercnv_obj@observation_grouped_cell_indices), infercnv_obj@reference_grouped_cell_indices)
}

hip_predict_CNV_via_HMM_on_whole_tumor_samples <- function(infercnv_obj, cluster_by_groups,
        cnv_mean_sd = get_spike_dists(infercnv_obj@.hspike),
        cnv_level_to_mean_sd_fit = get_hspike_cnv_mean_sd_trend_by_num_cells_fit(infercnv_obj@.hspike), t = 1e-6) {
    HMM_info <- .get_HMM(cnv_mean_sd, t)
    groups <- .icnv_whole_sample_groups(infercnv_obj, cluster_by_groups)
    .icnv_hmm_states(infercnv_obj, HMM_info, .icnv_i6_group_sd(groups, cnv_mean_sd, cnv_level_to_mean_sd_fit), groups)
}

hip_predict_CNV_via_HMM_on_tumor_subclusters <- function(infercnv_obj,
        cnv_mean_sd = get_spike_dists(infercnv_obj@.hspike),
        cnv_level_to_mean_sd_fit = get_hspike_cnv_mean_sd_trend_by_num_cells_fit(infercnv_obj@.hspike), t = 1e-6) {
    if (is.null(infercnv_obj@tumor_subclusters)) {
        ## R/inferCNV_HMM.R:358-361 reroutes to the whole-sample predictor (passing cnv_mean_sd in the position of
        ## cluster_by_groups); the groups it means are the annotation groups, i.e. cluster_by_groups = TRUE
        flog.warn("No subclusters defined, so instead running on whole samples")
        return(hip_predict_CNV_via_HMM_on_whole_tumor_samples(infercnv_obj, TRUE, cnv_mean_sd, cnv_level_to_mean_sd_fit, t))
    }
    HMM_info <- .get_HMM(cnv_mean_sd, t)
    groups <- unlist(infercnv_obj@tumor_subclusters[["subclusters"]], recursive = FALSE)
    .icnv_hmm_states(infercnv_obj, HMM_info, .icnv_i6_group_sd(groups, cnv_mean_sd, cnv_level_to_mean_sd_fit), groups)
}

## R/inferCNV_HMM.R:412-487: per chromosome its own list of subclusters (one device call per chromosome on that
## chromosome's rows), then every global subcluster receives its per-gene consensus state (:473-483: the
## get_predicted_CNV_regions(by = "subcluster") overwrite, done on the device by icnv_state_consensus)
hip_predict_CNV_via_HMM_on_tumor_subclusters_per_chr <- function(infercnv_obj, subclusters_per_chr,
        cnv_mean_sd = get_spike_dists(infercnv_obj@.hspike),
        cnv_level_to_mean_sd_fit = get_hspike_cnv_mean_sd_trend_by_num_cells_fit(infercnv_obj@.hspike), t = 1e-6) {
    if (is.null(subclusters_per_chr)) {
        flog.warn("No subclusters defined, so instead running on whole samples")
        return(hip_predict_CNV_via_HMM_on_whole_tumor_samples(infercnv_obj, TRUE, cnv_mean_sd, cnv_level_to_mean_sd_fit, t))
    }
    HMM_info <- .get_HMM(cnv_mean_sd, t)
    pm <- HMM_info[["state_emission_params"]]
    logPi <- log(HMM_info[["state_transitions"]]); logDelta <- log(HMM_info[["delta"]])
    x <- infercnv_obj@expr.data
    if (!is.matrix(x)) x <- as.matrix(x)
    hmm.data <- x
    hmm.data[, ] <- -1
    ## The reference lapply()s over the FACTOR unique(gene_order$chr) and indexes `subclusters_per_chr[[chr]]` with its elements,
    ## i.e. by the level's integer CODE (R/inferCNV_HMM.R:430, 443-447); its producer fills the list over levels(chr) under the
    ## level names (R/inferCNV_tumor_subclusters.R:651-652), so code, name and level order agree there.  Here: by name when the
    ## list carries this chromosome's name, else by the level code (factor column) / the position in unique() order (character column).
    chr_col <- infercnv_obj@gene_order[[C_CHR]]
    chrs <- unique(chr_col)
    for (k in seq_along(chrs)) {
        chr <- chrs[k]
        rows <- which(chr_col == chr)
        key <- as.character(chr)
        pos <- if (is.factor(chr_col)) as.integer(chr) else k
        groups <- if (!is.null(names(subclusters_per_chr)) && key %in% names(subclusters_per_chr)) subclusters_per_chr[[key]]
                  else if (pos <= length(subclusters_per_chr)) subclusters_per_chr[[pos]]
                  else NULL
        if (length(rows) == 0 || length(groups) == 0) next
        g <- .icnv_pack(groups)
        xc <- x[rows, , drop = FALSE]
        storage.mode(xc) <- "double"
        st <- .Call("icnv_R_viterbi_groups", xc, c(0L, length(rows)), g$idx, g$off, as.numeric(pm$mean),
                    as.numeric(.icnv_i6_group_sd(groups, cnv_mean_sd, cnv_level_to_mean_sd_fit)), logPi, logDelta)
        covered <- st != -1
        hmm.data[rows, ][covered] <- st[covered]
    }
    sub <- .icnv_pack(unlist(infercnv_obj@tumor_subclusters[["subclusters"]], recursive = FALSE))
    infercnv_obj@expr.data <- .Call("icnv_R_state_consensus_overwrite", hmm.data, sub$idx, sub$off)
    infercnv_obj
}

## ---- i3 (R/inferCNV_i3HMM.R).  Parameters exactly as the reference prepares them in R -- including the KS-based delta,
## which draws from R's RNG (get_HoneyBADGER_setGexpDev, :469-493) -- the Viterbi on the device.
.icnv_i3_states <- function(infercnv_obj, sd_trend, t, i3_p_val, use_KS, groups = NULL) {
    HMM_info <- .i3HMM_get_HMM(sd_trend, t = t, i3_p_val = i3_p_val, use_KS = use_KS)
    sd <- median(HMM_info[["state_emission_params"]]$sd)
    .icnv_hmm_states(infercnv_obj, HMM_info, if (is.null(groups)) sd else rep(sd, length(groups)), groups)
}

hip_i3HMM_predict_CNV_via_HMM_on_indiv_cells <- function(infercnv_obj, i3_p_val = 0.05,
        sd_trend = .i3HMM_get_sd_trend_by_num_cells_fit(infercnv_obj, i3_p_val), t = 1e-6, use_KS = TRUE)
    .icnv_i3_states(infercnv_obj, sd_trend, t, i3_p_val, use_KS)

hip_i3HMM_predict_CNV_via_HMM_on_whole_tumor_samples <- function(infercnv_obj, cluster_by_groups, i3_p_val = 0.05,
        sd_trend = .i3HMM_get_sd_trend_by_num_cells_fit(infercnv_obj, i3_p_val), t = 1e-6, use_KS = TRUE)
    .icnv_i3_states(infercnv_obj, sd_trend, t, i3_p_val, use_KS, .icnv_whole_sample_groups(infercnv_obj, cluster_by_groups))

hip_i3HMM_predict_CNV_via_HMM_on_tumor_subclusters <- function(infercnv_obj, i3_p_val = 0.05,
        sd_trend = .i3HMM_get_sd_trend_by_num_cells_fit(infercnv_obj, i3_p_val), t = 1e-6, use_KS = TRUE) {
    if (is.null(infercnv_obj@tumor_subclusters)) {          # R/inferCNV_i3HMM.R:259-262 (same positional quirk as the i6 one)
        flog.warn("No subclusters defined, so instead running on whole samples")
        return(hip_i3HMM_predict_CNV_via_HMM_on_whole_tumor_samples(infercnv_obj, TRUE, i3_p_val, sd_trend, t, use_KS))
    }
    .icnv_i3_states(infercnv_obj, sd_trend, t, i3_p_val, use_KS,
                    unlist(infercnv_obj@tumor_subclusters[["subclusters"]], recursive = FALSE))
}

## state -> proxy expression value (R/inferCNV_HMM.R:1191-1206, R/inferCNV_i3HMM.R:405-417); entries that are not a
## state of the model (e.g. -1) stay as they are, like the reference's masked assignments
.icnv_proxy <- function(infercnv_obj, K) {
    x <- infercnv_obj@expr.data
    if (!is.matrix(x)) x <- as.matrix(x)
    if (storage.mode(x) != "double") storage.mode(x) <- "double"
    infercnv_obj@expr.data <- .Call("icnv_R_states_to_proxy", x, as.integer(K))
    infercnv_obj
}
hip_assign_HMM_states_to_proxy_expr_vals <- function(infercnv_obj) .icnv_proxy(infercnv_obj, 6L)
hip_i3HMM_assign_HMM_states_to_proxy_expr_vals <- function(infercnv_obj) .icnv_proxy(infercnv_obj, 3L)

hip_apply_median_filtering <- function(infercnv_obj, window_size = 7, on_observations = TRUE, on_references = TRUE) {
    tiles <- list()
    if (on_observations) for (tt in names(infercnv_obj@observation_grouped_cell_indices))
        tiles <- c(tiles, infercnv_obj@tumor_subclusters[["subclusters"]][[tt]])
    if (on_references) tiles <- c(tiles, infercnv_obj@reference_grouped_cell_indices)
    lay <- .icnv_chr_layout(infercnv_obj)
    tl <- .icnv_pack(tiles)
    x <- .icnv_no_na(.icnv_matrix(infercnv_obj, lay), "apply_median_filtering")
    out <- .Call("icnv_R_median_filter", x, lay$chr_start, tl$idx, tl$off, as.integer(window_size))
    infercnv_obj@expr.data <- .icnv_unpermute(out, lay)
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

## Steps 2, 3, 4 of run() in one call from the raw counts (R/inferCNV_ops.R:560-620): require_above_min_mean_expr_cutoff,
## require_above_min_cells_ref, normalize_counts_by_seq_depth, log2xplus1.  The counts cross PCIe once as integers -- a
## dgCMatrix as its CSC slots, a dense matrix as int32 -- and the log-scale matrix of the kept genes comes back (and, with
## residency on, stays on the device for step 8).  Same result as the four step functions; counts must be integers.
hip_ingest_counts <- function(infercnv_obj, min_mean_expr_cutoff, min_cells_per_gene = 3, normalize_factor = NA_real_) {
    x <- infercnv_obj@expr.data
    G <- nrow(x); C <- ncol(x)
    res <- if (is(x, "dgCMatrix")) {
        if (any(x@x != round(x@x))) stop("hip_ingest_counts wants integer counts")
        .Call("icnv_R_ingest_counts", NULL, as.integer(x@p), as.integer(x@i), as.integer(round(x@x)), G, C,
              as.numeric(min_mean_expr_cutoff), as.integer(min_cells_per_gene), as.numeric(normalize_factor))
    } else {
        m <- as.matrix(x)
        if (any(m != round(m))) stop("hip_ingest_counts wants integer counts")
        storage.mode(m) <- "integer"
        .Call("icnv_R_ingest_counts", m, NULL, NULL, NULL, G, C, as.numeric(min_mean_expr_cutoff),
              as.integer(min_cells_per_gene), as.numeric(normalize_factor))
    }
    keep <- res[[2]]
    if (length(keep) < G) infercnv_obj <- remove_genes(infercnv_obj, setdiff(seq_len(G), keep))   # gene_order, count.data follow
    expr <- res[[1]]
    dimnames(expr) <- list(rownames(x)[keep], colnames(x))
    infercnv_obj@expr.data <- expr
    infercnv_obj
}

## Swap the package's step functions for the hip ones.  devices: 0 = every visible MI355X (cells are split into one
## contiguous block per GPU inside the library, one host thread per GPU), n = the first n, -1 = the current one only.
## residency: keep the last results on the device(s) so that the next step skips the upload of the matrix it was
## handed back.  A matrix is recognised by CONTENT (length, strided sample, then a 64-bit hash of every value computed
## on the host at memory speed): editing expr.data between steps, or R reusing a freed address, cannot serve stale
## device data -- a changed matrix hashes differently and is uploaded.
.icnv_enable_hip_backend <- function(devices = getOption("infercnv.hip.devices", 0L),
                                     residency = getOption("infercnv.hip.residency", FALSE)) {
    .Call("icnv_R_init", as.integer(devices), as.logical(residency))
    ns <- asNamespace("infercnv")
    swap <- c(subtract_ref_expr_from_obs = "hip_subtract_ref_expr_from_obs",
              get_average_bounds = "hip_get_average_bounds",
              apply_max_threshold_bounds = "hip_apply_max_threshold_bounds",
              remove_outliers_norm = "hip_remove_outliers_norm",
              scale_infercnv_expr = "hip_scale_infercnv_expr",
              smooth_by_chromosome = "hip_smooth_by_chromosome",
              center_cell_expr_across_chromosome = "hip_center_cell_expr_across_chromosome",
              invert_log2 = "hip_invert_log2",
              clear_noise_via_ref_mean_sd = "hip_clear_noise_via_ref_mean_sd",
              clear_noise = "hip_clear_noise",
              predict_CNV_via_HMM_on_indiv_cells = "hip_predict_CNV_via_HMM_on_indiv_cells",
              predict_CNV_via_HMM_on_tumor_subclusters = "hip_predict_CNV_via_HMM_on_tumor_subclusters",
              predict_CNV_via_HMM_on_tumor_subclusters_per_chr = "hip_predict_CNV_via_HMM_on_tumor_subclusters_per_chr",
              predict_CNV_via_HMM_on_whole_tumor_samples = "hip_predict_CNV_via_HMM_on_whole_tumor_samples",
              i3HMM_predict_CNV_via_HMM_on_indiv_cells = "hip_i3HMM_predict_CNV_via_HMM_on_indiv_cells",
              i3HMM_predict_CNV_via_HMM_on_tumor_subclusters = "hip_i3HMM_predict_CNV_via_HMM_on_tumor_subclusters",
              i3HMM_predict_CNV_via_HMM_on_whole_tumor_samples = "hip_i3HMM_predict_CNV_via_HMM_on_whole_tumor_samples",
              assign_HMM_states_to_proxy_expr_vals = "hip_assign_HMM_states_to_proxy_expr_vals",
              i3HMM_assign_HMM_states_to_proxy_expr_vals = "hip_i3HMM_assign_HMM_states_to_proxy_expr_vals",
              apply_median_filtering = "hip_apply_median_filtering")
    for (nm in names(swap)) {
        unlockBinding(nm, ns)
        assign(nm, get(swap[[nm]], envir = ns), envir = ns)
        lockBinding(nm, ns)
    }
    invisible(TRUE)
}

## The package may already have an .onLoad: do not define a second one here.  Add ONE line to the existing hook (or
## create R/zzz.R with it):
##     .onLoad <- function(libname, pkgname) { ...existing body...; .icnv_onLoad_hook() }
.icnv_onLoad_hook <- function() {
    if (.icnv_use_hip()) .icnv_enable_hip_backend()
    invisible(NULL)
}
