/*
 * icnv_shim.c -- .Call shim between the infercnv R package and libicnv_hip.so.
 *
 * R is not installed in the build image, so this file has never been compiled
 * against the real Rinternals.h; it IS compiled (-Wall -Wextra -Werror) and
 * driven from C against a mock of the R API subset it uses (rglue/mock/,
 * tests/test_host.py::test_r_shim_compiles_and_registers,
 * tests/test_gpu_entrypoints.py::test_r_shim_driven_from_c): types, arities,
 * PROTECT balance, error path and results are checked, R's own semantics of
 * those calls are not.  It contains no logic: it unpacks SEXPs
 * into the plain pointers/sizes of include/icnv.h, allocates the result under
 * PROTECT, and converts error codes to Rf_error() only AFTER the library has
 * returned (the library never longjmps and owns/frees its device memory).
 *
 * Build (inside the infercnv source tree):
 *   PKG_CPPFLAGS = -I$(ICNV_HOME)/include
 *   PKG_LIBS     = -L$(ICNV_HOME)/infercnv_amd -licnv_hip -Wl,-rpath,$(ICNV_HOME)/infercnv_amd
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdint.h>
#include <string.h>

#include "icnv.h"

static void fail(int rc) { Rf_error("libicnv_hip error %d: %s", rc, icnv_last_error()); }

/* .Call("icnv_R_smooth_chain", expr, chr_start, ref_idx, ref_off, window, max_thresh, use_bounds,
 *       sd_amplifier, noise_filter, stage_mask, want_pre, inv_log, noise_logistic)  ->  list(expr, pre | NULL)
 * expr: REALSXP matrix genes x cells (column-major == cell-major); indices 0-based INTSXP. */
SEXP icnv_R_smooth_chain(SEXP expr, SEXP chr_start, SEXP ref_idx, SEXP ref_off, SEXP window, SEXP max_thresh,
                         SEXP use_bounds, SEXP sd_amplifier, SEXP noise_filter, SEXP stage_mask, SEXP want_pre,
                         SEXP inv_log, SEXP noise_logistic) {
    if (!Rf_isReal(expr) || !Rf_isMatrix(expr)) Rf_error("expr must be a numeric matrix");
    icnv_chain_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.G = Rf_nrows(expr);
    cfg.C = Rf_ncols(expr);
    cfg.chr_start = (const int32_t *)INTEGER(chr_start);
    cfg.n_chr = (int32_t)(XLENGTH(chr_start) - 1);
    cfg.window_length = Rf_asInteger(window);
    cfg.max_thresh = Rf_asReal(max_thresh);       /* NA_real_ is a NaN: step 9 skipped */
    cfg.use_bounds = Rf_asLogical(use_bounds);
    cfg.inv_log = Rf_asLogical(inv_log) == TRUE;  /* stand-alone subtract_ref_expr_from_obs(inv_log = TRUE) only */
    cfg.sd_amplifier = Rf_asReal(sd_amplifier);
    cfg.noise_filter = Rf_asReal(noise_filter);
    cfg.stage_mask = (uint32_t)Rf_asInteger(stage_mask);
    cfg.noise_logistic = Rf_asLogical(noise_logistic) == TRUE;   /* step 22 as depress_log_signal_midpt_val */
    cfg.ref_idx = (const int32_t *)INTEGER(ref_idx);
    cfg.ref_off = (const int32_t *)INTEGER(ref_off);
    cfg.n_ref_grp = (int32_t)(XLENGTH(ref_off) - 1);
    const int pre = Rf_asLogical(want_pre);
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)cfg.G, (int)cfg.C));
    SEXP pre_m = PROTECT(pre ? Rf_allocMatrix(REALSXP, (int)cfg.G, (int)cfg.C) : R_NilValue);
    int rc = icnv_smooth_chain(REAL(expr), REAL(out), pre ? REAL(pre_m) : NULL, &cfg);
    if (rc) { UNPROTECT(2); fail(rc); }
    Rf_setAttrib(out, R_DimNamesSymbol, Rf_getAttrib(expr, R_DimNamesSymbol));
    if (pre) Rf_setAttrib(pre_m, R_DimNamesSymbol, Rf_getAttrib(expr, R_DimNamesSymbol));
    SEXP res = PROTECT(Rf_allocVector(VECSXP, 2));
    SET_VECTOR_ELT(res, 0, out);
    SET_VECTOR_ELT(res, 1, pre_m);
    UNPROTECT(3);
    return res;
}

SEXP icnv_R_average_bounds(SEXP expr) {
    double out2[2];
    int rc = icnv_average_bounds(REAL(expr), Rf_nrows(expr), Rf_ncols(expr), out2);
    if (rc) fail(rc);
    SEXP r = PROTECT(Rf_allocVector(REALSXP, 2));
    REAL(r)[0] = out2[0];
    REAL(r)[1] = out2[1];
    UNPROTECT(1);
    return r;
}

/* .Call("icnv_R_scale_genes", expr) -> matrix: t(scale(t(expr))) (scale_infercnv_expr, step 5 of run(), R/inferCNV_ops.R:3174-3185) */
SEXP icnv_R_scale_genes(SEXP expr) {
    if (!Rf_isReal(expr) || !Rf_isMatrix(expr)) Rf_error("expr must be a numeric matrix");
    const int64_t G = Rf_nrows(expr), C = Rf_ncols(expr);
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)G, (int)C));
    int rc = icnv_scale_genes(REAL(expr), REAL(out), G, C);
    if (rc) { UNPROTECT(1); fail(rc); }
    Rf_setAttrib(out, R_DimNamesSymbol, Rf_getAttrib(expr, R_DimNamesSymbol));
    UNPROTECT(1);
    return out;
}

/* .Call("icnv_R_remove_outliers", expr, lower_bound, upper_bound) -> matrix; NA bounds = out_method "average_bound"
 * (remove_outliers_norm, step 16 of run(), R/inferCNV_ops.R:1969-2054) */
SEXP icnv_R_remove_outliers(SEXP expr, SEXP lower_bound, SEXP upper_bound) {
    if (!Rf_isReal(expr) || !Rf_isMatrix(expr)) Rf_error("expr must be a numeric matrix");
    const int64_t G = Rf_nrows(expr), C = Rf_ncols(expr);
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)G, (int)C));
    int rc = icnv_remove_outliers(REAL(expr), REAL(out), G, C, Rf_asReal(lower_bound), Rf_asReal(upper_bound), NULL);   /* NA_real_ is a NaN */
    if (rc) { UNPROTECT(1); fail(rc); }
    Rf_setAttrib(out, R_DimNamesSymbol, Rf_getAttrib(expr, R_DimNamesSymbol));
    UNPROTECT(1);
    return out;
}

/* states are uint8 in the library; the reference stores them as numeric (R/inferCNV_HMM.R:294-295) */
static SEXP widen_states(const uint8_t *st, SEXP like) {
    const R_xlen_t n = XLENGTH(like);
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, Rf_nrows(like), Rf_ncols(like)));
    double *o = REAL(out);
    for (R_xlen_t i = 0; i < n; i++) o[i] = (st[i] == 0xFF) ? -1.0 : (double)st[i];
    Rf_setAttrib(out, R_DimNamesSymbol, Rf_getAttrib(like, R_DimNamesSymbol));
    UNPROTECT(1);
    return out;
}

/* .Call("icnv_R_viterbi_cells", expr, chr_start, mean, sd_shared, logPi, logDelta) */
SEXP icnv_R_viterbi_cells(SEXP expr, SEXP chr_start, SEXP mean, SEXP sd_shared, SEXP logPi, SEXP logDelta) {
    const int64_t G = Rf_nrows(expr), C = Rf_ncols(expr);
    uint8_t *st = (uint8_t *)R_alloc((size_t)G * (size_t)C, 1);
    int rc = icnv_viterbi_cells(REAL(expr), st, G, C, (const int32_t *)INTEGER(chr_start),
                                (int32_t)(XLENGTH(chr_start) - 1), (int32_t)XLENGTH(mean), REAL(mean),
                                Rf_asReal(sd_shared), REAL(logPi), REAL(logDelta));
    if (rc) fail(rc); /* ICNV_ERR_UNDERFLOW == the reference's stop("Problems With Underflow") */
    return widen_states(st, expr);
}

/* .Call("icnv_R_viterbi_groups", expr, chr_start, grp_idx, grp_off, mean, sd_per_grp, logPi, logDelta) */
SEXP icnv_R_viterbi_groups(SEXP expr, SEXP chr_start, SEXP grp_idx, SEXP grp_off, SEXP mean, SEXP sd_per_grp,
                           SEXP logPi, SEXP logDelta) {
    const int64_t G = Rf_nrows(expr), C = Rf_ncols(expr);
    uint8_t *st = (uint8_t *)R_alloc((size_t)G * (size_t)C, 1);
    int rc = icnv_viterbi_groups(REAL(expr), st, G, C, (const int32_t *)INTEGER(chr_start),
                                 (int32_t)(XLENGTH(chr_start) - 1), (const int32_t *)INTEGER(grp_idx),
                                 (const int32_t *)INTEGER(grp_off), (int32_t)(XLENGTH(grp_off) - 1),
                                 (int32_t)XLENGTH(mean), REAL(mean), REAL(sd_per_grp), REAL(logPi), REAL(logDelta));
    if (rc) fail(rc);
    return widen_states(st, expr);
}

/* .Call("icnv_R_median_filter", expr, chr_start, tile_idx, tile_off, window_size) */
SEXP icnv_R_median_filter(SEXP expr, SEXP chr_start, SEXP tile_idx, SEXP tile_off, SEXP window_size) {
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, Rf_nrows(expr), Rf_ncols(expr)));
    int rc = icnv_median_filter(REAL(expr), REAL(out), Rf_nrows(expr), Rf_ncols(expr),
                                (const int32_t *)INTEGER(chr_start), (int32_t)(XLENGTH(chr_start) - 1),
                                (const int32_t *)INTEGER(tile_idx), (const int32_t *)INTEGER(tile_off),
                                (int32_t)(XLENGTH(tile_off) - 1), Rf_asInteger(window_size));
    if (rc) { UNPROTECT(1); fail(rc); }
    Rf_setAttrib(out, R_DimNamesSymbol, Rf_getAttrib(expr, R_DimNamesSymbol));
    UNPROTECT(1);
    return out;
}

/* .Call("icnv_R_cell_distances", expr, cell_idx): full symmetric matrix of parallelDist(t(expr[, cells])) */
SEXP icnv_R_cell_distances(SEXP expr, SEXP cell_idx) {
    const R_xlen_t n = XLENGTH(cell_idx);
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)n, (int)n));
    int rc = icnv_cell_distances(REAL(expr), Rf_nrows(expr), Rf_ncols(expr), (const int32_t *)INTEGER(cell_idx), (int64_t)n,
                                 REAL(out));
    if (rc) { UNPROTECT(1); fail(rc); }
    UNPROTECT(1);
    return out;
}

/* .Call("icnv_R_states_to_proxy", states, K): states as the reference stores them (numeric, -1 = untouched); entries that
 * are not a state 1..K keep their value, like the reference's masked assignments (R/inferCNV_HMM.R:1195-1200) */
SEXP icnv_R_states_to_proxy(SEXP states, SEXP K) {
    const R_xlen_t n = XLENGTH(states);
    const double *x = REAL(states);
    const int k = Rf_asInteger(K);
    const double kmax = (double)k;   /* a value above the model's last state is not a state: it keeps its value */
    uint8_t *st = (uint8_t *)R_alloc((size_t)n, 1);
    for (R_xlen_t i = 0; i < n; i++) st[i] = (x[i] >= 1.0 && x[i] <= kmax && x[i] == (double)(int)x[i]) ? (uint8_t)x[i] : 0xFF;
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, Rf_nrows(states), Rf_ncols(states)));
    int rc = icnv_states_to_proxy(st, REAL(out), (int64_t)n, k);
    if (rc) { UNPROTECT(1); fail(rc); }
    double *o = REAL(out);
    for (R_xlen_t i = 0; i < n; i++)
        if (ISNAN(o[i])) o[i] = x[i];
    Rf_setAttrib(out, R_DimNamesSymbol, Rf_getAttrib(states, R_DimNamesSymbol));
    UNPROTECT(1);
    return out;
}

/* .Call("icnv_R_state_consensus_overwrite", states, grp_idx, grp_off): every member cell of a group receives the group's
 * per-gene consensus state (.get_state_consensus, R/inferCNV_HMM.R:977-987; the overwrite of :473-483) */
SEXP icnv_R_state_consensus_overwrite(SEXP states, SEXP grp_idx, SEXP grp_off) {
    const R_xlen_t n = XLENGTH(states);
    const double *x = REAL(states);
    uint8_t *st = (uint8_t *)R_alloc((size_t)n, 1);
    uint8_t *so = (uint8_t *)R_alloc((size_t)n, 1);
    for (R_xlen_t i = 0; i < n; i++) st[i] = (x[i] >= 0.0 && x[i] < 255.0) ? (uint8_t)x[i] : 0xFF;
    int rc = icnv_state_consensus(st, Rf_nrows(states), Rf_ncols(states), (const int32_t *)INTEGER(grp_idx),
                                  (const int32_t *)INTEGER(grp_off), (int32_t)(XLENGTH(grp_off) - 1), NULL, so);
    if (rc) fail(rc);
    return widen_states(so, states);
}

/* .Call("icnv_R_ingest_counts", counts, colptr, rowidx, vals, G, C, min_mean_expr_cutoff, min_cells_per_gene, normalize_factor)
 *   -> list(expr = G_out x C numeric matrix (log2(count / depth * factor + 1) of the kept genes), keep = 1-based kept genes,
 *           factor = normalisation factor used, h2d_bytes)
 * Steps 2, 3, 4 of run() in one call from the raw counts (R/inferCNV_ops.R:2128-2213, 3064-3111, 2756-2769).  Dense: `counts`
 * is an INTEGER matrix (G x C) and colptr / rowidx / vals are NULL.  Sparse: `counts` is NULL and colptr / rowidx / vals are the
 * @p, @i, @x slots of a dgCMatrix (x rounded to integer by the R wrapper); R has no 64-bit integers, so @p arrives as INTEGER
 * and is widened here. */
SEXP icnv_R_ingest_counts(SEXP counts, SEXP colptr, SEXP rowidx, SEXP vals, SEXP G_, SEXP C_, SEXP min_mean_expr_cutoff,
                          SEXP min_cells_per_gene, SEXP normalize_factor) {
    const int64_t G = Rf_asInteger(G_), C = Rf_asInteger(C_);
    icnv_counts cnt;
    memset(&cnt, 0, sizeof(cnt));
    int64_t *cp64 = NULL;
    if (counts != R_NilValue) {
        if (Rf_nrows(counts) != G || Rf_ncols(counts) != C) Rf_error("counts must be a G x C integer matrix");
        cnt.dense = (const int32_t *)INTEGER(counts);
    } else {
        if (XLENGTH(colptr) != C + 1) Rf_error("colptr must have C + 1 entries");
        cp64 = (int64_t *)R_alloc((size_t)C + 1, sizeof(int64_t));
        for (int64_t c = 0; c <= C; c++) cp64[c] = (int64_t)INTEGER(colptr)[c];
        cnt.colptr = cp64;
        cnt.rowidx = (const int32_t *)INTEGER(rowidx);
        cnt.vals = (const int32_t *)INTEGER(vals);
        cnt.nnz = (int64_t)XLENGTH(vals);
    }
    int32_t *keep = (int32_t *)R_alloc((size_t)G, sizeof(int32_t));
    double *buf = (double *)R_alloc((size_t)G * (size_t)C, sizeof(double));   /* capacity G x C, filled G_out x C */
    int64_t g_out = 0, up = 0;
    double used = 0.0;
    int rc = icnv_ingest_counts(&cnt, G, C, Rf_asReal(min_mean_expr_cutoff), Rf_asInteger(min_cells_per_gene), Rf_asReal(normalize_factor),
                                keep, &g_out, buf, &used, &up);
    if (rc) fail(rc);
    SEXP expr = PROTECT(Rf_allocMatrix(REALSXP, (int)g_out, (int)C));
    memcpy(REAL(expr), buf, (size_t)g_out * (size_t)C * sizeof(double));
    SEXP kept = PROTECT(Rf_allocVector(INTSXP, (R_xlen_t)g_out));
    for (int64_t j = 0; j < g_out; j++) INTEGER(kept)[j] = keep[j] + 1;
    SEXP f = PROTECT(Rf_allocVector(REALSXP, 2));
    REAL(f)[0] = used;
    REAL(f)[1] = (double)up;
    SEXP res = PROTECT(Rf_allocVector(VECSXP, 3));
    SET_VECTOR_ELT(res, 0, expr);
    SET_VECTOR_ELT(res, 1, kept);
    SET_VECTOR_ELT(res, 2, f);
    UNPROTECT(4);
    return res;
}

/* .Call("icnv_R_init", devices, residency): devices 0 = all visible GPUs, n = the first n, -1 = the current one only */
SEXP icnv_R_init(SEXP devices, SEXP residency) {
    const int nd = Rf_asInteger(devices);
    int rc = icnv_init(nd < 0 ? -1 : 0);
    if (!rc) rc = icnv_set_devices(nd < 0 ? 1 : nd);
    if (!rc) rc = icnv_residency(Rf_asLogical(residency) == TRUE);
    if (rc) fail(rc);
    return R_NilValue;
}

static const R_CallMethodDef call_methods[] = {
    {"icnv_R_smooth_chain", (DL_FUNC)&icnv_R_smooth_chain, 13},
    {"icnv_R_average_bounds", (DL_FUNC)&icnv_R_average_bounds, 1},
    {"icnv_R_remove_outliers", (DL_FUNC)&icnv_R_remove_outliers, 3},
    {"icnv_R_scale_genes", (DL_FUNC)&icnv_R_scale_genes, 1},
    {"icnv_R_viterbi_cells", (DL_FUNC)&icnv_R_viterbi_cells, 6},
    {"icnv_R_viterbi_groups", (DL_FUNC)&icnv_R_viterbi_groups, 8},
    {"icnv_R_median_filter", (DL_FUNC)&icnv_R_median_filter, 5},
    {"icnv_R_cell_distances", (DL_FUNC)&icnv_R_cell_distances, 2},
    {"icnv_R_states_to_proxy", (DL_FUNC)&icnv_R_states_to_proxy, 2},
    {"icnv_R_state_consensus_overwrite", (DL_FUNC)&icnv_R_state_consensus_overwrite, 3},
    {"icnv_R_ingest_counts", (DL_FUNC)&icnv_R_ingest_counts, 9},
    {"icnv_R_init", (DL_FUNC)&icnv_R_init, 2},
    {NULL, NULL, 0}};

void R_init_infercnv(DllInfo *dll) {
    R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
