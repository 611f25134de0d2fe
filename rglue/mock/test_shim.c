/*
 * C driver of rglue/src/icnv_shim.c against the mock R API -- TEST INFRASTRUCTURE (see Rinternals.h here).
 *
 *   test_shim cpu   registration table, arities (compile time and run time), the error path: the library has no
 *                   device, returns a code, the shim raises Rf_error AFTER the library returned
 *   test_shim gpu   the .Call routines on a small matrix against direct calls of the C ABI: smooth chain (+ pre-denoise
 *                   matrix, dimnames carried over), per-cell and group Viterbi (states widened, 0xFF -> -1), median
 *                   filter, state -> proxy tables (K = 6, K = 3 with values beyond K left untouched), PROTECT balance
 */
#include <stdio.h>

#include "mock_r.h"
#include "../src/icnv_shim.c"

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

/* ---- arities: the function types, at compile time ---- */
#define S1 SEXP
#define S2 S1, SEXP
#define S3 S2, SEXP
#define S5 S3, SEXP, SEXP
#define S6 S5, SEXP
#define S8 S6, SEXP, SEXP
#define S9 S8, SEXP
#define S12 S8, SEXP, SEXP, SEXP, SEXP
#define S13 S12, SEXP
#define ARITY(fn, n, ...) _Static_assert(__builtin_types_compatible_p(__typeof__(&fn), SEXP (*)(__VA_ARGS__)), #fn " does not take " #n " SEXPs");
ARITY(icnv_R_smooth_chain, 13, S13)
ARITY(icnv_R_average_bounds, 1, S1)
ARITY(icnv_R_remove_outliers, 3, S3)
ARITY(icnv_R_scale_genes, 1, S1)
ARITY(icnv_R_viterbi_cells, 6, S6)
ARITY(icnv_R_viterbi_groups, 8, S8)
ARITY(icnv_R_median_filter, 5, S5)
ARITY(icnv_R_cell_distances, 2, S2)
ARITY(icnv_R_states_to_proxy, 2, S2)
ARITY(icnv_R_state_consensus_overwrite, 3, S3)
ARITY(icnv_R_ingest_counts, 9, S9)
ARITY(icnv_R_init, 2, S2)
static const struct { const char *name; int n; } expected[] = {
    {"icnv_R_smooth_chain", 13}, {"icnv_R_average_bounds", 1}, {"icnv_R_remove_outliers", 3}, {"icnv_R_scale_genes", 1}, {"icnv_R_viterbi_cells", 6}, {"icnv_R_viterbi_groups", 8},
    {"icnv_R_median_filter", 5}, {"icnv_R_cell_distances", 2}, {"icnv_R_states_to_proxy", 2},
    {"icnv_R_state_consensus_overwrite", 3}, {"icnv_R_ingest_counts", 9}, {"icnv_R_init", 2}};

static int check_registration(void) {
    R_init_infercnv(mock_r_dll());
    const R_CallMethodDef *t = mock_r_registered();
    CHECK(t != NULL);
    CHECK(mock_r_dynamic_symbols() == 0);
    int n = 0;
    for (; t[n].name; n++) {
        CHECK(n < (int)(sizeof(expected) / sizeof(expected[0])));
        CHECK(strcmp(t[n].name, expected[n].name) == 0);
        CHECK(t[n].numArgs == expected[n].n);
        CHECK(t[n].fun != NULL);
    }
    CHECK(n == (int)(sizeof(expected) / sizeof(expected[0])));
    return 0;
}

static double val(int g, int c) {   /* deterministic, smooth along the genes, different per cell */
    return 1.0 + 0.3 * sin(0.013 * g + 0.7 * c) + 0.05 * cos(1.7 * g + c) + ((c % 4 == 1 && g < 120) ? 0.4 : 0.0);
}

static int run_cpu(void) {
    CHECK(check_registration() == 0);
    SEXP m = mock_r_real_matrix(3, 2);
    for (int i = 0; i < 6; i++) REAL(m)[i] = (double)i;
    int raised = 0;
    mock_r_try(raised, (void)icnv_R_average_bounds(m));
    CHECK(raised == 1);                                         /* no device here: code -> Rf_error, after the library returned */
    CHECK(strncmp(mock_r_last_error, "libicnv_hip error", 17) == 0);
    /* argument checks of the shim itself come before any library call */
    SEXP notm = mock_r_reals((const double[]){1.0, 2.0}, 2);
    mock_r_try(raised, (void)icnv_R_smooth_chain(notm, m, m, m, m, m, m, m, m, m, m, m, m));
    CHECK(raised == 1 && strstr(mock_r_last_error, "numeric matrix") != NULL);
    mock_r_reset();
    printf("SHIM_CPU_OK\n");
    return 0;
}

static int run_gpu(void) {
    CHECK(check_registration() == 0);
    int raised = 0;
    mock_r_try(raised, (void)icnv_R_init(mock_r_int(-1), mock_r_lgl(0)));
    CHECK(raised == 0);
    enum { G = 600, C = 40 };
    const int chr_start[3] = {0, 250, G};
    const int ref_idx[6] = {0, 4, 8, 12, 16, 20}, ref_off[3] = {0, 3, 6};
    SEXP x = mock_r_real_matrix(G, C);
    for (int c = 0; c < C; c++)
        for (int g = 0; g < G; g++) REAL(x)[g + G * c] = log2(1.0 + val(g, c));
    SEXP dn = Rf_allocVector(VECSXP, 2);
    Rf_setAttrib(x, R_DimNamesSymbol, dn);
    SEXP cs = mock_r_ints(chr_start, 3), ri = mock_r_ints(ref_idx, 6), ro = mock_r_ints(ref_off, 3);
    /* ---- smooth chain: all stages, pre-denoise matrix wanted ---- */
    SEXP res = NULL;
    mock_r_try(raised, res = icnv_R_smooth_chain(x, cs, ri, ro, mock_r_int(101), mock_r_real(3.0), mock_r_lgl(1), mock_r_real(1.5),
                                                  mock_r_real(NA_REAL), mock_r_int(0x7F), mock_r_lgl(1), mock_r_lgl(0), mock_r_lgl(0)));
    CHECK(raised == 0 && mock_r_protect_depth == 0);
    CHECK(XLENGTH(res) == 2);
    SEXP out = VECTOR_ELT(res, 0), pre = VECTOR_ELT(res, 1);
    CHECK(Rf_nrows(out) == G && Rf_ncols(out) == C && Rf_nrows(pre) == G && Rf_ncols(pre) == C);
    CHECK(Rf_getAttrib(out, R_DimNamesSymbol) == dn && Rf_getAttrib(pre, R_DimNamesSymbol) == dn);
    icnv_chain_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.G = G; cfg.C = C; cfg.chr_start = chr_start; cfg.n_chr = 2; cfg.ref_idx = ref_idx; cfg.ref_off = ref_off; cfg.n_ref_grp = 2;
    cfg.window_length = 101; cfg.max_thresh = 3.0; cfg.use_bounds = 1; cfg.sd_amplifier = 1.5; cfg.noise_filter = NAN;
    cfg.stage_mask = 0x7F;
    double *o2 = (double *)malloc(sizeof(double) * G * C), *p2 = (double *)malloc(sizeof(double) * G * C);
    CHECK(icnv_smooth_chain(REAL(x), o2, p2, &cfg) == 0);
    CHECK(memcmp(o2, REAL(out), sizeof(double) * G * C) == 0 && memcmp(p2, REAL(pre), sizeof(double) * G * C) == 0);
    /* without the pre-denoise matrix the second element is NULL; an even window is the library's error, raised by the shim */
    mock_r_try(raised, res = icnv_R_smooth_chain(x, cs, ri, ro, mock_r_int(101), mock_r_real(3.0), mock_r_lgl(1), mock_r_real(1.5),
                                                  mock_r_real(NA_REAL), mock_r_int(0x3F), mock_r_lgl(0), mock_r_lgl(0), mock_r_lgl(0)));
    CHECK(raised == 0 && VECTOR_ELT(res, 1) == R_NilValue && mock_r_protect_depth == 0);
    mock_r_try(raised, res = icnv_R_smooth_chain(x, cs, ri, ro, mock_r_int(100), mock_r_real(3.0), mock_r_lgl(1), mock_r_real(1.5),
                                                  mock_r_real(NA_REAL), mock_r_int(0x7F), mock_r_lgl(0), mock_r_lgl(0), mock_r_lgl(0)));
    CHECK(raised == 1 && strstr(mock_r_last_error, "odd") != NULL && mock_r_protect_depth == 0);
    /* ---- per-cell i6 Viterbi on the pre-denoise matrix ---- */
    const double mean6[6] = {0.01, 0.5, 1.0, 1.5, 2.0, 3.0};
    double logPi[36], logDelta[6];
    for (int j = 0; j < 6; j++) {
        logDelta[j] = log(j == 2 ? 1.0 - 5e-6 : 1e-6);
        for (int k = 0; k < 6; k++) logPi[j + 6 * k] = log(j == k ? 1.0 - 5e-6 : 1e-6);
    }
    SEXP mean = mock_r_reals(mean6, 6), lp = mock_r_reals(logPi, 36), ld = mock_r_reals(logDelta, 6);
    SEXP st = NULL;
    mock_r_try(raised, st = icnv_R_viterbi_cells(pre, cs, mean, mock_r_real(0.18), lp, ld));
    CHECK(raised == 0 && mock_r_protect_depth == 0 && Rf_nrows(st) == G && Rf_ncols(st) == C);
    uint8_t *s2 = (uint8_t *)malloc((size_t)G * C);
    CHECK(icnv_viterbi_cells(REAL(pre), s2, G, C, chr_start, 2, 6, mean6, 0.18, logPi, logDelta) == 0);
    int seen[8] = {0};
    for (int i = 0; i < G * C; i++) { CHECK(REAL(st)[i] == (double)s2[i]); seen[s2[i] & 7]++; }
    CHECK(seen[3] > 0 && seen[0] == 0 && seen[7] == 0);         /* states 1..6, the neutral one among them */
    /* ---- groups: cells 0..9 and 10..29, cells 30..39 in no group -> -1 ---- */
    int gidx[30], goff[3] = {0, 10, 30};
    for (int i = 0; i < 30; i++) gidx[i] = i;
    const double sdg[2] = {0.1, 0.07};
    mock_r_try(raised, st = icnv_R_viterbi_groups(pre, cs, mock_r_ints(gidx, 30), mock_r_ints(goff, 3), mean, mock_r_reals(sdg, 2), lp, ld));
    CHECK(raised == 0 && mock_r_protect_depth == 0);
    CHECK(icnv_viterbi_groups(REAL(pre), s2, G, C, chr_start, 2, gidx, goff, 2, 6, mean6, sdg, logPi, logDelta) == 0);
    for (int i = 0; i < G * C; i++) CHECK(REAL(st)[i] == (s2[i] == 0xFF ? -1.0 : (double)s2[i]));
    CHECK(REAL(st)[G * 35] == -1.0 && REAL(st)[0] >= 1.0);
    /* ---- state -> proxy: K = 6, then K = 3 where 4..6 and -1 are not states and keep their value ---- */
    SEXP pr = NULL;
    mock_r_try(raised, pr = icnv_R_states_to_proxy(st, mock_r_int(6)));
    CHECK(raised == 0 && mock_r_protect_depth == 0);
    const double tab6[7] = {0, 0.0, 0.5, 1.0, 1.5, 2.0, 3.0}, tab3[4] = {0, 0.5, 1.0, 1.5};
    for (int i = 0; i < G * C; i++) CHECK(REAL(pr)[i] == (REAL(st)[i] < 0 ? -1.0 : tab6[(int)REAL(st)[i]]));
    SEXP s3 = mock_r_real_matrix(7, 1);
    const double v3[7] = {1, 2, 3, 4, 5, 6, -1};
    memcpy(REAL(s3), v3, sizeof(v3));
    mock_r_try(raised, pr = icnv_R_states_to_proxy(s3, mock_r_int(3)));
    CHECK(raised == 0);
    for (int i = 0; i < 7; i++) CHECK(REAL(pr)[i] == (i < 3 ? tab3[i + 1] : v3[i]));
    /* ---- median filter over two tiles ---- */
    SEXP mf = NULL;
    mock_r_try(raised, mf = icnv_R_median_filter(out, cs, mock_r_ints(gidx, 30), mock_r_ints(goff, 3), mock_r_int(7)));
    CHECK(raised == 0 && mock_r_protect_depth == 0 && Rf_getAttrib(mf, R_DimNamesSymbol) == dn);
    CHECK(icnv_median_filter(REAL(out), o2, G, C, chr_start, 2, gidx, goff, 2, 7) == 0);
    CHECK(memcmp(o2, REAL(mf), sizeof(double) * G * C) == 0);
    /* ---- ingest from integer counts: dense and CSC give the same matrix; the kept genes are 1-based ---- */
    {
        enum { IG = 50, IC = 12 };
        SEXP cm = Rf_allocMatrix(INTSXP, IG, IC);
        int nnz = 0;
        for (int c = 0; c < IC; c++)
            for (int g = 0; g < IG; g++) {
                const int v = ((g * 7 + c * 3) % 11 < 4 && g % 10 != 3) ? (g + c) % 9 + 1 : 0;   /* genes 3, 13, ... never expressed */
                INTEGER(cm)[g + IG * c] = v;
                nnz += v != 0;
            }
        int *cp = (int *)malloc(sizeof(int) * (IC + 1)), *ri = (int *)malloc(sizeof(int) * (size_t)nnz), *vv = (int *)malloc(sizeof(int) * (size_t)nnz);
        int k = 0;
        for (int c = 0; c < IC; c++) {
            cp[c] = k;
            for (int g = 0; g < IG; g++)
                if (INTEGER(cm)[g + IG * c]) { ri[k] = g; vv[k] = INTEGER(cm)[g + IG * c]; k++; }
        }
        cp[IC] = k;
        SEXP rd = NULL, rs = NULL;
        mock_r_try(raised, rd = icnv_R_ingest_counts(cm, R_NilValue, R_NilValue, R_NilValue, mock_r_int(IG), mock_r_int(IC), mock_r_real(0.05),
                                                     mock_r_int(2), mock_r_real(NA_REAL)));
        CHECK(raised == 0 && mock_r_protect_depth == 0);
        mock_r_try(raised, rs = icnv_R_ingest_counts(R_NilValue, mock_r_ints(cp, IC + 1), mock_r_ints(ri, nnz), mock_r_ints(vv, nnz), mock_r_int(IG),
                                                     mock_r_int(IC), mock_r_real(0.05), mock_r_int(2), mock_r_real(NA_REAL)));
        CHECK(raised == 0 && mock_r_protect_depth == 0);
        SEXP ed = VECTOR_ELT(rd, 0), es = VECTOR_ELT(rs, 0), kd = VECTOR_ELT(rd, 1), ks = VECTOR_ELT(rs, 1);
        CHECK(Rf_ncols(ed) == IC && Rf_nrows(ed) == (int)XLENGTH(kd) && Rf_nrows(ed) < IG && Rf_nrows(ed) > 10);
        CHECK(XLENGTH(kd) == XLENGTH(ks) && memcmp(INTEGER(kd), INTEGER(ks), sizeof(int) * (size_t)XLENGTH(kd)) == 0);
        CHECK(memcmp(REAL(ed), REAL(es), sizeof(double) * (size_t)XLENGTH(ed)) == 0);
        for (R_xlen_t j = 0; j < XLENGTH(kd); j++) CHECK(INTEGER(kd)[j] >= 1 && INTEGER(kd)[j] <= IG && INTEGER(kd)[j] % 10 != 4);   /* 1-based; gene 3 (0-based) is gone */
        CHECK(REAL(VECTOR_ELT(rd, 2))[1] == (double)(IG * IC * 4) && REAL(VECTOR_ELT(rs, 2))[1] == (double)((IC + 1) * 8 + nnz * 8));
        free(cp); free(ri); free(vv);
    }
    /* ---- average bounds ---- */
    SEXP ab = NULL;
    mock_r_try(raised, ab = icnv_R_average_bounds(x));
    double ab2[2];
    CHECK(raised == 0 && icnv_average_bounds(REAL(x), G, C, ab2) == 0 && REAL(ab)[0] == ab2[0] && REAL(ab)[1] == ab2[1]);
    /* ---- step 16: the reference's own literal case (tests/testthat/test_infer_cnv.R:405-433): average bounds -0.5 / 17.75 ---- */
    {
        SEXP m = Rf_allocMatrix(REALSXP, 15, 4);
        for (int c = 0; c < 4; c++)
            for (int g = 0; g < 15; g++) REAL(m)[g + 15 * c] = g + 1;
        const double col2[15] = {-5, -4, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 21, 26};
        for (int g = 0; g < 15; g++) REAL(m)[g + 15] = col2[g];
        SEXP ro = NULL;
        mock_r_try(raised, ro = icnv_R_remove_outliers(m, mock_r_real(NA_REAL), mock_r_real(NA_REAL)));
        CHECK(raised == 0 && mock_r_protect_depth == 0 && Rf_nrows(ro) == 15 && Rf_ncols(ro) == 4);
        for (int c = 0; c < 4; c++)
            for (int g = 0; g < 15; g++) {
                double want = REAL(m)[g + 15 * c];
                if (c == 1 && g < 2) want = -0.5;
                if (c == 1 && g >= 13) want = 17.75;
                CHECK(REAL(ro)[g + 15 * c] == want);
            }
        mock_r_try(raised, ro = icnv_R_remove_outliers(m, mock_r_real(5.0), mock_r_real(10.0)));   /* hard thresholds */
        CHECK(raised == 0 && REAL(ro)[0] == 5.0 && REAL(ro)[14] == 10.0 && REAL(ro)[7] == 8.0);
        /* step 5: every gene (row) of t(scale(t(m))) has mean 0 and sample sd 1; row 0 of m is (1, -5, 1, 1) */
        SEXP sc = NULL;
        mock_r_try(raised, sc = icnv_R_scale_genes(m));
        CHECK(raised == 0 && mock_r_protect_depth == 0 && Rf_nrows(sc) == 15 && Rf_ncols(sc) == 4);
        CHECK(fabs(REAL(sc)[0] - 0.5) < 1e-14 && fabs(REAL(sc)[15] + 1.5) < 1e-14);              /* (1 - (-0.5)) / 3, (-5 + 0.5) / 3 */
    }
    free(o2); free(p2); free(s2);
    mock_r_reset();
    printf("SHIM_GPU_OK\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) return run_gpu();
    return run_cpu();
}
