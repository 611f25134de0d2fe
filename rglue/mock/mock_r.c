/* Toy implementation of the R C API subset declared in rglue/mock/Rinternals.h -- TEST INFRASTRUCTURE.
 * SEXPs are malloc'ed records that live until mock_r_reset(); Rf_error() formats the message into mock_r_last_error
 * and longjmps to the jmp_buf the test armed with mock_r_try (R's own Rf_error longjmps to the top level). */
#include "mock_r.h"

#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct mock_sexp {
    SEXPTYPE type;
    R_xlen_t len;
    int nrow, ncol, is_matrix;
    void *data;
    SEXP dimnames;
    struct mock_sexp *next_alloc;
};
struct mock_dllinfo { const R_CallMethodDef *call; int dynamic; };

static struct mock_sexp nil_rec = {NILSXP, 0, 0, 0, 0, NULL, NULL, NULL};
static struct mock_sexp dimnames_sym = {NILSXP, 0, 0, 0, 0, NULL, NULL, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_DimNamesSymbol = &dimnames_sym;
double R_NaReal = NAN;

char mock_r_last_error[1024];
int mock_r_protect_depth = 0;
jmp_buf *mock_r_jmp = NULL;
static struct mock_sexp *all_allocs = NULL;
static void *raw_allocs[4096];
static int n_raw = 0;
static struct mock_dllinfo the_dll;

void Rf_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(mock_r_last_error, sizeof(mock_r_last_error), fmt, ap);
    va_end(ap);
    if (mock_r_jmp) longjmp(*mock_r_jmp, 1);
    fprintf(stderr, "Rf_error outside mock_r_try: %s\n", mock_r_last_error);
    abort();
}

static size_t elt_size(SEXPTYPE t) {
    switch (t) {
    case REALSXP: return sizeof(double);
    case INTSXP: case LGLSXP: return sizeof(int);
    case VECSXP: case STRSXP: return sizeof(SEXP);
    default: return 1;
    }
}
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n) {
    struct mock_sexp *s = (struct mock_sexp *)calloc(1, sizeof(*s));
    s->type = type;
    s->len = n;
    s->data = calloc((size_t)(n > 0 ? n : 1), elt_size(type));
    s->dimnames = R_NilValue;
    if (type == VECSXP) for (R_xlen_t i = 0; i < n; i++) ((SEXP *)s->data)[i] = R_NilValue;
    s->next_alloc = all_allocs;
    all_allocs = s;
    return s;
}
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol) {
    SEXP s = Rf_allocVector(type, (R_xlen_t)nrow * (R_xlen_t)ncol);
    s->nrow = nrow;
    s->ncol = ncol;
    s->is_matrix = 1;
    return s;
}
Rboolean Rf_isReal(SEXP x) { return x->type == REALSXP ? TRUE : FALSE; }
Rboolean Rf_isMatrix(SEXP x) { return x->is_matrix ? TRUE : FALSE; }
int Rf_nrows(SEXP x) { if (!x->is_matrix) Rf_error("object is not a matrix"); return x->nrow; }
int Rf_ncols(SEXP x) { if (!x->is_matrix) Rf_error("object is not a matrix"); return x->ncol; }
int *INTEGER(SEXP x) { if (x->type != INTSXP && x->type != LGLSXP) Rf_error("INTEGER() can only be applied to a 'integer', not a type %u", x->type); return (int *)x->data; }
double *REAL(SEXP x) { if (x->type != REALSXP) Rf_error("REAL() can only be applied to a 'numeric', not a type %u", x->type); return (double *)x->data; }
R_xlen_t XLENGTH(SEXP x) { return x->len; }
int Rf_asInteger(SEXP x) {
    if (x->len < 1) return NA_INTEGER;
    if (x->type == INTSXP || x->type == LGLSXP) return ((int *)x->data)[0];
    if (x->type == REALSXP) { double v = ((double *)x->data)[0]; return isnan(v) ? NA_INTEGER : (int)v; }
    return NA_INTEGER;
}
double Rf_asReal(SEXP x) {
    if (x->len < 1) return R_NaReal;
    if (x->type == REALSXP) return ((double *)x->data)[0];
    if (x->type == INTSXP || x->type == LGLSXP) { int v = ((int *)x->data)[0]; return v == NA_INTEGER ? R_NaReal : (double)v; }
    return R_NaReal;
}
int Rf_asLogical(SEXP x) {
    if (x->len < 1) return NA_LOGICAL;
    if (x->type == LGLSXP || x->type == INTSXP) { int v = ((int *)x->data)[0]; return v == NA_INTEGER ? NA_LOGICAL : (v != 0); }
    if (x->type == REALSXP) { double v = ((double *)x->data)[0]; return isnan(v) ? NA_LOGICAL : (v != 0.0); }
    return NA_LOGICAL;
}
SEXP Rf_protect(SEXP x) { mock_r_protect_depth++; return x; }
void Rf_unprotect(int n) { mock_r_protect_depth -= n; }
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP value) { if (name == R_DimNamesSymbol && x != R_NilValue) x->dimnames = value; return value; }
SEXP Rf_getAttrib(SEXP x, SEXP name) { return (name == R_DimNamesSymbol && x != R_NilValue) ? x->dimnames : R_NilValue; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { if (x->type != VECSXP || i < 0 || i >= x->len) Rf_error("SET_VECTOR_ELT out of range"); ((SEXP *)x->data)[i] = v; return v; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { if (x->type != VECSXP || i < 0 || i >= x->len) Rf_error("VECTOR_ELT out of range"); return ((SEXP *)x->data)[i]; }
char *R_alloc(size_t n, int size) {
    void *p = calloc(n ? n : 1, (size_t)size);
    if (n_raw < 4096) raw_allocs[n_raw++] = p;
    return (char *)p;
}
int R_registerRoutines(DllInfo *info, const R_CMethodDef *const c, const R_CallMethodDef *const call,
                       const R_FortranMethodDef *const f, const R_ExternalMethodDef *const ext) {
    (void)c; (void)f; (void)ext;
    info->call = call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value) { Rboolean old = info->dynamic ? TRUE : FALSE; info->dynamic = value; return old; }

/* ---- helpers of the test driver ---- */
DllInfo *mock_r_dll(void) { the_dll.dynamic = 1; return &the_dll; }
const R_CallMethodDef *mock_r_registered(void) { return the_dll.call; }
int mock_r_dynamic_symbols(void) { return the_dll.dynamic; }
SEXP mock_r_real_matrix(int nrow, int ncol) { return Rf_allocMatrix(REALSXP, nrow, ncol); }
SEXP mock_r_ints(const int *v, R_xlen_t n) { SEXP s = Rf_allocVector(INTSXP, n); if (n) memcpy(s->data, v, (size_t)n * sizeof(int)); return s; }
SEXP mock_r_reals(const double *v, R_xlen_t n) { SEXP s = Rf_allocVector(REALSXP, n); if (n) memcpy(s->data, v, (size_t)n * sizeof(double)); return s; }
SEXP mock_r_int(int v) { return mock_r_ints(&v, 1); }
SEXP mock_r_real(double v) { return mock_r_reals(&v, 1); }
SEXP mock_r_lgl(int v) { SEXP s = Rf_allocVector(LGLSXP, 1); ((int *)s->data)[0] = v; return s; }
void mock_r_reset(void) {
    while (all_allocs) { struct mock_sexp *n = all_allocs->next_alloc; free(all_allocs->data); free(all_allocs); all_allocs = n; }
    for (int i = 0; i < n_raw; i++) free(raw_allocs[i]);
    n_raw = 0;
    mock_r_protect_depth = 0;
}
