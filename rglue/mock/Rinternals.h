/*
 * Mock of the part of R's C API that rglue/src/icnv_shim.c uses -- TEST INFRASTRUCTURE.
 *
 * R is not installed in the build image, so the shim cannot be compiled against the real <Rinternals.h>.  This header
 * declares the ~25 symbols the shim touches with R's own signatures (R 4.x, src/include/Rinternals.h and
 * R_ext/Rdynload.h) and rglue/mock/mock_r.c implements them on a toy SEXP, so that
 *   * `gcc -fsyntax-only -Wall -Wextra` type-checks every line of the shim (typos, arities, pointer types),
 *   * rglue/mock/test_shim.c can drive the .Call routines from C against libicnv_hip.so
 * (tests/test_host.py::test_r_shim_compiles_and_registers, tests/test_gpu_entrypoints.py::test_r_shim_driven_from_c).
 * It is NOT a substitute for building inside the package with R's headers.
 */
#ifndef ICNV_MOCK_RINTERNALS_H
#define ICNV_MOCK_RINTERNALS_H
#include <math.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mock_sexp *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE } Rboolean;
typedef unsigned int SEXPTYPE;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19

extern SEXP R_NilValue;
extern SEXP R_DimNamesSymbol;
extern double R_NaReal;
#define NA_REAL R_NaReal
#define NA_INTEGER (-2147483647 - 1)
#define NA_LOGICAL NA_INTEGER
#define ISNAN(x) (isnan(x) != 0)

void Rf_error(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
Rboolean Rf_isReal(SEXP x);
Rboolean Rf_isMatrix(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);
int *INTEGER(SEXP x);
double *REAL(SEXP x);
R_xlen_t XLENGTH(SEXP x);
int Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
int Rf_asLogical(SEXP x);
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP value);
SEXP Rf_getAttrib(SEXP x, SEXP name);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
char *R_alloc(size_t n, int size);

#ifdef __cplusplus
}
#endif
#endif
