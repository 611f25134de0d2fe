/* test-driver side of the mock R API (rglue/mock/mock_r.c) -- TEST INFRASTRUCTURE */
#ifndef ICNV_MOCK_R_DRIVER_H
#define ICNV_MOCK_R_DRIVER_H
#include <setjmp.h>

#include "Rinternals.h"
#include "R_ext/Rdynload.h"

extern char mock_r_last_error[1024];
extern int mock_r_protect_depth;   /* PROTECT / UNPROTECT balance: must be 0 after every .Call routine returns */
extern jmp_buf *mock_r_jmp;
/* run `stmt`; evaluates to 1 if it raised Rf_error (message in mock_r_last_error) */
#define mock_r_try(raised, stmt)              \
    do {                                      \
        jmp_buf jb_;                          \
        mock_r_jmp = &jb_;                    \
        if (setjmp(jb_) == 0) { stmt; (raised) = 0; } else { (raised) = 1; } \
        mock_r_jmp = NULL;                    \
    } while (0)
DllInfo *mock_r_dll(void);
const R_CallMethodDef *mock_r_registered(void);
int mock_r_dynamic_symbols(void);
SEXP mock_r_real_matrix(int nrow, int ncol);
SEXP mock_r_ints(const int *v, R_xlen_t n);
SEXP mock_r_reals(const double *v, R_xlen_t n);
SEXP mock_r_int(int v);
SEXP mock_r_real(double v);
SEXP mock_r_lgl(int v);
void mock_r_reset(void);
#endif
