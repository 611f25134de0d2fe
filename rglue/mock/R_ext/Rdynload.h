/* mock of <R_ext/Rdynload.h> for rglue/src/icnv_shim.c -- see ../Rinternals.h (real R declares the registration types
 * here, not in Rinternals.h: the shim has to include this header by name) */
#ifndef ICNV_MOCK_RDYNLOAD_H
#define ICNV_MOCK_RDYNLOAD_H
#include <Rinternals.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct { const char *name; DL_FUNC fun; int numArgs; void *types; } R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct mock_dllinfo DllInfo;
int R_registerRoutines(DllInfo *info, const R_CMethodDef *const c, const R_CallMethodDef *const call,
                       const R_FortranMethodDef *const f, const R_ExternalMethodDef *const ext);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);

#ifdef __cplusplus
}
#endif
#endif
