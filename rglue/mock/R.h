/* mock of <R.h> for rglue/src/icnv_shim.c -- see Rinternals.h in this directory */
#ifndef ICNV_MOCK_R_H
#define ICNV_MOCK_R_H
#include <stdlib.h>
#include <string.h>
#endif
