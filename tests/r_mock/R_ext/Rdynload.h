/* Test double, see ../Rinternals.h. */
#ifndef FLGP_R_MOCK_RDYNLOAD_H
#define FLGP_R_MOCK_RDYNLOAD_H
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *, const void *, const R_CallMethodDef *, const void *, const void *);
int R_useDynamicSymbols(DllInfo *, int);
#endif
