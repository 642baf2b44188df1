/* Test double: declarations of the handful of R C-API entry points flgp_rcall.c uses, written
 * from R's documented API ("Writing R Extensions", sections 5.9-5.10, 6.12) so that the shim can be
 * type-checked AND executed (rmock.c in this directory implements them) in an image without R.
 * NOT R's header. */
#ifndef FLGP_R_MOCK_RINTERNALS_H
#define FLGP_R_MOCK_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int Rboolean;
#define FALSE 0
#define TRUE 1
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NilValue, R_NamesSymbol, R_GlobalEnv;
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
void Rf_error(const char *, ...) __attribute__((noreturn));
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
R_xlen_t Rf_xlength(SEXP);
int Rf_length(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
const char *CHAR(SEXP);
Rboolean Rf_isString(SEXP);
Rboolean Rf_isMatrix(SEXP);
SEXP Rf_coerceVector(SEXP, unsigned int);
SEXP Rf_allocVector(unsigned int, R_xlen_t);
SEXP Rf_allocMatrix(unsigned int, int, int);
int *INTEGER(SEXP);
double *REAL(SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
double Rf_asReal(SEXP);
SEXP Rf_install(const char *);
SEXP Rf_mkString(const char *);
SEXP Rf_mkChar(const char *);
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarReal(double);
SEXP Rf_cons(SEXP, SEXP);
SEXP Rf_lcons(SEXP, SEXP);
void SET_TAG(SEXP, SEXP);
SEXP Rf_findFun(SEXP, SEXP);
SEXP Rf_eval(SEXP, SEXP);
SEXP R_FindNamespace(SEXP);
SEXP R_do_MAKE_CLASS(const char *);
SEXP R_do_new_object(SEXP);
SEXP R_do_slot_assign(SEXP, SEXP, SEXP);
char *R_alloc(size_t, int);
void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);            /* R_ext/Random.h (pulled in by R.h) */
/* unwind protection (R >= 3.5.0, "Writing R Extensions" 6.12): cleanfun runs whether fun returns or jumps */
SEXP R_MakeUnwindCont(void);
void R_ContinueUnwind(SEXP cont) __attribute__((noreturn));
SEXP R_UnwindProtect(SEXP (*fun)(void *data), void *data, void (*cleanfun)(void *data, Rboolean jump), void *cleandata, SEXP cont);
#endif
