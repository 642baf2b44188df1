/* Functional test double of the few R C-API entry points flgp_rcall.c uses (tests/r_mock/Rinternals.h), written from
 * R's documented API ("Writing R Extensions", sections 5.9-5.10, 6.12) so that the `.Call` shim can be EXECUTED in an
 * image without R: tagged heap objects, attributes (names, dim), pairlists and calls, S4 slots, a PROTECT counter,
 * Rf_error as a longjmp to the innermost handler (the harness' rmock_call or an R_UnwindProtect frame), the RNG-state
 * bracket as a counter, and "R functions" (base::sample, stats::kmeans ...) as C callbacks the test registers.
 * NOT R; nothing of R's source is used.  Objects live until rmock_reset(). */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "Rinternals.h"
#include "R_ext/Rdynload.h"

enum { T_NIL = 0, T_SYM = 1, T_LIST = 2, T_CLO = 3, T_LANG = 6, T_CHAR = 9, T_LGL = 10, T_S4 = 25, T_CONT = 98, T_CLASS = 99 };

struct SEXPREC {
  int type;
  R_xlen_t length;
  void *data;
  SEXP names, dim;
  SEXP car, cdr, tag;
  struct { SEXP sym, val; } slots[8];
  int nslots;
  const char *klass;
  SEXP (*fn)(SEXP);
  struct SEXPREC *next_alloc;
};

static struct SEXPREC nil_obj = {T_NIL, 0, NULL, NULL, NULL, NULL, NULL, NULL, {{NULL, NULL}}, 0, NULL, NULL, NULL};
static struct SEXPREC names_sym = {T_SYM, 0, (void *)"names", NULL, NULL, NULL, NULL, NULL, {{NULL, NULL}}, 0, NULL, NULL, NULL};
static struct SEXPREC global_env = {T_NIL, 0, NULL, NULL, NULL, NULL, NULL, NULL, {{NULL, NULL}}, 0, NULL, NULL, NULL};
SEXP R_NilValue = &nil_obj, R_NamesSymbol = &names_sym, R_GlobalEnv = &global_env;

static struct SEXPREC *all_objs = NULL;
static int protect_depth = 0, protect_max = 0, rng_get = 0, rng_put = 0, n_errors = 0;
static char last_error[1024];

/* ---- error handling: a stack of jump targets ---- */
#define MAX_HANDLERS 16
static jmp_buf *handlers[MAX_HANDLERS];
static int n_handlers = 0;

static SEXP new_obj(int type, R_xlen_t length, size_t elt) {
  struct SEXPREC *o = (struct SEXPREC *)calloc(1, sizeof(struct SEXPREC));
  o->type = type; o->length = length;
  o->names = o->dim = o->car = o->cdr = o->tag = R_NilValue;
  if (elt && length >= 0) o->data = calloc((size_t)(length > 0 ? length : 1), elt);
  o->next_alloc = all_objs; all_objs = o;
  return o;
}

void rmock_reset(void) {
  while (all_objs) { struct SEXPREC *n = all_objs->next_alloc; free(all_objs->data); free(all_objs); all_objs = n; }
  protect_depth = protect_max = rng_get = rng_put = n_errors = 0; n_handlers = 0; last_error[0] = 0;
}
int rmock_protect_depth(void) { return protect_depth; }
int rmock_protect_max(void) { return protect_max; }
int rmock_rng_gets(void) { return rng_get; }
int rmock_rng_puts(void) { return rng_put; }
const char *rmock_last_error(void) { return last_error; }

SEXP Rf_protect(SEXP s) { if (++protect_depth > protect_max) protect_max = protect_depth; return s; }
void Rf_unprotect(int n) { protect_depth -= n; if (protect_depth < 0) { fprintf(stderr, "rmock: PROTECT stack underflow\n"); abort(); } }

void Rf_error(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(last_error, sizeof last_error, fmt, ap); va_end(ap);
  ++n_errors;
  if (n_handlers == 0) { fprintf(stderr, "rmock: Rf_error outside rmock_call: %s\n", last_error); abort(); }
  longjmp(*handlers[n_handlers - 1], 1);
}

/* ---- vectors ---- */
SEXP Rf_allocVector(unsigned int type, R_xlen_t n) {
  switch (type) {
    case INTSXP: case T_LGL: return new_obj((int)type, n, sizeof(int));
    case REALSXP: return new_obj(REALSXP, n, sizeof(double));
    case STRSXP: case VECSXP: { SEXP o = new_obj((int)type, n, sizeof(SEXP)); for (R_xlen_t i = 0; i < n; ++i) ((SEXP *)o->data)[i] = R_NilValue; return o; }
    default: Rf_error("rmock: allocVector of type %u is not modelled", type);
  }
}
SEXP Rf_allocMatrix(unsigned int type, int nr, int nc) {
  SEXP o = Rf_allocVector(type, (R_xlen_t)nr * nc);
  SEXP d = Rf_allocVector(INTSXP, 2);
  ((int *)d->data)[0] = nr; ((int *)d->data)[1] = nc;
  o->dim = d;
  return o;
}
R_xlen_t Rf_xlength(SEXP s) { return s->type == T_LIST || s->type == T_LANG ? (s == R_NilValue ? 0 : 1 + Rf_xlength(s->cdr)) : s->length; }
int Rf_length(SEXP s) { return (int)Rf_xlength(s); }
int *INTEGER(SEXP s) { if (s->type != INTSXP && s->type != T_LGL) Rf_error("rmock: INTEGER() of a non-integer object (type %d)", s->type); return (int *)s->data; }
double *REAL(SEXP s) { if (s->type != REALSXP) Rf_error("rmock: REAL() of a non-double object (type %d)", s->type); return (double *)s->data; }
SEXP STRING_ELT(SEXP s, R_xlen_t i) { if (s->type != STRSXP || i < 0 || i >= s->length) Rf_error("rmock: STRING_ELT out of range"); return ((SEXP *)s->data)[i]; }
SEXP VECTOR_ELT(SEXP s, R_xlen_t i) { if (s->type != VECSXP || i < 0 || i >= s->length) Rf_error("rmock: VECTOR_ELT out of range"); return ((SEXP *)s->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP s, R_xlen_t i, SEXP v) { if (s->type != VECSXP || i < 0 || i >= s->length) Rf_error("rmock: SET_VECTOR_ELT out of range"); ((SEXP *)s->data)[i] = v; return v; }
void SET_STRING_ELT(SEXP s, R_xlen_t i, SEXP v) { if (s->type != STRSXP || i < 0 || i >= s->length || v->type != T_CHAR) Rf_error("rmock: SET_STRING_ELT misuse"); ((SEXP *)s->data)[i] = v; }
const char *CHAR(SEXP s) { if (s->type != T_CHAR) Rf_error("rmock: CHAR() of a non-CHARSXP"); return (const char *)s->data; }
SEXP Rf_mkChar(const char *c) { SEXP o = new_obj(T_CHAR, (R_xlen_t)strlen(c), 0); o->data = strdup(c); return o; }
SEXP Rf_mkString(const char *c) { SEXP o = Rf_allocVector(STRSXP, 1); SET_STRING_ELT(o, 0, Rf_mkChar(c)); return o; }
SEXP Rf_ScalarInteger(int v) { SEXP o = Rf_allocVector(INTSXP, 1); ((int *)o->data)[0] = v; return o; }
SEXP Rf_ScalarReal(double v) { SEXP o = Rf_allocVector(REALSXP, 1); ((double *)o->data)[0] = v; return o; }
Rboolean Rf_isString(SEXP s) { return s->type == STRSXP; }
Rboolean Rf_isMatrix(SEXP s) { return s->dim != R_NilValue && s->dim->length == 2; }
int Rf_nrows(SEXP s) { if (!Rf_isMatrix(s)) Rf_error("object is not a matrix"); return ((int *)s->dim->data)[0]; }
int Rf_ncols(SEXP s) { if (!Rf_isMatrix(s)) Rf_error("object is not a matrix"); return ((int *)s->dim->data)[1]; }
SEXP Rf_coerceVector(SEXP s, unsigned int type) {
  if ((unsigned)s->type == type) return s;
  if ((s->type == INTSXP || s->type == T_LGL) && type == REALSXP) {
    SEXP o = Rf_allocVector(REALSXP, s->length);
    for (R_xlen_t i = 0; i < s->length; ++i) ((double *)o->data)[i] = (double)((int *)s->data)[i];
    o->dim = s->dim; o->names = s->names;
    return o;
  }
  if (s->type == REALSXP && type == INTSXP) {
    SEXP o = Rf_allocVector(INTSXP, s->length);
    for (R_xlen_t i = 0; i < s->length; ++i) ((int *)o->data)[i] = (int)((double *)s->data)[i];
    o->dim = s->dim; o->names = s->names;
    return o;
  }
  Rf_error("rmock: coerceVector from type %d to %u is not modelled", s->type, type);
}
int Rf_asInteger(SEXP s) {
  if (s->length < 1) Rf_error("rmock: asInteger of an empty object");
  if (s->type == INTSXP || s->type == T_LGL) return ((int *)s->data)[0];
  if (s->type == REALSXP) return (int)((double *)s->data)[0];
  Rf_error("rmock: asInteger of type %d", s->type);
}
int Rf_asLogical(SEXP s) { return Rf_asInteger(s) != 0; }
double Rf_asReal(SEXP s) {
  if (s->length < 1) Rf_error("rmock: asReal of an empty object");
  if (s->type == REALSXP) return ((double *)s->data)[0];
  if (s->type == INTSXP || s->type == T_LGL) return (double)((int *)s->data)[0];
  Rf_error("rmock: asReal of type %d", s->type);
}

/* ---- symbols, attributes ---- */
SEXP Rf_install(const char *name) {
  if (!strcmp(name, "names")) return R_NamesSymbol;
  for (struct SEXPREC *o = all_objs; o; o = o->next_alloc)
    if (o->type == T_SYM && !strcmp((const char *)o->data, name)) return o;
  SEXP s = new_obj(T_SYM, 0, 0);
  s->data = strdup(name);
  return s;
}
SEXP Rf_getAttrib(SEXP s, SEXP sym) { if (sym == R_NamesSymbol) return s->names; if (!strcmp((const char *)sym->data, "dim")) return s->dim; return R_NilValue; }
SEXP Rf_setAttrib(SEXP s, SEXP sym, SEXP v) {
  if (sym == R_NamesSymbol) { if (v != R_NilValue && (v->type != STRSXP || v->length != Rf_xlength(s))) Rf_error("rmock: names of the wrong length"); s->names = v; }
  else if (!strcmp((const char *)sym->data, "dim")) s->dim = v;
  else Rf_error("rmock: attribute %s is not modelled", (const char *)sym->data);
  return v;
}

/* ---- pairlists, calls, "functions" ---- */
SEXP Rf_cons(SEXP car, SEXP cdr) { SEXP o = new_obj(T_LIST, 0, 0); o->car = car; o->cdr = cdr; return o; }
SEXP Rf_lcons(SEXP car, SEXP cdr) { SEXP o = Rf_cons(car, cdr); o->type = T_LANG; return o; }
void SET_TAG(SEXP cell, SEXP tag) { cell->tag = tag; }

struct fn_entry { char pkg[32], name[48]; SEXP (*fn)(SEXP); };
static struct fn_entry fns[16];
static int n_fns = 0;
void rmock_register_function(const char *pkg, const char *name, SEXP (*fn)(SEXP)) {
  for (int i = 0; i < n_fns; ++i)
    if (!strcmp(fns[i].pkg, pkg) && !strcmp(fns[i].name, name)) { fns[i].fn = fn; return; }
  if (n_fns == 16) abort();
  snprintf(fns[n_fns].pkg, sizeof fns[n_fns].pkg, "%s", pkg); snprintf(fns[n_fns].name, sizeof fns[n_fns].name, "%s", name);
  fns[n_fns++].fn = fn;
}
SEXP R_FindNamespace(SEXP name) { SEXP ns = new_obj(T_NIL, 0, 0); ns->klass = strdup(CHAR(STRING_ELT(name, 0))); ns->data = NULL; return ns; }
SEXP Rf_findFun(SEXP sym, SEXP ns) {
  for (int i = 0; i < n_fns; ++i)
    if (!strcmp(fns[i].name, (const char *)sym->data) && ns->klass && !strcmp(fns[i].pkg, ns->klass)) { SEXP f = new_obj(T_CLO, 0, 0); f->fn = fns[i].fn; return f; }
  Rf_error("could not find function \"%s\" in namespace %s (rmock: not registered)", (const char *)sym->data, ns->klass ? ns->klass : "?");
}
SEXP Rf_eval(SEXP call, SEXP env) {
  (void)env;
  if (call->type != T_LANG || call->car->type != T_CLO) Rf_error("rmock: eval of something that is not a call of a registered function");
  return call->car->fn(call->cdr);
}
/* the k-th argument of a call (0-based) or the one tagged `name` */
SEXP rmock_arg(SEXP args, const char *name, int k) {
  int i = 0;
  for (SEXP c = args; c != R_NilValue; c = c->cdr, ++i) {
    if (name && c->tag != R_NilValue && !strcmp((const char *)c->tag->data, name)) return c->car;
    if (!name && i == k) return c->car;
  }
  return R_NilValue;
}

/* ---- S4 objects (Matrix::dgRMatrix) ---- */
SEXP R_do_MAKE_CLASS(const char *what) { SEXP c = new_obj(T_CLASS, 0, 0); c->klass = strdup(what); return c; }
SEXP R_do_new_object(SEXP cls) { if (cls->type != T_CLASS) Rf_error("rmock: new_object of a non-class"); SEXP o = new_obj(T_S4, 0, 0); o->klass = cls->klass; return o; }
SEXP R_do_slot_assign(SEXP obj, SEXP sym, SEXP val) {
  if (obj->type != T_S4) Rf_error("rmock: slot assignment on a non-S4 object");
  for (int i = 0; i < obj->nslots; ++i) if (obj->slots[i].sym == sym) { obj->slots[i].val = val; return obj; }
  if (obj->nslots == 8) Rf_error("rmock: too many slots");
  obj->slots[obj->nslots].sym = sym; obj->slots[obj->nslots++].val = val;
  return obj;
}
SEXP rmock_slot(SEXP obj, const char *name) {
  if (obj->type != T_S4) return NULL;
  for (int i = 0; i < obj->nslots; ++i) if (!strcmp((const char *)obj->slots[i].sym->data, name)) return obj->slots[i].val;
  return NULL;
}
const char *rmock_class(SEXP obj) { return obj->klass ? obj->klass : ""; }
int rmock_type(SEXP obj) { return obj->type; }

/* ---- memory, RNG ---- */
char *R_alloc(size_t n, int size) { SEXP o = new_obj(T_NIL, 0, 0); o->data = calloc(n ? n : 1, (size_t)size); return (char *)o->data; }
void GetRNGstate(void) { ++rng_get; }
void PutRNGstate(void) { ++rng_put; }
static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
void rmock_set_seed(unsigned long long s) { rng_state = s ? s : 1; }
double unif_rand(void) {
  if (rng_get <= rng_put) Rf_error("rmock: unif_rand outside a GetRNGstate / PutRNGstate bracket");
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0);
}

/* ---- unwind protection (R >= 3.5): cleanfun runs whether fun returns or jumps ---- */
SEXP R_MakeUnwindCont(void) { return new_obj(T_CONT, 0, 0); }
void R_ContinueUnwind(SEXP cont) { (void)cont; if (n_handlers == 0) abort(); longjmp(*handlers[n_handlers - 1], 1); }
SEXP R_UnwindProtect(SEXP (*fun)(void *), void *data, void (*cleanfun)(void *, Rboolean), void *cleandata, SEXP cont) {
  jmp_buf jb;
  if (n_handlers == MAX_HANDLERS) abort();
  const int depth = protect_depth;
  handlers[n_handlers++] = &jb;
  if (setjmp(jb)) {                       /* fun jumped: clean up, then go on unwinding to the next handler */
    --n_handlers;
    protect_depth = depth;
    if (cleanfun) cleanfun(cleandata, TRUE);
    R_ContinueUnwind(cont);
  }
  SEXP res = fun(data);
  --n_handlers;
  if (cleanfun) cleanfun(cleandata, FALSE);
  return res;
}

/* ---- registration (R_init_FLGPhip) ---- */
static const R_CallMethodDef *registered = NULL;
static int dynamic_symbols = -1;
int R_registerRoutines(DllInfo *dll, const void *c, const R_CallMethodDef *call, const void *f, const void *e) { (void)dll; (void)c; (void)f; (void)e; registered = call; return 1; }
int R_useDynamicSymbols(DllInfo *dll, int v) { (void)dll; dynamic_symbols = v; return 1; }
int rmock_dynamic_symbols(void) { return dynamic_symbols; }
/* look a routine up the way .Call does after registration: by name AND arity */
void *rmock_lookup(const char *name, int nargs) {
  for (const R_CallMethodDef *d = registered; d && d->name; ++d)
    if (!strcmp(d->name, name)) return d->numArgs == nargs ? (void *)d->fun : NULL;
  return NULL;
}

/* ---- the harness: .Call(fn, args...) with R's top-level error handling ---- */
typedef SEXP (*F1)(SEXP); typedef SEXP (*F2)(SEXP, SEXP); typedef SEXP (*F3)(SEXP, SEXP, SEXP); typedef SEXP (*F4)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP); typedef SEXP (*F7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*F9)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP rmock_call(void *fn, int nargs, SEXP *a) {
  jmp_buf jb;
  const int depth = protect_depth;
  last_error[0] = 0;
  handlers[n_handlers++] = &jb;
  if (setjmp(jb)) { --n_handlers; protect_depth = depth; return NULL; }     /* R resets the protect stack to the context's depth */
  SEXP res = NULL;
  switch (nargs) {
    case 1: res = ((F1)fn)(a[0]); break;
    case 2: res = ((F2)fn)(a[0], a[1]); break;
    case 3: res = ((F3)fn)(a[0], a[1], a[2]); break;
    case 4: res = ((F4)fn)(a[0], a[1], a[2], a[3]); break;
    case 6: res = ((F6)fn)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
    case 7: res = ((F7)fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
    case 9: res = ((F9)fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
    default: snprintf(last_error, sizeof last_error, "rmock_call: %d arguments are not modelled", nargs); res = NULL;
  }
  --n_handlers;
  return res;
}

/* ---- constructors for the Python side ---- */
SEXP rmock_real_matrix(const double *v, int nr, int nc) { SEXP o = Rf_allocMatrix(REALSXP, nr, nc); memcpy(o->data, v, sizeof(double) * (size_t)nr * nc); return o; }
SEXP rmock_int_matrix(const int *v, int nr, int nc) { SEXP o = Rf_allocMatrix(INTSXP, nr, nc); memcpy(o->data, v, sizeof(int) * (size_t)nr * nc); return o; }
SEXP rmock_real_vector(const double *v, int n) { SEXP o = Rf_allocVector(REALSXP, n); memcpy(o->data, v, sizeof(double) * (size_t)n); return o; }
SEXP rmock_int_vector(const int *v, int n) { SEXP o = Rf_allocVector(INTSXP, n); memcpy(o->data, v, sizeof(int) * (size_t)n); return o; }
SEXP rmock_logical(int v) { SEXP o = Rf_allocVector(T_LGL, 1); ((int *)o->data)[0] = v; return o; }
SEXP rmock_named_list(int n, const char **names, SEXP *vals) {
  SEXP l = Rf_allocVector(VECSXP, n), nm = Rf_allocVector(STRSXP, n);
  for (int i = 0; i < n; ++i) { SET_VECTOR_ELT(l, i, vals[i]); SET_STRING_ELT(nm, i, Rf_mkChar(names[i])); }
  Rf_setAttrib(l, R_NamesSymbol, nm);
  return l;
}
SEXP rmock_list_get(SEXP l, const char *name) {
  if (l->type != VECSXP || l->names == R_NilValue) return NULL;
  for (R_xlen_t i = 0; i < l->length; ++i) if (!strcmp(CHAR(STRING_ELT(l->names, i)), name)) return VECTOR_ELT(l, i);
  return NULL;
}
void *rmock_data(SEXP s) { return s->data; }
long rmock_len(SEXP s) { return (long)Rf_xlength(s); }
