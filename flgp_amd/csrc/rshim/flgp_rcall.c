/*
 * flgp_rcall.c -- the R `.Call` shim: registers FLGP's own `_FLGP_*` entry points for the
 * graph-Laplacian / heat-kernel covariance path (same names, same arities as the table in the
 * reference's src/RcppExports.cpp:471-499) and forwards them to the C ABI of libflgp_hip.so
 * (include/flgp_hip.h).  With this object in place of the Rcpp bodies, R/RcppExports.R and
 * R/Fit.R:760-770 work unmodified.
 *
 * Build (where R is installed; it is not in the development image, so this file is only
 * syntax-checked there against tests/r_mock/):
 *     R CMD SHLIB -o FLGPhip.so flgp_rcall.c -I../../../include -L.. -lflgp_hip
 * and load it from the package with useDynLib(FLGPhip, .registration = TRUE), or link it into
 * FLGP.so next to the untouched fit_* drivers.
 *
 * Division of labour, as in the reference:
 *   - R owns its objects; inputs are read in place (REAL()/INTEGER() are exactly the column-
 *     major buffers the C ABI wants), outputs are freshly allocated R objects;
 *   - subsampling (subsample_cpp, src/Utils.cpp:32-68) stays in R ("lloyd" is an added device method): stats::kmeans,
 *     ClusterR::MiniBatchKmeans and sample() are called back from here on the main R thread;
 *   - errors: the C ABI returns a status and a message; this file raises them with Rf_error
 *     (the reference: Rcpp::stop -> R condition).  Only R-managed memory (PROTECT / R_alloc) is
 *     live at that point, so the longjmp leaks nothing.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdlib.h>
#include <string.h>

#include "flgp_hip.h"

static void chk(int rc) {
  if (rc != FLGP_OK) Rf_error("%s", flgp_last_error());
}

static SEXP list_get(SEXP list, const char *name) {
  SEXP names = Rf_getAttrib(list, R_NamesSymbol);
  for (R_xlen_t i = 0; i < Rf_xlength(list); ++i)
    if (names != R_NilValue && strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return VECTOR_ELT(list, i);
  Rf_error("models$%s is missing", name);
  return R_NilValue;
}

static const char *as_cstr(SEXP s) {
  if (!Rf_isString(s) || Rf_length(s) < 1) Rf_error("expected a character string");
  return CHAR(STRING_ELT(s, 0));
}

static SEXP as_real_matrix(SEXP x, const char *what) {
  if (!Rf_isMatrix(x)) Rf_error("%s must be a numeric matrix", what);
  return Rf_coerceVector(x, REALSXP); /* no copy when already double */
}

/* Matrix::dgRMatrix from the CSR triplet the C ABI fills (what Rcpp::wrap of an
 * Eigen::SparseMatrix<double,RowMajor> builds in the reference) */
static SEXP make_dgR(int n, int s, SEXP p, SEXP j, SEXP x) {
  SEXP cls = PROTECT(R_do_MAKE_CLASS("dgRMatrix"));
  SEXP obj = PROTECT(R_do_new_object(cls));
  SEXP dim = PROTECT(Rf_allocVector(INTSXP, 2));
  INTEGER(dim)[0] = n;
  INTEGER(dim)[1] = s;
  R_do_slot_assign(obj, Rf_install("p"), p);
  R_do_slot_assign(obj, Rf_install("j"), j);
  R_do_slot_assign(obj, Rf_install("x"), x);
  R_do_slot_assign(obj, Rf_install("Dim"), dim);
  UNPROTECT(3);
  return obj;
}

/* subsample_cpp (src/Utils.cpp:32-68): all three methods are R functions; returns s x d
 * ("random") or s x (d+1) with the cluster sizes in the last column. */
static SEXP call_in_ns(const char *pkg, const char *fun, SEXP args_pairlist) {
  SEXP ns = PROTECT(R_FindNamespace(Rf_mkString(pkg)));
  SEXP f = PROTECT(Rf_findFun(Rf_install(fun), ns));
  SEXP call = PROTECT(Rf_lcons(f, args_pairlist));
  SEXP res = Rf_eval(call, R_GlobalEnv);
  UNPROTECT(3);
  return res;
}

static SEXP tagged(SEXP head, const char *tag, SEXP value) { /* prepend tag=value to a pairlist */
  SEXP cell = PROTECT(Rf_cons(value, head));
  SET_TAG(cell, Rf_install(tag));
  UNPROTECT(1);
  return cell;
}

static SEXP subsample(SEXP X, int s, const char *method, int nstart) {
  const int n = Rf_nrows(X), d = Rf_ncols(X);
  if (strcmp(method, "kmeans") == 0) {
    SEXP args = R_NilValue;
    args = PROTECT(tagged(args, "nstart", Rf_ScalarInteger(nstart)));
    args = PROTECT(tagged(args, "iter.max", Rf_ScalarInteger(100)));
    args = PROTECT(tagged(args, "centers", Rf_ScalarInteger(s)));
    args = PROTECT(tagged(args, "x", X));
    SEXP km = PROTECT(call_in_ns("stats", "kmeans", args));
    SEXP centers = PROTECT(Rf_coerceVector(list_get(km, "centers"), REALSXP));
    SEXP size = PROTECT(Rf_coerceVector(list_get(km, "size"), REALSXP));
    SEXP U = PROTECT(Rf_allocMatrix(REALSXP, s, d + 1));
    memcpy(REAL(U), REAL(centers), sizeof(double) * (size_t)s * d);
    memcpy(REAL(U) + (size_t)s * d, REAL(size), sizeof(double) * (size_t)s);
    UNPROTECT(8);
    return U;
  }
  if (strcmp(method, "random") == 0) {
    SEXP args = R_NilValue;
    args = PROTECT(Rf_cons(Rf_ScalarInteger(s), args));
    args = PROTECT(Rf_cons(Rf_ScalarInteger(n), args));
    SEXP rows = PROTECT(Rf_coerceVector(call_in_ns("base", "sample", args), INTSXP)); /* Rcpp::sample(n, s) */
    SEXP U = PROTECT(Rf_allocMatrix(REALSXP, s, d));
    for (int k = 0; k < d; ++k)
      for (int i = 0; i < s; ++i) REAL(U)[(size_t)k * s + i] = REAL(X)[(size_t)k * n + (INTEGER(rows)[i] - 1)];
    UNPROTECT(4);
    return U;
  }
  if (strcmp(method, "minibatchkmeans") == 0) {
    SEXP args = R_NilValue;
    args = PROTECT(tagged(args, "num_init", Rf_ScalarInteger(nstart)));
    args = PROTECT(tagged(args, "init_fraction", Rf_ScalarReal(s * 20.0 / n)));
    args = PROTECT(tagged(args, "batch_size", Rf_ScalarInteger(s * 10)));
    args = PROTECT(tagged(args, "clusters", Rf_ScalarInteger(s)));
    args = PROTECT(tagged(args, "data", X));
    SEXP mb = PROTECT(call_in_ns("ClusterR", "MiniBatchKmeans", args));
    SEXP cen = PROTECT(Rf_coerceVector(list_get(mb, "centroids"), REALSXP));
    SEXP U = PROTECT(Rf_allocMatrix(REALSXP, s, d + 1));
    memcpy(REAL(U), REAL(cen), sizeof(double) * (size_t)s * d);
    /* cluster sizes = 1-NN counts (src/Utils.cpp:59-62): one pass of the k-NN kernel with r = 1 */
    int *lab = (int *)R_alloc((size_t)n, sizeof(int));
    chk(flgp_knn(REAL(X), n, d, REAL(cen), s, 1, "Euclidean", lab, NULL));
    double *cnt = REAL(U) + (size_t)s * d;
    for (int i = 0; i < s; ++i) cnt[i] = 0.0;
    for (int i = 0; i < n; ++i) cnt[lab[i]] += 1.0;
    UNPROTECT(8);
    return U;
  }
  if (strcmp(method, "lloyd") == 0) {
    /* extension: Lloyd k-means on the device, nstart starts drawn with base::sample as stats::kmeans
     * draws its own; same s x (d+1) result as "kmeans", iter.max = 100 (src/Utils.cpp:41) */
    if (nstart < 1) nstart = 1;
    int *rows = (int *)R_alloc((size_t)nstart * s, sizeof(int));
    for (int j = 0; j < nstart; ++j) {
      SEXP args = R_NilValue;
      args = PROTECT(Rf_cons(Rf_ScalarInteger(s), args));
      args = PROTECT(Rf_cons(Rf_ScalarInteger(n), args));
      SEXP draw = PROTECT(Rf_coerceVector(call_in_ns("base", "sample", args), INTSXP));
      for (int i = 0; i < s; ++i) rows[(size_t)j * s + i] = INTEGER(draw)[i] - 1;
      UNPROTECT(3);
    }
    SEXP U = PROTECT(Rf_allocMatrix(REALSXP, s, d + 1));
    chk(flgp_kmeans_lloyd(REAL(X), n, d, s, rows, nstart, 100, REAL(U), NULL, NULL));
    UNPROTECT(1);
    return U;
  }
  if (strcmp(method, "minibatch") == 0) {
    /* extension: the algorithm and arguments of the "minibatchkmeans" branch above (src/Utils.cpp:49-62) on
     * the device instead of through ClusterR; the seed is drawn from R's RNG (the caller holds GetRNGstate) */
    if (nstart < 1) nstart = 1;
    const unsigned long long seed = (unsigned long long)(unif_rand() * 9007199254740992.0);
    SEXP U = PROTECT(Rf_allocMatrix(REALSXP, s, d + 1));
    chk(flgp_kmeans_minibatch(REAL(X), n, d, s, -1, nstart, 100, -1.0, 10, seed, REAL(U), NULL, NULL));
    UNPROTECT(1);
    return U;
  }
  Rf_error("The subsample method is not supported!"); /* src/Utils.cpp:64 */
  return R_NilValue;
}

/* Rcpp::RNGScope (src/RcppExports.cpp:19): GetRNGstate() ... PutRNGstate() around everything that may draw from R's RNG.
 * subsample() can leave by longjmp (Rf_error here, or an R-level error inside stats::kmeans / ClusterR), and Rcpp's scope
 * object is unwound in that case; in C the same is R_UnwindProtect, whose clean-up function runs on both ways out. */
struct sub_args { SEXP X; int s; const char *method; int nstart; };
static SEXP sub_body(void *p) {
  struct sub_args *a = (struct sub_args *)p;
  return subsample(a->X, a->s, a->method, a->nstart);
}
static void sub_clean(void *p, Rboolean jump) { (void)p; (void)jump; PutRNGstate(); }
static SEXP subsample_rng(SEXP X, int s, const char *method, int nstart) {
  struct sub_args a = {X, s, method, nstart};
  SEXP cont = PROTECT(R_MakeUnwindCont());
  GetRNGstate();
  SEXP U = R_UnwindProtect(sub_body, &a, sub_clean, NULL, cont);
  UNPROTECT(1);
  return U;
}

/* FLGP_DEVICES="0,1,2,3" in the environment of the R session: the GPUs the rows are sharded over (one host thread each
 * inside the library, RCCL over xGMI between them); a single id selects that GPU; unset: the current device.  A list that
 * is not a comma-separated sequence of non-negative integers is an error, not a silent fall-back to one GPU. */
static int devices_from_env(int *devs, int cap) {
  const char *env = getenv("FLGP_DEVICES");
  int ndev = 0;
  if (!env || !*env) return 0;
  for (;;) {
    char *end = NULL;
    const long v = strtol(env, &end, 10);
    if (end == env || v < 0 || v > 4096) Rf_error("FLGP_DEVICES=\"%s\" is not a comma-separated list of device numbers", getenv("FLGP_DEVICES"));
    if (ndev == cap) Rf_error("FLGP_DEVICES lists more than %d devices", cap);
    devs[ndev++] = (int)v;
    if (*end == 0) break;
    if (*end != ',') Rf_error("FLGP_DEVICES=\"%s\" is not a comma-separated list of device numbers", getenv("FLGP_DEVICES"));
    env = end + 1;
  }
  return ndev;
}

static void sort_row(int *j, double *x, int r) { /* ascending column index inside a row */
  for (int a = 1; a < r; ++a) {
    int tj = j[a];
    double tx = x ? x[a] : 0.0;
    int p = a;
    while (p > 0 && j[p - 1] > tj) { j[p] = j[p - 1]; if (x) x[p] = x[p - 1]; --p; }
    j[p] = tj;
    if (x) x[p] = tx;
  }
}

/* ---- _FLGP_subsample_cpp (4 args, src/RcppExports.cpp:361-372) ---- */
SEXP FLGP_subsample_cpp(SEXP XS, SEXP sS, SEXP methodS, SEXP nstartS) {
  SEXP X = PROTECT(as_real_matrix(XS, "X"));
  SEXP U = subsample_rng(X, Rf_asInteger(sS), as_cstr(methodS), Rf_asInteger(nstartS));
  UNPROTECT(1);
  return U;
}

/* ---- _FLGP_KNN_cpp (6 args, src/RcppExports.cpp:375-388) ---- */
SEXP FLGP_KNN_cpp(SEXP XS, SEXP US, SEXP rS, SEXP distanceS, SEXP outputS, SEXP batchS) {
  (void)batchS; /* no numerical effect in the reference either */
  SEXP X = PROTECT(as_real_matrix(XS, "X"));
  SEXP U = PROTECT(as_real_matrix(US, "U"));
  const int n = Rf_nrows(X), d = Rf_ncols(X), s = Rf_nrows(U), r = Rf_asInteger(rS);
  const int output = Rf_asLogical(outputS);
  if (Rf_ncols(U) != d) Rf_error("X and U must have the same number of columns");
  SEXP ind = PROTECT(Rf_allocMatrix(INTSXP, n, r));
  double *dist = output ? (double *)R_alloc((size_t)n * r, sizeof(double)) : NULL;
  chk(flgp_knn(REAL(X), n, d, REAL(U), s, r, as_cstr(distanceS), INTEGER(ind), dist));
  SEXP res = PROTECT(Rf_allocVector(VECSXP, output ? 2 : 1));
  SEXP names = PROTECT(Rf_allocVector(STRSXP, output ? 2 : 1));
  SET_VECTOR_ELT(res, 0, ind);
  SET_STRING_ELT(names, 0, Rf_mkChar("ind_knn"));
  if (output) {
    SEXP p = PROTECT(Rf_allocVector(INTSXP, n + 1));
    SEXP j = PROTECT(Rf_allocVector(INTSXP, (R_xlen_t)n * r));
    SEXP x = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)n * r));
    for (int i = 0; i <= n; ++i) INTEGER(p)[i] = i * r;
    for (int i = 0; i < n; ++i) {
      for (int a = 0; a < r; ++a) {
        INTEGER(j)[(size_t)i * r + a] = INTEGER(ind)[(size_t)a * n + i];
        REAL(x)[(size_t)i * r + a] = dist[(size_t)a * n + i];
      }
      sort_row(INTEGER(j) + (size_t)i * r, REAL(x) + (size_t)i * r, r);
    }
    SET_VECTOR_ELT(res, 1, make_dgR(n, s, p, j, x));
    SET_STRING_ELT(names, 1, Rf_mkChar("distances_sp"));
    UNPROTECT(3);
  }
  Rf_setAttrib(res, R_NamesSymbol, names);
  UNPROTECT(5);
  return res;
}

static SEXP similarity_out(int n, int s, int r, SEXP *p, SEXP *j, SEXP *x) {
  *p = PROTECT(Rf_allocVector(INTSXP, n + 1));
  *j = PROTECT(Rf_allocVector(INTSXP, (R_xlen_t)n * r));
  *x = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)n * r));
  (void)s;
  return R_NilValue;
}

/* ---- _FLGP_LAE_cpp (3 args, src/RcppExports.cpp:421-431) ---- */
SEXP FLGP_LAE_cpp(SEXP XS, SEXP US, SEXP rS) {
  SEXP X = PROTECT(as_real_matrix(XS, "X"));
  SEXP U = PROTECT(as_real_matrix(US, "U"));
  const int n = Rf_nrows(X), d = Rf_ncols(X), s = Rf_nrows(U), r = Rf_asInteger(rS);
  if (Rf_ncols(U) != d) Rf_error("X and U must have the same number of columns");
  SEXP p, j, x;
  similarity_out(n, s, r, &p, &j, &x);
  chk(flgp_lae(REAL(X), n, d, REAL(U), s, r, INTEGER(p), INTEGER(j), REAL(x)));
  SEXP Z = make_dgR(n, s, p, j, x);
  UNPROTECT(5);
  return Z;
}

/* ---- _FLGP_cross_similarity_lae_cpp (4 args, src/RcppExports.cpp:347-358) ---- */
SEXP FLGP_cross_similarity_lae_cpp(SEXP XS, SEXP US, SEXP rS, SEXP glS) {
  SEXP X = PROTECT(as_real_matrix(XS, "X"));
  SEXP U = PROTECT(as_real_matrix(US, "U"));
  const int n = Rf_nrows(X), d = Rf_ncols(X), s = Rf_nrows(U), r = Rf_asInteger(rS);
  SEXP p, j, x;
  similarity_out(n, s, r, &p, &j, &x);
  chk(flgp_cross_similarity_lae(REAL(X), n, d, REAL(U), s, Rf_ncols(U), r, as_cstr(glS), INTEGER(p), INTEGER(j), REAL(x)));
  SEXP Z = make_dgR(n, s, p, j, x);
  UNPROTECT(5);
  return Z;
}

/* ---- _FLGP_local_anchor_embedding_cpp (2 args, src/RcppExports.cpp:434-443) ---- */
SEXP FLGP_local_anchor_embedding_cpp(SEXP xS, SEXP US) {
  SEXP x = PROTECT(Rf_coerceVector(xS, REALSXP));
  SEXP U = PROTECT(as_real_matrix(US, "U"));
  const int r = Rf_nrows(U), d = Rf_ncols(U);
  if (Rf_length(x) != d) Rf_error("x and U must have the same dimension");
  SEXP z = PROTECT(Rf_allocMatrix(REALSXP, 1, r)); /* Eigen::RowVectorXd -> 1 x r matrix */
  chk(flgp_local_anchor_embedding(REAL(x), d, REAL(U), r, REAL(z)));
  UNPROTECT(3);
  return z;
}

/* ---- _FLGP_v_to_z_cpp (1 arg, src/RcppExports.cpp:446-454) ---- */
SEXP FLGP_v_to_z_cpp(SEXP vS) {
  SEXP v = PROTECT(Rf_coerceVector(vS, REALSXP));
  const int r = Rf_length(v);
  SEXP z = PROTECT(Rf_allocMatrix(REALSXP, 1, r));
  chk(flgp_v_to_z(REAL(v), r, REAL(z)));
  UNPROTECT(2);
  return z;
}

/* ---- _FLGP_heat_kernel_covariance_cpp (9 args, src/RcppExports.cpp:328-344) ---- */
SEXP FLGP_heat_kernel_covariance_cpp(SEXP XS, SEXP XnewS, SEXP sS, SEXP rS, SEXP tS, SEXP KS, SEXP modelsS,
                                     SEXP nstartS, SEXP epsilonS) {
  SEXP X = PROTECT(as_real_matrix(XS, "X"));
  SEXP Xnew = PROTECT(as_real_matrix(XnewS, "X_new"));
  const int m = Rf_nrows(X), mnew = Rf_nrows(Xnew), d = Rf_ncols(X), n = m + mnew;
  const int s = Rf_asInteger(sS), r = Rf_asInteger(rS);
  int K = Rf_asInteger(KS);
  if (Rf_ncols(Xnew) != d) Rf_error("X and X_new must have the same number of columns");
  if (K < 0) K = s; /* src/Spectrum.cpp:31-33 */
  /* X_all = [X; X_new]  (src/Spectrum.cpp:50-53) */
  SEXP Xall = PROTECT(Rf_allocMatrix(REALSXP, n, d));
  for (int k = 0; k < d; ++k) {
    memcpy(REAL(Xall) + (size_t)k * n, REAL(X) + (size_t)k * m, sizeof(double) * (size_t)m);
    memcpy(REAL(Xall) + (size_t)k * n + m, REAL(Xnew) + (size_t)k * mnew, sizeof(double) * (size_t)mnew);
  }
  int devs[16];
  const int ndev = devices_from_env(devs, 16);       /* (before anything is drawn or computed: a malformed list is an error) */
  SEXP U = PROTECT(subsample_rng(Xall, s, as_cstr(list_get(modelsS, "subsample")), Rf_asInteger(nstartS)));
  SEXP H = PROTECT(Rf_allocMatrix(REALSXP, n, m));
  if (ndev >= 1)       /* (one id: the library switches to that GPU for the call and back) */
    chk(flgp_heat_kernel_covariance_multi(REAL(Xall), n, m, d, REAL(U), s, Rf_ncols(U), r, Rf_asReal(tS), K,
                                          as_cstr(list_get(modelsS, "kernel")), as_cstr(list_get(modelsS, "gl")),
                                          Rf_asLogical(list_get(modelsS, "root")), Rf_asReal(epsilonS), ndev, devs, REAL(H)));
  else
    chk(flgp_heat_kernel_covariance(REAL(Xall), n, m, d, REAL(U), s, Rf_ncols(U), r, Rf_asReal(tS), K,
                                    as_cstr(list_get(modelsS, "kernel")), as_cstr(list_get(modelsS, "gl")),
                                    Rf_asLogical(list_get(modelsS, "root")), Rf_asReal(epsilonS), REAL(H)));
  UNPROTECT(5);
  return H;
}

/* ---- _FLGP_lae_eigenmap (7 args, src/RcppExports.cpp:311-325) ---- */
SEXP FLGP_lae_eigenmap(SEXP XS, SEXP sS, SEXP rS, SEXP ndimS, SEXP subsampleS, SEXP normS, SEXP nstartS) {
  SEXP X = PROTECT(as_real_matrix(XS, "X"));
  const int n = Rf_nrows(X), d = Rf_ncols(X), s = Rf_asInteger(sS), ndim = Rf_asInteger(ndimS);
  SEXP U = PROTECT(subsample_rng(X, s, as_cstr(subsampleS), Rf_asInteger(nstartS)));
  SEXP ev = PROTECT(Rf_allocVector(REALSXP, ndim));
  SEXP vec = PROTECT(Rf_allocMatrix(REALSXP, n, ndim));
  chk(flgp_lae_eigenmap(REAL(X), n, d, REAL(U), s, Rf_ncols(U), Rf_asInteger(rS), ndim, as_cstr(normS), REAL(ev), REAL(vec)));
  SEXP res = PROTECT(Rf_allocVector(VECSXP, 2));
  SEXP names = PROTECT(Rf_allocVector(STRSXP, 2));
  SET_VECTOR_ELT(res, 0, ev);
  SET_VECTOR_ELT(res, 1, vec);
  SET_STRING_ELT(names, 0, Rf_mkChar("eigenvalues"));
  SET_STRING_ELT(names, 1, Rf_mkChar("eigenvectors"));
  Rf_setAttrib(res, R_NamesSymbol, names);
  UNPROTECT(6);
  return res;
}

/* same names and arities as the reference's CallEntries[] (src/RcppExports.cpp:471-499) */
static const R_CallMethodDef CallEntries[] = {
    {"_FLGP_lae_eigenmap", (DL_FUNC)&FLGP_lae_eigenmap, 7},
    {"_FLGP_heat_kernel_covariance_cpp", (DL_FUNC)&FLGP_heat_kernel_covariance_cpp, 9},
    {"_FLGP_cross_similarity_lae_cpp", (DL_FUNC)&FLGP_cross_similarity_lae_cpp, 4},
    {"_FLGP_subsample_cpp", (DL_FUNC)&FLGP_subsample_cpp, 4},
    {"_FLGP_KNN_cpp", (DL_FUNC)&FLGP_KNN_cpp, 6},
    {"_FLGP_LAE_cpp", (DL_FUNC)&FLGP_LAE_cpp, 3},
    {"_FLGP_local_anchor_embedding_cpp", (DL_FUNC)&FLGP_local_anchor_embedding_cpp, 2},
    {"_FLGP_v_to_z_cpp", (DL_FUNC)&FLGP_v_to_z_cpp, 1},
    {NULL, NULL, 0}};

void R_init_FLGPhip(DllInfo *dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
