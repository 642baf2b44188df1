// flgp_cpp_adapters.cpp -- the reference's INTERNAL C++ entry points of the graph-Laplacian / heat-kernel path, with
// the reference's own signatures (Eigen / Rcpp types), implemented by forwarding to the C ABI of libflgp_hip.so.
//
// The twelve fit_* drivers of FLGP (src/Fit.cpp) and the training objectives (src/train.cpp, src/Predict.cpp) reach the
// path through plain C++ calls -- heat_kernel_spectrum_cpp (src/Fit.cpp:42,543,628), KNN_cpp / graphLaplacian_cpp /
// spectrum_from_Z_cpp (the SE bandwidth loop, src/Fit.cpp:127-158,694-723,820-850), HK_from_spectrum_cpp
// (src/Fit.cpp:84-85,...; src/train.cpp:17,30,363,471,687; src/Predict.cpp:49-85) -- not through `.Call`, so the R shim
// alone does not make them drop-in.  This file does: compile it INTO the package in place of the bodies it restates
//     src/Spectrum.cpp   (all of it: lae_eigenmap, heat_kernel_covariance_cpp, heat_kernel_spectrum_cpp,
//                         HK_from_spectrum_cpp, cross_similarity_lae_cpp, cross_similarity_se_cpp, spectrum_from_Z_cpp)
//     src/lae.cpp        (all of it: LAE_cpp, local_anchor_embedding_cpp, v_to_z_cpp)
//     (src/TruncatedSVD.cpp stays in the package untouched: truncated_SVD_cpp is called from spectrum_from_Z_cpp only,
//      src/Spectrum.cpp:155, so with the body above replaced nothing reaches it -- or RSpectra -- on this path any more)
//     src/Utils.cpp:72-212 (KNN_cpp, graphLaplacian_cpp; the rest of Utils.cpp -- subsample_cpp, the posterior
//                         formulas -- stays)
// and src/Fit.cpp, src/train.cpp, src/Predict.cpp, src/Multiclassification.cpp compile and run UNCHANGED against the
// reference's own headers (src/Spectrum.h:45-124, src/lae.h:34-60, src/Utils.h:35-62).  The generated glue
// (src/RcppExports.cpp) keeps working too, since it calls these same functions.  INTEGRATION.md lists the lines.
//
// Arithmetic: every function is one call of the C ABI (include/flgp_hip.h); Eigen objects are read through
// .data() / .rows() / .cols() (column-major doubles, exactly what the ABI takes) and results are written into Eigen
// objects of the reference's types.  Errors: the ABI's status + message become Rcpp::stop, as in the reference.
//
// Type-checked and RUN in the development image (no Eigen, no Rcpp there) against a functional test double of the few
// Eigen / Rcpp members used (tests/eigen_mock/, tests/c/adapters_check.cpp, tests/test_adapters.py).
#ifdef FLGP_ADAPTERS_TEST
#include "RcppEigen.h"        // tests/eigen_mock/: the test double
#include "ref_decls.h"        // tests/eigen_mock/: the reference's declarations, restated (interface only)
#else
#include <RcppEigen.h>
#include "Spectrum.h"
#include "Utils.h"
#include "lae.h"
#endif
#include <string>
#include <vector>

#include "flgp_hip.h"

namespace {

typedef Eigen::SparseMatrix<double, Eigen::RowMajor> SpR;

void chk(int rc) {
  if (rc != FLGP_OK) Rcpp::stop(flgp_last_error());
}

// A similarity matrix of the path has exactly r stored entries in every row (src/lae.cpp:60-67 inserts r per row,
// src/Utils.cpp:162-167 likewise, explicit zeros kept): the C ABI takes it as (column indices, values) of n x r.
int entries_per_row(const SpR &Z) {
  const long n = Z.rows(), nnz = Z.nonZeros();
  if (n <= 0 || nnz % n != 0) Rcpp::stop("the similarity matrix must hold the same number of entries in every row (k-NN / LAE output)");
  const int r = (int)(nnz / n);
  const int *p = Z.outerIndexPtr();
  for (long i = 0; i <= n; ++i)
    if (p[i] != (int)(i * r)) Rcpp::stop("the similarity matrix must be compressed with r entries per row (k-NN / LAE output)");
  return r;
}

SpR make_csr(int n, int s, int r) {       // n x s with n r slots, row pointers set
  SpR Z(n, s);
  Z.resizeNonZeros((long)n * r);
  int *p = Z.outerIndexPtr();
  for (int i = 0; i <= n; ++i) p[i] = i * r;
  return Z;
}

std::string str(const Rcpp::String &s) { return std::string(s.get_cstring()); }

}  // namespace

// ---- src/lae.cpp:137-153 ----
Eigen::RowVectorXd v_to_z_cpp(const Eigen::RowVectorXd &v) {
  Eigen::RowVectorXd z(v.size());
  chk(flgp_v_to_z(v.data(), (int)v.size(), z.data()));
  return z;
}

// ---- src/lae.cpp:76-133 ----
Eigen::RowVectorXd local_anchor_embedding_cpp(const Eigen::RowVectorXd &x, const Eigen::MatrixXd &U) {
  if (U.cols() != x.size()) Rcpp::stop("x and U must have the same dimension");
  Eigen::RowVectorXd z(U.rows());
  chk(flgp_local_anchor_embedding(x.data(), (int)x.size(), U.data(), (int)U.rows(), z.data()));
  return z;
}

// ---- src/lae.cpp:48-70 ----
SpR LAE_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &U, int r) {
  const int n = (int)X.rows(), d = (int)X.cols(), s = (int)U.rows();
  if (U.cols() != d) Rcpp::stop("X and U must have the same number of columns");
  SpR Z = make_csr(n, s, r);
  chk(flgp_lae(X.data(), n, d, U.data(), s, r, Z.outerIndexPtr(), Z.innerIndexPtr(), Z.valuePtr()));
  return Z;
}

// ---- src/Utils.cpp:102-192 (batch has no numerical effect there either) ----
Rcpp::List KNN_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &U, int r, std::string distance, bool output, int batch) {
  (void)batch;
  const int n = (int)X.rows(), d = (int)X.cols(), s = (int)U.rows();
  if (U.cols() != d) Rcpp::stop("X and U must have the same number of columns");
  Eigen::MatrixXi ind(n, r);
  std::vector<double> dist(output ? (size_t)n * r : 0);
  chk(flgp_knn(X.data(), n, d, U.data(), s, r, distance.c_str(), ind.data(), output ? dist.data() : nullptr));
  if (!output) return Rcpp::List::create(Rcpp::Named("ind_knn") = ind);
  // the n x s sparse matrix of the r distances per row, columns ascending inside a row (what Eigen's insert() leaves)
  SpR D = make_csr(n, s, r);
  int *j = D.innerIndexPtr();
  double *x = D.valuePtr();
  for (int i = 0; i < n; ++i) {
    for (int a = 0; a < r; ++a) { j[(size_t)i * r + a] = ind.data()[(size_t)a * n + i]; x[(size_t)i * r + a] = dist[(size_t)a * n + i]; }
    for (int a = 1; a < r; ++a) {                 // insertion sort by column
      const int tj = j[(size_t)i * r + a]; const double tx = x[(size_t)i * r + a];
      int p = a;
      while (p > 0 && j[(size_t)i * r + p - 1] > tj) { j[(size_t)i * r + p] = j[(size_t)i * r + p - 1]; x[(size_t)i * r + p] = x[(size_t)i * r + p - 1]; --p; }
      j[(size_t)i * r + p] = tj; x[(size_t)i * r + p] = tx;
    }
  }
  return Rcpp::List::create(Rcpp::Named("ind_knn") = ind, Rcpp::Named("distances_sp") = D);
}

// ---- src/Utils.cpp:195-212: in place ----
void graphLaplacian_cpp(SpR &Z, std::string gl, const Eigen::VectorXd &num_class) {
  const int r = entries_per_row(Z);
  if (gl == "cluster-normalized" && num_class.size() != Z.cols()) Rcpp::stop("gl=\"cluster-normalized\" needs one cluster size per anchor");
  chk(flgp_graph_laplacian(Z.innerIndexPtr(), Z.valuePtr(), (int)Z.rows(), (int)Z.cols(), r, gl.c_str(),
                           num_class.size() ? num_class.data() : nullptr));
}

// ---- src/Spectrum.cpp:101-117 ----
SpR cross_similarity_lae_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &U, int r, Rcpp::String gl) {
  const int n = (int)X.rows(), d = (int)X.cols(), s = (int)U.rows();
  SpR Z = make_csr(n, s, r);
  chk(flgp_cross_similarity_lae(X.data(), n, d, U.data(), s, (int)U.cols(), r, str(gl).c_str(), Z.outerIndexPtr(), Z.innerIndexPtr(), Z.valuePtr()));
  return Z;
}

// ---- src/Spectrum.cpp:120-142 ----
SpR cross_similarity_se_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &U, int r, Rcpp::String gl, double epsilon) {
  const int n = (int)X.rows(), d = (int)X.cols(), s = (int)U.rows();
  SpR Z = make_csr(n, s, r);
  chk(flgp_cross_similarity_se(X.data(), n, d, U.data(), s, (int)U.cols(), r, str(gl).c_str(), epsilon, Z.outerIndexPtr(), Z.innerIndexPtr(),
                               Z.valuePtr()));
  return Z;
}

// ---- src/Spectrum.cpp:146-161 (+ src/TruncatedSVD.cpp:9-34 behind it) ----
EigenPair spectrum_from_Z_cpp(const SpR &Z, int K, bool root) {
  const int n = (int)Z.rows(), s = (int)Z.cols(), r = entries_per_row(Z);
  if (K < 0) K = s;
  Eigen::VectorXd values(K);
  Eigen::MatrixXd vectors(n, K);
  chk(flgp_spectrum_from_Z(Z.innerIndexPtr(), Z.valuePtr(), n, s, r, K, root ? 1 : 0, values.data(), vectors.data()));
  return EigenPair(values, vectors);
}

// ---- src/Spectrum.cpp:83-94 ----
Eigen::MatrixXd HK_from_spectrum_cpp(const EigenPair &eigenpair, int K, double t, const Eigen::VectorXi &idx0, const Eigen::VectorXi &idx1) {
  Eigen::MatrixXd H(idx0.size(), idx1.size());
  chk(flgp_hk_from_spectrum(eigenpair.values.data(), eigenpair.vectors.data(), (int)eigenpair.vectors.rows(), K, t, idx0.data(),
                            (int)idx0.size(), idx1.data(), (int)idx1.size(), H.data()));
  return H;
}

// ---- src/Spectrum.cpp:48-76: stack [X; X_new], subsample (the reference's own subsample_cpp: R call-backs), spectrum ----
EigenPair heat_kernel_spectrum_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &X_new, int s, int r, int K, const Rcpp::List &models,
                                   int nstart, double epsilon) {
  const int m = (int)X.rows(), m_new = (int)X_new.rows(), d = (int)X.cols(), n = m + m_new;
  if (X_new.cols() != d) Rcpp::stop("X and X_new must have the same number of columns");
  if (K < 0) K = s;
  Eigen::MatrixXd X_all(n, d);
  for (int k = 0; k < d; ++k) {
    for (int i = 0; i < m; ++i) X_all.data()[(size_t)k * n + i] = X.data()[(size_t)k * m + i];
    for (int i = 0; i < m_new; ++i) X_all.data()[(size_t)k * n + m + i] = X_new.data()[(size_t)k * m_new + i];
  }
  const Eigen::MatrixXd U = subsample_cpp(X_all, s, Rcpp::as<std::string>(models["subsample"]), nstart);
  const std::string kernel = Rcpp::as<std::string>(models["kernel"]), gl = Rcpp::as<std::string>(models["gl"]);
  const bool root = models["root"];
  Eigen::VectorXd values(K);
  Eigen::MatrixXd vectors(n, K);
  chk(flgp_heat_kernel_spectrum(X_all.data(), n, d, U.data(), s, (int)U.cols(), r, K, kernel.c_str(), gl.c_str(), root ? 1 : 0, epsilon,
                                values.data(), vectors.data()));
  return EigenPair(values, vectors);
}

// ---- src/Spectrum.cpp:28-43 ----
Eigen::MatrixXd heat_kernel_covariance_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &X_new, int s, int r, double t, int K,
                                           Rcpp::List models, int nstart, double epsilon) {
  if (K < 0) K = s;
  const EigenPair ep = heat_kernel_spectrum_cpp(X, X_new, s, r, K, models, nstart, epsilon);
  const int m = (int)X.rows(), n = m + (int)X_new.rows();
  Eigen::VectorXi idx0(n), idx1(m);
  for (int i = 0; i < n; ++i) idx0.data()[i] = i;
  for (int i = 0; i < m; ++i) idx1.data()[i] = i;
  return HK_from_spectrum_cpp(ep, K, t, idx0, idx1);
}

// ---- src/Spectrum.cpp:17-25 ----
Rcpp::List lae_eigenmap(const Eigen::MatrixXd &X, int s, int r, int ndim, std::string subsample, std::string norm, int nstart) {
  const Eigen::MatrixXd U = subsample_cpp(X, s, subsample, nstart);
  const SpR Z = cross_similarity_lae_cpp(X, U, r, Rcpp::String(norm.c_str()));
  const EigenPair ep = spectrum_from_Z_cpp(Z, ndim, true);
  Eigen::VectorXd ev(ndim);
  for (int k = 0; k < ndim; ++k) ev.data()[k] = 1.0 - ep.values.data()[k];
  return Rcpp::List::create(Rcpp::Named("eigenvalues") = ev, Rcpp::Named("eigenvectors") = ep.vectors);
}
