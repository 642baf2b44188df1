// Runs the reference's INTERNAL call sequences through flgp_amd/csrc/rshim/flgp_cpp_adapters.cpp (built against the test
// double of Eigen / Rcpp in tests/eigen_mock/), the way the untouched drivers of src/Fit.cpp do:
//   (1) fit_lae_*:  heat_kernel_spectrum_cpp -> HK_from_spectrum_cpp(idx0, idx0) / (idx1, idx0)     (src/Fit.cpp:42,84-85)
//   (2) fit_se_*:   KNN_cpp(output = true) -> Z.coeffs() = exp(-d / (a2 mean d)) -> graphLaplacian_cpp -> spectrum_from_Z_cpp
//                   -> HK_from_spectrum_cpp, for every a2 of the grid                                  (src/Fit.cpp:127-158)
//   (3) heat_kernel_covariance_cpp and lae_eigenmap with their Rcpp::List arguments / results.
// Input: a binary file written by tests/test_adapters.py (n, d, s, r, K, m, t, gl, root, X, U); output: the matrices, for
// comparison with the golden fixture and the oracle.  usage: adapters_check <in.bin> <out.bin>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "RcppEigen.h"
#include "ref_decls.h"

static Eigen::MatrixXd g_U;      // what the reference's subsample_cpp would return (stats::kmeans): the fixture's anchors
Eigen::MatrixXd subsample_cpp(const Eigen::MatrixXd &, int s, std::string, int) {
  if (g_U.rows() != s) Rcpp::stop("subsample_cpp (test stand-in): s does not match the fixture");
  return g_U;
}

static void put(FILE *f, const double *p, size_t n) { fwrite(p, sizeof(double), n, f); }

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  int hdr[8]; double t; char gl[32];
  if (fread(hdr, sizeof(int), 8, f) != 8 || fread(&t, sizeof(double), 1, f) != 1 || fread(gl, 1, 32, f) != 32) return 2;
  const int n = hdr[0], d = hdr[1], s = hdr[2], r = hdr[3], K = hdr[4], m = hdr[5], root = hdr[6], ucols = hdr[7];
  Eigen::MatrixXd X(n, d);
  g_U.resize(s, ucols);
  if (fread(X.data(), sizeof(double), (size_t)n * d, f) != (size_t)n * d || fread(g_U.data(), sizeof(double), (size_t)s * ucols, f) != (size_t)s * ucols) return 2;
  fclose(f);
  FILE *o = fopen(argv[2], "wb");
  try {
    Eigen::MatrixXd Xt(m, d), Xn(n - m, d);
    for (int k = 0; k < d; ++k) {
      for (int i = 0; i < m; ++i) Xt(i, k) = X(i, k);
      for (int i = m; i < n; ++i) Xn(i - m, k) = X(i, k);
    }
    Rcpp::List models = Rcpp::List::create(Rcpp::Named("subsample") = std::string("kmeans"), Rcpp::Named("kernel") = std::string("lae"),
                                           Rcpp::Named("gl") = std::string(gl), Rcpp::Named("root") = (bool)root);
    // (1) the fit_lae_* sequence
    EigenPair ep = heat_kernel_spectrum_cpp(Xt, Xn, s, r, K, models, 1);
    Eigen::VectorXi idx0(m), idx1(n - m);
    for (int i = 0; i < m; ++i) idx0(i) = i;
    for (int i = 0; i < n - m; ++i) idx1(i) = m + i;
    Eigen::MatrixXd Cvv = HK_from_spectrum_cpp(ep, K, t, idx0, idx0), Cnv = HK_from_spectrum_cpp(ep, K, t, idx1, idx0);
    put(o, ep.values.data(), K); put(o, Cvv.data(), (size_t)m * m); put(o, Cnv.data(), (size_t)(n - m) * m);
    // (3) the exported wrappers
    Eigen::MatrixXd H = heat_kernel_covariance_cpp(Xt, Xn, s, r, t, K, models, 1, 0.1);
    put(o, H.data(), (size_t)n * m);
    Rcpp::List em = lae_eigenmap(X, s, r, 3, "kmeans", gl, 1);
    Eigen::VectorXd ev = em["eigenvalues"];
    Eigen::MatrixXd evec = em["eigenvectors"];
    if (evec.rows() != n || evec.cols() != 3) Rcpp::stop("lae_eigenmap: wrong shape");
    put(o, ev.data(), 3);
    // (2) the fit_se_* sequence (src/Fit.cpp:127-158), two bandwidths
    Eigen::MatrixXd U0(s, d);
    for (int k = 0; k < d; ++k) for (int i = 0; i < s; ++i) U0(i, k) = g_U(i, k);
    Eigen::VectorXd sizes(s);
    for (int i = 0; i < s; ++i) sizes(i) = ucols > d ? g_U(i, d) : 1.0;
    Rcpp::List res_knn = KNN_cpp(X, U0, r, "Euclidean", true);
    Eigen::MatrixXi ind_knn = res_knn["ind_knn"];
    FlgpSpR distances_sp = res_knn["distances_sp"];
    double sum = 0.0;
    for (long e = 0; e < distances_sp.nonZeros(); ++e) sum += distances_sp.valuePtr()[e];
    const double distances_mean = sum / ((double)n * r);
    for (double a2 : {0.5, 2.0}) {
      FlgpSpR Z = distances_sp;
      for (long e = 0; e < Z.nonZeros(); ++e) Z.valuePtr()[e] = std::exp(-distances_sp.valuePtr()[e] / (a2 * distances_mean));
      if (std::string(gl) == "cluster-normalized") graphLaplacian_cpp(Z, gl, sizes);
      else graphLaplacian_cpp(Z, gl);
      EigenPair es = spectrum_from_Z_cpp(Z, K, (bool)root);
      Eigen::MatrixXd Cs = HK_from_spectrum_cpp(es, K, t, idx0, idx0);
      put(o, es.values.data(), K); put(o, Cs.data(), (size_t)m * m);
    }
    // error behaviour: the ABI's message arrives as Rcpp::stop (an exception here)
    bool threw = false;
    try { KNN_cpp(X, U0, r, "geodesic", false); } catch (const std::exception &e) { threw = std::string(e.what()).find("not supported") != std::string::npos; }
    if (!threw) Rcpp::stop("KNN_cpp with an unsupported distance did not stop");
  } catch (const std::exception &e) {
    fprintf(stderr, "adapters_check: %s\n", e.what());
    fclose(o);
    return 1;
  }
  fclose(o);
  printf("adapters_check ok\n");
  return 0;
}
