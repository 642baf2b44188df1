// The declarations of the reference's headers that flgp_cpp_adapters.cpp implements or uses, restated for the test build
// (interface only: names, argument types and defaults of src/Spectrum.h:45-124, src/lae.h:34-60, src/Utils.h:35-62).
#pragma once
#include "RcppEigen.h"
struct EigenPair {                       // src/Spectrum.h:117-124
  Eigen::VectorXd values;
  Eigen::MatrixXd vectors;
  EigenPair(const Eigen::VectorXd &values, const Eigen::MatrixXd &vectors) : values(values), vectors(vectors) {}
  EigenPair() {}
};
typedef Eigen::SparseMatrix<double, Eigen::RowMajor> FlgpSpR;
Rcpp::List lae_eigenmap(const Eigen::MatrixXd &X, int s, int r = 3, int ndim = 2, std::string subsample = "kmeans",
                        std::string norm = "cluster-normalized", int nstart = 1);
Eigen::MatrixXd heat_kernel_covariance_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &X_new, int s, int r, double t, int K,
                                           Rcpp::List models, int nstart, double epsilon);
EigenPair heat_kernel_spectrum_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &X_new, int s, int r, int K, const Rcpp::List &models,
                                   int nstart = 1, double epsilon = 0.1);
Eigen::MatrixXd HK_from_spectrum_cpp(const EigenPair &eigenpair, int K, double t, const Eigen::VectorXi &idx0, const Eigen::VectorXi &idx1);
FlgpSpR cross_similarity_lae_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &U, int r = 3, Rcpp::String gl = "rw");
FlgpSpR cross_similarity_se_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &U, int r, Rcpp::String gl, double epsilon);
EigenPair truncated_SVD_cpp(const FlgpSpR &Z, int K = -1);
EigenPair spectrum_from_Z_cpp(const FlgpSpR &Z, int K, bool root = false);
FlgpSpR LAE_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &U, int r = 3);
Eigen::RowVectorXd local_anchor_embedding_cpp(const Eigen::RowVectorXd &x, const Eigen::MatrixXd &U);
Eigen::RowVectorXd v_to_z_cpp(const Eigen::RowVectorXd &v);
Eigen::MatrixXd subsample_cpp(const Eigen::MatrixXd &X, int s, std::string method = "kmeans", int nstart = 1);   // stays the reference's (src/Utils.cpp:32-68)
void graphLaplacian_cpp(FlgpSpR &Z, std::string gl = "rw", const Eigen::VectorXd &num_class = Eigen::VectorXd());
Rcpp::List KNN_cpp(const Eigen::MatrixXd &X, const Eigen::MatrixXd &U, int r = 3, std::string distance = "Euclidean", bool output = false,
                   int batch = 100);
