// Functional test double of the few Eigen / Rcpp members flgp_cpp_adapters.cpp uses, written from their documented
// public interfaces (dense Matrix: rows / cols / size / data / resize, column-major storage; SparseMatrix<double, RowMajor>
// in compressed form: outerIndexPtr / innerIndexPtr / valuePtr / nonZeros / resizeNonZeros; Rcpp::List with named
// elements, Rcpp::Named, Rcpp::String, Rcpp::as, Rcpp::stop) so that the adapters can be compiled AND run in an image
// without either library.  NOT Eigen, NOT Rcpp.
#pragma once
#include <any>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace Eigen {
enum { ColMajor = 0, RowMajor = 1 };
typedef long Index;
template <class T, bool IsVector = false, bool IsRow = false> class Dense {
  std::vector<T> v_;
  Index r_ = 0, c_ = 0;
 public:
  Dense() {}
  Dense(Index r, Index c) : v_((size_t)(r * c)), r_(r), c_(c) {}
  explicit Dense(Index n) : v_((size_t)n), r_(IsRow ? 1 : n), c_(IsRow ? n : 1) {}
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  Index size() const { return r_ * c_; }
  T *data() { return v_.data(); }
  const T *data() const { return v_.data(); }
  void resize(Index r, Index c) { v_.assign((size_t)(r * c), T()); r_ = r; c_ = c; }
  T &operator()(Index i, Index j) { return v_[(size_t)(j * r_ + i)]; }
  const T &operator()(Index i, Index j) const { return v_[(size_t)(j * r_ + i)]; }
  T &operator()(Index i) { return v_[(size_t)i]; }
  const T &operator()(Index i) const { return v_[(size_t)i]; }
};
typedef Dense<double> MatrixXd;
typedef Dense<int> MatrixXi;
typedef Dense<double, true, false> VectorXd;
typedef Dense<int, true, false> VectorXi;
typedef Dense<double, true, true> RowVectorXd;

template <class T, int Order> class SparseMatrix {
  static_assert(Order == RowMajor, "the path's sparse matrices are row-major (dgRMatrix)");
  Index r_ = 0, c_ = 0;
  std::vector<int> outer_, inner_;
  std::vector<T> val_;
 public:
  SparseMatrix() : outer_(1, 0) {}
  SparseMatrix(Index r, Index c) : r_(r), c_(c), outer_((size_t)r + 1, 0) {}
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  Index nonZeros() const { return (Index)val_.size(); }
  void resize(Index r, Index c) { r_ = r; c_ = c; outer_.assign((size_t)r + 1, 0); inner_.clear(); val_.clear(); }
  void resizeNonZeros(Index nnz) { inner_.resize((size_t)nnz); val_.resize((size_t)nnz); }
  bool isCompressed() const { return true; }
  int *outerIndexPtr() { return outer_.data(); }
  const int *outerIndexPtr() const { return outer_.data(); }
  int *innerIndexPtr() { return inner_.data(); }
  const int *innerIndexPtr() const { return inner_.data(); }
  T *valuePtr() { return val_.data(); }
  const T *valuePtr() const { return val_.data(); }
};
}  // namespace Eigen

namespace Rcpp {
[[noreturn]] inline void stop(const std::string &msg) { throw std::runtime_error(msg); }
class String {
  std::string s_;
 public:
  String(const char *c = "") : s_(c) {}
  String(const std::string &c) : s_(c) {}
  const char *get_cstring() const { return s_.c_str(); }
};
struct NamedValue { std::string name; std::any value; };
struct Named {
  std::string name;
  explicit Named(const std::string &n) : name(n) {}
  template <class T> NamedValue operator=(const T &v) const { return NamedValue{name, std::any(v)}; }
};
class List {
  std::vector<NamedValue> items_;
 public:
  struct Proxy {
    const std::any *a;
    template <class T> operator T() const { return std::any_cast<T>(*a); }
    operator bool() const { return std::any_cast<bool>(*a); }
  };
  static List create() { return List(); }
  template <class... A> static List create(const A &...a) { List l; (l.items_.push_back(a), ...); return l; }
  Proxy operator[](const std::string &name) const {
    for (const auto &it : items_) if (it.name == name) return Proxy{&it.value};
    stop("list has no element named " + name);
  }
  bool has(const std::string &name) const { for (const auto &it : items_) if (it.name == name) return true; return false; }
  size_t size() const { return items_.size(); }
};
template <class T> T as(const List::Proxy &p) {
  if (const std::string *s = std::any_cast<std::string>(p.a)) return T(*s);
  if (const char *const *c = std::any_cast<const char *>(p.a)) return T(*c);
  return std::any_cast<T>(*p.a);
}
}  // namespace Rcpp
