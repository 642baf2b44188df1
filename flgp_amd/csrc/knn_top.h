// The sorted candidate list of the k-NN kernels (knn.hip, knn_wide.hip).
#pragma once
#include "common.h"

namespace flgp {

// Sorted list of the RCAP smallest (value, index) pairs, ascending.  The first RCAP - r slots
// are pinned by -inf sentinels so that the r-th best real candidate is always bd[RCAP-1]
// (a compile-time register), whatever the run-time r.
template <int RCAP>
struct TopList {
  double bd[RCAP];
  int bi[RCAP];
  __device__ __forceinline__ void init(int r) {
#pragma unroll
    for (int k = 0; k < RCAP; ++k) {
      bd[k] = (k < RCAP - r) ? -__builtin_inf() : __builtin_inf();
      bi[k] = (k < RCAP - r) ? -1 : 0x7fffffff;
    }
  }
  __device__ __forceinline__ double thr() const { return bd[RCAP - 1]; }
  // strict '<': an equal distance never moves ahead of an earlier (lower) index
  __device__ __forceinline__ void insert(double D, int j) {
#pragma unroll
    for (int k = RCAP - 1; k >= 1; --k) {
      const bool c1 = D < bd[k - 1];
      const bool c0 = D < bd[k];
      bd[k] = c1 ? bd[k - 1] : (c0 ? D : bd[k]);
      bi[k] = c1 ? bi[k - 1] : (c0 ? j : bi[k]);
    }
    const bool c0 = D < bd[0];
    bd[0] = c0 ? D : bd[0];
    bi[0] = c0 ? j : bi[0];
  }
  // order by (distance, index): the list a scan in ascending index with the strict '<' above ends with, whatever the
  // order the candidates come in.  D must be below +inf (the scan never takes such an anchor).
  __device__ __forceinline__ bool before(double D, int j, int k) const { return D < bd[k] || (D == bd[k] && j < bi[k]); }
  __device__ __forceinline__ void insert_lex(double D, int j) {
#pragma unroll
    for (int k = RCAP - 1; k >= 1; --k) {
      const bool c1 = before(D, j, k - 1);
      const bool c0 = before(D, j, k);
      bd[k] = c1 ? bd[k - 1] : (c0 ? D : bd[k]);
      bi[k] = c1 ? bi[k - 1] : (c0 ? j : bi[k]);
    }
    const bool c0 = before(D, j, 0);
    bd[0] = c0 ? D : bd[0];
    bi[0] = c0 ? j : bi[0];
  }
};

}  // namespace flgp
