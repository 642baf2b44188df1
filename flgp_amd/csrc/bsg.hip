// Block-sparse products with the Gram matrix for the eigensolver's filter: ordering, block lists, product kernels.
// See bsg.h for the idea.  Everything here works on G alone (no coordinates) and is deterministic.
#include "bsg.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace flgp {

typedef double bd4 __attribute__((ext_vector_type(4)));
typedef double bd2 __attribute__((ext_vector_type(2)));

enum { BSG_M_NNZ = 0, BSG_M_OFF = 1, BSG_M_TOTAL = 2, BSG_M_RNNZ = 3, BSG_M_MAXNK = 4, BSG_M_NPART = 5, BSG_M_NSLAB = 6, BSG_M_MAXPART = 7 };
constexpr unsigned BSG_BUF_WORD3 = 0x00020000u;   // raw buffer resource, 32-bit offsets, no swizzle
typedef unsigned int bu4 __attribute__((ext_vector_type(4)));
constexpr int BSG_BLK = BSG_SK * BSG_TM;   // doubles per kept block
constexpr int BSG_DENSE_MIN = 16;          // a 16 x 64 block with at least this many non-zeros goes to the MFMA product

static size_t al(size_t x) { return (x + 255) / 256 * 256; }

// ------------------------------------------------------------------------------------------
// pass over G: per row (= column, G is symmetric) the number of non-zeros, the absolute sum, the diagonal entry
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bsg_scan_kernel(const double *__restrict__ G, int ldg, int s, int *__restrict__ rcnt,
                                                       double *__restrict__ colabs, double *__restrict__ diag) {
  const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= s) return;
  const double *col = G + (size_t)j * ldg;
  double a = 0.0;
  int n = 0;
  for (int i0 = 0; i0 < s; i0 += 256) {     // four independent loads in flight per lane
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + 64 * u + lane; v[u] = (i < s) ? col[i] : 0.0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { a += __builtin_fabs(v[u]); n += (v[u] != 0.0); }
  }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); n += __shfl_xor(n, off, 64); }
  if (lane == 0) { rcnt[j] = n; colabs[j] = a; diag[j] = col[j]; }
}

// ptr = exclusive scan of cnt[0..n) (one workgroup); total -> meta[slot]; meta[BSG_M_OFF] is raised when total > cap
__global__ __launch_bounds__(1024) void bsg_ptr_kernel(const int *__restrict__ cnt, int n, int *__restrict__ ptr,
                                                       int *__restrict__ meta, int slot, long cap) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int a = tid * per, e = (a + per < n) ? a + per : n;
  int sum = 0;
  for (int i = a; i < e; ++i) sum += cnt[i];
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (tid >= off) ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
  for (int i = a; i < e; ++i) { ptr[i] = run; run += cnt[i]; }
  if (tid == 1023) {
    ptr[n] = part[1023];
    meta[slot] = part[1023];
    if ((long)part[1023] > cap) meta[BSG_M_OFF] = 1;
  }
}

// CSR of G: row j = the non-zeros of column j, ascending
__global__ __launch_bounds__(256) void bsg_csr_fill_kernel(const double *__restrict__ G, int ldg, int s,
                                                           const int *__restrict__ gptr, const int *__restrict__ meta,
                                                           int *__restrict__ gcol, double *__restrict__ gval) {
  if (meta[BSG_M_OFF]) return;
  const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= s) return;
  const double *col = G + (size_t)j * ldg;
  int base = gptr[j];
  for (int i0 = 0; i0 < s; i0 += 64) {
    const int i = i0 + lane;
    const double v = (i < s) ? col[i] : 0.0;
    const bool take = v != 0.0;
    const unsigned long long m = __ballot(take);
    if (take) {
      const int o = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
      gcol[o] = i;
      gval[o] = v;
    }
    base += __builtin_popcountll(m);
  }
}

// out[0] = max colabs (the 1-norm of G), out[1] = sum of the diagonal; one workgroup, fixed order
__global__ __launch_bounds__(256) void bsg_bounds_kernel(const double *__restrict__ colabs, const double *__restrict__ diag,
                                                         int s, double *__restrict__ out) {
  __shared__ double mx[256], tr[256];
  double m = 0.0, t = 0.0;
  for (int i = threadIdx.x; i < s; i += 256) { m = colabs[i] > m ? colabs[i] : m; t += diag[i]; }
  mx[threadIdx.x] = m; tr[threadIdx.x] = t;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      mx[threadIdx.x] = mx[threadIdx.x + off] > mx[threadIdx.x] ? mx[threadIdx.x + off] : mx[threadIdx.x];
      tr[threadIdx.x] += tr[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = mx[0]; out[1] = tr[0]; }
}

// ------------------------------------------------------------------------------------------
// ordering: BSG_SEEDS seed anchors, three hops of diffusion, every anchor joins the seed it received most from,
// two rounds of label smoothing, cluster-to-cluster weights for the host's greedy chain.
// E blocks are s x 64 row-major (the 64 seed values of one anchor contiguous): a wave owns a row, a lane a seed.
// ------------------------------------------------------------------------------------------
__global__ void bsg_seed_gather_kernel(const double *__restrict__ G, int ldg, int s, double *__restrict__ E) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)s * BSG_SEEDS) return;
  const int q = (int)(e & (BSG_SEEDS - 1)), i = (int)(e >> 6);
  const int seed = (int)(((long)q * s) / BSG_SEEDS);
  E[e] = G[(size_t)seed * ldg + i];       // (G e_seed)(i)
}

// lowest lane holding the largest strictly positive value; BSG_SEEDS when there is none
__device__ __forceinline__ int wave_argmax_pos(double v, int lane) {
  int q = (v > 0.0) ? lane : BSG_SEEDS;
  if (!(v > 0.0)) v = 0.0;
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_xor(v, off, 64);
    const int oq = __shfl_xor(q, off, 64);
    if (ov > v || (ov == v && oq < q)) { v = ov; q = oq; }
  }
  return q;
}

// Eout(i, :) = sum_e gval[e] Ein(gcol[e], :) ; lab_out (optional) = argmax of the row
__global__ __launch_bounds__(256) void bsg_hop_kernel(const int *__restrict__ gptr, const int *__restrict__ gcol,
                                                      const double *__restrict__ gval, const int *__restrict__ meta,
                                                      const double *__restrict__ Ein, int s, double *__restrict__ Eout,
                                                      int *__restrict__ lab_out) {
  if (meta[BSG_M_OFF]) return;
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= s) return;
  const int e0 = gptr[i], e1 = gptr[i + 1];
  double acc = 0.0;
  for (int eb = e0; eb < e1; eb += 64) {
    const int ne = (e1 - eb < 64) ? e1 - eb : 64;
    const int kc = (lane < ne) ? gcol[eb + lane] : 0;
    const double vc = (lane < ne) ? gval[eb + lane] : 0.0;
    for (int j = 0; j < ne; j += 8) {
      double x[8], vv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jj = (j + u < ne) ? j + u : j;
        const int k = __shfl(kc, jj, 64);
        vv[u] = (j + u < ne) ? __shfl(vc, jj, 64) : 0.0;
        x[u] = Ein[(size_t)k * BSG_SEEDS + lane];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += vv[u] * x[u];
    }
  }
  Eout[(size_t)i * BSG_SEEDS + lane] = acc;
  if (lab_out) {
    const int q = wave_argmax_pos(acc, lane);
    if (lane == 0) lab_out[i] = q;
  }
}

// Y(i, q) = sum of gval[e] over the entries of row i whose column carries label q; lab_out = argmax (optional),
// Yout (optional) keeps the row.  The bins live in LDS; a wave's LDS adds are applied in program / lane order.
__global__ __launch_bounds__(256) void bsg_label_hop_kernel(const int *__restrict__ gptr, const int *__restrict__ gcol,
                                                            const double *__restrict__ gval, const int *__restrict__ meta,
                                                            const int *__restrict__ lab_in, int s, int *__restrict__ lab_out,
                                                            double *__restrict__ Yout) {
  __shared__ double bins[4][BSG_SEEDS];
  if (meta[BSG_M_OFF]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i = blockIdx.x * 4 + wv;
  if (i >= s) return;
  bins[wv][lane] = 0.0;
  const int e0 = gptr[i], e1 = gptr[i + 1];
  for (int e = e0 + lane; e < e1; e += 64) {
    const int q = lab_in[gcol[e]];
    if (q < BSG_SEEDS) atomicAdd(&bins[wv][q], gval[e]);
  }
  const double v = bins[wv][lane];
  if (Yout) Yout[(size_t)i * BSG_SEEDS + lane] = v;
  if (lab_out) {
    const int q = wave_argmax_pos(v, lane);
    if (lane == 0) lab_out[i] = q;
  }
}

// C(p, q) = sum over the rows with label p of Y(row, q), in two levels with a fixed order: a workgroup adds its chunk
// of 64 rows into a 65 x 64 table in LDS (thread q owns column q: no conflicts, rows in ascending order), a second
// launch adds the chunk tables in ascending chunk order.
constexpr int BSG_CW_ROWS = 64;
__global__ __launch_bounds__(64) void bsg_cluster_partial_kernel(const double *__restrict__ Y, const int *__restrict__ lab,
                                                                 const int *__restrict__ meta, int s,
                                                                 double *__restrict__ part) {
  __shared__ double tab[(BSG_SEEDS + 1) * BSG_SEEDS];
  if (meta[BSG_M_OFF]) return;
  const int q = threadIdx.x, i0 = blockIdx.x * BSG_CW_ROWS;
  for (int p = 0; p <= BSG_SEEDS; ++p) tab[p * BSG_SEEDS + q] = 0.0;
  const int i1 = (i0 + BSG_CW_ROWS < s) ? i0 + BSG_CW_ROWS : s;
  for (int i = i0; i < i1; ++i) tab[lab[i] * BSG_SEEDS + q] += Y[(size_t)i * BSG_SEEDS + q];
  for (int p = 0; p < BSG_SEEDS; ++p) part[((size_t)blockIdx.x * BSG_SEEDS + p) * BSG_SEEDS + q] = tab[p * BSG_SEEDS + q];
}
__global__ __launch_bounds__(256) void bsg_cluster_sum_kernel(const double *__restrict__ part, int nchunk,
                                                              const int *__restrict__ meta, double *__restrict__ C) {
  if (meta[BSG_M_OFF]) return;
  const int e = blockIdx.x * 256 + threadIdx.x;    // e = p * 64 + q
  double acc = 0.0;
  for (int c = 0; c < nchunk; ++c) acc += part[(size_t)c * BSG_SEEDS * BSG_SEEDS + e];
  C[e] = acc;
}

__global__ void bsg_iperm_kernel(const int *__restrict__ perm, int s, int *__restrict__ iperm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < s) iperm[perm[i]] = i;
}

// non-zeros per (64-row tile, 16-deep stage) of P G P^T: one workgroup per tile gathers its 64 rows' entries into a
// histogram in LDS (no global atomics: the kept blocks take ~500 increments each)
__global__ __launch_bounds__(256) void bsg_count_kernel(const int *__restrict__ gptr, const int *__restrict__ gcol,
                                                        const int *__restrict__ perm, const int *__restrict__ iperm, int s,
                                                        int nstage, int *__restrict__ cnt) {
  extern __shared__ int hist[];
  const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int g = threadIdx.x; g < nstage; g += 256) hist[g] = 0;
  __syncthreads();
  for (int r = wave; r < BSG_TM; r += 4) {
    const int ip = t * BSG_TM + r;
    if (ip >= s) break;
    const int i = perm[ip];
    const int e1 = gptr[i + 1];
    for (int e = gptr[i] + lane; e < e1; e += 64) atomicAdd(&hist[iperm[gcol[e]] >> 4], 1);
  }
  __syncthreads();
  for (int g = threadIdx.x; g < nstage; g += 256) cnt[(size_t)t * nstage + g] = hist[g];
}

// One workgroup turns the counts into the block lists: per tile the stages with >= BSG_DENSE_MIN non-zeros (ascending),
// their positions in the packed array, the tiles ordered by descending list length (longest first in the work queue).
__global__ __launch_bounds__(1024) void bsg_lists_kernel(const int *__restrict__ cnt, int ntile, int nstage, long pack_cap,
                                                         int *__restrict__ blkpos, int *__restrict__ nk,
                                                         int *__restrict__ off, int *__restrict__ klist,
                                                         int *__restrict__ order, int *__restrict__ meta, int part_cap,
                                                         int part_max, int slab_cap, int4 *__restrict__ parts_u,
                                                         int4 *__restrict__ parts, int *__restrict__ slab0) {
  __shared__ int over;
  __shared__ int nk_s[1024];          // the list lengths, for the serial passes below (ntile <= 1024: s <= 65536): from global
                                      // memory every one of their ~80 iterations was a dependent trip to L2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) over = 0;
  for (int t = wave; t < ntile; t += 16) {
    int n = 0;
    for (int g0 = 0; g0 < nstage; g0 += 64) {
      const int g = g0 + lane;
      const bool d = g < nstage && cnt[(size_t)t * nstage + g] >= BSG_DENSE_MIN;
      const unsigned long long m = __ballot(d);
      if (g < nstage)
        blkpos[(size_t)t * nstage + g] =
            d ? n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0)) : -1;
      n += __builtin_popcountll(m);
    }
    if (lane == 0) { nk[t] = n; nk_s[t] = n; }
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0, mx = 0;
    for (int t = 0; t < ntile; ++t) { off[t] = run; run += nk_s[t]; mx = nk_s[t] > mx ? nk_s[t] : mx; }
    off[ntile] = run;
    meta[BSG_M_TOTAL] = run;
    meta[BSG_M_MAXNK] = mx;
    if ((long)run > pack_cap) { meta[BSG_M_OFF] = 1; over = 1; }
  }
  __syncthreads();
  if (over) return;
  for (int t = tid; t < ntile; t += 1024) {
    int rank = 0;
    const int mine = nk_s[t];
    for (int u = 0; u < ntile; ++u) rank += (nk_s[u] > mine) || (nk_s[u] == mine && u < t);
    order[rank] = t;
  }
  for (int t = wave; t < ntile; t += 16) {
    const int o = off[t];
    for (int g = lane; g < nstage; g += 64) {
      const int p = blkpos[(size_t)t * nstage + g];
      if (p >= 0) { blkpos[(size_t)t * nstage + g] = o + p; klist[o + p] = g; }
    }
  }
  // Tasks of the product: a tile, or -- part_cap > 0 -- the equal parts of a tile whose list is longer than part_cap stages
  // (their partial sums meet in slabs, bsg_gemm_kernel), longest first.
  __shared__ int np_sh;
  if (tid == 0) {
    int cap = part_cap;
    for (int pass = 0; pass < 2; ++pass) {
      int np = 0, nslab = 0;
      for (int t = 0; t < ntile; ++t) {
        const int P = (cap > 0 && nk_s[t] > cap) ? (nk_s[t] + cap - 1) / cap : 1;
        np += P; nslab += (P > 1) ? P : 0;
      }
      if (np <= part_max && nslab <= slab_cap) break;
      cap = 0;                                   // does not fit: whole tiles
    }
    int np = 0, nslab = 0, mxp = 0;
    for (int t = 0; t < ntile; ++t) {
      const int P = (cap > 0 && nk_s[t] > cap) ? (nk_s[t] + cap - 1) / cap : 1;
      slab0[t] = (P > 1) ? nslab : -1;
      for (int q = 0; q < P; ++q) {
        const int lo = (int)(((long)q * nk_s[t]) / P), hi = (int)(((long)(q + 1) * nk_s[t]) / P);
        parts_u[np + q] = make_int4(t, lo, hi - lo, q | (P << 16));
        mxp = (hi - lo > mxp) ? hi - lo : mxp;
      }
      np += P; nslab += (P > 1) ? P : 0;
    }
    np_sh = np;
    meta[BSG_M_NPART] = np; meta[BSG_M_NSLAB] = nslab; meta[BSG_M_MAXPART] = mxp;
  }
  __syncthreads();
  const int np = np_sh;
  for (int u = tid; u < np; u += 1024) {
    const int4 me = parts_u[u];
    int rank = 0;
    for (int v = 0; v < np; ++v) { const int nv = parts_u[v].z; rank += (nv > me.z) || (nv == me.z && v < u); }
    parts[rank] = me;
  }
}

// kept block (tile t, stage g) copied out: pack[pos][kk][r] = (P G P^T)(t*64 + r, g*16 + kk), zero outside the matrix
__global__ __launch_bounds__(256) void bsg_pack_kernel(const double *__restrict__ G, int ldg, int s, int nstage,
                                                       const int *__restrict__ perm, const int *__restrict__ blkpos,
                                                       const int *__restrict__ meta, double *__restrict__ pack) {
  if (meta[BSG_M_OFF]) return;
  const int g = blockIdx.x, t = blockIdx.y;
  const int pos = blkpos[(size_t)t * nstage + g];
  if (pos < 0) return;
#pragma unroll
  for (int rep = 0; rep < 4; ++rep) {
    const int x = threadIdx.x + 256 * rep;
    const int kk = x >> 6, r = x & 63;
    const int i = t * BSG_TM + r, k = g * BSG_SK + kk;
    pack[(size_t)pos * BSG_BLK + x] = (i < s && k < s) ? G[(size_t)perm[k] * ldg + perm[i]] : 0.0;
  }
}

// the rest: entries of permuted row i that sit in blocks the MFMA product skips; count pass (fill == 0) and fill pass
__global__ __launch_bounds__(256) void bsg_rem_kernel(const int *__restrict__ gptr, const int *__restrict__ gcol,
                                                      const double *__restrict__ gval, const int *__restrict__ perm,
                                                      const int *__restrict__ iperm, const int *__restrict__ blkpos,
                                                      const int *__restrict__ meta, int s, int nstage,
                                                      const int *__restrict__ rptr, int *__restrict__ rcnt,
                                                      int *__restrict__ rcol, double *__restrict__ rval, int fill) {
  if (meta[BSG_M_OFF]) return;
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= s) return;
  const int src = perm[i];
  const int *bp = blkpos + (size_t)(i / BSG_TM) * nstage;
  const int e0 = gptr[src], e1 = gptr[src + 1];
  int n = 0;
  const int base = fill ? rptr[i] : 0;
  for (int eb = e0; eb < e1; eb += 64) {
    const int e = eb + lane;
    int k = 0;
    bool take = false;
    if (e < e1) { k = iperm[gcol[e]]; take = bp[k >> 4] < 0; }
    const unsigned long long m = __ballot(take);
    if (fill && take) {
      const int o = base + n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
      rcol[o] = k;
      rval[o] = gval[e];
    }
    n += __builtin_popcountll(m);
  }
  if (!fill && lane == 0) rcnt[i] = n;
}

// ------------------------------------------------------------------------------------------
// product, first launch: out(:, i) = beta E(:, i) + gamma E2(:, i) + alpha * sum over the scattered entries e of row i of
// rval[e] X(:, rcol[e]).  One workgroup per row; the four waves split a long row's entries (a hub anchor has a couple
// of hundred), partial sums added in wave order.  Also resets the work-queue head of the launch that follows.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bsg_pre_kernel(const int *__restrict__ rptr, const int *__restrict__ rcol,
                                                      const double *__restrict__ rval, const double *__restrict__ Xt, int b,
                                                      double alpha, double beta, const double *__restrict__ Et, double gamma,
                                                      const double *__restrict__ E2t, double *__restrict__ out,
                                                      int *__restrict__ head) {
  __shared__ double part[3][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x;
  if (i == 0 && threadIdx.x == 0) *head = 0;
  const int e0 = rptr[i], e1 = rptr[i + 1];
  const int n = e1 - e0;
  if (n <= 8 && wave > 0) return;                         // short rows: wave 0 alone (uniform over the workgroup)
  const int q = (n <= 8) ? n : (n + 3) / 4;
  const int my0 = e0 + wave * q, my1 = (my0 + q < e1) ? my0 + q : e1;
  for (int c0 = 0; c0 < b; c0 += 256) {
    const int c = c0 + 4 * lane;
    const bool okc = c + 3 < b;            // b is a multiple of 16: a lane's four columns are all in or all out
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int eb = my0; eb < my1; eb += 64) {
      const int ne = (my1 - eb < 64) ? my1 - eb : 64;
      const int kc = (lane < ne) ? rcol[eb + lane] : 0;
      const double vc = (lane < ne) ? rval[eb + lane] : 0.0;
      for (int j = 0; j < ne; j += 8) {
        bd2 xa[8], xb[8];
        double vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int jj = (j + u < ne) ? j + u : j;
          // the entry is the same for every lane: read it from lane jj into scalar registers (a shuffle is a trip through the
          // LDS crossbar per 32 bits, three per entry, in front of every gather's address)
          const int k = __builtin_amdgcn_readlane(kc, jj);
          const unsigned long long vbits = __builtin_bit_cast(unsigned long long, vc);
          const unsigned vlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)vbits, jj);
          const unsigned vhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(vbits >> 32), jj);
          const double vsel = __builtin_bit_cast(double, ((unsigned long long)vhi << 32) | vlo);
          vv[u] = (j + u < ne) ? vsel : 0.0;
          const bd2 *x = (const bd2 *)(Xt + (size_t)k * b + (okc ? c : 0));
          xa[u] = x[0];
          xb[u] = x[1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a0 += vv[u] * xa[u][0]; a1 += vv[u] * xa[u][1]; a2 += vv[u] * xb[u][0]; a3 += vv[u] * xb[u][1];
        }
      }
    }
    if (n > 8) {
      if (wave > 0) {
        part[wave - 1][4 * lane + 0] = a0; part[wave - 1][4 * lane + 1] = a1;
        part[wave - 1][4 * lane + 2] = a2; part[wave - 1][4 * lane + 3] = a3;
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          a0 += part[w][4 * lane + 0]; a1 += part[w][4 * lane + 1]; a2 += part[w][4 * lane + 2]; a3 += part[w][4 * lane + 3];
        }
      }
      __syncthreads();
    }
    if (wave == 0 && okc) {
      const size_t o = (size_t)i * b + c;
      bd2 r0 = bd2{alpha * a0, alpha * a1}, r1 = bd2{alpha * a2, alpha * a3};
      if (Et) {
        const bd2 x0 = *(const bd2 *)(Et + o), x1 = *(const bd2 *)(Et + o + 2);
        r0 += beta * x0; r1 += beta * x1;
      }
      if (E2t) {
        const bd2 x0 = *(const bd2 *)(E2t + o), x1 = *(const bd2 *)(E2t + o + 2);
        r0 += gamma * x0; r1 += gamma * x1;
      }
      *(bd2 *)(out + o) = r0;
      *(bd2 *)(out + o + 2) = r1;
    }
  }
}

// ------------------------------------------------------------------------------------------
// product, second launch: out(c, i) += alpha * sum over the kept blocks of the row tile of i.
// A task = (64-anchor tile, 64-column tile of the block); workgroups pull tasks from a queue ordered by descending
// list length, so no output tile is split and nothing is reduced afterwards.  Per stage the packed 16 x 64 block of
// P G P^T and the 16 rows of X_t it multiplies go global -> registers (two stages ahead) -> LDS (one stage ahead);
// four waves of 32 x 32 (2 x 2 v_mfma_f64_16x16x4).  LDS row stride 80 doubles: the k and k+1 fragment rows of a
// ds_read_b64 fall into opposite halves of the 64 banks.
// ------------------------------------------------------------------------------------------
constexpr int BSG_LD = 64;
#ifndef BSG_PF
#define BSG_PF 4
#endif
constexpr int BSG_LROWS = 2 * BSG_SK;     // k rows per LDS buffer: two list entries per barrier

// Eight waves per workgroup, two per SIMD, so that one wave's LDS hand-over (stores, barrier, fragment reads) passes
// behind the other's MFMAs -- with one wave per SIMD the two phases simply added up (measured: 21 us of hand-over +
// 36 us of MFMA for the longest tile).  Two list entries (32 k) go through LDS per barrier: the upper four waves
// multiply the second entry, the lower four the first, and their accumulators are added, lower half first, when the
// tile is done.
__global__ __launch_bounds__(512, 2) void bsg_gemm_kernel(const double *__restrict__ Xt, double *__restrict__ out, int b,
                                                          int s, double alpha, const int *__restrict__ order,
                                                          const int *__restrict__ nk, const int *__restrict__ off,
                                                          const int *__restrict__ klist, const double *__restrict__ pack,
                                                          int ntile, int nbt, int *__restrict__ head,
                                                          const int4 *__restrict__ parts, int npart,
                                                          const int *__restrict__ slab0, double *__restrict__ slab,
                                                          int *__restrict__ tcnt, int xmap,
                                                          long long *__restrict__ trace) {
  __shared__ double sm[4 * BSG_LROWS * BSG_LD];
  double (*As2)[BSG_LROWS * BSG_LD] = (double (*)[BSG_LROWS * BSG_LD])sm;
  double (*Bs2)[BSG_LROWS * BSG_LD] = (double (*)[BSG_LROWS * BSG_LD])(sm + 2 * BSG_LROWS * BSG_LD);
  __shared__ int task_sh, arrived_sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int khalf = wave >> 2;                      // which entry of a pair this wave multiplies
  const int wr = ((wave >> 1) & 1) * 32, wc = (wave & 1) * 32;
  const int fr = lane & 15, fk = lane >> 4;
  const int sw = (fk & 1) << 4;                     // LDS swizzle: odd k rows hold their 16-column groups swapped in pairs
  const int kq = tid >> 5, pr = 2 * (tid & 31);     // staging: row kq (0..15) of both entries, column pair pr
  const int ntask = npart * nbt;
  const size_t x_bytes = (size_t)s * b * 8;
  const __amdgpu_buffer_rsrc_t rs_b =
      __builtin_amdgcn_make_buffer_rsrc((void *)Xt, 0, x_bytes < 0xfffffff0u ? (int)(unsigned)x_bytes : (int)0xfffffff0u, BSG_BUF_WORD3);
  // A workgroup's first task is fixed by its number, the rest come from the queue (which starts behind the fixed ones):
  // a couple of hundred workgroups asking one word for their first task at the same moment queue up at that word.
  // xmap: workgroups are dealt round-robin to the 8 XCDs; the fixed tasks are laid out so that the column tiles of a
  // tile run on ONE XCD (they read the same packed blocks: one fetch into that L2 instead of nbt) and every XCD gets
  // every 8th tile of the length order.
  int next_fixed;
  {
    const int bid = blockIdx.x, G = gridDim.x;
    if (xmap) { const int x = bid & 7, j = bid >> 3; next_fixed = ((j / nbt) * 8 + x) * nbt + j % nbt; }
    else next_fixed = bid;
    (void)G;
  }
  for (bool fixed = true;; fixed = false) {
    if (!fixed) {
      if (tid == 0) task_sh = (int)gridDim.x + atomicAdd(head, 1);
      __syncthreads();
    }
    // (made wave-uniform for the compiler: the stage numbers kl[si] then come through the scalar cache and its own
    //  counter -- as vector loads they were the youngest entry of the in-order vmcnt queue, and waiting for one drained
    //  every operand load in flight behind it, i.e. the whole prefetch)
    const int task = fixed ? next_fixed : __builtin_amdgcn_readfirstlane(task_sh);
    if (!fixed) __syncthreads();
    if (task >= ntask) return;
    if (trace && tid == 0) {                        // diagnostic timeline (flgp_dev_bsg_set_trace): start, end, place, length
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      trace[4 * (size_t)task + 0] = (long long)wall_clock64();
      trace[4 * (size_t)task + 2] = ((long long)(xcc & 0xf) << 56) | ((long long)(hw & 0xffff) << 40) | (clock64() & 0xffffffffffll);
    }
    const int4 part = parts[task / nbt];            // (tile, first entry, entries, part | parts << 16): scalar loads
    const int t = part.x, tb = task % nbt;
    const int ns = part.z;
    const int nparts = part.w >> 16;
    if (ns == 0) continue;                          // (an empty list: the tile is one part, nothing to add)
    const int np = (ns + 1) >> 1;                   // pairs of list entries
    const int first = off[t] + part.y;
    const int *kl = klist + first;
    const int col0 = tb * BSG_TN;
    const int cl = (col0 + pr < b) ? col0 + pr : b - 2;      // columns past the block read a valid pair (never stored)
    const int last = ns - 1;
    // Loads are issued unconditionally, so that the number of loads in flight does not depend on the path taken -- with
    // loads under `if` the compiler's wait-count pass gave up at the joins and waited for all of them every stage.  Both
    // operands come through buffer resources: the packed blocks' one ends with this task's list, so an entry past the
    // end reads zeros from the bounds check (a choice on the loaded DATA, `live ? a : 0`, had been compiled as load ->
    // s_waitcnt vmcnt(0) -> v_cndmask right at the load: every stage waited for the loads it had just issued, round 3's
    // timeline: 1.6 us per pair of stages); X_t's ends with the matrix, so the rows of a last, partial stage read zeros
    // too.  An address is one v_add of a scalar to the lane's constant: beside the MFMA stream every instruction counts.
    const __amdgpu_buffer_rsrc_t rs_a =
        __builtin_amdgcn_make_buffer_rsrc((void *)(pack + (size_t)first * BSG_BLK), 0, ns * BSG_BLK * 8, BSG_BUF_WORD3);
    const unsigned lane_a = (unsigned)((kq * BSG_TM + pr) * 8), lane_b = (unsigned)((kq * b + cl) * 8);
    const unsigned stage_b = (unsigned)(BSG_SK * b * 8);
    auto load = [&](int pj, bd2 (&ra)[2], bd2 (&rb)[2]) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int si = 2 * pj + e;
        const int sb = si < last ? si : last;
        ra[e] = __builtin_bit_cast(bd2, __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane_a + (unsigned)si * (BSG_BLK * 8), 0, 0));
        rb[e] = __builtin_bit_cast(bd2, __builtin_amdgcn_raw_buffer_load_b128(rs_b, lane_b + (unsigned)kl[sb] * stage_b, 0, 0));
      }
    };
    auto stash = [&](int buf, const bd2 (&ra)[2], const bd2 (&rb)[2]) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        *(bd2 *)(&As2[buf][(e * BSG_SK + kq) * BSG_LD + (pr ^ ((kq & 1) << 4))]) = ra[e];
        *(bd2 *)(&Bs2[buf][(e * BSG_SK + kq) * BSG_LD + (pr ^ ((kq & 1) << 4))]) = rb[e];
      }
    };
    bd4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = bd4{0.0, 0.0, 0.0, 0.0};
    auto mma = [&](int buf) {
      const double *As = As2[buf] + (khalf * BSG_SK + fk) * BSG_LD + wr + fr;
      const double *Bs = Bs2[buf] + (khalf * BSG_SK + fk) * BSG_LD + wc + fr;
#pragma unroll
      for (int kk = 0; kk < BSG_SK; kk += 4) {
        double fa[2], fb[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[mi] = As[kk * BSG_LD + ((mi * 16) ^ sw)];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[ni] = Bs[kk * BSG_LD + ((ni * 16) ^ sw)];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
      }
    };
    // What the first launch left in this tile of `out` is fetched now, so that the trip to the memory side passes behind
    // the main loop (fetched in the epilogue, the dependent load -> add -> store cost 10 us of a 50 us launch).
    double base[2][2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int i = t * BSG_TM + wr + mi * 16 + fk + 4 * reg, c = col0 + wc + ni * 16 + fr;
          base[mi][ni][reg] = (khalf == 0 && nparts == 1 && i < s && c < b) ? out[(size_t)i * b + c] : 0.0;
        }
    // the operands of BSG_PF pairs are in flight in registers (pair j in slot j % BSG_PF), one further pair sits in LDS
    bd2 ra[BSG_PF][2], rb[BSG_PF][2];
#pragma unroll
    for (int j = 0; j < BSG_PF; ++j) load(j, ra[j], rb[j]);
    stash(0, ra[0], rb[0]);
    load(BSG_PF, ra[0], rb[0]);
    __syncthreads();
    int p0 = 0;
    for (; p0 + BSG_PF <= np; p0 += BSG_PF) {      // whole groups: straight-line code, BSG_PF - 1 pairs of loads in flight
#pragma unroll
      for (int u = 0; u < BSG_PF; ++u) {
        stash((u + 1) & 1, ra[(u + 1) % BSG_PF], rb[(u + 1) % BSG_PF]);
        load(p0 + u + 1 + BSG_PF, ra[(u + 1) % BSG_PF], rb[(u + 1) % BSG_PF]);
        mma(u & 1);
        __syncthreads();
      }
    }
    static_assert(BSG_PF % 2 == 0, "the LDS buffer of a pair is its parity: groups must be even");
#pragma unroll
    for (int u = 0; u < BSG_PF - 1; ++u) {         // the last, partial group
      if (p0 + u < np) {                           // uniform over the workgroup
        stash((u + 1) & 1, ra[(u + 1) % BSG_PF], rb[(u + 1) % BSG_PF]);
        mma(u & 1);
        __syncthreads();
      }
    }
    // the upper waves hand their sums over through LDS (the stage buffers are free: everybody is past the last barrier)
    double *xch = sm;                              // 4 waves x 16 values x 64 lanes = 4096 doubles
    static_assert(4 * BSG_LROWS * BSG_LD >= 4096, "exchange buffer");
    if (khalf == 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) xch[(((wave & 3) * 16) + (mi * 2 + ni) * 4 + reg) * 64 + lane] = acc[mi][ni][reg];
    }
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) acc[mi][ni][reg] += xch[((wave * 16) + (mi * 2 + ni) * 4 + reg) * 64 + lane];
    }
    bool write = true;                              // uniform over the workgroup
    if (nparts > 1) {
      // The partial sum goes to this part's slab, write-through (sc1: the parts of a tile may sit on different XCDs, whose
      // L2s do not see each other); every storing wave drains its stores, the workgroup meets, one lane draws a ticket.
      // The part that draws the last ticket adds the slabs in part order (sc1 loads: they pass the L1) on top of what
      // the first launch left in `out`.  Deterministic: the sum does not depend on who arrives last.
      const int tile_slab = slab0[t];
      double *my = slab + ((size_t)(tile_slab + (part.w & 0xffff)) * nbt + tb) * (BSG_TM * BSG_TN);
      if (khalf == 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)my, 0, BSG_TM * BSG_TN * 8, BSG_BUF_WORD3);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bd2 v = bd2{acc[mi][ni][2 * h], acc[mi][ni][2 * h + 1]};
              const unsigned o = (unsigned)(((wave * 8 + (mi * 2 + ni) * 2 + h) * 64 + lane) * 16);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu4, v), rs, o, 0, 16);
            }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        arrived_sh = __hip_atomic_fetch_add(&tcnt[t * nbt + tb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      write = arrived_sh == nparts - 1;
      if (write) {
        if (tid == 0) __hip_atomic_store(&tcnt[t * nbt + tb], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (khalf == 0) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                const int i = t * BSG_TM + wr + mi * 16 + fk + 4 * reg, c = col0 + wc + ni * 16 + fr;
                base[mi][ni][reg] = (i < s && c < b) ? out[(size_t)i * b + c] : 0.0;
                acc[mi][ni][reg] = 0.0;
              }
          for (int q = 0; q < nparts; ++q) {
            const double *sl = slab + ((size_t)(tile_slab + q) * nbt + tb) * (BSG_TM * BSG_TN);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)sl, 0, BSG_TM * BSG_TN * 8, BSG_BUF_WORD3);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const unsigned o = (unsigned)(((wave * 8 + (mi * 2 + ni) * 2 + h) * 64 + lane) * 16);
                  const bd2 v = __builtin_bit_cast(bd2, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 16));
                  acc[mi][ni][2 * h] += v[0];
                  acc[mi][ni][2 * h + 1] += v[1];
                }
          }
        }
      }
    }
    if (khalf == 0 && write) {
      // D(row = fk + 4 reg, col = fr) of each 16 x 16 tile; the 16 lanes of one fk write 128 contiguous bytes
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int c = col0 + wc + ni * 16 + fr;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int i = t * BSG_TM + wr + mi * 16 + fk + 4 * reg;
            if (i < s && c < b) out[(size_t)i * b + c] = base[mi][ni][reg] + alpha * acc[mi][ni][reg];
          }
        }
    }
    __syncthreads();                               // the exchange buffer is stage memory of the next task
    if (trace && tid == 0) {
      trace[4 * (size_t)task + 1] = (long long)wall_clock64();
      const long long c0 = trace[4 * (size_t)task + 2];
      trace[4 * (size_t)task + 2] = (c0 & ~0xffffffffffll) | ((clock64() - c0) & 0xffffffffffll);   // shader cycles of the task
      trace[4 * (size_t)task + 3] = ((long long)blockIdx.x << 32) | (unsigned)ns;
    }
  }
}

// ------------------------------------------------------------------------------------------
// A few Lanczos steps on the CSR copy of G: an estimate of lambda_min, the far end of the interval the Chebyshev
// filter damps (eig.hip).  Single vector, no re-orthogonalisation (only the extreme Ritz value is wanted, 20 steps);
// runs on a second stream beside the rest of the set-up.  Two launches per step:
//   w = G v - beta v_prev, partial sums of w . v                     (one wave per row)
//   alpha = sum; w -= alpha v; beta = |w|; v_prev = v; v = w / beta   (one workgroup)
// ------------------------------------------------------------------------------------------
constexpr int BSG_LANCZOS = 20;

__global__ __launch_bounds__(256) void bsg_lz_spmv_kernel(const int *__restrict__ gptr, const int *__restrict__ gcol,
                                                          const double *__restrict__ gval, const int *__restrict__ meta,
                                                          const double *__restrict__ v, const double *__restrict__ vprev,
                                                          const double *__restrict__ ab, int step, int s,
                                                          double *__restrict__ w, double *__restrict__ part) {
  __shared__ double red[4];
  if (meta[BSG_M_OFF]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i = blockIdx.x * 4 + wv;
  double acc = 0.0;
  if (i < s) {
    const int e1 = gptr[i + 1];
    for (int e = gptr[i] + lane; e < e1; e += 64) acc += gval[e] * v[gcol[e]];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  double dot = 0.0;
  if (i < s && lane == 0) {
    const double beta = step > 0 ? ab[2 * (step - 1) + 1] : 0.0;
    const double wi = acc - beta * vprev[i];
    w[i] = wi;
    dot = wi * v[i];
  }
  if (lane == 0) red[wv] = dot;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(1024) void bsg_lz_update_kernel(const double *__restrict__ part, int nparts, int s,
                                                             const int *__restrict__ meta, double *__restrict__ w,
                                                             double *__restrict__ v, double *__restrict__ vprev,
                                                             double *__restrict__ ab, int step) {
  __shared__ double red[1024];
  __shared__ double alpha_s, beta_s;
  if (meta[BSG_M_OFF]) return;
  const int tid = threadIdx.x;
  double a = 0.0;
  for (int q = tid; q < nparts; q += 1024) a += part[q];
  red[tid] = a;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
  if (tid == 0) alpha_s = red[0];
  __syncthreads();
  const double alpha = alpha_s;
  double nn = 0.0;
  for (int i = tid; i < s; i += 1024) { const double x = w[i] - alpha * v[i]; w[i] = x; nn += x * x; }
  __syncthreads();
  red[tid] = nn;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
  if (tid == 0) { beta_s = __builtin_sqrt(red[0]); ab[2 * step] = alpha; ab[2 * step + 1] = beta_s; }
  __syncthreads();
  const double inv = beta_s > 0.0 ? 1.0 / beta_s : 0.0;
  for (int i = tid; i < s; i += 1024) { vprev[i] = v[i]; v[i] = w[i] * inv; }
}

__global__ void bsg_lz_start_kernel(double *__restrict__ v, double *__restrict__ vprev, int s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s) return;
  unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  v[i] = (((double)(z >> 11) + 0.5) * (2.0 / 9007199254740992.0) - 1.0) / __builtin_sqrt((double)s / 3.0);   // |v| ~ 1
  vprev[i] = 0.0;
}

// smallest eigenvalue of the k x k symmetric tridiagonal (alpha, beta) by bisection on the Sturm count
static double tridiag_min(const double *ab, int k) {
  double lo = 1e300, hi = -1e300;
  for (int i = 0; i < k; ++i) {
    const double r = (i > 0 ? std::fabs(ab[2 * (i - 1) + 1]) : 0.0) + (i + 1 < k ? std::fabs(ab[2 * i + 1]) : 0.0);
    lo = std::min(lo, ab[2 * i] - r); hi = std::max(hi, ab[2 * i] + r);
  }
  for (int it = 0; it < 100; ++it) {
    const double mid = 0.5 * (lo + hi);
    int below = 0;            // eigenvalues < mid
    double d = 1.0;
    for (int i = 0; i < k; ++i) {
      const double b2 = i > 0 ? ab[2 * (i - 1) + 1] * ab[2 * (i - 1) + 1] : 0.0;
      d = (ab[2 * i] - mid) - (i > 0 ? b2 / d : 0.0);
      if (d == 0.0) d = 1e-300;
      below += d < 0.0;
    }
    if (below >= 1) hi = mid; else lo = mid;
  }
  return 0.5 * (lo + hi);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static size_t bsg_csr_cap(int s) { return (size_t)s * s / 4 + 64; }   // denser than 25 %: not worth a block-sparse product
static size_t bsg_pack_cap(int s) {
  const size_t nt = (s + BSG_TM - 1) / BSG_TM, ng = (s + BSG_SK - 1) / BSG_SK;
  return nt * ng / 2 + 1;
}

void bsg_host_slots(BsG &g, void *slots, void *big, size_t big_bytes) {
  g.h_big = big; g.h_big_bytes = big_bytes;
  double *d = (double *)slots;
  g.h_bounds = d; g.h_ab = d + 2; g.h_meta = (int *)(d + 2 + 64);
  for (int q = 0; q < 2 + 64; ++q) d[q] = 0.0;
  for (int q = 0; q < BSG_META; ++q) g.h_meta[q] = 0;
}

size_t bsg_workspace_bytes(int s, int b) {   // what bsg_carve takes, measured by carving at address 0
  BsG g;
  char *p = nullptr;
  bsg_carve(g, p, s, b);
  return (size_t)(p - (char *)nullptr) + 256;
}

void bsg_carve(BsG &g, char *&p, int s, int b) {
  auto take = [&](size_t bytes) { char *q = p; p += al(bytes); return q; };
  const size_t nt = (s + BSG_TM - 1) / BSG_TM, ng = (s + BSG_SK - 1) / BSG_SK;
  g.s = s; g.ntile = (int)nt; g.nstage = (int)ng; g.nbt = (b + BSG_TN - 1) / BSG_TN;
  g.csr_cap = bsg_csr_cap(s); g.rem_cap = g.csr_cap / 4 + 64; g.pack_cap = bsg_pack_cap(s);
  g.gptr = (int *)take(sizeof(int) * (s + 1)); g.gcol = (int *)take(sizeof(int) * g.csr_cap);
  g.gval = (double *)take(sizeof(double) * g.csr_cap);
  // C, meta and the labels are read back by the host in one copy: adjacent, in this order
  g.Cw = (double *)take(sizeof(double) * BSG_SEEDS * BSG_SEEDS + sizeof(int) * (BSG_META + (size_t)s));
  g.meta = (int *)(g.Cw + BSG_SEEDS * BSG_SEEDS);
  g.lab = g.meta + BSG_META;
  int *lab2 = (int *)take(sizeof(int) * (size_t)s);
  g.perm = (int *)take(sizeof(int) * (size_t)s); g.iperm = (int *)take(sizeof(int) * (size_t)s);
  g.rcnt = lab2;   // the second label buffer is free again once the ordering is done; (s + 1 ints are not needed: s suffice)
  // (rows rounded up to whole chunks of BSG_CW_ROWS: E0 later holds one 64 x 64 table per chunk, and with the exact size the
  //  last chunk's table ran 28 KB into E1 at s = 5000 -- the rows other workgroups of the same launch were still reading: the
  //  cluster weights, hence the ordering, depended on timing until round 3 made the Lanczos steps truly concurrent and it showed)
  const size_t rows_up = ((size_t)s + BSG_CW_ROWS - 1) / BSG_CW_ROWS * BSG_CW_ROWS;
  g.E0 = (double *)take(sizeof(double) * rows_up * BSG_SEEDS); g.E1 = (double *)take(sizeof(double) * rows_up * BSG_SEEDS);
  g.cnt = (int *)take(sizeof(int) * nt * ng); g.blkpos = (int *)take(sizeof(int) * nt * ng);
  g.nk = (int *)take(sizeof(int) * (nt + 1)); g.off = (int *)take(sizeof(int) * (nt + 1));
  g.order = (int *)take(sizeof(int) * (nt + 1)); (void)take(sizeof(int) * (nt + 1));
  g.klist = (int *)take(sizeof(int) * nt * ng);
  {
    const size_t nbt = (size_t)(b + BSG_TN - 1) / BSG_TN;
    g.part_max = g.pack_cap / 8 + nt;
    g.slab_cap = std::min(g.part_max, (size_t)(256u << 20) / (nbt * BSG_TM * BSG_TN * sizeof(double)));
    g.slab_cap = std::min(g.slab_cap, (size_t)1024);
    g.parts = (int *)take(sizeof(int) * 4 * g.part_max); g.parts_u = (int *)take(sizeof(int) * 4 * g.part_max);
    g.slab0 = (int *)take(sizeof(int) * (nt + 1));
    g.tcnt = (int *)take(sizeof(int) * (nt * nbt + 4));
    g.slab = (double *)take(sizeof(double) * g.slab_cap * nbt * BSG_TM * BSG_TN);
  }
  g.pack = (double *)take(sizeof(double) * BSG_BLK * g.pack_cap);
  g.rptr = (int *)take(sizeof(int) * (s + 1)); (void)take(sizeof(int) * (s + 1));
  g.rcol = (int *)take(sizeof(int) * g.rem_cap); g.rval = (double *)take(sizeof(double) * g.rem_cap);
  g.colabs = (double *)take(sizeof(double) * (size_t)s); g.diag = (double *)take(sizeof(double) * (size_t)s);
  g.bounds = (double *)take(sizeof(double) * 2);
  g.head = (int *)take(sizeof(int) * 4);
  g.lz = (double *)take(sizeof(double) * (3 * (size_t)s + (size_t)(s + 3) / 4 + 2 * 32 + 8));
  for (int q = 0; q < 3; ++q) g.T[q] = (double *)take(sizeof(double) * (size_t)s * b);
}

int bsg_setup(hipStream_t st, const double *dG, int ldg, int s, BsG &g, hipStream_t side, hipEvent_t side_ev) {
  g.built = false; g.on = false; g.lanczos = false;
  const int rows4 = ceil_div(s, 4);
  int *lab2 = g.rcnt;
  FLGP_HIP(hipMemsetAsync(g.meta, 0, sizeof(int) * BSG_META, st));
  hipLaunchKernelGGL(bsg_scan_kernel, dim3(rows4), dim3(256), 0, st, dG, ldg, s, g.rcnt, g.colabs, g.diag);   // counts -> rcnt (free yet)
  hipLaunchKernelGGL(bsg_ptr_kernel, dim3(1), dim3(1024), 0, st, g.rcnt, s, g.gptr, g.meta, (int)BSG_M_NNZ, (long)g.csr_cap);
  hipLaunchKernelGGL(bsg_csr_fill_kernel, dim3(rows4), dim3(256), 0, st, dG, ldg, s, g.gptr, g.meta, g.gcol, g.gval);
  hipLaunchKernelGGL(bsg_bounds_kernel, dim3(1), dim3(256), 0, st, g.colabs, g.diag, s, g.bounds);
  FLGP_TRY(check_launch("bsg csr"));
  FLGP_HIP(hipMemcpyAsync(g.h_bounds, g.bounds, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
  // (round 4: the ordering's launches are ENQUEUED before the Lanczos run's 41 -- the host needs 0.2-0.5 ms for those, and the
  //  ordering, which the host then waits for, used to sit behind them; the event that releases the second stream is recorded first)
  const bool lz = side && side_ev && tuning("eig_lanczos_lo", 1);
  if (lz) FLGP_HIP(hipEventRecord(side_ev, st));
  // ordering
  hipLaunchKernelGGL(bsg_seed_gather_kernel, dim3(ceil_div((long)s * BSG_SEEDS, 256)), dim3(256), 0, st, dG, ldg, s, g.E0);
  hipLaunchKernelGGL(bsg_hop_kernel, dim3(rows4), dim3(256), 0, st, g.gptr, g.gcol, g.gval, g.meta, g.E0, s, g.E1, (int *)nullptr);
  hipLaunchKernelGGL(bsg_hop_kernel, dim3(rows4), dim3(256), 0, st, g.gptr, g.gcol, g.gval, g.meta, g.E1, s, g.E0, g.lab);
  hipLaunchKernelGGL(bsg_label_hop_kernel, dim3(rows4), dim3(256), 0, st, g.gptr, g.gcol, g.gval, g.meta, g.lab, s, lab2,
                     (double *)nullptr);
  hipLaunchKernelGGL(bsg_label_hop_kernel, dim3(rows4), dim3(256), 0, st, g.gptr, g.gcol, g.gval, g.meta, lab2, s, g.lab,
                     (double *)nullptr);
  hipLaunchKernelGGL(bsg_label_hop_kernel, dim3(rows4), dim3(256), 0, st, g.gptr, g.gcol, g.gval, g.meta, g.lab, s,
                     (int *)nullptr, g.E1);
  {
    const int nchunk = ceil_div(s, BSG_CW_ROWS);     // chunk tables: s/64 x 4096 doubles = the size of E0, free by now
    hipLaunchKernelGGL(bsg_cluster_partial_kernel, dim3(nchunk), dim3(64), 0, st, g.E1, g.lab, g.meta, s, g.E0);
    hipLaunchKernelGGL(bsg_cluster_sum_kernel, dim3(BSG_SEEDS * BSG_SEEDS / 256), dim3(256), 0, st, g.E0, nchunk, g.meta, g.Cw);
  }
  FLGP_TRY(check_launch("bsg ordering"));
  const int p = BSG_SEEDS;
  // one copy for the three things the host needs: C (p x p doubles), meta, labels are adjacent in the workspace
  const size_t xfer_doubles = (size_t)p * p + (BSG_META * sizeof(int) + sizeof(int) * (size_t)s + 7) / 8 + 1;
  const bool big_ok = g.h_big && g.h_big_bytes >= sizeof(double) * xfer_doubles + sizeof(int) * (size_t)s;
  std::vector<double> xfer_own(big_ok ? 0 : xfer_doubles);
  double *xfer_p = big_ok ? (double *)g.h_big : xfer_own.data();
  struct { double *p; double *data() const { return p; } } xfer{xfer_p};
  FLGP_HIP(hipMemcpyAsync(xfer.data(), g.Cw, sizeof(double) * p * p + sizeof(int) * (BSG_META + (size_t)s), hipMemcpyDeviceToHost, st));
  if (lz) {
    // lambda_min estimate on the second stream, beside the ordering and the first iterations
    double *v = g.lz, *vp = g.lz + s, *w = g.lz + 2 * (size_t)s, *part = g.lz + 3 * (size_t)s, *ab = part + (s + 3) / 4;
    FLGP_HIP(hipStreamWaitEvent(side, side_ev, 0));
    hipLaunchKernelGGL(bsg_lz_start_kernel, dim3(ceil_div(s, 256)), dim3(256), 0, side, v, vp, s);
    for (int k = 0; k < BSG_LANCZOS; ++k) {
      hipLaunchKernelGGL(bsg_lz_spmv_kernel, dim3(rows4), dim3(256), 0, side, g.gptr, g.gcol, g.gval, g.meta, v, vp, ab, k, s, w, part);
      hipLaunchKernelGGL(bsg_lz_update_kernel, dim3(1), dim3(1024), 0, side, part, rows4, s, g.meta, w, v, vp, ab, k);
    }
    FLGP_TRY(check_launch("bsg lanczos"));
    FLGP_HIP(hipMemcpyAsync(g.h_ab, ab, sizeof(double) * 2 * BSG_LANCZOS, hipMemcpyDeviceToHost, side));
    FLGP_HIP(hipEventRecord(side_ev, side));
    g.lanczos = true; g.lz_ev = side_ev;
  }
  FLGP_HIP(stream_wait(st));
  const double *C = xfer.data();
  memcpy(g.h_meta, xfer.data() + (size_t)p * p, sizeof(int) * BSG_META);
  const int *lab = (const int *)(xfer.data() + (size_t)p * p) + BSG_META;
  for (int q = 0; q < p * p; ++q)
    if (!std::isfinite(C[q])) { set_error("block-sparse ordering: the matrix is not finite"); return FLGP_ERR_INVALID; }
  if (g.h_meta[BSG_M_OFF]) return FLGP_OK;     // denser than the CSR allows: the caller stays with the dense products
  // chain the clusters: start at the heaviest, always continue with the unused cluster most strongly tied to the last
  std::vector<int> order, rank(p + 1, p);
  std::vector<char> used(p, 0);
  {
    int start = 0; double best = -1.0;
    for (int q = 0; q < p; ++q) {
      double w = 0.0;
      for (int q2 = 0; q2 < p; ++q2) if (q2 != q) w += std::fabs(C[(size_t)q2 * p + q]);
      if (w > best) { best = w; start = q; }
    }
    order.push_back(start); used[start] = 1;
    while ((int)order.size() < p) {
      const int cur = order.back();
      int nxt = -1; double bw = -1.0;
      for (int q = 0; q < p; ++q)
        if (!used[q] && std::fabs(C[(size_t)q * p + cur]) > bw) { bw = std::fabs(C[(size_t)q * p + cur]); nxt = q; }
      order.push_back(nxt); used[nxt] = 1;
    }
    for (int q = 0; q < p; ++q) rank[order[q]] = q;   // label p (isolated anchors) keeps rank p: last
  }
  std::vector<int> &perm = g.h_perm;
  perm.assign(s, 0);
  std::vector<int> start(p + 2, 0);
  for (int i = 0; i < s; ++i) {
    if (lab[i] < 0 || lab[i] > p) { set_error("block-sparse ordering: bad label"); return FLGP_ERR_HIP; }
    ++start[rank[lab[i]] + 1];
  }
  for (int q = 0; q <= p; ++q) start[q + 1] += start[q];
  for (int i = 0; i < s; ++i) perm[start[rank[lab[i]]]++] = i;   // stable counting sort by cluster rank
  const int *perm_src = perm.data();
  if (big_ok) {                                  // upload from the pinned area (behind the part the download used)
    int *pp = (int *)((char *)g.h_big + sizeof(double) * xfer_doubles);
    memcpy(pp, perm.data(), sizeof(int) * (size_t)s);
    perm_src = pp;
  }
  FLGP_HIP(hipMemcpyAsync(g.perm, perm_src, sizeof(int) * s, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(bsg_iperm_kernel, dim3(ceil_div(s, 256)), dim3(256), 0, st, g.perm, s, g.iperm);
  hipLaunchKernelGGL(bsg_count_kernel, dim3(g.ntile), dim3(256), sizeof(int) * g.nstage, st, g.gptr, g.gcol, g.perm, g.iperm, s,
                     g.nstage, g.cnt);
  FLGP_HIP(hipMemsetAsync(g.tcnt, 0, sizeof(int) * ((size_t)g.ntile * g.nbt + 4), st));
  const int part_cap = tuning("eig_bs_part_cap", 0) < 8 ? 0 : tuning("eig_bs_part_cap", 0);
  hipLaunchKernelGGL(bsg_lists_kernel, dim3(1), dim3(1024), 0, st, g.cnt, g.ntile, g.nstage, (long)g.pack_cap, g.blkpos,
                     g.nk, g.off, g.klist, g.order, g.meta, part_cap, (int)g.part_max, (int)g.slab_cap, (int4 *)g.parts_u,
                     (int4 *)g.parts, g.slab0);
  hipLaunchKernelGGL(bsg_pack_kernel, dim3(g.nstage, g.ntile), dim3(256), 0, st, dG, ldg, s, g.nstage, g.perm, g.blkpos,
                     g.meta, g.pack);
  hipLaunchKernelGGL(bsg_rem_kernel, dim3(rows4), dim3(256), 0, st, g.gptr, g.gcol, g.gval, g.perm, g.iperm, g.blkpos, g.meta,
                     s, g.nstage, (const int *)nullptr, g.rcnt, (int *)nullptr, (double *)nullptr, 0);
  hipLaunchKernelGGL(bsg_ptr_kernel, dim3(1), dim3(1024), 0, st, g.rcnt, s, g.rptr, g.meta, (int)BSG_M_RNNZ, (long)g.rem_cap);
  hipLaunchKernelGGL(bsg_rem_kernel, dim3(rows4), dim3(256), 0, st, g.gptr, g.gcol, g.gval, g.perm, g.iperm, g.blkpos, g.meta,
                     s, g.nstage, g.rptr, g.rcnt, g.rcol, g.rval, 1);
  FLGP_TRY(check_launch("bsg lists"));
  FLGP_HIP(hipMemcpyAsync(g.h_meta, g.meta, sizeof(int) * BSG_META, hipMemcpyDeviceToHost, st));
  g.built = true;       // (g.h_perm, the source of the copy above, lives as long as g)
  g.launches = 0;
  return FLGP_OK;
}

void bsg_finish(BsG &g) {
  g.on = false; g.lambda_lo = 0.0;
  if (!g.built || g.h_meta[BSG_M_OFF]) return;
  if (g.lanczos && hipEventSynchronize(g.lz_ev) == hipSuccess) {
    // Ritz values of a Krylov space lie inside the spectrum: theta_min >= lambda_min, approaching it from above.  What is
    // used is the last estimate less three times its last improvement (15 -> 20 steps): at BASELINE configs[2]
    // 0.114 - 3 x 0.004 = 0.10 against lambda_min = 0.1133 (scripts/model_chfsi.py, lo=...).  eig.hip watches the
    // Ritz values of its block for anything this bound would have let grow and drops it if it sees some.
    bool ok = true;
    for (int q = 0; q < 2 * BSG_LANCZOS; ++q) ok = ok && std::isfinite(g.h_ab[q]);
    if (ok && g.h_ab[2 * (BSG_LANCZOS - 1) + 1] >= 0.0) {
      const double t15 = tridiag_min(g.h_ab, 15), t20 = tridiag_min(g.h_ab, BSG_LANCZOS);
      double lo = t20 - 3.0 * std::max(0.0, t15 - t20);
      lo = std::min(lo, 0.9 * t20);
      g.lambda_lo = lo > 0.0 ? lo : 0.0;
      if (tuning("eig_verbose", 0)) fprintf(stderr, "[flgp eig] Lanczos lambda_min estimates: %.6g (15 steps), %.6g (20) -> lower end %.6g\n", t15, t20, g.lambda_lo);
    }
  }
  const double frac = (double)g.h_meta[BSG_M_TOTAL] / ((double)g.ntile * g.nstage);
  if (tuning("eig_verbose", 0))
    fprintf(stderr, "[flgp eig] block-sparse G: %d non-zeros, %.1f %% of the %dx%d blocks kept for the MFMA product (max %d of %d stages "
            "per tile), %d scattered non-zeros\n", g.h_meta[BSG_M_NNZ], 100.0 * frac, BSG_TM, BSG_SK, g.h_meta[BSG_M_MAXNK], g.nstage,
            g.h_meta[BSG_M_RNNZ]);
  if (tuning("eig_verbose", 0) > 1) {     // the remainder: entries per row, and per tile
    std::vector<int> rp(g.s + 1);
    if (hipMemcpy(rp.data(), g.rptr, sizeof(int) * (g.s + 1), hipMemcpyDeviceToHost) == hipSuccess) {
      int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const int edge[8] = {0, 4, 8, 16, 32, 64, 128, 1 << 30};
      for (int i = 0; i < g.s; ++i) { const int n = rp[i + 1] - rp[i]; int q = 0; while (n > edge[q]) ++q; ++hist[q]; }
      fprintf(stderr, "[flgp eig] remainder rows by entries: 0:%d  1-4:%d  5-8:%d  9-16:%d  17-32:%d  33-64:%d  65-128:%d  >128:%d\n",
              hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
      fprintf(stderr, "[flgp eig] remainder entries per tile:");
      for (int t = 0; t < g.ntile; ++t) fprintf(stderr, " %d", rp[std::min(g.s, (t + 1) * BSG_TM)] - rp[t * BSG_TM]);
      fprintf(stderr, "\n");
    }
  }
  if (tuning("eig_verbose", 0) > 1) {     // list lengths of the tiles, longest first
    std::vector<int> nk(g.ntile);
    if (hipMemcpy(nk.data(), g.nk, sizeof(int) * g.ntile, hipMemcpyDeviceToHost) == hipSuccess) {
      std::sort(nk.begin(), nk.end(), [](int a, int b) { return a > b; });
      fprintf(stderr, "[flgp eig] kept stages per tile:");
      for (int v : nk) fprintf(stderr, " %d", v);
      fprintf(stderr, "\n");
    }
  }
  g.on = frac <= 0.01 * std::min(50, tuning("eig_bs_max_pct", 50));
}

static long long *g_bsg_trace = nullptr;     // diagnostic: 4 words per task of the product kernel (see flgp_dev_bsg_set_trace)

int bsg_product(hipStream_t st, BsG &g, const double *Xt, int b, double alpha, double beta, const double *Et, double gamma,
                const double *E2t, double *out_t) {
  const int s = g.s;
  if (beta == 0.0) Et = nullptr;
  if (gamma == 0.0) E2t = nullptr;
  int *head = g.head;
  {
    ProfScope ps("bsg_pre_kernel", st, 24.0 * (double)s * b);
    hipLaunchKernelGGL(bsg_pre_kernel, dim3(s), dim3(256), 0, st, g.rptr, g.rcol, g.rval, Xt, b, alpha, beta, Et, gamma, E2t,
                       out_t, head);
  }
  FLGP_TRY(check_launch("bsg_pre_kernel"));
  const int nbt = ceil_div(b, BSG_TN);
  if (nbt != g.nbt) { set_error("bsg_product: block width differs from the one the workspace was carved for"); return FLGP_ERR_INVALID; }
  const int npart = g.h_meta[BSG_M_NPART];
  const int ntask = npart * nbt;
  int grid = std::max(64, tuning("eig_bs_wgs", 256));
  int xmap = tuning("eig_bs_xmap", 1);
  if (xmap) {                                       // the layout of the fixed tasks needs (grid / 8) % nbt == 0
    const int unit = 8 * nbt;
    if (grid >= unit) grid = grid / unit * unit; else xmap = 0;
  }
  if (!xmap) grid = std::min(grid, ntask);
  {
    const double fl = 2.0 * (double)BSG_BLK * (double)g.h_meta[BSG_M_TOTAL] * (double)b;
    ProfScope ps("bsg_gemm_kernel", st, fl);
    hipLaunchKernelGGL(bsg_gemm_kernel, dim3(grid), dim3(512), 0, st, Xt, out_t, b, s, alpha, g.order, g.nk, g.off, g.klist,
                       g.pack, g.ntile, nbt, head, (const int4 *)g.parts, npart, g.slab0, g.slab, g.tcnt, xmap, g_bsg_trace);
  }
  FLGP_TRY(check_launch("bsg_gemm_kernel"));
  ++g.launches;
  return FLGP_OK;
}

}  // namespace flgp

using namespace flgp;

// tiled transposes between the solver's s x b column-major blocks and the b x s layout of the products
__global__ void bsg_to_t_kernel(const double *__restrict__ in, int s, int b, const int *__restrict__ perm,
                                double *__restrict__ out_t) {   // out_t(c, i') = in(perm[i'], c)
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)s * b) return;
  const int c = (int)(e % b), ip = (int)(e / b);
  out_t[e] = in[(size_t)c * s + perm[ip]];
}
__global__ void bsg_from_t_kernel(const double *__restrict__ in_t, int s, int b, const int *__restrict__ perm,
                                  double *__restrict__ out) {   // out(perm[i'], c) = in_t(c, i')
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)s * b) return;
  const int c = (int)(e % b), ip = (int)(e / b);
  out[(size_t)c * s + perm[ip]] = in_t[e];
}

extern "C" size_t flgp_dev_bsg_workspace(int s, int b) { return bsg_workspace_bytes(s, b) + 1024; }

extern "C" void flgp_dev_bsg_set_trace(void *d_trace) { flgp::g_bsg_trace = (long long *)d_trace; }

extern "C" int flgp_dev_bsg_apply(void *stream, const double *dG, int ldg, int s, const double *dX, int b, double alpha,
                                  double beta, const double *dE, double *dOut, void *d_work, size_t work_bytes,
                                  int *info) {
  hipStream_t st = (hipStream_t)stream;
  FLGP_REQUIRE(s >= 64 && b >= 16 && b % 16 == 0 && ldg >= s, "bsg_apply: need s >= 64, b a multiple of 16");
  FLGP_REQUIRE(work_bytes >= bsg_workspace_bytes(s, b), "bsg_apply: workspace too small");
  BsG g;
  char *p = (char *)d_work;
  bsg_carve(g, p, s, b);
  FLGP_TRY(bsg_setup(st, dG, ldg, s, g));
  FLGP_HIP(hipStreamSynchronize(st));
  bsg_finish(g);
  if (info) {
    info[0] = g.on ? 1 : 0; info[1] = g.h_meta[BSG_M_NNZ]; info[2] = g.h_meta[BSG_M_TOTAL]; info[3] = g.h_meta[BSG_M_RNNZ];
    info[4] = g.h_meta[BSG_M_NPART]; info[5] = g.h_meta[BSG_M_NSLAB];
  }
  if (!g.built || g.h_meta[BSG_M_OFF]) { set_error("bsg_apply: the matrix is too dense for the block-sparse product"); return FLGP_ERR_INVALID; }
  const long tot = (long)s * b;
  hipLaunchKernelGGL(bsg_to_t_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, st, dX, s, b, g.perm, g.T[0]);
  if (dE) hipLaunchKernelGGL(bsg_to_t_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, st, dE, s, b, g.perm, g.T[1]);
  FLGP_TRY(bsg_product(st, g, g.T[0], b, alpha, beta, dE ? g.T[1] : nullptr, 0.0, nullptr, g.T[2]));
  hipLaunchKernelGGL(bsg_from_t_kernel, dim3(ceil_div(tot, 256)), dim3(256), 0, st, g.T[2], s, b, g.perm, dOut);
  FLGP_TRY(check_launch("bsg_apply"));
  FLGP_HIP(hipStreamSynchronize(st));
  return FLGP_OK;
}
