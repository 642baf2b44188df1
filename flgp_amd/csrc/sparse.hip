// k5 and the sparse half of k6: everything that touches the n x s similarity matrix Z / A in
// its ELL form (row-major idx[n][r], val[n][r], rows sorted by column).
//
//   - CSC view of the pattern by a stable counting sort (deterministic, rows ascending);
//   - column sums in row-ascending order == `RowVectorXd::Ones(n) * Z` on a row-major sparse
//     matrix (reference src/Utils.cpp:200,203, src/Spectrum.cpp:149), bit-exact vs the oracle;
//   - column / row scalings of graphLaplacian_cpp (src/Utils.cpp:195-212) and of
//     spectrum_from_Z_cpp (src/Spectrum.cpp:150);
//   - Gram matrix G = A^T A in a fixed summation order (the s x s operator RSpectra::svds
//     iterates on, src/TruncatedSVD.cpp:23-28);
//   - u_k = A v_k / sigma_k * sqrt(n) (svds' left vectors + src/Spectrum.cpp:157-158).
//
// All of these are HBM / latency bound integer-and-fp64 streaming work: no MFMA here.
#include "common.h"

namespace flgp {

// ----------------------------------------------------------------------------------------
// CSC build: stable counting sort of the nnz = n*r entries by column.
//   pass 1  per-chunk column histograms (LDS int atomics: counts are order independent)
//   pass 2  per column: exclusive prefix over chunks, column totals
//   pass 3  exclusive scan of the totals -> colptr
//   pass 4  one wave per chunk walks its entries in order, 64 at a time; lanes that hit the
//           same column in one step are ranked by lane id (ballot-built match mask), so the
//           order inside a column is exactly the entry order == ascending row.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csc_hist_kernel(const int *__restrict__ ell_idx, long nnz, int s,
                                                       int chunk, int *__restrict__ hist) {
  extern __shared__ int lh[];
  for (int j = threadIdx.x; j < s; j += blockDim.x) lh[j] = 0;
  __syncthreads();
  const long e0 = (long)blockIdx.x * chunk;
  long e1 = e0 + chunk;
  if (e1 > nnz) e1 = nnz;
  for (long e = e0 + threadIdx.x; e < e1; e += blockDim.x) atomicAdd(&lh[ell_idx[e]], 1);
  __syncthreads();
  int *out = hist + (size_t)blockIdx.x * s;
  for (int j = threadIdx.x; j < s; j += blockDim.x) out[j] = lh[j];
}

__global__ void csc_colscan_kernel(int *__restrict__ hist, int nchunks, int s, int *__restrict__ tot) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= s) return;
  int run = 0;
  // 16 loads in flight per thread: one load, one store, one add at a time (the compiler cannot move a load above
  // the preceding store to the same array) made each of the ~2000 steps a full memory round trip
  constexpr int CB = 16;
  int c = 0;
  for (; c + CB <= nchunks; c += CB) {
    int h[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) h[u] = hist[(size_t)(c + u) * s + j];
#pragma unroll
    for (int u = 0; u < CB; ++u) { hist[(size_t)(c + u) * s + j] = run; run += h[u]; }
  }
  for (; c < nchunks; ++c) {
    const int h = hist[(size_t)c * s + j];
    hist[(size_t)c * s + j] = run;
    run += h;
  }
  tot[j] = run;
}

// single-block exclusive scan of tot[0..s) -> colptr[0..s]
__global__ __launch_bounds__(1024) void csc_scan_kernel(const int *__restrict__ tot, int s, int *__restrict__ colptr) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (s + 1023) / 1024;
  const int lo = t * per, hi = (lo + per < s) ? lo + per : s;
  int sum = 0;
  for (int j = lo; j < hi; ++j) sum += tot[j];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = (t == 0) ? 0 : part[t - 1];
  for (int j = lo; j < hi; ++j) { colptr[j] = run; run += tot[j]; }
  if (t == 1023) colptr[s] = part[1023];
}

__global__ __launch_bounds__(64) void csc_scatter_kernel(const int *__restrict__ ell_idx, long nnz, int s,
                                                         int chunk, int nbits, const int *__restrict__ hist,
                                                         const int *__restrict__ colptr, int *__restrict__ pos) {
  extern __shared__ int cur[];  // running count per column inside this chunk
  const int lane = threadIdx.x;
  for (int j = lane; j < s; j += 64) cur[j] = 0;
  __syncthreads();
  const int *hoff = hist + (size_t)blockIdx.x * s;
  const long e0 = (long)blockIdx.x * chunk;
  long e1 = e0 + chunk;
  if (e1 > nnz) e1 = nnz;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (long eb = e0; eb < e1; eb += 64) {
    const long e = eb + lane;
    const bool act = e < e1;
    const int col = act ? ell_idx[e] : 0;
    unsigned long long m = __ballot(act);
    for (int b = 0; b < nbits; ++b) {
      const bool bit = (col >> b) & 1;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    if (act) {
      const int rank = __popcll(m & lt);
      const int base = cur[col];
      pos[(size_t)colptr[col] + hoff[col] + base + rank] = (int)e;
      if (rank == 0) cur[col] = base + __popcll(m);  // one writer per column group; all read before (same instr)
    }
    __syncthreads();  // single wave: orders the LDS read-modify-write between steps
  }
}

// Column sums in the fixed two-level order of the oracle (oracle/flgp_oracle.c, flgp_oracle_colsum): chunks of
// COLSUM_CHUNK consecutive rows; inside a chunk every column's entries are added in row order, then the chunk totals
// in chunk order.  One wave owns a chunk and a table of s sums in LDS, streams the chunk's entries 64 at a time
// (coalesced: the round-1 kernel walked the CSC lists instead and fetched a 128-byte line for every 8-byte value,
// 1.2 GB for 80 MB) and adds them with ds_add_f64.  Lanes of one step that hit the same column are ranked by lane
// (= entry order) with the ballot match of csc_scatter_kernel and applied in as many passes as the largest rank
// needs; the LDS unit executes a wave's instructions in order, so pass k+1 sees pass k.
constexpr int COLSUM_CHUNK = 1024;     // rows; FLGP_COLSUM_CHUNK of the oracle

// WIN: the table holds the columns [w0, w0 + wn) of window blockIdx.y only (s beyond what LDS holds: the chunk is walked once
// per window, entries of other windows are skipped -- the order inside a column is untouched)
template <bool WIN>
__global__ __launch_bounds__(64) void colsum_chunk_kernel(const int *__restrict__ ell_idx, const double *__restrict__ val,
                                                          int n, int r, int s, int nbits, double *__restrict__ part, int wmax,
                                                          int direct) {
  extern __shared__ double bins[];
  const int lane = threadIdx.x;
  const int w0 = WIN ? (int)blockIdx.y * wmax : 0;
  const int wn = WIN ? ((s - w0 < wmax) ? s - w0 : wmax) : s;
  for (int j = lane; j < wn; j += 64) bins[j] = 0.0;
  __syncthreads();
  const long i0 = (long)blockIdx.x * COLSUM_CHUNK;
  const long i1 = (i0 + COLSUM_CHUNK < n) ? i0 + COLSUM_CHUNK : n;
  const long e0 = i0 * r, e1 = i1 * r;
  if (direct) {
    // ONE ds_add_f64 per step of 64 entries: lanes that hit the same column are applied by the LDS unit in ascending lane
    // order, which is the entries' order -- the rule gram_kernel's row groups rest on (bit-exact there and here on all 1e6
    // rows of configs[2]); the LDS unit executes a wave's instructions in order, so step follows step.  Eight steps' loads
    // are in flight (with one the wave -- alone on its SIMD: 977 chunks on 1024 SIMDs -- waited a memory latency per step).
    constexpr int CU = 8;
    for (long eb = e0; eb < e1; eb += 64 * CU) {
      int cq[CU];
      double vq[CU];
#pragma unroll
      for (int u = 0; u < CU; ++u) {
        const long e = eb + 64 * u + lane;
        const long ec = e < e1 ? e : e1 - 1;             // (unconditional loads: a valid address, masked below)
        cq[u] = ell_idx[ec];
        vq[u] = val[ec];
      }
#pragma unroll
      for (int u = 0; u < CU; ++u) {
        const bool on = eb + 64 * u + lane < e1 && (!WIN || (unsigned)(cq[u] - w0) < (unsigned)wn);
        const int bin = on ? cq[u] - w0 : 0;
        if (on) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)&bins[bin], vq[u]);
      }
    }
    __syncthreads();
    double *outd = part + (size_t)blockIdx.x * s + w0;
    for (int j = lane; j < wn; j += 64) outd[j] = bins[j];
    return;
  }
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // the next step's entries are loaded while this step's are ranked and added
  long e = e0 + lane;
  int col = (e < e1) ? ell_idx[e] : 0;
  double v = (e < e1) ? val[e] : 0.0;
  for (long eb = e0; eb < e1; eb += 64) {
    const long en = eb + 64 + lane;
    const int coln = (en < e1) ? ell_idx[en] : 0;
    const double vn = (en < e1) ? val[en] : 0.0;
    const bool act = eb + lane < e1 && (!WIN || (unsigned)(col - w0) < (unsigned)wn);
    unsigned long long m = __ballot(act);
    for (int b = 0; b < nbits; ++b) {
      const bool bit = (col >> b) & 1;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const int rank = act ? __popcll(m & lt) : -1;
    // pass k adds the lanes of rank k; the ranks of a column's lanes are 0, 1, 2, ... without gaps, so the first k that
    // nobody holds ends the step (a ballot per pass; the wave-wide maximum by six dependent shuffles that stood here was
    // most of the step: ~700 of its ~1200 cycles)
    for (int k = 0; __ballot(rank == k) != 0ull; ++k)
      if (rank == k)
        __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)&bins[col - w0], v);
    col = coln; v = vn;
  }
  __syncthreads();
  double *out = part + (size_t)blockIdx.x * s + w0;
  for (int j = lane; j < wn; j += 64) out[j] = bins[j];
}

// colsum[j] = chunk totals added in chunk order, from 0.0
__global__ void colsum_reduce_kernel(const double *__restrict__ part, int nchunks, int s, double *__restrict__ colsum) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= s) return;
  double acc = 0.0;
  int c = 0;
  for (; c + 32 <= nchunks; c += 32) {    // 32 loads in flight, added in order (s / 256 workgroups: a few waves on the whole chip,
    double p[32];                         //  every batch a memory latency -- with eight in flight 46 us for 977 chunks)
#pragma unroll
    for (int u = 0; u < 32; ++u) p[u] = part[(size_t)(c + u) * s + j];
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += p[u];
  }
  for (; c + 8 <= nchunks; c += 8) {
    double p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = part[(size_t)(c + u) * s + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += p[u];
  }
  for (; c < nchunks; ++c) acc += part[(size_t)c * s + j];
  colsum[j] = acc;
}

__global__ void col_scale_kernel(const int *__restrict__ ell_idx, double *__restrict__ val, long nnz,
                                 const double *__restrict__ colsum, const double *__restrict__ num_class,
                                 int mode) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  const int j = ell_idx[e];
  const double c = colsum[j];
  double v = val[e];
  if (mode == 0) {
    v = v * (1.0 / (c + 1e-9));                    // src/Utils.cpp:201,204
    if (num_class) v = v * num_class[j];           // src/Utils.cpp:205
  } else {
    v = v * (1.0 / __builtin_sqrt(__builtin_fabs(c) + 1e-9));  // src/Spectrum.cpp:150
  }
  val[e] = v;
}

// 256 rows per workgroup, staged through LDS: the rows are r doubles apart, so a thread walking its own row in
// global memory touched a different cache line than its neighbours on every load (measured: 4x the bytes fetched and
// written); the block's 256 r values are one contiguous run, read and written back coalesced.
// With `idx` the column scaling of col_scale_kernel's mode 0 is applied on the way in (the same two or three rounded
// multiplications per entry, then the same row sums): graphLaplacian_cpp's two passes over the values in one.
__global__ __launch_bounds__(256) void row_normalize_kernel(double *__restrict__ val, int n, int r,
                                                            const int *__restrict__ idx = nullptr,
                                                            const double *__restrict__ colsum = nullptr,
                                                            const double *__restrict__ num_class = nullptr) {
  extern __shared__ double rn_rows[];
  const long i0 = (long)blockIdx.x * 256;
  const int rows = (n - i0 < 256) ? (int)(n - i0) : 256;
  double *g = val + (size_t)i0 * r;
  const int cnt = rows * r;
  if (idx) {
    const int *gi = idx + (size_t)i0 * r;
    for (int e = threadIdx.x; e < cnt; e += 256) {
      const int j = gi[e];
      double v = g[e];
      v = v * (1.0 / (colsum[j] + 1e-9));              // src/Utils.cpp:201,204
      if (num_class) v = v * num_class[j];             // src/Utils.cpp:205
      rn_rows[e] = v;
    }
  } else
  for (int e = threadIdx.x; e < cnt; e += 256) rn_rows[e] = g[e];
  __syncthreads();
  if ((int)threadIdx.x < rows) {
    double *row = rn_rows + (size_t)threadIdx.x * r;
    double rs = 0.0;
    for (int a = 0; a < r; ++a) rs += row[a];        // ascending column order (src/Utils.cpp:210)
    const double inv = 1.0 / (rs + 1e-9);
    for (int a = 0; a < r; ++a) row[a] = inv * row[a];  // src/Utils.cpp:211
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cnt; e += 256) g[e] = rn_rows[e];
}

// ----------------------------------------------------------------------------------------
// Gram: one wave per column j1.  The wave owns row j1 of G in LDS (s doubles) and walks the column's entries in
// ascending row order: every G(j1, j2) is summed in ascending row order -- deterministic, and identical to the oracle.
// A row's r products go to r distinct bins; consecutive rows of a column share anchors (they all have j1, and usually
// more).  One ds_add_f64 serves a group of 64 / LPR rows: lanes that hit the same bin are applied in ascending lane order
// (= ascending row), and the LDS unit executes a wave's instructions in order, so the sums stay sequential without a
// round trip through registers (see the note at `apply` below).
//
// Where the time goes (round 2, scripts/ubench_ldsatomic.hip): a ds_add_f64 costs ~60 cycles per instruction per
// wave whatever the number of active lanes, and four waves per CU get four times that throughput -- 2000 rows per
// column, 5000 columns, 1024 resident waves: 0.25 ms.  The round-1 kernel took 2.4 ms because each step of six rows
// first chased pos -> entry -> row (an integer division and two dependent gathers) with a single step of look-ahead.
// Here LPR lanes serve a row (64 / LPR rows per group), the loads of GRAM_PF groups are in flight while the previous
// GRAM_PF groups are applied, the next 64 positions are fetched a block ahead, and entry / r is a multiplication.
// ----------------------------------------------------------------------------------------
constexpr int GRAM_PF = 4;      // groups of rows whose loads are in flight together

// WIN: the wave owns the columns [w0, w0 + wn) of row j1 only (window blockIdx.y); products for other windows are skipped
template <int LPR, bool WIN>
__global__ __launch_bounds__(64) void gram_kernel(const int *__restrict__ ell_idx, const double *__restrict__ val,
                                                  int s, int r, double inv_r, const int *__restrict__ colptr,
                                                  const int *__restrict__ pos, double *__restrict__ G, int ldg, int wmax) {
  extern __shared__ double acc[];
  constexpr int RPG = 64 / LPR;     // rows per group
  constexpr int NG = 64 / RPG;      // groups per block of 64 positions
  const int lane = threadIdx.x;
  const int j1 = blockIdx.x;
  const int w0 = WIN ? (int)blockIdx.y * wmax : 0;
  const int wn = WIN ? ((s - w0 < wmax) ? s - w0 : wmax) : s;
  for (int j = lane; j < wn; j += 64) acc[j] = 0.0;
  __syncthreads();
  const int sub = lane / LPR;       // which row of a group this lane serves
  const int a = lane % LPR;         // slot inside the row
  const bool slot_ok = a < r;
  const int ac = slot_ok ? a : 0;
  const int p0 = colptr[j1], p1 = colptr[j1 + 1];
  int epos = (p0 + lane < p1) ? pos[p0 + lane] : 0;
  for (int pc = p0; pc < p1; pc += 64) {
    const int pn = pc + 64 + lane;
    const int epos_next = (pn < p1) ? pos[pn] : 0;                 // a block ahead
    const int cnt = (p1 - pc < 64) ? p1 - pc : 64;
    struct Grp { int j2; double v, vj; bool act; };
    // loads are unconditional (inactive slots read entry 0 of row 0: a valid address), so that the number in flight
    // does not depend on the path and the compiler's wait counts stay tight
    auto fetch = [&](int g) {
      Grp o;
      const int slot = g * RPG + sub;
      o.act = slot_ok && slot < cnt;
      const int e = __shfl(epos, slot & 63, 64);
      const int ee = o.act ? e : 0;
      const int row = (int)(((double)ee + 0.5) * inv_r);            // == ee / r exactly for ee < 2^31, r <= 32
      const size_t rowbase = (size_t)row * r;
      o.j2 = ell_idx[rowbase + ac];
      o.v = val[rowbase + ac];
      o.vj = val[ee];
      return o;
    };
    // ONE ds_add_f64 per group of RPG rows: lanes of different rows that hit the same bin (bin j1 always, usually more)
    // are applied by the LDS unit in ascending lane order = ascending row, which is the oracle's order.  (Until round 4
    // the source spelled this as RPG instructions, one row each; per thread that is the same program, the compiler had
    // merged them all along, and with a window test in the condition it split them in another order -- 1 ulp off in 0.5 %
    // of the entries.  The lane-order rule is what the bit-exact tests on all 1e6 rows of configs[2] have been checking.)
    auto apply = [&](const Grp &o) {
      const double prod = o.vj * o.v;
      const bool on = o.act && (!WIN || (unsigned)(o.j2 - w0) < (unsigned)wn);
      const int bin = on ? o.j2 - w0 : 0;
      if (on) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)&acc[bin], prod);
    };
    Grp cur[GRAM_PF], nxt[GRAM_PF];
#pragma unroll
    for (int u = 0; u < GRAM_PF; ++u) cur[u] = fetch(u);
    for (int g0 = 0; g0 < NG; g0 += GRAM_PF) {
      if (g0 * RPG >= cnt) break;                                   // uniform: the rest of the block is empty
#pragma unroll
      for (int u = 0; u < GRAM_PF; ++u) nxt[u] = fetch(g0 + GRAM_PF + u);   // (past the block: slot >= cnt, inactive)
#pragma unroll
      for (int u = 0; u < GRAM_PF; ++u) apply(cur[u]);
#pragma unroll
      for (int u = 0; u < GRAM_PF; ++u) cur[u] = nxt[u];
    }
    epos = epos_next;
  }
  __syncthreads();
  double *out = G + (size_t)j1 * ldg + w0;
  for (int j = lane; j < wn; j += 64) out[j] = acc[j];
}

// vectors(i,k) = (sum_a A(i,a) V(idx(i,a),k)) / sigma_k * scale ; sigma_k = sqrt(max(eig_k,0))
__global__ __launch_bounds__(256) void u_recover_kernel(const int *__restrict__ ell_idx,
                                                        const double *__restrict__ val, int n, int r,
                                                        const double *__restrict__ V, int ldv,
                                                        const double *__restrict__ eig, int K, double scale,
                                                        double *__restrict__ out, int ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k0 = blockIdx.y * 8;
  double acc[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) acc[kk] = 0.0;
  for (int a = 0; a < r; ++a) {  // a ascending per (i,k): the oracle's order; mul then add
    const int id = ell_idx[(size_t)i * r + a];
    const double va = val[(size_t)i * r + a];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      if (k0 + kk < K) acc[kk] += va * V[(size_t)(k0 + kk) * ldv + id];
  }
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int k = k0 + kk;
    if (k < K) {
      const double ev = eig[k];
      const double sigma = __builtin_sqrt(ev > 0.0 ? ev : 0.0);
      // sigma = 0 (K reaches into the null space of A): a zero column instead of Inf / NaN; the host entry points refuse
      out[(size_t)k * ldo + i] = sigma > 0.0 ? (acc[kk] / sigma) * scale : 0.0;
    }
  }
}

// Vt(j, k) = V(j, k) stored k-contiguous: Vt[j*K + k]  (V: s x K column-major, ld = ldv)
__global__ void transpose_v_kernel(const double *__restrict__ V, int ldv, int s, int K, double *__restrict__ Vt) {
  __shared__ double t[32][33];
  const int j0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
  for (int kk = ty; kk < 32; kk += 8) {
    const int j = j0 + tx, k = k0 + kk;
    t[kk][tx] = (j < s && k < K) ? V[(size_t)k * ldv + j] : 0.0;
  }
  __syncthreads();
  for (int jj = ty; jj < 32; jj += 8) {
    const int j = j0 + jj, k = k0 + tx;
    if (j < s && k < K) Vt[(size_t)j * K + k] = t[tx][jj];
  }
}

// U-recovery, gather form: a workgroup owns 64 rows x 64 eigen-columns.  Lanes run along k, so the
// r rows of Vt a point needs are read as contiguous 512-byte pieces (they come from L2: Vt is
// s*K*8 bytes); the 64 x 64 tile is turned through LDS so that the column-major output is written
// in 512-byte pieces as well.  Arithmetic as u_recover_kernel: acc = 0; acc += A(i,a) V(idx,k)
// (a ascending, mul then add); (acc / sigma_k) * scale.
#ifndef U_RECOVER_UNROLL
#define U_RECOVER_UNROLL 8
#endif
__global__ __launch_bounds__(256) void u_recover_tiled_kernel(const int *__restrict__ ell_idx,
                                                              const double *__restrict__ val, int n, int r,
                                                              const double *__restrict__ Vt,
                                                              const double *__restrict__ eig, int K, double scale,
                                                              double *__restrict__ out, int ldo) {
  __shared__ double tile[64][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long i0 = (long)blockIdx.x * 64;
  const int k0 = blockIdx.y * 64;
  const int k = k0 + lane;
  const bool kok = k < K;
  double sigma = 1.0;
  if (kok) { const double ev = eig[k]; sigma = __builtin_sqrt(ev > 0.0 ? ev : 0.0); }
  // UR rows in flight per wave: their UR r gathers are issued before the first one is consumed
  constexpr int UR = U_RECOVER_UNROLL;
  const double inv_guard = kok ? 1.0 : 0.0;
  const int kk = kok ? k : 0;
  for (int il = wave * 16; il < wave * 16 + 16; il += UR) {
    double acc[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) acc[u] = 0.0;
    for (int a = 0; a < r; ++a) {
      double v[UR], z[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const long i = (i0 + il + u < n) ? i0 + il + u : n - 1;   // wave-uniform: scalar loads
        z[u] = val[(size_t)i * r + a];
        v[u] = Vt[(size_t)ell_idx[(size_t)i * r + a] * K + kk];
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) acc[u] += z[u] * v[u];
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) tile[lane][il + u] = sigma > 0.0 ? ((acc[u] * inv_guard) / sigma) * scale : 0.0;
  }
  __syncthreads();
  const int i = tid & 63;
  if (i0 + i < n)
    for (int kq = tid >> 6; kq < 64; kq += 4)
      if (k0 + kq < K) out[(size_t)(k0 + kq) * ldo + i0 + i] = tile[kq][i];
}

// The same with KPL eigen-columns per lane (64 KPL per workgroup; K a multiple of KPL): a row of Vt is K * 8 contiguous
// bytes, so one 8 KPL-byte load per lane takes 512 KPL bytes of it -- 1 / KPL as many gather instructions as the
// 64-column kernel, and K = 200 is two column tiles at KPL = 2 (128 + 72) or one at KPL = 4 instead of four
// (64 + 64 + 64 + 8, the last one all overhead).  ROWS rows per workgroup (a quarter per wave).
template <int KPL, int ROWS>
__global__ __launch_bounds__(256) void u_recover_wide_kernel(const int *__restrict__ ell_idx,
                                                             const double *__restrict__ val, int n, int r,
                                                             const double *__restrict__ Vt,
                                                             const double *__restrict__ eig, int K, double scale,
                                                             double *__restrict__ out, int ldo) {
  typedef double vec_t __attribute__((ext_vector_type(KPL)));
  constexpr int KT = 64 * KPL, RPW = ROWS / 4;
  constexpr int UR = (RPW < U_RECOVER_UNROLL) ? RPW : U_RECOVER_UNROLL;
  __shared__ double tile[KT][ROWS + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long i0 = (long)blockIdx.x * ROWS;
  const int k0 = blockIdx.y * KT;
  const int k = k0 + KPL * lane;
  const bool kok = k < K;                  // K % KPL == 0: a lane's columns are in or out together
  double sg[KPL];
#pragma unroll
  for (int c = 0; c < KPL; ++c) { const double e = kok ? eig[k + c] : 1.0; sg[c] = __builtin_sqrt(e > 0.0 ? e : 0.0); }
  const double inv_guard = kok ? 1.0 : 0.0;
  const int kk = kok ? k : 0;
  for (int il = wave * RPW; il < wave * RPW + RPW; il += UR) {
    vec_t acc[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u)
#pragma unroll
      for (int c = 0; c < KPL; ++c) acc[u][c] = 0.0;
    for (int a = 0; a < r; ++a) {
      vec_t v[UR];
      double z[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const long i = (i0 + il + u < n) ? i0 + il + u : n - 1;   // wave-uniform: scalar loads
        z[u] = val[(size_t)i * r + a];
        v[u] = *(const vec_t *)(Vt + (size_t)ell_idx[(size_t)i * r + a] * K + kk);
      }
#pragma unroll
      for (int u = 0; u < UR; ++u)
#pragma unroll
        for (int c = 0; c < KPL; ++c) acc[u][c] += z[u] * v[u][c];
    }
#pragma unroll
    for (int u = 0; u < UR; ++u)
#pragma unroll
      for (int c = 0; c < KPL; ++c)
        tile[KPL * lane + c][il + u] = sg[c] > 0.0 ? ((acc[u][c] * inv_guard) / sg[c]) * scale : 0.0;
  }
  __syncthreads();
  for (int e = tid; e < KT * ROWS; e += 256) {
    const int kq = e / ROWS, i = e % ROWS;
    if (k0 + kq < K && i0 + i < n) out[(size_t)(k0 + kq) * ldo + i0 + i] = tile[kq][i];
  }
}

__global__ void values_out_kernel(const double *__restrict__ eig, int K, int root, double *__restrict__ values) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const double ev = eig[k] > 0.0 ? eig[k] : 0.0;
  values[k] = root ? __builtin_sqrt(ev) : ev;   // values = d^2 (src/TruncatedSVD.cpp:29), sqrt if root (src/Spectrum.cpp:153-155)
}

struct CscPlan {
  int nchunks, chunk, nbits;
  size_t hist_bytes, tot_bytes;
};

static CscPlan csc_plan(long nnz, int s) {
  CscPlan p;
  long nc = (nnz + 2047) / 2048;
  if (nc > 2048) nc = 2048;
  if (nc < 1) nc = 1;
  long chunk = (nnz + nc - 1) / nc;
  chunk = (chunk + 63) / 64 * 64;
  if (chunk < 64) chunk = 64;
  p.chunk = (int)chunk;
  p.nchunks = (int)((nnz + chunk - 1) / chunk);
  if (p.nchunks < 1) p.nchunks = 1;
  p.nbits = 1;
  while ((1 << p.nbits) < s) ++p.nbits;
  p.hist_bytes = sizeof(int) * (size_t)p.nchunks * s;
  p.tot_bytes = sizeof(int) * (size_t)s;
  return p;
}

}  // namespace flgp

using namespace flgp;

extern "C" size_t flgp_dev_csc_workspace(int n, int s, int r) {
  const CscPlan p = csc_plan((long)n * r, s);
  return p.hist_bytes + p.tot_bytes + 256;
}

extern "C" int flgp_dev_csc_build(void *stream, const int *d_ell_idx, int n, int s, int r, int *d_colptr,
                                  int *d_pos, void *d_work, size_t work_bytes) {
  hipStream_t st = (hipStream_t)stream;
  const long nnz = (long)n * r;
  FLGP_REQUIRE(nnz < 2147483647L, "CSC: n*r = %ld does not fit int32 positions", nnz);
  FLGP_REQUIRE(s >= 1 && s <= FLGP_SMAX, "CSC: kernels are built for s <= %d (got %d)", FLGP_SMAX, s);
  FLGP_REQUIRE(work_bytes >= flgp_dev_csc_workspace(n, s, r), "CSC: workspace too small");
  const CscPlan p = csc_plan(nnz, s);
  int *hist = (int *)d_work;
  int *tot = (int *)((char *)d_work + p.hist_bytes);
  const size_t lds = sizeof(int) * (size_t)s;
  if (lds > 48 * 1024) {
    FLGP_HIP(hipFuncSetAttribute((const void *)csc_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FLGP_HIP(hipFuncSetAttribute((const void *)csc_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  ProfScope ps("csc_build", st, 12.0 * (double)nnz);
  hipLaunchKernelGGL(csc_hist_kernel, dim3(p.nchunks), dim3(256), lds, st, d_ell_idx, nnz, s, p.chunk, hist);
  FLGP_TRY(check_launch("csc_hist_kernel"));
  hipLaunchKernelGGL(csc_colscan_kernel, dim3(ceil_div(s, 256)), dim3(256), 0, st, hist, p.nchunks, s, tot);
  FLGP_TRY(check_launch("csc_colscan_kernel"));
  hipLaunchKernelGGL(csc_scan_kernel, dim3(1), dim3(1024), 0, st, tot, s, d_colptr);
  FLGP_TRY(check_launch("csc_scan_kernel"));
  hipLaunchKernelGGL(csc_scatter_kernel, dim3(p.nchunks), dim3(64), lds, st, d_ell_idx, nnz, s, p.chunk, p.nbits,
                     hist, d_colptr, d_pos);
  return check_launch("csc_scatter_kernel");
}

// columns of a per-column table of doubles that one workgroup keeps in LDS: all s of them up to 20000 (what rounds 1-3 were
// built for), windows of 16384 beyond (the kernels then walk their input once per window)
static int lds_window(int s) {
  const int forced = tuning("sparse_window", 0);      // tests: windows on small inputs
  if (forced > 0 && forced < s) return forced;
  return s <= 20000 ? s : 16384;
}

extern "C" size_t flgp_dev_colsum_workspace(int n, int s) {
  return sizeof(double) * (size_t)ceil_div(n > 0 ? n : 1, COLSUM_CHUNK) * (size_t)s + 256;
}

extern "C" int flgp_dev_colsum(void *stream, const int *d_ell_idx, const double *d_ell_val, int n, int r, int s,
                               double *d_colsum, void *d_work, size_t work_bytes) {
  hipStream_t st = (hipStream_t)stream;
  FLGP_REQUIRE(n >= 0 && r >= 1 && s >= 1 && s <= FLGP_SMAX, "colsum: need 1 <= s <= %d (got %d)", FLGP_SMAX, s);
  FLGP_REQUIRE(work_bytes >= flgp_dev_colsum_workspace(n, s), "colsum: workspace too small");
  const int nchunks = ceil_div(n > 0 ? n : 1, COLSUM_CHUNK);
  int nbits = 1;
  while ((1 << nbits) < s) ++nbits;
  const int wmax = lds_window(s);                    // s itself while a table of s doubles fits LDS
  const size_t lds = sizeof(double) * (size_t)wmax;
  const void *fn = wmax < s ? (const void *)colsum_chunk_kernel<true> : (const void *)colsum_chunk_kernel<false>;
  if (lds > 48 * 1024) FLGP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope ps("colsum_kernel", st, 12.0 * (double)n * r);
  if (wmax < s)
    hipLaunchKernelGGL(colsum_chunk_kernel<true>, dim3(nchunks, ceil_div(s, wmax)), dim3(64), lds, st, d_ell_idx, d_ell_val, n, r, s,
                       nbits, (double *)d_work, wmax, tuning("colsum_direct", 1));
  else
    hipLaunchKernelGGL(colsum_chunk_kernel<false>, dim3(nchunks), dim3(64), lds, st, d_ell_idx, d_ell_val, n, r, s, nbits,
                       (double *)d_work, wmax, tuning("colsum_direct", 1));
  FLGP_TRY(check_launch("colsum_chunk_kernel"));
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(ceil_div(s, 64)), dim3(64), 0, st, (const double *)d_work, nchunks, s, d_colsum);
  return check_launch("colsum_reduce_kernel");
}

extern "C" int flgp_dev_col_scale(void *stream, const int *d_ell_idx, double *d_ell_val, int n, int r,
                                  const double *d_colsum, const double *d_num_class, int mode) {
  FLGP_REQUIRE(mode == 0 || mode == 1, "col_scale: mode must be 0 or 1");
  const long nnz = (long)n * r;
  if (nnz == 0) return FLGP_OK;
  hipLaunchKernelGGL(col_scale_kernel, dim3(ceil_div(nnz, 256)), dim3(256), 0, (hipStream_t)stream, d_ell_idx,
                     d_ell_val, nnz, d_colsum, d_num_class, mode);
  return check_launch("col_scale_kernel");
}

extern "C" int flgp_dev_row_normalize(void *stream, double *d_ell_val, int n, int r) {
  if (n == 0) return FLGP_OK;
  FLGP_REQUIRE(r >= 1 && r <= FLGP_RMAX, "row_normalize: r = %d outside 1..%d", r, FLGP_RMAX);
  const size_t lds = sizeof(double) * 256 * (size_t)r;
  if (lds > 48 * 1024)
    FLGP_HIP(hipFuncSetAttribute((const void *)row_normalize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(row_normalize_kernel, dim3(ceil_div(n, 256)), dim3(256), lds, (hipStream_t)stream, d_ell_val,
                     n, r);
  return check_launch("row_normalize_kernel");
}

// graphLaplacian_cpp's column scaling and row normalisation in one pass over the values (reference src/Utils.cpp:199-211):
// the same operations per entry as flgp_dev_col_scale(mode 0) followed by flgp_dev_row_normalize, the same bits
extern "C" int flgp_dev_col_scale_row_normalize(void *stream, const int *d_ell_idx, double *d_ell_val, int n, int r,
                                                const double *d_colsum, const double *d_num_class) {
  FLGP_REQUIRE(r >= 1 && r <= FLGP_RMAX && d_ell_idx && d_ell_val && d_colsum, "col_scale_row_normalize: bad arguments");
  if (n == 0) return FLGP_OK;
  const size_t lds = sizeof(double) * 256 * (size_t)r;
  if (lds > 48 * 1024)
    FLGP_HIP(hipFuncSetAttribute((const void *)row_normalize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(row_normalize_kernel, dim3(ceil_div(n, 256)), dim3(256), lds, (hipStream_t)stream, d_ell_val, n, r, d_ell_idx,
                     d_colsum, d_num_class);
  return check_launch("row_normalize_kernel");
}

extern "C" int flgp_dev_gram(void *stream, const int *d_ell_idx, const double *d_ell_val, int n, int s, int r,
                             const int *d_colptr, const int *d_pos, double *dG, int ldg) {
  (void)n;
  FLGP_REQUIRE(s >= 1 && s <= FLGP_SMAX, "Gram: need 1 <= s <= %d (got %d)", FLGP_SMAX, s);
  FLGP_REQUIRE(r >= 1 && r <= FLGP_RMAX && ldg >= s, "Gram: bad r / ldg");
  const int wmax = lds_window(s);
  const size_t lds = sizeof(double) * (size_t)wmax;
  const bool win = wmax < s;
  const void *fn = r <= 16 ? (win ? (const void *)gram_kernel<16, true> : (const void *)gram_kernel<16, false>)
                           : (win ? (const void *)gram_kernel<32, true> : (const void *)gram_kernel<32, false>);
  if (lds > 48 * 1024) FLGP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope ps("gram_kernel", (hipStream_t)stream, 12.0 * (double)n * r + 8.0 * (double)s * s);
  const dim3 grid(s, win ? ceil_div(s, wmax) : 1);
#define GRAM_LAUNCH(LPRv, WINv)                                                                                                \
  hipLaunchKernelGGL((gram_kernel<LPRv, WINv>), grid, dim3(64), lds, (hipStream_t)stream, d_ell_idx, d_ell_val, s, r,         \
                     1.0 / (double)r, d_colptr, d_pos, dG, ldg, wmax)
  if (r <= 16) { if (win) GRAM_LAUNCH(16, true); else GRAM_LAUNCH(16, false); }
  else { if (win) GRAM_LAUNCH(32, true); else GRAM_LAUNCH(32, false); }
#undef GRAM_LAUNCH
  return check_launch("gram_kernel");
}

// ---- upper triangle of a symmetric matrix as one contiguous run (exchange 3 of the row-sharded path sends s(s+1)/2
// doubles instead of s^2; mirroring after the reduction also makes G symmetric bit for bit, which a ring all-reduce
// of the full square does not: (i,j) and (j,i) travel in different chunks and are added in different rank orders).
// Column j's rows 0..j sit at packed[j(j+1)/2 ..].
__global__ __launch_bounds__(256) void sym_pack_kernel(const double *__restrict__ G, int ldg, int s, double *__restrict__ packed) {
  const int j = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i <= j) packed[(size_t)j * (j + 1) / 2 + i] = G[(size_t)j * ldg + i];
}
// 32 x 32 tiles (ti <= tj): the upper tile is copied, its mirror image written through an LDS transpose
__global__ __launch_bounds__(256) void sym_unpack_kernel(const double *__restrict__ packed, int s, double *__restrict__ G, int ldg) {
  __shared__ double tile[32][33];
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (ti > tj) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int c = ty; c < 32; c += 8) {
    const int i = ti * 32 + tx, j = tj * 32 + c;
    double v = 0.0;
    if (i < s && j < s) {
      v = (i <= j) ? packed[(size_t)j * (j + 1) / 2 + i] : packed[(size_t)i * (i + 1) / 2 + j];   // i > j only on diagonal tiles
      G[(size_t)j * ldg + i] = v;
    }
    tile[c][tx] = v;
  }
  __syncthreads();
  if (ti == tj) return;
  for (int c = ty; c < 32; c += 8) {
    const int i = tj * 32 + tx, j = ti * 32 + c;     // element (i, j) of the lower tile = (j, i) of the upper one
    if (i < s && j < s) G[(size_t)j * ldg + i] = tile[tx][c];
  }
}

extern "C" int flgp_dev_sym_pack(void *stream, const double *dG, int ldg, int s, double *d_packed) {
  FLGP_REQUIRE(s >= 1 && ldg >= s, "sym_pack: bad shape");
  hipLaunchKernelGGL(sym_pack_kernel, dim3(ceil_div(s, 256), s), dim3(256), 0, (hipStream_t)stream, dG, ldg, s, d_packed);
  return check_launch("sym_pack_kernel");
}
extern "C" int flgp_dev_sym_unpack(void *stream, const double *d_packed, int s, double *dG, int ldg) {
  FLGP_REQUIRE(s >= 1 && ldg >= s, "sym_unpack: bad shape");
  const int nt = ceil_div(s, 32);
  hipLaunchKernelGGL(sym_unpack_kernel, dim3(nt, nt), dim3(256), 0, (hipStream_t)stream, d_packed, s, dG, ldg);
  return check_launch("sym_unpack_kernel");
}

extern "C" size_t flgp_dev_u_recover_workspace(int s, int K) { return sizeof(double) * (size_t)s * K + 256; }

extern "C" int flgp_dev_u_recover(void *stream, const int *d_ell_idx, const double *d_ell_val, int n, int r,
                                  const double *dV, int ldv, int s, const double *d_eig, int K, double scale,
                                  int root, double *d_vectors, int ldo, double *d_values_out, double *d_work) {
  hipStream_t st = (hipStream_t)stream;
  FLGP_REQUIRE(r >= 1 && r <= FLGP_RMAX && K >= 1 && ldo >= n, "u_recover: bad shape");
  if (n > 0) {
    ProfScope ps("u_recover_kernel", st, 12.0 * (double)n * r + 8.0 * (double)s * K + 8.0 * (double)n * K);
    if (d_work) {
      hipLaunchKernelGGL(transpose_v_kernel, dim3(ceil_div(s, 32), ceil_div(K, 32)), dim3(256), 0, st, dV, ldv, s, K,
                         d_work);
      const int wide = tuning("u_recover_wide", 4);
      if (wide >= 4 && K % 4 == 0) {
        hipLaunchKernelGGL((u_recover_wide_kernel<4, 32>), dim3(ceil_div(n, 32), ceil_div(K, 256)), dim3(256), 0, st, d_ell_idx,
                           d_ell_val, n, r, d_work, d_eig, K, scale, d_vectors, ldo);
      } else if (wide >= 2 && K % 2 == 0) {
        hipLaunchKernelGGL((u_recover_wide_kernel<2, 64>), dim3(ceil_div(n, 64), ceil_div(K, 128)), dim3(256), 0, st, d_ell_idx,
                           d_ell_val, n, r, d_work, d_eig, K, scale, d_vectors, ldo);
      } else {
        hipLaunchKernelGGL(u_recover_tiled_kernel, dim3(ceil_div(n, 64), ceil_div(K, 64)), dim3(256), 0, st, d_ell_idx,
                           d_ell_val, n, r, d_work, d_eig, K, scale, d_vectors, ldo);
      }
    } else {
      hipLaunchKernelGGL(u_recover_kernel, dim3(ceil_div(n, 256), ceil_div(K, 8)), dim3(256), 0, st, d_ell_idx,
                         d_ell_val, n, r, dV, ldv, d_eig, K, scale, d_vectors, ldo);
    }
    FLGP_TRY(check_launch("u_recover_kernel"));
  }
  if (d_values_out) {
    hipLaunchKernelGGL(values_out_kernel, dim3(ceil_div(K, 256)), dim3(256), 0, st, d_eig, K, root, d_values_out);
    FLGP_TRY(check_launch("values_out_kernel"));
  }
  return FLGP_OK;
}
