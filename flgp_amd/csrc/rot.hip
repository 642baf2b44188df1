// The eigensolver's rotation  out = alpha X W (+ beta E):  X is s x b (column-major, s in the thousands), W is b x b, b <= 256.
//
// gemm_f64_kernel<64> spends 1.4 us on a 16-deep stage of this shape whose MFMAs take 0.2 us (NOTEBOOK r04-8): operands go
// global -> registers -> select -> LDS with 64-bit address arithmetic per load, the k-contiguous operand's LDS stores land on a
// third of the banks, and every k step waits for its own LDS reads.  The operand traffic alone takes 8 us of the 30
// (scripts/ubench_rotload.hip).  This kernel is the same product -- one k-ascending MFMA chain per element, bit for bit the
// GEMM's -- written for the one shape:
//   * W arrives TRANSPOSED-major (WT[k * b + j] = W(k, j): its producers write it that way), so both operands are read in
//     16-byte pieces along their contiguous dimension and stored to LDS as 16-byte pieces: no transposing store, no conflicts;
//   * 32-deep stages (half the barriers), LDS rows 80 doubles apart (the four k rows of a fragment read fall on disjoint banks);
//   * no selects: rows past s read a clamped address and are simply not stored;
//   * the output tile is produced transposed (W side as the MFMA's first operand), so a store instruction writes 128
//     contiguous bytes of a column of out.
// Tile 64 x 64, four waves of 32 x 32; gridDim.y = 2 runs a second rotation by the same W (the Rayleigh-Ritz pair).
#include "common.h"
#include <type_traits>

namespace flgp {

typedef double rd4 __attribute__((ext_vector_type(4)));
typedef double rd2 __attribute__((ext_vector_type(2)));

constexpr int ROT_GK = 32;      // k depth of a stage
constexpr int ROT_LD = 80;      // LDS row stride (doubles): 64 + 16

struct RotArgs {
  const double *X, *X2;         // s x b, leading dimension s
  const double *WT;             // b x b, WT[k * b + j] = W(k, j)
  const double *E, *E2;         // optional, like out
  double *out, *out2;
  int s, b;
  double alpha, beta;
};

// TM = rows of X per tile: 64 (four waves of 32 x 32) or 32 (four waves of 16 x 32: twice the tiles, for a chip that 64-row
// tiles leave unevenly loaded -- 5000 rows are 316 tiles of 64 on 256 CUs, and the kernel is MFMA-bound on the CUs with two)
template <int TM, int GKR, bool HAS_E>
__global__ __launch_bounds__(256, 2) void rot_kernel(RotArgs g) {   // (GKR: k depth of a stage)
  constexpr int MI = TM / 32;                        // 16-row pieces per wave
  constexpr int XLD = TM + 16;                       // LDS row stride of the X tile (doubles): the fragment rows fall on disjoint banks
  constexpr int XP = TM / 2, XK = 256 / XP, XREP = GKR / XK;     // (GKR >= XK)
  constexpr int WREP = GKR / 8;   // load mapping of X: XP row pairs x XK k per pass, XREP passes
  __shared__ double Xs[2][GKR * XLD];
  __shared__ double Ws[2][GKR * ROT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = g.s, b = g.b;
  const int ntm = b / 64, ntn = (s + TM - 1) / TM, nt = ntm * ntn;
  // XCD-aware tile order (as gemm_f64_kernel): consecutive workgroup ids share an XCD every 8; each XCD gets a band of tiles
  int bid;
  const double *X = g.X, *E = g.E;
  double *out = g.out;
  {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, ntt = (int)gridDim.y * nt;
    const int q = ntt / 8, rem = ntt % 8, xcd = lin % 8;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + lin / 8;
    if (bid >= nt) { bid -= nt; X = g.X2; E = g.E2; out = g.out2; }
  }
  const int tm = bid % ntm, tn = bid / ntm;          // the column tiles of one row block are neighbours: X's panel is shared through L2
  const int i0 = tn * TM, j0 = tm * 64;

  // ---- operand loads: thread = (pair of adjacent rows / columns 2 l, 2 l + 1; k = kq + (passes) of the stage)
  const int l2 = 2 * (tid & 31), kq = tid >> 5;      // W: 32 column pairs x 8 k per pass, 4 passes
  const int xl2 = 2 * (tid % XP), xkq = tid / XP;    // X: XP row pairs x XK k per pass
  int ix = i0 + xl2;
  if (ix > s - 2) ix = s - 2;                        // rows past the end: a valid address, never stored (s is even, >= 64)
  const double *px = X + ix + (size_t)xkq * s;
  const double *pw = g.WT + (size_t)kq * b + j0 + l2;
  const size_t xstep = (size_t)XK * s, wstep = (size_t)8 * b;
  rd2 rx[2][XREP], rw[2][WREP];                         // two register sets: stage st + 1 waits in one while stage st + 2 is fetched into the other
  auto fetch = [&](auto set_c, int st) {
    constexpr int SET = decltype(set_c)::value;
    const double *qx = px + (size_t)st * GKR * s;
    const double *qw = pw + (size_t)st * GKR * b;
#pragma unroll
    for (int rep = 0; rep < XREP; ++rep) rx[SET][rep] = *(const rd2 *)(qx + rep * xstep);
#pragma unroll
    for (int rep = 0; rep < WREP; ++rep) rw[SET][rep] = *(const rd2 *)(qw + rep * wstep);
  };
  auto put = [&](auto set_c, int buf) {
    constexpr int SET = decltype(set_c)::value;
    double *dx = &Xs[buf][xkq * XLD + xl2], *dw = &Ws[buf][kq * ROT_LD + l2];
#pragma unroll
    for (int rep = 0; rep < XREP; ++rep) *(rd2 *)(dx + rep * XK * XLD) = rx[SET][rep];
#pragma unroll
    for (int rep = 0; rep < WREP; ++rep) *(rd2 *)(dw + rep * 8 * ROT_LD) = rw[SET][rep];
  };

  // ---- fragments: lane (fr, fk); the wave's (16 MI) x 32 part of the tile: rows wr.., columns wc..
  const int fr = lane & 15, fk = lane >> 4;
  const int wr = (wave >> 1) * (16 * MI), wc = (wave & 1) * 32;
  rd4 acc[2][MI];                                    // [column sub-tile ni][row sub-tile mi]
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = rd4{0.0, 0.0, 0.0, 0.0};

  using std::integral_constant;
  const int nst = b / GKR;                        // even: b is a multiple of 64
  fetch(integral_constant<int, 0>{}, 0);
  fetch(integral_constant<int, 1>{}, 1);
  put(integral_constant<int, 0>{}, 0);
  __syncthreads();
  // stage st (P = st & 1): fetch stage st + 2 into set P (free: stage st sits in LDS), multiply buffer P, then store stage
  // st + 1 (set P ^ 1, fetched a whole stage ago) into buffer P ^ 1, whose last readers passed the barrier that ended stage st - 1
  auto stage = [&](auto p_c, int st) {
    constexpr int P = decltype(p_c)::value;
    if (st + 2 < nst) fetch(integral_constant<int, P>{}, st + 2);
    const double *xs = &Xs[P][fk * XLD + wr + fr], *ws = &Ws[P][fk * ROT_LD + wc + fr];
    double xb[2][MI], wa[2][2];                      // [parity of the k step][sub-tile]
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xb[0][mi] = xs[16 * mi];
    wa[0][0] = ws[0]; wa[0][1] = ws[16];
#pragma unroll
    for (int kk = 0; kk < GKR / 4; ++kk) {
      const int q = kk & 1;
      if (kk + 1 < GKR / 4) {                     // the next k step's fragments are on their way while this one multiplies
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xb[q ^ 1][mi] = xs[(kk + 1) * 4 * XLD + 16 * mi];
        wa[q ^ 1][0] = ws[(kk + 1) * 4 * ROT_LD]; wa[q ^ 1][1] = ws[(kk + 1) * 4 * ROT_LD + 16];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[q][ni], xb[q][mi], acc[ni][mi], 0, 0, 0);
    }
    if (st + 1 < nst) put(integral_constant<int, P ^ 1>{}, P ^ 1);
    __syncthreads();
  };
  for (int st = 0; st < nst; st += 2) {
    stage(integral_constant<int, 0>{}, st);
    stage(integral_constant<int, 1>{}, st + 1);
  }

  // ---- epilogue: D(column j = fk + 4 reg, row i = fr) of each 16 x 16 piece
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int i = i0 + wr + mi * 16 + fr;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int j = j0 + wc + ni * 16 + fk + 4 * reg;
        if (i < s) {
          double o = g.alpha * acc[ni][mi][reg];
          if (HAS_E) o += g.beta * E[(size_t)j * s + i];
          out[(size_t)j * s + i] = o;
        }
      }
    }
}

// whether rot_launch takes the shape (else the caller multiplies with gemm_launch on the same operands)
bool rot_applicable(int s, int b, const double *X, const double *X2, const double *WT, const double *out, const double *out2) {
  auto al = [](const void *p) { return (((size_t)p) & 15) == 0; };
  return tuning("eig_rot_kernel", 1) && b >= 64 && b <= 256 && b % 64 == 0 && s >= 64 && s % 2 == 0 && al(X) && al(WT) && al(out) &&
         (!X2 || (al(X2) && al(out2)));
}

// out = alpha X W + beta E   (and out2 = alpha X2 W + beta E2 when X2 is given)
int rot_launch(hipStream_t st, int s, int b, double alpha, const double *X, const double *X2, const double *WT, double beta,
               const double *E, const double *E2, double *out, double *out2) {
  RotArgs g;
  g.X = X; g.X2 = X2; g.WT = WT; g.E = (beta == 0.0) ? nullptr : E; g.E2 = (beta == 0.0) ? nullptr : E2;
  g.out = out; g.out2 = out2; g.s = s; g.b = b; g.alpha = alpha; g.beta = beta;
  // 32-row tiles while 64-row tiles would leave the chip unevenly loaded (fewer than four tiles per CU)
  int n_cu = 256;
  {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      if (!cached[dev]) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess) cached[dev] = pr.multiProcessorCount; }
      if (cached[dev] > 0) n_cu = cached[dev];
    }
  }
  const int np = X2 ? 2 : 1;
  const int tm_knob = tuning("eig_rot_tile", 0);
  const bool tm32 = tm_knob ? tm_knob == 32 : ((long)(b / 64) * ceil_div(s, 64) * np < 4L * n_cu);
  const int nt = (b / 64) * ceil_div(s, tm32 ? 32 : 64);
  const dim3 grid(nt, np);
  const double fl = 2.0 * (double)s * b * b * np;
  ProfScope ps("gemm_f64_kernel", st, fl);
  ProfScope ps2("gemm_medium", st, fl);
  if (tm32 && tuning("eig_rot_gk16", 1)) {
    if (g.E) hipLaunchKernelGGL((rot_kernel<32, 16, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((rot_kernel<32, 16, false>), grid, dim3(256), 0, st, g);
  } else if (tm32) {
    if (g.E) hipLaunchKernelGGL((rot_kernel<32, 32, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((rot_kernel<32, 32, false>), grid, dim3(256), 0, st, g);
  } else {
    if (g.E) hipLaunchKernelGGL((rot_kernel<64, 32, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((rot_kernel<64, 32, false>), grid, dim3(256), 0, st, g);
  }
  return check_launch("rot_kernel");
}

// ------------------------------------------------------------------------------------------
// The other solver shape: S = Xa^T Xb (b x b) with the long dimension s reduced -- gemm_f64_kernel<64> needs 23 us + the reduction
// for it.  Here both operands are read along their contiguous dimension (the rows k of a column) in 16-byte pieces and kept
// k-contiguous in LDS, [column][k] with a row stride of 34 doubles: a fragment read (16 columns x 4 k) then falls on 64
// distinct banks per half-wave.  Tile 64 x 64 over a range of rows (blockIdx.y), partial planes in the layout gemm.hip's
// reduction kernels read (plane[jc * b + ic]), k ascending from zero inside a plane: the same bits as the GEMM's planes when
// the ranges coincide.
// ------------------------------------------------------------------------------------------
constexpr int GRK_LD = 34;      // doubles between two columns of a stage in LDS (32 k + 2)

__global__ __launch_bounds__(256, 2) void gramk_kernel(const double *__restrict__ Xa, const double *__restrict__ Xb, int s, int b,
                                                        int klen, double *__restrict__ part) {
  __shared__ double As[2][64 * GRK_LD];
  __shared__ double Bs[2][64 * GRK_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntb = b / 64;
  const int ti = blockIdx.x % ntb, tj = blockIdx.x / ntb;
  const int i0 = ti * 64, j0 = tj * 64;
  const int kbeg = blockIdx.y * klen;
  const int kend = (kbeg + klen < s) ? kbeg + klen : s;
  const int nst = (kend - kbeg + ROT_GK - 1) / ROT_GK;

  // loads: thread = (k pair 2 kp, 2 kp + 1 of the stage; column cq + 16 rep)
  const int kp2 = 2 * (tid & 15), cq = tid >> 4;
  const double *pa = Xa + (size_t)(i0 + cq) * s + kp2;
  const double *pb = Xb + (size_t)(j0 + cq) * s + kp2;
  const size_t cstep = (size_t)16 * s;
  rd2 ra[2][4], rb[2][4];
  auto fetch = [&](auto set_c, int st) {
    constexpr int SET = decltype(set_c)::value;
    int k = kbeg + st * ROT_GK;
    if (k + kp2 > s - 2) k = s - 2 - kp2;                // past the end: a valid address; zeroed in `put` (s is even)
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      ra[SET][rep] = *(const rd2 *)(pa + rep * cstep + k);
      rb[SET][rep] = *(const rd2 *)(pb + rep * cstep + k);
    }
  };
  auto put = [&](auto set_c, int buf, int st) {
    constexpr int SET = decltype(set_c)::value;
    const bool in = kbeg + st * ROT_GK + kp2 < kend;     // (kend - kbeg is even: a pair is in or out together)
    double *da = &As[buf][cq * GRK_LD + kp2], *db = &Bs[buf][cq * GRK_LD + kp2];
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      *(rd2 *)(da + rep * 16 * GRK_LD) = in ? ra[SET][rep] : rd2{0.0, 0.0};
      *(rd2 *)(db + rep * 16 * GRK_LD) = in ? rb[SET][rep] : rd2{0.0, 0.0};
    }
  };
  const int fr = lane & 15, fk = lane >> 4;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;      // the wave's rows (ic) and columns (jc) of the tile
  rd4 acc[2][2];                                              // [column piece nj][row piece mi]
#pragma unroll
  for (int nj = 0; nj < 2; ++nj)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) acc[nj][mi] = rd4{0.0, 0.0, 0.0, 0.0};
  using std::integral_constant;
  if (nst > 0) {
    fetch(integral_constant<int, 0>{}, 0);
    fetch(integral_constant<int, 1>{}, nst > 1 ? 1 : 0);
    put(integral_constant<int, 0>{}, 0, 0);
  }
  __syncthreads();
  auto stage = [&](auto p_c, int st) {
    constexpr int P = decltype(p_c)::value;
    if (st + 2 < nst) fetch(integral_constant<int, P>{}, st + 2);
    const double *as = &As[P][(wr + fr) * GRK_LD + fk], *bs = &Bs[P][(wc + fr) * GRK_LD + fk];
    double af[2][2], bf[2][2];
    af[0][0] = as[0]; af[0][1] = as[16 * GRK_LD]; bf[0][0] = bs[0]; bf[0][1] = bs[16 * GRK_LD];
#pragma unroll
    for (int kk = 0; kk < ROT_GK / 4; ++kk) {
      const int q = kk & 1;
      if (kk + 1 < ROT_GK / 4) {
        const int o = (kk + 1) * 4;
        af[q ^ 1][0] = as[o]; af[q ^ 1][1] = as[o + 16 * GRK_LD]; bf[q ^ 1][0] = bs[o]; bf[q ^ 1][1] = bs[o + 16 * GRK_LD];
      }
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[nj][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[q][nj], af[q][mi], acc[nj][mi], 0, 0, 0);
    }
    if (st + 1 < nst) put(integral_constant<int, P ^ 1>{}, P ^ 1, st + 1);
    __syncthreads();
  };
  for (int st = 0; st < nst; st += 2) {
    stage(integral_constant<int, 0>{}, st);
    if (st + 1 < nst) stage(integral_constant<int, 1>{}, st + 1);
  }
  // plane[jc * b + ic], D(jc = fk + 4 reg, ic = fr)
  double *plane = part + (size_t)blockIdx.y * b * b;
#pragma unroll
  for (int nj = 0; nj < 2; ++nj)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        plane[(size_t)(j0 + wc + nj * 16 + fk + 4 * reg) * b + i0 + wr + mi * 16 + fr] = acc[nj][mi][reg];
}

static int gramk_split(int s, int b, int *klen_out) {
  const int ntiles = (b / 64) * (b / 64);
  int nsplit = tuning("gemm_tile64_blocks", 256) / ntiles;          // as gemm_launch splits the same product
  const int maxk = s / (tuning("gemm_min_stages", 5) * 16);
  if (nsplit > maxk) nsplit = maxk;
  if (nsplit < 1) nsplit = 1;
  int klen = ceil_div(s, nsplit);
  klen = (klen + ROT_GK - 1) / ROT_GK * ROT_GK;
  *klen_out = klen;
  return ceil_div(s, klen);
}

bool gramk_applicable(int s, int b, const double *Xa, const double *Xb, size_t work_elems) {
  auto al = [](const void *p) { return (((size_t)p) & 15) == 0; };
  if (!(tuning("eig_gram_kernel", 1) && b >= 64 && b <= 256 && b % 64 == 0 && s >= 256 && s % 2 == 0 && al(Xa) && al(Xb))) return false;
  int klen;
  const int nsplit = gramk_split(s, b, &klen);
  return nsplit >= 2 && (size_t)nsplit * b * b <= work_elems;
}

int gramk_launch(hipStream_t st, int s, int b, const double *Xa, const double *Xb, double *out, double *work, size_t work_elems,
                 GemmFusedReduce *fused) {
  (void)work_elems;
  int klen;
  const int nsplit = gramk_split(s, b, &klen);
  {
    const double fl = 2.0 * (double)s * b * b;
    ProfScope ps("gemm_f64_kernel", st, fl);
    ProfScope ps2("gemm_medium", st, fl);
    hipLaunchKernelGGL(gramk_kernel, dim3((b / 64) * (b / 64), nsplit), dim3(256), 0, st, Xa, Xb, s, b, klen, work);
  }
  FLGP_TRY(check_launch("gramk_kernel"));
  return gemm_reduce_square(st, b, work, nsplit, out, fused);
}

}  // namespace flgp

using namespace flgp;

// test / benchmark entry: out (b x b, column-major) = Xa^T Xb with the solver's own Gram kernel; d_work: at least 64 b^2 doubles
extern "C" int flgp_dev_gram_small(void *stream, int s, int b, const double *Xa, const double *Xb, double *out, double *d_work,
                                   size_t work_elems) {
  FLGP_REQUIRE(Xa && Xb && out && d_work, "gram_small: bad arguments");
  FLGP_REQUIRE(gramk_applicable(s, b, Xa, Xb, work_elems), "gram_small: the kernel is built for 64 <= b <= 256 (multiples of 64), even s >= 256, "
                                                            "16-byte aligned operands and a workspace for its planes");
  return gramk_launch((hipStream_t)stream, s, b, Xa, Xb, out, d_work, work_elems, nullptr);
}

// test / benchmark entry: out = alpha X W + beta E with W given k-major (WT[k * b + j] = W(k, j)); X2 / out2 optional
extern "C" int flgp_dev_rotate(void *stream, int s, int b, double alpha, const double *X, const double *X2, const double *WT,
                               double beta, const double *E, double *out, double *out2) {
  FLGP_REQUIRE(X && WT && out && (!X2 || out2), "rotate: bad arguments");
  FLGP_REQUIRE(rot_applicable(s, b, X, X2, WT, out, out2), "rotate: the kernel is built for 64 <= b <= 256 (multiples of 64), even s >= 64, 16-byte aligned operands");
  FLGP_REQUIRE(beta == 0.0 || (E && !X2), "rotate: E goes with a single product");
  return rot_launch((hipStream_t)stream, s, b, alpha, X, X2, WT, beta, E, nullptr, out, out2);
}
