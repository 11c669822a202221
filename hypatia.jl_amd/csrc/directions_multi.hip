// Two stepper directions per pass.  CombinedStepper (/root/reference/src/Solvers/steppers/combined.jl:53-95)
// asks for four directions per iteration; (cent, pred) do not depend on each other and neither do
// (centadj, predadj), so each pair is solved together: the products with G (804 MB at config 2), the
// triangular solves with the 100 MB factor and the cone Hessian products are each ONE pass that serves
// both right-hand sides.  Arithmetic per column is the reference's (systemsolvers/common.jl:15-182,
// qrchol.jl:16-98); a column whose residual calls for refinement continues alone through the
// single-right-hand-side routines of syssolver.hip.
#include <cstring>
#include "syssolver.hpp"
#include <chrono>

namespace hyp {

void seg_dots(Ctx& c, int B, int len, const double* a, const double* b, int off, int stride, double* out);   // syssolver.hip
constexpr int MR = 2;   // right-hand sides per pass

// ---- Y[:, r] = alpha A' X[:, r] + beta Y[:, r]: one workgroup per column of A, A read once --------------
// TPB threads, U independent 16-byte loads of the column in flight per thread (256 x 2, 256 x 4, 512 x 2, 512 x 4,
// 1024 x 1, 1024 x 2 all measure 4.0-4.2 TB/s at q = 20100, n = 5000: the two X vectors re-read from L2 by every
// workgroup cost as much as the column itself; one right-hand side reaches 4.8 TB/s).
template <int NR, int TPB, int U>
__global__ __launch_bounds__(TPB) void gemv_t_multi_kernel(int m, double alpha, const double* __restrict__ A, long lda,
                                                           const double* __restrict__ X, long ldx, double beta, double* __restrict__ Y,
                                                           long ldy) {
  constexpr int NW = TPB / 64;
  __shared__ double red[NR][NW];
  const int col = blockIdx.x;
  const double* a = A + (long)col * lda;
  double s[NR][2 * U];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int u = 0; u < 2 * U; ++u) s[r][u] = 0.0;
  int i = 0;
  typedef double d2_t __attribute__((ext_vector_type(2)));
  const bool wide = (((uintptr_t)a | (uintptr_t)X | ((uintptr_t)ldx * sizeof(double))) & 15) == 0;   // 16-byte pairs everywhere
  if (wide) {
    constexpr int BLK = 2 * TPB * U;    // rows per iteration: thread t takes the pairs at 2 t + 2 TPB u
    const int nfull = m / BLK;
    for (int b = 0; b < nfull; ++b) {
      const int j = BLK * b + 2 * threadIdx.x;
      d2_t av[U];
#pragma unroll
      for (int u = 0; u < U; ++u) av[u] = *reinterpret_cast<const d2_t*>(a + j + 2 * TPB * u);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const d2_t xv = *reinterpret_cast<const d2_t*>(X + (long)r * ldx + j + 2 * TPB * u);
          s[r][2 * u] += av[u].x * xv.x;
          s[r][2 * u + 1] += av[u].y * xv.y;
        }
      }
    }
    i = BLK * nfull + threadIdx.x;   // the tail below is element-wise
  } else {
    i = threadIdx.x;
    for (; i + (2 * U - 1) * TPB < m; i += 2 * U * TPB) {   // 2 U independent column loads in flight per thread
      double av[2 * U];
#pragma unroll
      for (int u = 0; u < 2 * U; ++u) av[u] = a[i + u * TPB];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const double* x = X + (long)r * ldx + i;
#pragma unroll
        for (int u = 0; u < 2 * U; ++u) s[r][u] += av[u] * x[u * TPB];
      }
    }
  }
  for (; i < m; i += TPB) {
    const double a0 = a[i];
#pragma unroll
    for (int r = 0; r < NR; ++r) s[r][0] += a0 * X[(long)r * ldx + i];
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    double pu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pu[u] = s[r][2 * u] + s[r][2 * u + 1];
#pragma unroll
    for (int step = 1; step < U; step *= 2)
#pragma unroll
      for (int u = 0; u + step < U; u += 2 * step) pu[u] += pu[u + step];
    double t = pu[0];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
    if ((threadIdx.x & 63) == 0) red[r][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x < NR) {
    const int r = threadIdx.x;
    double pr[NW];   // fixed pairwise tree over the wavefronts' partial sums
#pragma unroll
    for (int w = 0; w < NW; ++w) pr[w] = red[r][w];
#pragma unroll
    for (int step = 1; step < NW; step *= 2)
#pragma unroll
      for (int w = 0; w + step < NW; w += 2 * step) pr[w] += pr[w + step];
    const double t = pr[0];
    double* y = Y + (long)r * ldy + col;
    *y = alpha * t + (beta != 0.0 ? beta * (*y) : 0.0);
  }
}

// The same product with CB columns of A per workgroup: the X values of a row block are loaded once and serve all CB
// columns (with one column per workgroup every workgroup streams both X vectors from L2, twice the traffic of the column
// itself).  Per-thread row assignment, partial sums and reduction tree are those of gemv_t_multi_kernel<NR, 256, 2>, so
// the results are bitwise the same.  Requires 16-byte aligned columns and X (the launcher checks).
template <int NR, int CB>
__global__ __launch_bounds__(256) void gemv_t_multi_cb_kernel(int m, int n, double alpha, const double* __restrict__ A, long lda,
                                                              const double* __restrict__ X, long ldx, double beta, double* __restrict__ Y,
                                                              long ldy) {
  __shared__ double red[CB][NR][4];
  typedef double d2_t __attribute__((ext_vector_type(2)));
  const int col0 = blockIdx.x * CB;
  const int ncol = min(CB, n - col0);
  const double* a[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) a[c] = A + (long)(col0 + (c < ncol ? c : 0)) * lda;   // (surplus columns re-read column 0, results dropped)
  double s[CB][NR][4];
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int r = 0; r < NR; ++r) s[c][r][0] = s[c][r][1] = s[c][r][2] = s[c][r][3] = 0.0;
  const int nfull = m / 1024;
  for (int b = 0; b < nfull; ++b) {
    const int j = 1024 * b + 2 * threadIdx.x;
    d2_t av[CB][2];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      av[c][0] = *reinterpret_cast<const d2_t*>(a[c] + j);
      av[c][1] = *reinterpret_cast<const d2_t*>(a[c] + j + 512);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const d2_t x0 = *reinterpret_cast<const d2_t*>(X + (long)r * ldx + j), x1 = *reinterpret_cast<const d2_t*>(X + (long)r * ldx + j + 512);
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        s[c][r][0] += av[c][0].x * x0.x;
        s[c][r][1] += av[c][0].y * x0.y;
        s[c][r][2] += av[c][1].x * x1.x;
        s[c][r][3] += av[c][1].y * x1.y;
      }
    }
  }
  for (int i = 1024 * nfull + threadIdx.x; i < m; i += 256) {   // element-wise tail
    double xv[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) xv[r] = X[(long)r * ldx + i];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const double a0 = a[c][i];
#pragma unroll
      for (int r = 0; r < NR; ++r) s[c][r][0] += a0 * xv[r];
    }
  }
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      double t = (s[c][r][0] + s[c][r][1]) + (s[c][r][2] + s[c][r][3]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
      if ((threadIdx.x & 63) == 0) red[c][r][threadIdx.x >> 6] = t;
    }
  __syncthreads();
  if (threadIdx.x < CB * NR) {
    const int c = threadIdx.x / NR, r = threadIdx.x % NR;
    if (c < ncol) {
      const double t = (red[c][r][0] + red[c][r][1]) + (red[c][r][2] + red[c][r][3]);
      double* y = Y + (long)r * ldy + col0 + c;
      *y = alpha * t + (beta != 0.0 ? beta * (*y) : 0.0);
    }
  }
}

// ---- Y[:, r] = alpha A X[:, r] + beta Y[:, r]: partial sums over 256-column chunks, then an ordered reduce
constexpr int GM_CHUNK = 256;
template <int NR>
__global__ __launch_bounds__(256) void gemv_n_multi_partial_kernel(int m, int n, const double* __restrict__ A, long lda,
                                                                   const double* __restrict__ X, long ldx, double* __restrict__ partial) {
  __shared__ double xs[NR][GM_CHUNK];
  const int row = blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * GM_CHUNK;
  const int nc = min(GM_CHUNK, n - c0);
#pragma unroll
  for (int r = 0; r < NR; ++r) xs[r][threadIdx.x] = (threadIdx.x < nc) ? X[(long)r * ldx + c0 + threadIdx.x] : 0.0;
  __syncthreads();
  if (row >= m) return;
  const double* a = A + (long)c0 * lda + row;
  // (two interleaved partial sums per 256-column chunk, then the chunks in order; the one-right-hand-side product of gemv() runs
  //  through this kernel with NR = 1 since round 3, so that a column has the same sums whether it is computed alone, in a pair or
  //  as the third column of the first paired solve: tools/diag_const3.py)
  double s[NR][2];
#pragma unroll
  for (int r = 0; r < NR; ++r) s[r][0] = s[r][1] = 0.0;
  int c = 0;
  for (; c + 1 < nc; c += 2) {
    const double a0 = a[(long)c * lda], a1 = a[(long)(c + 1) * lda];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      s[r][0] += a0 * xs[r][c];
      s[r][1] += a1 * xs[r][c + 1];
    }
  }
  if (c < nc) {
    const double a0 = a[(long)c * lda];
#pragma unroll
    for (int r = 0; r < NR; ++r) s[r][0] += a0 * xs[r][c];
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) partial[((long)blockIdx.y * NR + r) * m + row] = s[r][0] + s[r][1];
}
template <int NR>
__global__ void gemv_n_multi_reduce_kernel(int m, int nchunks, double alpha, const double* __restrict__ partial, double beta,
                                           double* __restrict__ Y, long ldy) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (row >= m) return;
  double s = 0.0;   // (strictly in chunk order; eight loads in flight at a time)
  int k = 0;
  for (; k + 8 <= nchunks; k += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[((long)(k + u) * NR + r) * m + row];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < nchunks; ++k) s += partial[((long)k * NR + r) * m + row];
  double* y = Y + (long)r * ldy + row;
  *y = alpha * s + (beta != 0.0 ? beta * (*y) : 0.0);
}

template <int NR>
static void gemv_multi_t(Ctx& c, bool trans, int m, int n, double alpha, const double* A, long lda, const double* X, long ldx, double beta, double* Y,
                         long ldy) {
  if (trans) {
    if (n <= 0) return;
    static const int cb_env = [] { const char* e = getenv("HYP_GEMVT_CB"); return e ? atoi(e) : 4; }();
    const bool wide = ((((uintptr_t)A | (uintptr_t)X) & 15) == 0) && (lda % 2 == 0) && (ldx % 2 == 0);
    if (wide && cb_env == 4) {
      hipLaunchKernelGGL((gemv_t_multi_cb_kernel<NR, 4>), dim3((n + 3) / 4), dim3(256), 0, c.stream, m, n, alpha, A, lda, X, ldx, beta, Y, ldy);
    } else if (wide && cb_env == 2) {
      hipLaunchKernelGGL((gemv_t_multi_cb_kernel<NR, 2>), dim3((n + 1) / 2), dim3(256), 0, c.stream, m, n, alpha, A, lda, X, ldx, beta, Y, ldy);
    } else if (wide && cb_env == 8) {
      hipLaunchKernelGGL((gemv_t_multi_cb_kernel<NR, 8>), dim3((n + 7) / 8), dim3(256), 0, c.stream, m, n, alpha, A, lda, X, ldx, beta, Y, ldy);
    } else {
      hipLaunchKernelGGL((gemv_t_multi_kernel<NR, 256, 2>), dim3(n), dim3(256), 0, c.stream, m, alpha, A, lda, X, ldx, beta, Y, ldy);
    }
  } else {
    if (m <= 0) return;
    const int nchunks = (n + GM_CHUNK - 1) / GM_CHUNK;
    c.scratch.ensure(std::max<size_t>((size_t)nchunks * NR * m * sizeof(double), 4096));
    if (nchunks > 0)
      hipLaunchKernelGGL((gemv_n_multi_partial_kernel<NR>), dim3((m + 255) / 256, nchunks), dim3(256), 0, c.stream, m, n, A, lda, X, ldx,
                         c.scratch.d());
    hipLaunchKernelGGL((gemv_n_multi_reduce_kernel<NR>), dim3((m + 255) / 256, NR), dim3(256), 0, c.stream, m, nchunks, alpha, c.scratch.d(),
                       beta, Y, ldy);
  }
  HYP_CHECK(hipGetLastError());
}
// ---- BOTH products of one matrix in ONE pass over it: Yn[:, r] = A Xn[:, r] + beta_n Yn[:, r] and Yt[:, r] = A' Xt[:, r] + beta_t Yt[:, r].
// Where the algorithm needs G x and G' z of independent vectors at the same moment (the residual of a pair of directions, apply_lhs
// common.jl:79-121; the residuals of calc_convergence_params, Solvers.jl:425-483) the two HBM-bound passes over the q x n block
// (804 MB at config 2, 8.3 GB at config 4) become one.  Workgroup = 64 columns (a 16-column strip per wavefront) x a range of rows
// walked in blocks of 256; lane l owns rows 4 l .. 4 l + 3 of the block (32-byte loads, a wavefront reads 2 KB of a column).  The
// A' part keeps 16 column sums per lane over the whole range (one shuffle tree at the end, partial per row range); the A part sums a
// lane's 4 rows over the strip, the four strips meet in LDS, partial per 64-column chunk.  Two ordered reductions finish (the
// kernels of the one-sided products).  Fixed assignment and order: bitwise reproducible; the sums are NOT those of the one-sided
// kernels (another order), so a call site uses one form or the other, never a mixture across iterations.
template <int NR>
__global__ __launch_bounds__(256) void gemv_both_kernel(int m, int n, const double* __restrict__ A, long lda, const double* __restrict__ Xn, long ldxn,
                                                        const double* __restrict__ Xt, long ldxt, double* __restrict__ part_n,
                                                        double* __restrict__ part_t, int rows_per_split) {
  typedef double d4v_t __attribute__((ext_vector_type(4)));
  __shared__ double red[2][4][NR][256];   // double-buffered: one barrier per row block
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = blockIdx.x * 64 + 16 * w;
  const int rbeg = blockIdx.y * rows_per_split, rend = min(m, rbeg + rows_per_split);
  double xn[NR][16];
  const double* a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = c0 + j;
    a[j] = A + (long)min(c, n - 1) * lda;
#pragma unroll
    for (int r = 0; r < NR; ++r) xn[r][j] = (c < n) ? Xn[(long)r * ldxn + c] : 0.0;
  }
  double st[NR][16];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) st[r][j] = 0.0;
  for (int row0 = rbeg; row0 < rend; row0 += 256) {
    const int r4 = row0 + 4 * lane;
    const bool full = (r4 + 3 < rend);
    double xt[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) xt[r][i] = (r4 + i < rend) ? Xt[(long)r * ldxt + r4 + i] : 0.0;
    double sn[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r) sn[r][0] = sn[r][1] = sn[r][2] = sn[r][3] = 0.0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      d4v_t av[8];
      if (full) {
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = *reinterpret_cast<const d4v_t*>(a[8 * half + j] + r4);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) av[j][i] = (r4 + i < rend) ? a[8 * half + j][r4 + i] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            sn[r][i] = fma(av[j][i], xn[r][8 * half + j], sn[r][i]);
            st[r][8 * half + j] = fma(av[j][i], xt[r][i], st[r][8 * half + j]);
          }
    }
    // the four strips' sums of a row meet in LDS.  A wavefront that runs ahead writes buffer b again two blocks later, i.e. after
    // the next block's barrier, which nobody passes before every reader of this block is done: one barrier per block suffices
    const int b = ((row0 - rbeg) >> 8) & 1;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[b][w][r][4 * lane + i] = sn[r][i];
    __syncthreads();
    if (row0 + tid < rend) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
        part_n[((long)blockIdx.x * NR + r) * m + row0 + tid] = (red[b][0][r][tid] + red[b][1][r][tid]) + (red[b][2][r][tid] + red[b][3][r][tid]);
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double t = st[r][j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
      if (lane == 0 && c0 + j < n) part_t[((long)blockIdx.y * NR + r) * n + c0 + j] = t;
    }
}

bool gemv_both_ok(int m, int n, const double* A, long lda) {
  static const bool on = [] { const char* e = getenv("HYP_GEMV_BOTH"); return !(e && e[0] == '0'); }();
  return on && m >= 1024 && n >= 64 && (lda % 4 == 0) && ((uintptr_t)A % 32 == 0);
}

template <int NR>
static void gemv_both_t(Ctx& c, int m, int n, const double* A, long lda, const double* Xn, long ldxn, double beta_n, double* Yn, long ldyn,
                        const double* Xt, long ldxt, double beta_t, double* Yt, long ldyt) {
  const int nchunks = (n + 63) / 64;
  int splits = std::max(1, std::min((1536 + nchunks - 1) / nchunks, (m + 1023) / 1024));
  const int rows_per_split = (((m + splits - 1) / splits) + 255) / 256 * 256;
  splits = (m + rows_per_split - 1) / rows_per_split;
  const size_t pn = (size_t)nchunks * NR * m, pt = (size_t)splits * NR * n;
  c.scratch.ensure((pn + pt) * sizeof(double));
  double* part_n = c.scratch.d();
  double* part_t = c.scratch.d() + pn;
  hipLaunchKernelGGL((gemv_both_kernel<NR>), dim3(nchunks, splits), dim3(256), 0, c.stream, m, n, A, lda, Xn, ldxn, Xt, ldxt, part_n, part_t,
                     rows_per_split);
  hipLaunchKernelGGL((gemv_n_multi_reduce_kernel<NR>), dim3((m + 255) / 256, NR), dim3(256), 0, c.stream, m, nchunks, 1.0, part_n, beta_n, Yn, ldyn);
  hipLaunchKernelGGL((gemv_n_multi_reduce_kernel<NR>), dim3((n + 255) / 256, NR), dim3(256), 0, c.stream, n, splits, 1.0, part_t, beta_t, Yt, ldyt);
  HYP_CHECK(hipGetLastError());
}
// nr = 1 or 2 columns each way; the caller has checked gemv_both_ok
void gemv_both(Ctx& c, int m, int n, int nr, const double* A, long lda, const double* Xn, long ldxn, double beta_n, double* Yn, long ldyn,
               const double* Xt, long ldxt, double beta_t, double* Yt, long ldyt) {
  HYP_REQUIRE(nr == 1 || nr == 2, "gemv_both: 1 or 2 right-hand sides");
  if (nr == 1) gemv_both_t<1>(c, m, n, A, lda, Xn, ldxn, beta_n, Yn, ldyn, Xt, ldxt, beta_t, Yt, ldyt);
  else gemv_both_t<2>(c, m, n, A, lda, Xn, ldxn, beta_n, Yn, ldyn, Xt, ldxt, beta_t, Yt, ldyt);
}

// y = alpha A x + beta y with the sums of the multi-column kernel (gemv() routes its one-right-hand-side product here)
void gemv_n_one(Ctx& c, int m, int n, double alpha, const double* A, long lda, const double* x, double beta, double* y) {
  gemv_multi_t<1>(c, false, m, n, alpha, A, lda, x, (long)n, beta, y, (long)m);
}
void gemv_t_one(Ctx& c, int m, int n, double alpha, const double* A, long lda, const double* x, double beta, double* y) {
  gemv_multi_t<1>(c, true, m, n, alpha, A, lda, x, (long)m, beta, y, (long)n);
}

// nr = 1, 2 or 3 right-hand sides per pass over A (3: the constant column of update_lhs rides along with the first pair of
// directions); every column's sums are those of the same column in a pass of its own width class (fixed per-thread row
// assignment and reduction tree)
void gemv_multi(Ctx& c, bool trans, int m, int n, int nr, double alpha, const double* A, long lda, const double* X, long ldx, double beta,
                double* Y, long ldy) {
  if (nr == 1) {
    gemv(c, trans, m, n, alpha, A, lda, X, beta, Y);
    return;
  }
  HYP_REQUIRE(nr == MR || nr == MR + 1, "gemv_multi: 1, 2 or 3 right-hand sides");
  if (nr == MR) gemv_multi_t<MR>(c, trans, m, n, alpha, A, lda, X, ldx, beta, Y, ldy);
  else gemv_multi_t<MR + 1>(c, trans, m, n, alpha, A, lda, X, ldx, beta, Y, ldy);
}

// ---- super-block triangular solves with two right-hand sides (see TriSolvePlan in dense.hip) -------------
// out[j, r] = base[j, r] + alpha sum_{i in rows(j)} M[i, j] v[i, r]; M is read once for both columns
__global__ __launch_bounds__(256) void coldot2_kernel(int m, int ncols, int mode, const double* __restrict__ M, long ld,
                                                      const double* __restrict__ v, long ldv, const double* base, long ldb, double alpha,
                                                      double* out, long ldo) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= ncols) return;
  const int lane = threadIdx.x & 63;
  int i0 = 0, i1 = m;
  if (mode == 1) i1 = min(j + 1, m);
  else if (mode == 2) i0 = min(j, m);
  const double* a = M + (long)j * ld;
  double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
  int i = i0 + lane;
  for (; i + 64 < i1; i += 128) {
    const double a0 = a[i], a1 = a[i + 64];
    s0 += a0 * v[i];
    s1 += a1 * v[i + 64];
    t0 += a0 * v[ldv + i];
    t1 += a1 * v[ldv + i + 64];
  }
  if (i < i1) {
    const double a0 = a[i];
    s0 += a0 * v[i];
    t0 += a0 * v[ldv + i];
  }
  double s = s0 + s1, t = t0 + t1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off);
    t += __shfl_down(t, off);
  }
  if (lane == 0) {
    out[j] = (base ? base[j] : 0.0) + alpha * s;
    out[ldo + j] = (base ? base[ldb + j] : 0.0) + alpha * t;
  }
}
// the same sums in the same order for m <= 1024 with all loads of a lane in flight at once (see coldot_batched_kernel, dense.hip)
__global__ __launch_bounds__(256) void coldot2_batched_kernel(int m, int ncols, int mode, const double* __restrict__ M, long ld,
                                                              const double* __restrict__ v, long ldv, const double* base, long ldb, double alpha,
                                                              double* out, long ldo) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= ncols) return;
  const int lane = threadIdx.x & 63;
  int i0 = 0, i1 = m;
  if (mode == 1) i1 = min(j + 1, m);
  else if (mode == 2) i0 = min(j, m);
  const double* a = M + (long)j * ld;
  const int ilast = max(i1 - 1, 0);
  double av[16], v0[16], v1[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = min(i0 + lane + 64 * k, ilast);
    av[k] = a[i];
    v0[k] = v[i];
    v1[k] = v[ldv + i];
  }
  const double b0 = base ? base[j] : 0.0, b1 = base ? base[ldb + j] : 0.0;
  double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int i = i0 + lane + 128 * g;
    if (i + 64 < i1) {
      s0 += av[2 * g] * v0[2 * g];
      s1 += av[2 * g + 1] * v0[2 * g + 1];
      t0 += av[2 * g] * v1[2 * g];
      t1 += av[2 * g + 1] * v1[2 * g + 1];
    } else if (i < i1) {
      s0 += av[2 * g] * v0[2 * g];
      t0 += av[2 * g] * v1[2 * g];
    }
  }
  double s = s0 + s1, t = t0 + t1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off);
    t += __shfl_down(t, off);
  }
  if (lane == 0) {
    out[j] = b0 + alpha * s;
    out[ldo + j] = b1 + alpha * t;
  }
}
bool coldot_batched_on();   // dense.hip (HYP_COLDOT_BATCH)
static void coldot2(Ctx& c, int m, int ncols, int mode, const double* M, long ld, const double* v, long ldv, const double* base, long ldb,
                    double alpha, double* out, long ldo) {
  if (ncols <= 0) return;
  if (m <= 1024 && coldot_batched_on())
    hipLaunchKernelGGL(coldot2_batched_kernel, dim3((ncols + 3) / 4), dim3(256), 0, c.stream, m, ncols, mode, M, ld, v, ldv, base, ldb, alpha, out, ldo);
  else
    hipLaunchKernelGGL(coldot2_kernel, dim3((ncols + 3) / 4), dim3(256), 0, c.stream, m, ncols, mode, M, ld, v, ldv, base, ldb, alpha, out, ldo);
}

// Three columns at once, M read once: columns 0 and 1 (v, v + ldv) with EXACTLY the sums of coldot2_batched_kernel, the third (v3)
// with exactly those of coldot_batched_kernel (dense.hip) -- the constant column of update_lhs rides along with the first pair of
// directions through the super-block solves (30 launches per triangular solve fewer per iteration) and every number stays what the
// separate solves gave.  m <= 1024.
__global__ __launch_bounds__(256) void coldot3_batched_kernel(int m, int ncols, int mode, const double* __restrict__ M, long ld,
                                                              const double* __restrict__ v, long ldv, const double* __restrict__ v3,
                                                              const double* base, long ldb, const double* base3, double alpha, double* out,
                                                              long ldo, double* out3) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= ncols) return;
  const int lane = threadIdx.x & 63;
  int i0 = 0, i1 = m;
  if (mode == 1) i1 = min(j + 1, m);
  else if (mode == 2) i0 = min(j, m);
  const double* a = M + (long)j * ld;
  const int ilast = max(i1 - 1, 0);
  double av[16], v0[16], v1[16], v2[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = min(i0 + lane + 64 * k, ilast);
    av[k] = a[i];
    v0[k] = v[i];
    v1[k] = v[ldv + i];
    v2[k] = v3[i];
  }
  const double b0 = base ? base[j] : 0.0, b1 = base ? base[ldb + j] : 0.0, b2 = base3 ? base3[j] : 0.0;
  double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {   // (columns 0, 1: coldot2_batched_kernel)
    const int i = i0 + lane + 128 * g;
    if (i + 64 < i1) {
      s0 += av[2 * g] * v0[2 * g];
      s1 += av[2 * g + 1] * v0[2 * g + 1];
      t0 += av[2 * g] * v1[2 * g];
      t1 += av[2 * g + 1] * v1[2 * g + 1];
    } else if (i < i1) {
      s0 += av[2 * g] * v0[2 * g];
      t0 += av[2 * g] * v1[2 * g];
    }
  }
  double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {   // (column 2: coldot_batched_kernel)
    const int i = i0 + lane + 256 * g;
    if (i + 192 < i1) {
      u0 += av[4 * g] * v2[4 * g];
      u1 += av[4 * g + 1] * v2[4 * g + 1];
      u2 += av[4 * g + 2] * v2[4 * g + 2];
      u3 += av[4 * g + 3] * v2[4 * g + 3];
    } else {
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (i + 64 * t < i1) u0 += av[4 * g + t] * v2[4 * g + t];
    }
  }
  double s = s0 + s1, t = t0 + t1, u = (u0 + u1) + (u2 + u3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off);
    t += __shfl_down(t, off);
    u += __shfl_down(u, off);
  }
  if (lane == 0) {
    out[j] = b0 + alpha * s;
    out[ldo + j] = b1 + alpha * t;
    out3[j] = b2 + alpha * u;
  }
}
void coldot_single(Ctx& c, int m, int ncols, int mode, const double* M, long ld, const double* v, const double* base, double alpha, double* out);   // dense.hip
static void coldot3(Ctx& c, int m, int ncols, int mode, const double* M, long ld, const double* v, long ldv, const double* v3, const double* base,
                    long ldb, const double* base3, double alpha, double* out, long ldo, double* out3) {
  if (ncols <= 0) return;
  if (m <= 1024 && coldot_batched_on()) {
    hipLaunchKernelGGL(coldot3_batched_kernel, dim3((ncols + 3) / 4), dim3(256), 0, c.stream, m, ncols, mode, M, ld, v, ldv, v3, base, ldb, base3,
                       alpha, out, ldo, out3);
  } else {   // (longer columns: the two separate kernels)
    coldot2(c, m, ncols, mode, M, ld, v, ldv, base, ldb, alpha, out, ldo);
    coldot_single(c, m, ncols, mode, M, ld, v3, base3, alpha, out3);
  }
}

// the pair x[:, 0:2] (leading dimension ldx) and a third right-hand side x3 through the super-block sweeps together: the steps of
// solve_multi (nr = 2) and solve() with every product of the three columns in one launch
void TriSolvePlan::solve_multi3(Ctx& c, const double* U, long ldu, bool trans, double* x, long ldx, double* x3) {
  if (ol_usable(c, ldu, MR + 1)) { ol_sweep(c, U, trans ? 0 : 1, x, ldx, x3, MR + 1); return; }
  const int nsb = (n + sb - 1) / sb;
  const size_t blk = (size_t)sb * sb;
  work.ensure((size_t)2 * (MR + 1) * sb * sizeof(double));
  double* t = work.d();                   // [sb x 2]
  double* e = work.d() + MR * sb;         // [sb x 2]
  double* t3 = work.d() + 2 * MR * sb;    // [sb]
  double* e3 = t3 + sb;                   // [sb]
  for (int s = 0; s < nsb; ++s) {
    const int b = trans ? s : nsb - 1 - s;
    const int r0 = b * sb, m = std::min(sb, n - r0);
    double* xb = x + r0;
    double* xb3 = x3 + r0;
    const double* Dm = trans ? U + (long)r0 * ldu + r0 : UT.d() + (long)r0 * n + r0;
    const long ldd = trans ? ldu : n;
    const double* Bm = (trans ? Binv.d() : BinvT.d()) + b * blk;
    const int mode = trans ? 1 : 2;
    coldot3(c, m, m, mode, Bm, sb, xb, ldx, xb3, nullptr, 0, nullptr, 1.0, t, sb, t3);                                   // t = B x_b
    for (int it = 0; it < refine; ++it) {
      coldot3(c, m, m, mode, Dm, ldd, t, sb, t3, xb, ldx, xb3, -1.0, e, sb, e3);                                         // e = x_b - T t
      const bool last = (it + 1 == refine);
      coldot3(c, m, m, mode, Bm, sb, e, sb, e3, t, sb, t3, 1.0, last ? xb : t, last ? ldx : sb, last ? xb3 : t3);        // t += B e
    }
    if (refine == 0) {
      HYP_CHECK(hipMemcpy2DAsync(xb, ldx * sizeof(double), t, sb * sizeof(double), m * sizeof(double), MR, hipMemcpyDeviceToDevice, c.stream));
      HYP_CHECK(hipMemcpyAsync(xb3, t3, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, c.stream));
    }
    if (trans) {
      const int rest = n - (r0 + m);
      coldot3(c, m, rest, 0, U + (long)(r0 + m) * ldu + r0, ldu, xb, ldx, xb3, x + r0 + m, ldx, x3 + r0 + m, -1.0, x + r0 + m, ldx, x3 + r0 + m);
    } else {
      coldot3(c, m, r0, 0, UT.d() + r0, n, xb, ldx, xb3, x, ldx, x3, -1.0, x, ldx, x3);
    }
  }
  HYP_CHECK(hipGetLastError());
}

void TriSolvePlan::solve_both(Ctx& c, const double* U, long ldu, double* x, long ldx, int nr, double* x3) {
  HYP_REQUIRE(nr >= 1 && nr <= MR + 1 && (!x3 || nr == MR + 1), "TriSolvePlan::solve_both: 1, 2 or 3 right-hand sides");
  static const bool fused = [] { const char* e = getenv("HYP_TRSV_ONE_LAUNCH"); return !(e && atoi(e) == 1); }();   // (1: one launch per sweep)
  if (fused && ol_usable(c, ldu, nr)) { ol_sweep(c, U, 2, x, ldx, x3, nr); return; }
  for (int pass = 0; pass < 2; ++pass) {
    if (x3) solve_multi3(c, U, ldu, pass == 0, x, ldx, x3);
    else solve_multi(c, U, ldu, pass == 0, x, ldx, nr);
  }
}

void TriSolvePlan::solve_multi(Ctx& c, const double* U, long ldu, bool trans, double* x, long ldx, int nr) {
  if (ol_usable(c, ldu, nr)) { ol_sweep(c, U, trans ? 0 : 1, x, ldx, nullptr, nr); return; }
  if (nr == 1) {
    solve(c, U, ldu, trans, x);
    return;
  }
  if (nr == MR + 1) {   // a third column: the pair through the two-column sweeps, the third through the one-column ones
    solve_multi(c, U, ldu, trans, x, ldx, MR);
    solve(c, U, ldu, trans, x + (long)MR * ldx);
    return;
  }
  HYP_REQUIRE(nr == MR, "TriSolvePlan: 1, 2 or 3 right-hand sides");
  const int nsb = (n + sb - 1) / sb;
  const size_t blk = (size_t)sb * sb;
  work.ensure((size_t)2 * MR * sb * sizeof(double));
  double* t = work.d();               // [sb x 2]
  double* e = work.d() + MR * sb;     // [sb x 2]
  for (int s = 0; s < nsb; ++s) {
    const int b = trans ? s : nsb - 1 - s;
    const int r0 = b * sb, m = std::min(sb, n - r0);
    double* xb = x + r0;
    const double* Dm = trans ? U + (long)r0 * ldu + r0 : UT.d() + (long)r0 * n + r0;
    const long ldd = trans ? ldu : n;
    const double* Bm = (trans ? Binv.d() : BinvT.d()) + b * blk;
    const int mode = trans ? 1 : 2;
    coldot2(c, m, m, mode, Bm, sb, xb, ldx, nullptr, 0, 1.0, t, sb);                 // t = B x_b
    for (int it = 0; it < refine; ++it) {
      coldot2(c, m, m, mode, Dm, ldd, t, sb, xb, ldx, -1.0, e, sb);                   // e = x_b - T t
      const bool last = (it + 1 == refine);
      coldot2(c, m, m, mode, Bm, sb, e, sb, t, sb, 1.0, last ? xb : t, last ? ldx : sb);   // t += B e
    }
    if (refine == 0) {
      HYP_CHECK(hipMemcpy2DAsync(xb, ldx * sizeof(double), t, sb * sizeof(double), m * sizeof(double), MR, hipMemcpyDeviceToDevice, c.stream));
    }
    if (trans) {
      const int rest = n - (r0 + m);
      coldot2(c, m, rest, 0, U + (long)(r0 + m) * ldu + r0, ldu, xb, ldx, x + r0 + m, ldx, -1.0, x + r0 + m, ldx);
    } else {
      coldot2(c, m, r0, 0, UT.d() + r0, n, xb, ldx, x, ldx, -1.0, x, ldx);
    }
  }
  HYP_CHECK(hipGetLastError());
}

// out[:, r] = k0_r x0[:, r] (*) k1_r x1[:, r] (*) k2_r x2[:, r] for r < nr columns in ONE launch, with exactly the roundings of the
// dev_scale_copy / dev_axpby sequence it replaces (a product, then one fused multiply-add per further term; x1 / x2 may be null);
// a leading dimension of 0 = the same vector for every column.  The paired solve issued 9 of those small launches per column.
// A coefficient may live on the device (p0 / p1 non-null: the tau of a solve whose scalars the host has not seen, round 6).
struct LinK { double k0[3], k1[3], k2[3]; const double* p0[3]; const double* p1[3]; };
// (no __restrict__: out may be x0 or x1 -- in-place updates, entry by entry)
__global__ void lincomb_cols_kernel(int n, const double* x0, long ld0, const double* x1, long ld1, const double* x2, long ld2, double* out, long ldo,
                                    LinK k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (i >= n) return;
  const double k0 = k.p0[r] ? *k.p0[r] : k.k0[r];
  const double k1 = k.p1[r] ? *k.p1[r] : k.k1[r];
  double v = __dmul_rn(k0, x0[(long)r * ld0 + i]);
  if (x1) v = __fma_rn(k1, x1[(long)r * ld1 + i], v);
  if (x2) v = __fma_rn(k.k2[r], x2[(long)r * ld2 + i], v);
  out[(long)r * ldo + i] = v;
}
static void lincomb_cols(Ctx& c, int n, int nr, const double* x0, long ld0, const double* x1, long ld1, const double* x2, long ld2, double* out,
                         long ldo, const LinK& k) {
  if (n <= 0 || nr <= 0) return;
  hipLaunchKernelGGL(lincomb_cols_kernel, dim3((n + 255) / 256, nr), dim3(256), 0, c.stream, n, x0, ld0, x1, ld1, x2, ld2, out, ldo, k);
  HYP_CHECK(hipGetLastError());
}

// ---- solve_subsystem3 for nr columns (qrchol.jl:39-85 with p = 0: Q = I) --------------------------------
void SysSolver::solve3_multi(double* sol, const double* rhs, int nr, double* x_third) {
  const size_t d = sizeof(double);
  const long ld3 = n + q;
  HYP_REQUIRE(p == 0, "solve3_multi: p = 0 only");
  if (sol != rhs) ctx.d2d(sol, rhs, (size_t)nr * ld3 * d);
  // x <- lhs^-1 (x + G' z)
  if (dist()) {   // sum the ranks' partial G' z, then add the replicated x once
    m_t.ensure((size_t)nr * n * d);
    gemv_multi(ctx, true, q, n, nr, 1.0, G.d(), q, sol + n, ld3, 0.0, m_t.d(), n);
    allreduce_dev(m_t.d(), (long)nr * n, 0, 1);
    for (int r = 0; r < nr; ++r) dev_axpby(ctx, n, 1.0, m_t.d() + (long)r * n, 1.0, sol + r * ld3);
  } else {
    gemv_multi(ctx, true, q, n, nr, 1.0, G.d(), q, sol + n, ld3, 1.0, sol, ld3);
  }
  join_plan();
  if (tri.ready(nmp)) {
    double* y = use_bk ? bk.gather(ctx, sol, ld3, nr) : sol;   // Bunch-Kaufman factor: P before, D^-1 between, P' after
    const long ldy = use_bk ? nmp : ld3;
    if (x_third && !use_bk && nr == MR) {   // (the constant column's triangular solves ride along: see step_directions)
      tri.solve_both(ctx, lhs_fact.d(), nmp, y, ldy, MR + 1, x_third);
    } else if (!x_third && !use_bk && nr == MR + 1) {   // (the constant column as third column of the pair: the same launches)
      tri.solve_both(ctx, lhs_fact.d(), nmp, y, ldy, MR + 1, y + (long)MR * ldy);
    } else if (!use_bk) {
      HYP_REQUIRE(!x_third, "solve3_multi: the third column needs the Cholesky factor's plan");
      tri.solve_both(ctx, lhs_fact.d(), nmp, y, ldy, nr);
    } else {
      HYP_REQUIRE(!x_third, "solve3_multi: the third column needs the Cholesky factor's plan");
      tri.solve_multi(ctx, lhs_fact.d(), nmp, true, y, ldy, nr);
      bk.dsolve(ctx, y, ldy, nr);
      tri.solve_multi(ctx, lhs_fact.d(), nmp, false, y, ldy, nr);
    }
    if (use_bk) bk.scatter(ctx, y, sol, ld3, nr);
  } else {
    for (int r = 0; r < nr; ++r) tri_solves(sol + r * ld3);
  }
  // z <- H (G x) - z
  gemv_multi(ctx, false, q, n, nr, 1.0, G.d(), q, sol, ld3, 0.0, m_Gx.d(), q);
  for (size_t k = 0; k < cones.size(); ++k) {
    Cone* ck = cones[k];
    if (const int used = run_hess_prod(k, m_HGx.d() + offs[k], q, m_Gx.d() + offs[k], q, nr)) { k += used - 1; continue; }
    if (ck->use_dual_barrier) ck->inv_hess_prod(m_HGx.d() + offs[k], q, m_Gx.d() + offs[k], q, nr);
    else ck->hess_prod(m_HGx.d() + offs[k], q, m_Gx.d() + offs[k], q, nr);
  }
  {
    LinK k{};   // (HGx - z: the value of dev_axpby(1, HGx, -1, z))
    for (int r = 0; r < nr; ++r) { k.k0[r] = 1.0; k.k1[r] = -1.0; }
    lincomb_cols(ctx, q, nr, m_HGx.d(), q, sol + n, ld3, nullptr, 0, sol + n, ld3, k);
  }
}

void SysSolver::get_directions2(double* h_dirs, const double* h_rhss, double mu, double taubar, int max_ref_steps, double res_norm_cutoff,
                                double min_impr_tol, double* res_norms, int* n_solves) {
  HYP_REQUIRE(model_loaded, "sys: load_model first");
  const size_t d = sizeof(double);
  const int dv = dimv(), it = n + p + q, ik = dv - 1;
  *n_solves = 0;
  if (p > 0) {   // (equalities kept: the pair goes through the single-column routine)
    for (int r = 0; r < MR; ++r) {
      int ns = 0;
      res_norms[r] = get_directions(h_dirs + (long)r * dv, h_rhss + (long)r * dv, mu, taubar, max_ref_steps, res_norm_cutoff, min_impr_tol, &ns);
      *n_solves += ns;
    }
    return;
  }
  m_rhs.ensure((size_t)MR * dv * d);
  double* rhs = m_rhs.d();
  ctx.h2d(rhs, h_rhss, (size_t)MR * dv * d);
  Scal rs[MR], dsc[MR];
  for (int r = 0; r < MR; ++r) {
    ctx.zero(rhs + (long)r * dv + it, d);   // tau / kap travel as host scalars
    ctx.zero(rhs + (long)r * dv + ik, d);
    rs[r] = Scal{h_rhss[(long)r * dv + it], h_rhss[(long)r * dv + ik]};
  }
  pair_solve_device(rhs, rs, mu, taubar, max_ref_steps, res_norm_cutoff, min_impr_tol, dsc, res_norms, n_solves);
  ctx.d2h(h_dirs, m_dir.d(), (size_t)MR * dv * d);
  ctx.sync();
  for (int r = 0; r < MR; ++r) {
    h_dirs[(long)r * dv + it] = dsc[r].tau;
    h_dirs[(long)r * dv + ik] = dsc[r].kap;
  }
}

// ---- scalars of a solve on the device (round 6) ---------------------------------------------------------------
// solve_subsystem4's tau (common.jl:155-161) and the kap that follows from it, for nr columns, with the host's operations in the
// host's order (no contraction: the host code is compiled for x86-64 without FMA).  const_mode: dot_const = sc[4] + sc[5] (the
// constant column came out of THIS solve: its two scalar products are the third column's) instead of the argument.
// base_on: the direction's scalars are base - (the solve's) (refinement: dir = dir_best - correction).
struct TauArgs { double rs_tau[2], rs_kap[2], base_tau[2], base_kap[2]; double mu, taubar, dot_const, seq; int nr, const_mode, base_on, hz_sep; };
__global__ void cols_tau_kernel(double* __restrict__ sc, TauArgs a) {
#pragma clang fp contract(off)
  if (threadIdx.x != 0) return;
  double dc = a.dot_const;
  // (hz_sep: cone-sharded -- the h'z of the columns, summed over the ranks, sit in slots of their own)
  if (a.const_mode) dc = sc[SysSolver::SC_SOLVE + 4] + (a.hz_sep ? sc[SysSolver::SC_HZ + 2] : sc[SysSolver::SC_SOLVE + 5]);
  sc[SysSolver::SC_DOTC] = dc;
  const double m2 = a.mu / a.taubar / a.taubar;
  for (int r = 0; r < a.nr; ++r) {
    const double dot_sub = sc[SysSolver::SC_SOLVE + 2 * r] + (a.hz_sep ? sc[SysSolver::SC_HZ + r] : sc[SysSolver::SC_SOLVE + 2 * r + 1]);
    const double sol_tau = (a.rs_tau[r] + a.rs_kap[r] + dot_sub) / (m2 - dc);
    const double sol_kap = -a.mu / a.taubar / a.taubar * sol_tau + a.rs_kap[r];
    sc[SysSolver::SC_CSC + 2 * r] = sol_tau;
    sc[SysSolver::SC_CSC + 2 * r + 1] = sol_kap;
    sc[SysSolver::SC_DSC + 2 * r] = a.base_on ? a.base_tau[r] - sol_tau : sol_tau;
    sc[SysSolver::SC_DSC + 2 * r + 1] = a.base_on ? a.base_kap[r] - sol_kap : sol_kap;
  }
}

// one-column products of column-wise cones go through the multi-column kernels (a column has the same sums wherever it is formed);
// only ever switched on for ONE column -- the paired refinement, which runs on models of column-wise cones only
struct GemvOneGuard {
  Ctx& c;
  bool keep;
  GemvOneGuard(Ctx& ctx, bool on) : c(ctx), keep(ctx.gemv_one) { if (on) c.gemv_one = true; }
  ~GemvOneGuard() { c.gemv_one = keep; }
};

void SysSolver::ensure_d_sc() {
  if (d_sc.bytes >= SC_TOTAL * sizeof(double)) return;
  d_sc.alloc(SC_TOTAL * sizeof(double));
  ctx.zero(d_sc.p, SC_TOTAL * sizeof(double));   // (the tickets of the column maxima start at zero)
}

static bool dir_poll_on() {
  static const bool on = [] { const char* e = getenv("HYP_DIR_POLL"); return !(e && e[0] == '0'); }();
  return on;
}

bool SysSolver::dirs_resident() const {
  static const bool on = [] { const char* e = getenv("HYP_DIR_RESIDENT"); return !(e && e[0] == '0'); }();
  // cone-sharded (round 6, HYP_DIST_RESIDENT, default on): the scalars that cross the ranks -- h'z of a solve, h'z and the norm of a
  // residual -- are summed / max-ed ON THE DEVICE (all-reduces of device slots, the maxima through a slot per rank) and the host of
  // every rank reads the finished block; needs the communicator's layout (the fused exchange)
  static const bool dist_on = [] { const char* e = getenv("HYP_DIST_RESIDENT"); return !(e && e[0] == '0'); }();
  return on && p == 0 && (!dist() || (dist_on && fused_ok()));
}

// solve_system for nr columns (common.jl:129-182, qrchol.jl:16-37); see syssolver.hpp
// with_const: the constant column of update_lhs (qrchol.jl:191-197: rhs_const = (-c, H h), solved once per iteration) rides
// along as a THIRD column of this call's solve_subsystem3 -- its two passes over G, its cone product and its scalar products
// cost the pair nothing extra; sol_const / dot_const are set before the pair's tau is formed from them (common.jl:155-161)
void SysSolver::cols_solve(double* sol, const double* rhs, int nr, const Scal* rs, double mu, double taubar, bool with_const, bool joint_const,
                           bool resident, bool both, const Scal* base, Scal* dsc_host) {
  const size_t d = sizeof(double);
  const int dv = dimv();
  HYP_REQUIRE(p == 0 && nr >= 1 && nr <= MR, "cols_solve: p = 0, one or two columns");
  HYP_REQUIRE(!(with_const || joint_const) || nr == MR, "cols_solve: the constant column rides with a pair");
  const int oz = n, os = n + q + 1;
  const long ld3 = n + q;
  for (DBuf* b : {&m_subr, &m_subs}) b->ensure((size_t)(MR + 1) * ld3 * d);
  for (DBuf* b : {&m_Gx, &m_HGx, &m_Gxd}) b->ensure((size_t)(MR + 1) * q * d);
  ensure_d_sc();
  double* sr = m_subr.d();
  double* ss = m_subs.d();
  double* sc = d_sc.d();
  // with_const: the constant column is a genuine third INPUT column, (x, z, s) = (-c, -h, 0), so that its right-hand side H h
  // (= -H (-h) - 0 below) and its H (G x) come out of the same three-column cone products (z_const = H G x - H h cancels to ~mu
  // of its terms late in a solve)
  const int ncol = with_const ? nr + 1 : nr;
  if (with_const) {
    double* rc = const_cast<double*>(rhs) + (long)MR * dv;                    // (the caller's buffer holds MR + 1 Point vectors)
    ctx.zero(rc, (size_t)dv * d);
    dev_scale_copy(ctx, n, -1.0, mc.d(), rc);
    dev_scale_copy(ctx, q, -1.0, mh.d(), rc + oz);
  }
  for (int r = 0; r < ncol; ++r) ctx.d2d(sr + r * ld3, rhs + (long)r * dv, (size_t)n * d);
  LinK neg2{};   // y <- -y - x: the value of dev_axpby(-1, x, -1, y) (two exact negations, one rounded sum)
  for (int r = 0; r < MR + 1; ++r) { neg2.k0[r] = -1.0; neg2.k1[r] = -1.0; }
  for (size_t k = 0; k < cones.size(); ++k) {
    Cone* ck = cones[k];
    const int o = offs[k], dk = ck->dim;
    if (ck->use_dual_barrier) {
      for (int r = 0; r < ncol; ++r) {
        double* tmp = ss + r * ld3 + oz + o;
        dev_scale_copy(ctx, dk, -1.0, rhs + (long)r * dv + oz + o, tmp);
        dev_axpby(ctx, dk, -1.0, rhs + (long)r * dv + os + o, 1.0, tmp);
      }
      ck->inv_hess_prod(sr + oz + o, ld3, ss + oz + o, ld3, ncol);
    } else if (const int used = run_hess_prod(k, sr + oz + o, ld3, rhs + oz + o, dv, ncol)) {
      lincomb_cols(ctx, offs[k + used] - o, ncol, sr + oz + o, ld3, rhs + os + o, dv, nullptr, 0, sr + oz + o, ld3, neg2);   // -H z - s, all columns in one launch
      k += used - 1;
    } else {
      ck->hess_prod(sr + oz + o, ld3, rhs + oz + o, dv, ncol);
      lincomb_cols(ctx, dk, ncol, sr + oz + o, ld3, rhs + os + o, dv, nullptr, 0, sr + oz + o, ld3, neg2);
    }
  }
  {
    GemvOneGuard g1(ctx, ncol == 1);
    solve3_multi(ss, sr, ncol, joint_const ? sol_const.d() : nullptr);
  }
  const bool dist_res = resident && dist();
  {
    DotSpecs sp;
    for (int r = 0; r < ncol; ++r) {
      sp.add(n, mc.d(), ss + r * ld3, sc + SC_SOLVE + 2 * r);
      sp.add(q, mh.d(), ss + r * ld3 + oz, dist_res ? sc + SC_HZ + r : sc + SC_SOLVE + 2 * r + 1);
    }
    if (joint_const) {   // the rest of the constant solve (solve3 after its triangular solves) and its two dot products
      update_const_post();
      sp.add(n, mc.d(), sol_const.d(), sc + SC_SOLVE + 2 * MR);
      sp.add(q, mh.d(), sol_const.d() + n, sc + SC_SOLVE + 2 * MR + 1);
    }
    dev_dots(ctx, sp);
  }
  if (dist_res) allreduce_dev(sc + SC_HZ, ncol, 0, 2);   // h' z over all ranks' rows, on the device
  const bool own_const = with_const || joint_const;
  LinK k{};
  if (resident) {
    TauArgs a{};
    for (int r = 0; r < nr; ++r) {
      a.rs_tau[r] = rs[r].tau; a.rs_kap[r] = rs[r].kap;
      if (base) { a.base_tau[r] = base[r].tau; a.base_kap[r] = base[r].kap; }
    }
    a.mu = mu; a.taubar = taubar; a.dot_const = dot_const; a.nr = nr; a.const_mode = own_const ? 1 : 0; a.base_on = base ? 1 : 0;
    a.seq = (double)(++sc_seq);
    a.hz_sep = dist_res ? 1 : 0;
    hipLaunchKernelGGL(cols_tau_kernel, dim3(1), dim3(64), 0, ctx.stream, sc, a);
    HYP_CHECK(hipGetLastError());
    for (int r = 0; r < nr; ++r) { k.k0[r] = 1.0; k.p1[r] = sc + SC_CSC + 2 * r; }
  } else {
    double* hp = ctx.h_sc();
    ctx.d2h(hp, sc, 2 * (MR + 1) * d);
    ctx.sync();
    if (joint_const) dot_const = hp[2 * MR] + hp[2 * MR + 1];
    if (dist()) {   // h' z over all ranks' rows
      double hz[MR + 1] = {hp[1], hp[3], hp[5]};
      allreduce_host(hz, ncol, 0, 2);
      hp[1] = hz[0];
      hp[3] = hz[1];
      hp[5] = hz[2];
    }
    if (with_const) dot_const = hp[2 * MR] + hp[2 * MR + 1];
    for (int r = 0; r < nr; ++r) {
      const double dot_sub = hp[2 * r] + hp[2 * r + 1];
      const double sol_tau = (rs[r].tau + rs[r].kap + dot_sub) / (mu / taubar / taubar - dot_const);
      const double sol_kap = -mu / taubar / taubar * sol_tau + rs[r].kap;
      dsc_host[r].tau = base ? base[r].tau - sol_tau : sol_tau;
      dsc_host[r].kap = base ? base[r].kap - sol_kap : sol_kap;
      k.k0[r] = 1.0; k.k1[r] = sol_tau;
    }
  }
  if (with_const) ctx.d2d(sol_const.p, ss + (long)MR * ld3, (size_t)ld3 * d);
  lincomb_cols(ctx, (int)ld3, nr, ss, ld3, sol_const.d(), 0, nullptr, 0, sol, dv, k);   // sol = sol_sub + sol_tau sol_const
  // sol.s = h tau - rhs.z - G sol.x  (G sol.x from the rounded sol.x, see solve_system; kept for the residual)
  // (both: with a residual of THIS vector to follow, G' sol.z of apply_lhs rides along in the same pass over G)
  if (both) {
    m_t.ensure((size_t)MR * n * d);
    gemv_both(ctx, q, n, nr, G.d(), q, sol, dv, 0.0, m_Gxd.d(), q, sol + oz, dv, 0.0, m_t.d(), n);
  } else {
    GemvOneGuard g1(ctx, nr == 1);
    gemv_multi(ctx, false, q, n, nr, 1.0, G.d(), q, sol, dv, 0.0, m_Gxd.d(), q);
  }
  {
    LinK k2{};
    for (int r = 0; r < nr; ++r) { k2.k0[r] = k.k1[r]; k2.p0[r] = k.p1[r]; k2.k1[r] = -1.0; k2.k2[r] = -1.0; }
    lincomb_cols(ctx, q, nr, mh.d(), 0, rhs + oz, dv, m_Gxd.d(), q, sol + os, dv, k2);
  }
}

// Behind the fused exchange of a cone-sharded residual (tail = [h'z sums (nr) | a slot per rank and column: the rank's maximum over
// its rows]): the summed h'z and the residual's maximum over all ranks' rows and the replicated x rows (NaN wins) into the scalar
// block -- what allreduce_fused's caller does on the host, with the host's rules
__global__ void dist_resid_finish_kernel(const double* __restrict__ tail, int nr, int world, const double* __restrict__ xmax, double* __restrict__ sc) {
  const int r = threadIdx.x;
  if (r >= nr) return;
  sc[SysSolver::SC_RESD + 2 * r + 1] = tail[r];
  double m = tail[nr + r];
  bool bad = (m != m);
  for (int w = 1; w < world; ++w) {
    const double v = tail[nr + w * nr + r];
    bad = bad || (v != v);
    if (v > m) m = v;
  }
  const double b = xmax[r];
  sc[SysSolver::SC_AMAX + r] = (bad || b != b) ? __builtin_nan("") : fmax(m, b);
}

// residual of nr directions (apply_lhs, common.jl:79-121, minus rhs); see syssolver.hpp
void SysSolver::cols_residual(double* res, const double* dir, const double* rhs, int nr, const Scal* dsc_host, bool resident, bool fresh, bool both) {
  const size_t d = sizeof(double);
  const int dv = dimv();
  const int oz = n, os = n + q + 1;
  double* sc = d_sc.d();
  if (fresh && both) {
    m_t.ensure((size_t)MR * n * d);
    gemv_both(ctx, q, n, nr, G.d(), q, dir, dv, 0.0, m_Gxd.d(), q, dir + oz, dv, 0.0, m_t.d(), n);
  } else if (fresh) {
    GemvOneGuard g1(ctx, nr == 1);
    gemv_multi(ctx, false, q, n, nr, 1.0, G.d(), q, dir, dv, 0.0, m_Gxd.d(), q);
  }
  {
    LinK k{};
    for (int r = 0; r < nr; ++r) {
      if (resident) k.p0[r] = sc + SC_DSC + 2 * r;
      else k.k0[r] = dsc_host[r].tau;
      k.k1[r] = -1.0; k.k2[r] = -1.0;
    }
    if (both && !dist()) {   // res.x = c tau + G' z in one launch (the value of the separate addition: a rounded product, one rounded sum)
      LinK kx = k;
      for (int r = 0; r < nr; ++r) kx.k1[r] = 1.0;
      lincomb_cols(ctx, n, nr, mc.d(), 0, m_t.d(), n, nullptr, 0, res, dv, kx);
    } else {
      lincomb_cols(ctx, n, nr, mc.d(), 0, nullptr, 0, nullptr, 0, res, dv, k);                     // res.x = c tau (+ G' z below)
    }
    lincomb_cols(ctx, q, nr, mh.d(), 0, dir + os, dv, m_Gxd.d(), q, res + oz, dv, k);              // res.z = h tau - s - G x
  }
  // Sharded, round 4: the residual's three exchanges -- G' z (n-vectors, SUM), the h' z of the directions (SUM) and the residual
  // norms (MAX) -- travel in ONE all-reduce (allreduce_fused): everything a rank contributes is local (the z / s rows of the
  // residual do not depend on the summed G' z; only the x rows do, and those are replicated afterwards), so the local maxima
  // and scalar products are formed first and ride behind the n-vectors.
  const bool fuse = dist() && fused_ok();
  if (dist() || both) {
    if (!both) {
      m_t.ensure((size_t)MR * n * d);
      gemv_multi(ctx, true, q, n, nr, 1.0, G.d(), q, dir + oz, dv, 0.0, m_t.d(), n);
    }
    if (dist() && !fuse) allreduce_dev(m_t.d(), (long)nr * n, 0, 3);
    if (dist() && !fuse)
      for (int r = 0; r < nr; ++r) dev_axpby(ctx, n, 1.0, m_t.d() + (long)r * n, 1.0, res + (long)r * dv);
  } else {
    GemvOneGuard g1(ctx, nr == 1);
    gemv_multi(ctx, true, q, n, nr, 1.0, G.d(), q, dir + oz, dv, 1.0, res, dv);
  }
  for (size_t k = 0; k < cones.size(); ++k) {   // res.s_k = H_k prim_dir_k + dual_dir_k
    Cone* ck = cones[k];
    const int o = offs[k], dk = ck->dim;
    const int po = ck->use_dual_barrier ? oz + o : os + o, du = ck->use_dual_barrier ? os + o : oz + o;
    LinK add2{};   // y <- y + x
    for (int r = 0; r < MR; ++r) { add2.k0[r] = 1.0; add2.k1[r] = 1.0; }
    if (const int used = run_hess_prod(k, res + os + o, dv, dir + po, dv, nr)) {   // (PosSemidefTri: hess_prod_slow! = hess_prod!)
      lincomb_cols(ctx, offs[k + used] - o, nr, res + os + o, dv, dir + du, dv, nullptr, 0, res + os + o, dv, add2);
      k += used - 1;
      continue;
    }
    ck->hess_prod_slow(res + os + o, dv, dir + po, dv, nr);
    lincomb_cols(ctx, dk, nr, res + os + o, dv, dir + du, dv, nullptr, 0, res + os + o, dv, add2);
  }
  {
    DotSpecs sp;
    for (int r = 0; r < nr; ++r) {
      sp.add(n, mc.d(), dir + (long)r * dv, sc + SC_RESD + 2 * r);
      sp.add(q, mh.d(), dir + (long)r * dv + oz, sc + SC_RESD + 2 * r + 1);
    }
    dev_dots(ctx, sp);
  }
  double* hp = ctx.h_sc();
  if (fuse && resident) {   // (round 6) the same exchange with its sums and maxima finished ON THE DEVICE: no host round trip here
    double* ds = ctx.dscal.d();
    for (int r = 0; r < nr; ++r) dev_sub_absmax(ctx, dv - oz, res + (long)r * dv + oz, rhs + (long)r * dv + oz, ds + 8 + r);   // this rank's rows
    if (m_t.bytes < ((size_t)MR * n + 64 + 2 * (size_t)MR * comm_world_) * d) {   // (room for the tail behind the n-vectors)
      DBuf bigger(((size_t)MR * n + 64 + 2 * (size_t)MR * comm_world_) * d);
      ctx.d2d(bigger.p, m_t.p, (size_t)MR * n * d);
      ctx.sync();
      m_t = std::move(bigger);
    }
    FusedTail t;
    t.nsum = nr; t.nmax = nr;
    for (int r = 0; r < nr; ++r) { t.sum_src[r] = sc + SC_RESD + 2 * r + 1; t.max_src[r] = ds + 8 + r; }
    allreduce_fused_dev(m_t.d(), (long)nr * n, t, 3);
    for (int r = 0; r < nr; ++r) dev_axpby(ctx, n, 1.0, m_t.d() + (long)r * n, 1.0, res + (long)r * dv);
    for (int r = 0; r < nr; ++r) dev_sub_absmax(ctx, oz, res + (long)r * dv, rhs + (long)r * dv, ds + 12 + r);                      // the replicated x rows
    hipLaunchKernelGGL(dist_resid_finish_kernel, dim3(1), dim3(64), 0, ctx.stream, m_t.d() + (long)nr * n, nr, comm_world_, ds + 12, sc);
    HYP_CHECK(hipGetLastError());
    return;
  }
  if (fuse) {
    HYP_REQUIRE(nr == MR, "cols_residual: the fused exchange carries a pair");
    double* ds = ctx.dscal.d();
    for (int r = 0; r < MR; ++r) dev_sub_absmax(ctx, dv - oz, res + (long)r * dv + oz, rhs + (long)r * dv + oz, ds + 8 + r);   // this rank's rows
    if (m_t.bytes < ((size_t)MR * n + 64 + 2 * (size_t)MR * comm_world_) * d) {   // (room for the tail behind the n-vectors)
      DBuf bigger(((size_t)MR * n + 64 + 2 * (size_t)MR * comm_world_) * d);
      ctx.d2d(bigger.p, m_t.p, (size_t)MR * n * d);
      ctx.sync();
      m_t = std::move(bigger);
    }
    FusedTail t;
    t.nsum = MR; t.nmax = MR;
    for (int r = 0; r < MR; ++r) { t.sum_src[r] = sc + SC_RESD + 2 * r + 1; t.max_src[r] = ds + 8 + r; }
    double ho[2 * MR];
    allreduce_fused(m_t.d(), (long)MR * n, t, ho, 3);
    for (int r = 0; r < MR; ++r) dev_axpby(ctx, n, 1.0, m_t.d() + (long)r * n, 1.0, res + (long)r * dv);
    for (int r = 0; r < MR; ++r) dev_sub_absmax(ctx, oz, res + (long)r * dv, rhs + (long)r * dv, sc + SC_AMAX + r);                // the replicated x rows
    ctx.d2h(hp, sc, SC_N * d);
    ctx.sync();
    for (int r = 0; r < MR; ++r) {
      hp[SC_RESD + 2 * r + 1] = ho[r];
      const double a = ho[MR + r], b = hp[SC_AMAX + r];
      hp[SC_AMAX + r] = (a != a || b != b) ? __builtin_nan("") : std::max(a, b);
    }
    return;
  }
  dev_sub_absmax_cols(ctx, dv, nr, res, rhs, dv, sc + SC_AMAX, sc + SC_WORK);   // (one launch for the pair: a maximum does not depend on the order)
  if (dist()) {
    ctx.d2h(hp, sc, SC_N * d);
    ctx.sync();
    double hz[MR] = {hp[SC_RESD + 1], hp[SC_RESD + 3]};
    allreduce_host(hz, MR, 0, 4);
    hp[SC_RESD + 1] = hz[0];
    hp[SC_RESD + 3] = hz[1];
    double v[2 * MR];   // residual norms: max over the ranks, NaN flags first
    for (int r = 0; r < MR; ++r) {
      const double m = hp[SC_AMAX + r];
      v[r] = (m != m) ? 1.0 : 0.0;
      v[MR + r] = (m != m) ? 0.0 : m;
    }
    allreduce_host(v, 2 * MR, 1, 5);
    for (int r = 0; r < MR; ++r) hp[SC_AMAX + r] = (v[r] > 0.5) ? __builtin_nan("") : v[MR + r];
  }
}

// The scalar block written into the pinned mirror by ONE wavefront: every word but the stamp, a system-scope fence, then the stamp --
// so that a host that sees the stamp sees the block.  (First form of round 6: a plain copy of the block with the stamp as its last
// word.  A copy kernel's lanes store independently and their writes reach host memory in no particular order: one solve in ~50 read
// a stale word and took another path -- tools/stress_determinism.py, EXPERIMENTS r06-11.)
__global__ __launch_bounds__(64) void publish_scalars_kernel(const double* __restrict__ sc, double* __restrict__ mirror, double seq,
                                                             const int* __restrict__ fact_info) {
  const int i = threadIdx.x;
  if (i < SysSolver::SC_N && i != SysSolver::SC_SEQ) {
    // (slot SC_INFO: the info word of the factorization queued in front of this solve rides along -- the host reads it from the block)
    const double v = (i == SysSolver::SC_INFO) ? (fact_info ? (double)fact_info[0] : 0.0) : sc[i];
    __hip_atomic_store(mirror + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  if (i == 0) __hip_atomic_store(mirror + SysSolver::SC_SEQ, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void SysSolver::cols_read_scalars(bool resident) {
  if (dist() && !resident) return;   // (the host-scalar sharded residual has read and completed the mirror itself)
  if (dir_poll_on() && ctx.h_sc_dev != nullptr) {
    hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(64), 0, ctx.stream, d_sc.d(), ctx.h_sc_dev, (double)sc_seq, (const int*)d_info.p);
    HYP_CHECK(hipGetLastError());
    return;
  }
  ctx.d2h(ctx.h_sc(), d_sc.d(), SC_N * sizeof(double));
}

// The host's wait for the scalars of the solve it queued last (resident flow).  A stream synchronisation blocks in the runtime until
// the queue's completion signal has travelled back -- tens of microseconds, and far more under a profiler --, while the device sits
// idle; the mirror block itself tells when it has landed: its last word is the solve's sequence number, written by
// publish_scalars_kernel BEHIND a system-scope fence (when the stamp is there, the block is, and so is everything queued in front
// of that kernel).  The host spins on that word; every few thousand spins it asks the stream
// for an error (a faulted queue would never deliver the stamp), and a stream found idle with the old stamp is an error too.
void SysSolver::wait_scalars() {
  if (!dir_poll_on() || ctx.h_sc_dev == nullptr) { ctx.sync(); return; }
  volatile const double* seqp = ctx.h_sc() + SC_SEQ;
  const double want = (double)sc_seq;
  unsigned spins = 0;
  int idle_seen = 0;
  while (*seqp != want) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFu) == 0) {
      const hipError_t e = hipStreamQuery(ctx.stream);
      if (e == hipSuccess) {
        if (++idle_seen > 4) {   // (the queue has drained and the stamp is not ours: take the runtime's word and re-read)
          ctx.sync();
          HYP_REQUIRE(*seqp == want, "step_directions: the scalar mirror was not updated by the queued solve");
          break;
        }
      } else if (e != hipErrorNotReady) {
        HYP_CHECK(e);
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (ctx.ol_abort_host && *ctx.ol_abort_host) ctx.check_persistent_abort();
}

void SysSolver::cols_finish(int nr, const Scal* rs, double mu, double taubar, bool resident, Scal* dsc, Scal* rsc, double* res_norms) {
  const double* hp = ctx.h_sc();
  for (int r = 0; r < nr; ++r) {
    if (resident) dsc[r] = Scal{hp[SC_DSC + 2 * r], hp[SC_DSC + 2 * r + 1]};
    rsc[r].tau = -hp[SC_RESD + 2 * r] - hp[SC_RESD + 2 * r + 1] - dsc[r].kap - rs[r].tau;
    rsc[r].kap = mu / taubar * dsc[r].tau / taubar + dsc[r].kap - rs[r].kap;
    const double m = hp[SC_AMAX + r];
    res_norms[r] = (m != m || rsc[r].tau != rsc[r].tau || rsc[r].kap != rsc[r].kap)
                       ? __builtin_nan("")
                       : std::max(m, std::max(std::fabs(rsc[r].tau), std::fabs(rsc[r].kap)));
  }
}

// The refinement loop of get_directions (common.jl:38-72) for the columns of a pair that need it, through the column routines:
// a step = solve_system on the residual(s), dir = dir_best - correction, the new residual(s), ONE host round trip; two columns
// that both need a step share its passes over G and the factor (each column's sums are those it gets alone).  v_tmp holds
// the best directions so far.  Every column follows its own acceptance / stopping rule, exactly as SysSolver::refine does.
void SysSolver::refine_cols(double* rhs, double* dir, double* res, const Scal* rs, Scal* dsc, Scal* rsc, double* res_norms, double mu,
                            double taubar, int max_ref_steps, double res_norm_cutoff, double min_impr_tol, int* n_solves, bool resident) {
  const size_t d = sizeof(double);
  const int dv = dimv();
  v_tmp.ensure((size_t)MR * dv * d);
  double* tmp = v_tmp.d();
  bool active[MR], prev_slow[MR];
  Scal tsc[MR];
  double prev_norm[MR];
  int steps[MR];
  for (int r = 0; r < MR; ++r) {
    active[r] = res_norms[r] > res_norm_cutoff;
    prev_slow[r] = false;
    tsc[r] = dsc[r];
    prev_norm[r] = res_norms[r];
    steps[r] = 0;
    if (active[r]) ctx.d2d(tmp + (long)r * dv, dir + (long)r * dv, (size_t)dv * d);
  }
  const bool both = gemv_both_ok(q, n, G.d(), q);
  while (true) {
    for (int r = 0; r < MR; ++r) active[r] = active[r] && steps[r] < max_ref_steps;
    if (!active[0] && !active[1]) break;
    const int c0 = active[0] ? 0 : 1, nr = (active[0] && active[1]) ? 2 : 1;
    const long o = (long)c0 * dv;
    Scal csc[MR];
    // dir = dir_best - solve(res)
    cols_solve(dir + o, res + o, nr, rsc + c0, mu, taubar, false, false, resident, false, tsc + c0, csc);
    *n_solves += nr;
    for (int r = 0; r < nr; ++r) dev_axpby(ctx, dv, 1.0, tmp + o + (long)r * dv, -1.0, dir + o + (long)r * dv);
    cols_residual(res + o, dir + o, rhs + o, nr, csc, resident, true, both);
    cols_read_scalars(resident);
    if (resident) wait_scalars();
    else ctx.sync();
    Scal rsc2[MR];
    double nn[MR];
    cols_finish(nr, rs + c0, mu, taubar, resident, csc, rsc2, nn);
    for (int i = 0; i < nr; ++i) {
      const int r = c0 + i;
      ++steps[r];
      if (!(nn[i] < res_norms[r])) {   // (>= or NaN: keep the previous direction)
        ctx.d2d(dir + (long)r * dv, tmp + (long)r * dv, (size_t)dv * d);
        dsc[r] = tsc[r];
        active[r] = false;
        continue;
      }
      ctx.d2d(tmp + (long)r * dv, dir + (long)r * dv, (size_t)dv * d);
      dsc[r] = csc[i];
      tsc[r] = csc[i];
      rsc[r] = rsc2[i];
      res_norms[r] = nn[i];
      if (res_norms[r] < res_norm_cutoff) { active[r] = false; continue; }
      const bool cur_slow = res_norms[r] > min_impr_tol * prev_norm[r];
      if (prev_slow[r] && cur_slow) { active[r] = false; continue; }
      prev_norm[r] = res_norms[r];
      prev_slow[r] = cur_slow;
    }
  }
}

// two right-hand sides already on the device (rhs2 = two Point vectors, tau / kap slots zero, scalars in rs): directions are left
// in m_dir, their residuals in m_res, the scalars in d_sc (and, queued, in its pinned mirror)
void SysSolver::pair_enqueue(double* rhs, const Scal* rs, double mu, double taubar, int max_ref_steps, bool with_const, bool joint_const,
                             bool resident, Scal* dsc_host, bool read_scalars) {
  const size_t d = sizeof(double);
  const int dv = dimv(), it = n + p + q, ik = dv - 1;
  HYP_REQUIRE(p == 0, "pair_solve_device: p = 0 only");
  for (DBuf* b : {&m_dir, &m_res}) b->ensure((size_t)MR * dv * d);
  double* dir = m_dir.d();
  double* res = m_res.d();
  {   // (the tau / kap slots of the work vectors stay zero: eight doubles, one launch)
    ZeroSlots z;
    for (int r = 0; r < MR; ++r) {
      z.add(dir + (long)r * dv + it); z.add(dir + (long)r * dv + ik);
      z.add(res + (long)r * dv + it); z.add(res + (long)r * dv + ik);
    }
    dev_zero_slots(ctx, z);
  }
  const bool both = (max_ref_steps > 0) && gemv_both_ok(q, n, G.d(), q);
  cols_solve(dir, rhs, MR, rs, mu, taubar, with_const, joint_const, resident, both, nullptr, dsc_host);
  if (max_ref_steps > 0) cols_residual(res, dir, rhs, MR, dsc_host, resident, false, both);
  if (read_scalars) cols_read_scalars(resident);
}

void SysSolver::pair_finish(double* rhs, const Scal* rs, double mu, double taubar, int max_ref_steps, double res_norm_cutoff,
                            double min_impr_tol, bool with_const, bool joint_const, bool resident, Scal* dsc, double* res_norms, int* n_solves) {
  const size_t d = sizeof(double);
  const int dv = dimv();
  double* dir = m_dir.d();
  double* res = m_res.d();
  const double* hp = ctx.h_sc();
  if (resident && (with_const || joint_const)) dot_const = hp[SC_DOTC];
  *n_solves += MR;
  Scal rsc[MR];
  if (max_ref_steps <= 0) {
    for (int r = 0; r < MR; ++r) {
      if (resident) dsc[r] = Scal{hp[SC_DSC + 2 * r], hp[SC_DSC + 2 * r + 1]};
      res_norms[r] = 0.0;
    }
    return;
  }
  cols_finish(MR, rs, mu, taubar, resident, dsc, rsc, res_norms);
  Gx_dir_valid = false;
  if (!(res_norms[0] > res_norm_cutoff) && !(res_norms[1] > res_norm_cutoff)) return;
  bool cones_ok = true;
  for (const Cone* ck : cones) cones_ok = cones_ok && ck->products_columnwise();
  static const bool paired = [] { const char* e = getenv("HYP_REFINE_PAIRED"); return !(e && e[0] == '0'); }();
  if (paired && cones_ok && (!dist() || resident)) {
    refine_cols(rhs, dir, res, rs, dsc, rsc, res_norms, mu, taubar, max_ref_steps, res_norm_cutoff, min_impr_tol, n_solves, resident);
    return;
  }
  // ---- a column that needs refinement continues alone (single-column routines)
  for (int r = 0; r < MR; ++r) {
    if (!(res_norms[r] > res_norm_cutoff)) continue;
    double* tmp = v_tmp.d();
    ctx.d2d(tmp, dir + (long)r * dv, (size_t)dv * d);
    res_norms[r] = refine(rhs + (long)r * dv, dir + (long)r * dv, res + (long)r * dv, tmp, rs[r], dsc[r], rsc[r], res_norms[r], mu, taubar,
                          max_ref_steps, res_norm_cutoff, min_impr_tol, n_solves);
  }
}

void SysSolver::pair_solve_device(double* rhs, const Scal* rs, double mu, double taubar, int max_ref_steps, double res_norm_cutoff,
                                  double min_impr_tol, Scal* dsc, double* res_norms, int* n_solves, bool with_const, bool joint_const) {
  const bool resident = dirs_resident();
  pair_enqueue(rhs, rs, mu, taubar, max_ref_steps, with_const, joint_const, resident, dsc);
  ctx.sync();
  pair_finish(rhs, rs, mu, taubar, max_ref_steps, res_norm_cutoff, min_impr_tol, with_const, joint_const, resident, dsc, res_norms, n_solves);
}


// ---- right-hand sides of the stepper on the device (steppers/common.jl:7-118) -----------------------------
// stage 0: columns (cent, pred); stage 1: columns (centadj from dir_cent, predadj from dir_pred).  rhs2 = two
// Point vectors on the device (tau / kap slots zero, the scalars go to rs).
// The acceptance tests of the third-order terms (steppers/common.jl:37-55, 96-113) taken on the device (round 6): for cone
// blockIdx.y of nc cones of dk rows each, from its four scalar products, the centering column gets dder3 and the prediction column
// H dir + dder3 where the test passes (the host's comparison in the host's operations; the columns were zeroed before)
__global__ void dder3_gate_kernel(int dk, const double* __restrict__ dots, double irtrtmu, double rteps, const double* __restrict__ D3c,
                                  const double* __restrict__ Hqp, const double* __restrict__ D3p, double* __restrict__ c0, double* __restrict__ c1) {
#pragma clang fp contract(off)
  const int k = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dk) return;
  const double* dd = dots + 4 * k;
  const long e = (long)k * dk + i;
  {
    const double dot1 = dd[0], dot2 = dd[1];
    if (fabs(dot1 - dot2) / (rteps + fabs(dot2)) < 1e-4) c0[e] = D3c[e];
  }
  {
    const double dot1 = dd[2], dot2 = irtrtmu * dd[3];
    if (fabs(dot1 - dot2) / (rteps + fabs(dot2)) < 1e-4) c1[e] = Hqp[e] + D3p[e];
  }
}

void SysSolver::build_rhs_pair(int stage, double* rhs2, const double* pt, double mu, double tau, double kap, double tau_residual,
                               const double* dirs2, const double* dir_tau2, double* rs_flat, bool resident) {
  const size_t d = sizeof(double);
  const int dv = dimv(), oz = n + p, os = n + p + q + 1;
  ctx.zero(rhs2, (size_t)MR * dv * d);
  double* c0 = rhs2;
  double* c1 = rhs2 + dv;
  Scal* rs = reinterpret_cast<Scal*>(rs_flat);
  if (stage == 0) {
    const double rtmu = std::sqrt(mu);
    ctx.d2d(c1, s_resid.d(), (size_t)n * d);                       // pred: x, z residuals (:7-24); cent: zeros (:63-84)
    ctx.d2d(c1 + oz, s_resid.d() + n + p, (size_t)q * d);
    if (const PsdRun* r = whole_model_run()) {   // one run of equal PSD cones: the same three vector operations over all of them
      run_g.ensure((size_t)q * d);
      run_grad(*r, run_g.d());
      dev_scale_copy(ctx, q, -1.0, pt + oz, c0 + os);
      dev_axpby(ctx, q, -rtmu, run_g.d(), 1.0, c0 + os);
      dev_scale_copy(ctx, q, -1.0, pt + oz, c1 + os);
      rs[0] = Scal{0.0, -kap + mu / tau};
      rs[1] = Scal{tau_residual, -kap};
      return;
    }
    for (size_t k = 0; k < cones.size(); ++k) {
      Cone* ck = cones[k];
      const int o = offs[k], dk = ck->dim;
      const double* dual = pt + (ck->use_dual_barrier ? os : oz) + o;
      dev_scale_copy(ctx, dk, -1.0, dual, c0 + os + o);            // cent: -dual - sqrt(mu) grad
      dev_axpby(ctx, dk, -rtmu, ck->get_grad(), 1.0, c0 + os + o);
      dev_scale_copy(ctx, dk, -1.0, dual, c1 + os + o);            // pred: -dual
    }
    rs[0] = Scal{0.0, -kap + mu / tau};
    rs[1] = Scal{tau_residual, -kap};
    return;
  }
  // stage 1: third-order adjustments; all cone products first, the acceptance tests (host scalars) after ONE sync
  const double rteps = std::sqrt(2.220446049250313e-16);
  const double irtrtmu = 1.0 / std::sqrt(std::sqrt(mu));
  for (DBuf* b : {&m_Gx, &m_HGx, &m_Gxd}) b->ensure((size_t)MR * q * d);
  double* scal = m_Gx.d();     // [q x 2] irtrtmu * prim_dir
  double* Hq = m_HGx.d();      // [q x 2] H * (scaled / unscaled) prim_dir
  double* D3 = m_Gxd.d();      // [q x 2] dder3
  const size_t nc = cones.size();
  s_dots.ensure(std::max<size_t>(4 * nc, 4) * d);
  double* dots = s_dots.d();
  const PsdRun* wr = whole_model_run();
  if (wr) {   // one run of equal PSD cones: scaled directions, H products, dder3 and the four scalar products per cone, batched
    for (int r = 0; r < MR; ++r) {
      const double* prim = dirs2 + (long)r * dv + os;
      dev_scale_copy(ctx, q, irtrtmu, prim, scal + (long)r * q);
      const int used = run_hess_prod(0, Hq + (long)r * q, q, r == 0 ? scal : prim, q, 1);
      HYP_REQUIRE(used == (int)nc, "build_rhs_pair: run product");
      run_dder3(*wr, scal + (long)r * q, D3 + (long)r * q);
      seg_dots(ctx, (int)nc, cones[0]->dim, D3 + (long)r * q, wr->point, 2 * r, 4, dots);
      seg_dots(ctx, (int)nc, cones[0]->dim, scal + (long)r * q, Hq + (long)r * q, 2 * r + 1, 4, dots);
    }
  }
  // (round 6, HYP_DDER3_PAIRED, default on) PosSemidefTri: the two columns of a cone through ONE set of launches -- the Hessian products
  // as a two-column product, the third-order term's five GEMMs stacked / batched over the columns, the four scalar products in one
  // launch; every column gets the sums it gets alone (tests/test_hip_switches.py)
  static const bool d3_paired = [] { const char* e = getenv("HYP_DDER3_PAIRED"); return !(e && e[0] == '0'); }();
  std::vector<char> done_k(nc, 0);
  for (size_t k = 0; k < nc && !wr && d3_paired; ++k) {
    PsdCone* pk = dynamic_cast<PsdCone*>(cones[k]);
    if (!pk || !pk->use_dder3() || MR != 2) continue;
    const int o = offs[k], dk = pk->dim;
    const int po = (pk->use_dual_barrier ? oz : os) + o;
    m_subs.ensure((size_t)(MR + 1) * (n + q) * d);
    double* Hin = m_subs.d();   // [dk x 2]: (scaled dir_cent, dir_pred) -- free between the two paired solves
    LinK ks{}, kh{};
    for (int r = 0; r < MR; ++r) { ks.k0[r] = irtrtmu; kh.k0[r] = (r == 0) ? irtrtmu : 1.0; }
    lincomb_cols(ctx, dk, MR, dirs2 + po, dv, nullptr, 0, nullptr, 0, scal + o, q, ks);
    lincomb_cols(ctx, dk, MR, dirs2 + po, dv, nullptr, 0, nullptr, 0, Hin, dk, kh);
    pk->hess_prod_slow(Hq + o, q, Hin, dk, MR);                 // centadj: H (scaled dir); predadj: H dir
    pk->dder3_cols(scal + o, q, MR, D3 + o, q);
    done_k[k] = 1;
  }
  for (size_t k = 0; k < nc && !wr; ++k) {
    Cone* ck = cones[k];
    if (!ck->use_dder3() || done_k[k]) continue;
    const int o = offs[k], dk = ck->dim;
    const int po = (ck->use_dual_barrier ? oz : os) + o;
    for (int r = 0; r < MR; ++r) {
      const double* prim = dirs2 + (long)r * dv + po;
      double* sc = scal + (long)r * q + o;
      double* hh = Hq + (long)r * q + o;
      dev_scale_copy(ctx, dk, irtrtmu, prim, sc);
      ck->hess_prod_slow(hh, q, r == 0 ? sc : prim, q, 1);          // centadj: H (scaled dir); predadj: H dir
      const double* d3 = ck->dder3(sc);
      ctx.d2d(D3 + (long)r * q + o, d3, (size_t)dk * d);
    }
  }
  for (size_t k = 0; k < nc && !wr; ++k) {   // (after all cone calls: they use ctx.dscal themselves)
    Cone* ck = cones[k];
    if (!ck->use_dder3()) continue;
    const int o = offs[k], dk = ck->dim;
    if (d3_paired) {   // (one launch; every sum is the one dot_kernel forms)
      DotSpecs sp;
      for (int r = 0; r < MR; ++r) {
        sp.add(dk, D3 + (long)r * q + o, ck->point.d(), dots + 4 * k + 2 * r);
        sp.add(dk, scal + (long)r * q + o, Hq + (long)r * q + o, dots + 4 * k + 2 * r + 1);
      }
      dev_dots(ctx, sp);
      continue;
    }
    for (int r = 0; r < MR; ++r) {
      dev_dot(ctx, dk, D3 + (long)r * q + o, ck->point.d(), dots + 4 * k + 2 * r);
      dev_dot(ctx, dk, scal + (long)r * q + o, Hq + (long)r * q + o, dots + 4 * k + 2 * r + 1);
    }
  }
  const double tc = dir_tau2[0] / tau, tp = dir_tau2[1] / tau;
  rs[0] = Scal{0.0, tc * mu / tau * tc};
  rs[1] = Scal{0.0, tp * mu / tau * (1.0 + tp)};
  if (resident) {   // the tests on the device: no host round trip between the two pairs of solves
    if (wr) {       // (equal cones back to back: one launch)
      const int dk = cones[0]->dim;
      hipLaunchKernelGGL(dder3_gate_kernel, dim3((dk + 255) / 256, (unsigned)nc), dim3(256), 0, ctx.stream, dk, dots, irtrtmu, rteps, D3, Hq + q,
                         D3 + q, c0 + os, c1 + os);
    } else {
      for (size_t k = 0; k < nc; ++k) {
        Cone* ck = cones[k];
        if (!ck->use_dder3()) continue;
        const int o = offs[k], dk = ck->dim;
        hipLaunchKernelGGL(dder3_gate_kernel, dim3((dk + 255) / 256, 1), dim3(256), 0, ctx.stream, dk, dots + 4 * k, irtrtmu, rteps, D3 + o,
                           Hq + q + o, D3 + q + o, c0 + os + o, c1 + os + o);
      }
    }
    HYP_CHECK(hipGetLastError());
    return;
  }
  std::vector<double> hd(4 * std::max<size_t>(nc, 1));
  ctx.d2h(hd.data(), dots, 4 * nc * d);
  ctx.sync();
  for (size_t k = 0; k < nc; ++k) {
    Cone* ck = cones[k];
    if (!ck->use_dder3()) continue;
    const int o = offs[k], dk = ck->dim;
    {   // centadj (:96-113): rhs.s_k = dder3
      const double dot1 = hd[4 * k], dot2 = hd[4 * k + 1];
      if (std::fabs(dot1 - dot2) / (rteps + std::fabs(dot2)) < 1e-4) ctx.d2d(c0 + os + o, D3 + o, (size_t)dk * d);
    }
    {   // predadj (:37-55): rhs.s_k = H dir + dder3
      const double dot1 = hd[4 * k + 2], dot2 = irtrtmu * hd[4 * k + 3];
      if (std::fabs(dot1 - dot2) / (rteps + std::fabs(dot2)) < 1e-4) {
        ctx.d2d(c1 + os + o, Hq + q + o, (size_t)dk * d);
        dev_axpby(ctx, dk, 1.0, D3 + q + o, 1.0, c1 + os + o);
      }
    }
  }
}

void SysSolver::step_directions(const double* h_point, const double* h_res, double tau_residual, double mu, int max_ref_steps,
                                double res_norm_cutoff, double min_impr_tol, double* h_dirs, double* res_norms, int* n_solves,
                                int* use_sqrt_out, int* info, int* used_fallback, double* h_sol_const) {
  HYP_REQUIRE(model_loaded, "sys: load_model first");
  HYP_REQUIRE(p == 0, "step_directions: p = 0 only");
  const size_t d = sizeof(double);
  const int dv = dimv(), it = n + p + q, ik = dv - 1;
  *n_solves = 0;
  *info = 0;
  *used_fallback = 0;
  s_resident = false;
  // The caller's vectors are pageable: a hipMemcpyAsync on them stops the host until the copy is done, which left the device
  // idle for 80-170 us at each of the two direction downloads in the middle of this call (profiles/r02_iteration_timeline.txt).
  // Everything travels through the library's pinned staging instead ([point | residuals | sol_const | four directions]); the
  // uploads are queued in front of the Schur assembly, the downloads are copied out after the one synchronisation at the end.
  double* hs_point = ctx.stage_host((size_t)dv + 2 * (size_t)it + 2 * (size_t)MR * dv);
  double* hs_res = hs_point + dv;
  double* hs_const = hs_res + it;
  double* hs_dirs = hs_const + it;
  s_point.ensure((size_t)dv * d);
  s_resid.ensure((size_t)it * d);
  // (round 6) The point and the residuals are needed by the right-hand sides, not by the Schur assembly: in the resident flow they are
  // staged and uploaded AFTER the assembly and the factorization have been queued -- on the helper stream, behind an event that orders
  // them after whatever still reads the previous iterate on the main stream -- so that neither the host's copy into pinned memory
  // (0.4 ms at config 4's 3.3 MB point) nor the transfer sits between two iterations with the device idle
  auto upload = [&](hipStream_t st) {
    std::memcpy(hs_point, h_point, (size_t)dv * d);
    std::memcpy(hs_res, h_res, (size_t)it * d);
    HYP_CHECK(hipMemcpyAsync(s_point.p, hs_point, (size_t)dv * d, hipMemcpyHostToDevice, st));
    HYP_CHECK(hipMemcpyAsync(s_resid.p, hs_res, (size_t)it * d, hipMemcpyHostToDevice, st));
  };
  const bool resident_flow = dirs_resident() && nmp > 0 && !getenv_on("HYP_FORCE_BK") && !getenv_on("HYP_FORCE_FACT_FAIL");
  static const bool late_upload = [] { const char* e = getenv("HYP_LATE_UPLOAD"); return !(e && e[0] == '0'); }();
  // (measured, same box, alternating: config 4 -- a 3.3 MB point -- 108.0 / 108.1 ms late against 109.0 / 108.6 early; config 2 -- 0.36 MB --
  //  17.67 / 17.54 late against 17.48 / 17.48: the cross-stream hand-over costs a small model more than its copies do: from 1 MB on)
  const bool upload_late = resident_flow && late_upload && ctx.stream == ctx.stream_primary && (size_t)dv * d >= (1u << 20);
  if (upload_late) {
    if (!up_ev0) {
      HYP_CHECK(hipEventCreateWithFlags(&up_ev0, hipEventDisableTiming));
      HYP_CHECK(hipEventCreateWithFlags(&up_ev1, hipEventDisableTiming));
    }
    HYP_CHECK(hipEventRecord(up_ev0, ctx.stream));
  }   // (everything queued so far: the last readers of the old point)
  // Smaller points: the same stream, but still BEHIND the assembly and the factorization in its queue (their kernels do not read the
  // point) -- the host's copy into pinned memory (0.7 MB at config 2: ~80 us) then runs while the device already multiplies instead of
  // in the gap between two iterations; no cross-stream hand-over.  HYP_UPLOAD_AFTER=0: in front, as before.
  static const bool after_env = [] { const char* e = getenv("HYP_UPLOAD_AFTER"); return !(e && e[0] == '0'); }();
  const bool upload_after = resident_flow && !upload_late && after_env;
  if (!upload_late && !upload_after) upload(ctx.stream);
  const auto t0 = std::chrono::steady_clock::now();
  const double tau = h_point[it], kap = h_point[ik];
  m_rhs.ensure((size_t)(MR + 1) * dv * d);   // (+ the constant column of the first pair)
  v_tmp.ensure((size_t)MR * dv * d);
  // Round 6 (HYP_DIR_RESIDENT, default on): the factorization's info word is NOT waited for.  The first pair of solves -- right-hand
  // sides, constant column, both passes, residuals -- is queued behind the Cholesky as if it had succeeded (it almost always has);
  // the host reads info together with that pair's scalars.  Behind a failed Cholesky the pair's numbers are discarded, the fall-back
  // chain runs and the call continues the way it always did.
  const bool resident = resident_flow;
  if (!resident) {
    if (nmp > 0) update_lhs_fact(info, used_fallback);                 // combined.jl:64
    if (use_sqrt_out)
      for (size_t k = 0; k < cones.size(); ++k) use_sqrt_out[k] = use_sqrt[k];
    if (*info != 0) { ctx.sync(); return; }
    step_directions_rest(false, tau, kap, tau_residual, mu, max_ref_steps, res_norm_cutoff, min_impr_tol, h_dirs, res_norms, n_solves,
                         h_sol_const, hs_const, hs_dirs, false, nullptr, nullptr);
    last_update_lhs_s = last_rest_update_lhs_s + std::chrono::duration<double>(t_rest0 - t0).count();
    return;
  }
  assemble_lhs();
  factor_lhs_begin();
  if (upload_after) {
    hs_point = ctx.stage_host((size_t)dv + 2 * (size_t)it + 2 * (size_t)MR * dv);   // (taken again: see below)
    hs_res = hs_point + dv;
    hs_const = hs_res + it;
    hs_dirs = hs_const + it;
    upload(ctx.stream);
  }
  if (upload_late) {
    // (the staging block is taken again: a cone oracle of the assembly may have asked for a larger one meanwhile)
    hs_point = ctx.stage_host((size_t)dv + 2 * (size_t)it + 2 * (size_t)MR * dv);
    hs_res = hs_point + dv;
    hs_const = hs_res + it;
    hs_dirs = hs_const + it;
    HYP_CHECK(hipStreamWaitEvent(ctx.stream2, up_ev0, 0));
    upload(ctx.stream2);
    HYP_CHECK(hipEventRecord(up_ev1, ctx.stream2));
    HYP_CHECK(hipStreamWaitEvent(ctx.stream, up_ev1, 0));
  }
  if (use_sqrt_out)
    for (size_t k = 0; k < cones.size(); ++k) use_sqrt_out[k] = use_sqrt[k];
  bool cones_ok = true;
  for (const Cone* ck : cones) cones_ok = cones_ok && ck->products_columnwise();
  const bool const3 = const3_on() && cones_ok;
  const bool joint = !const3 && tri3_on() && !dist() && tri.ready(nmp) && tri.sb > 0 && tri.sb <= 1024;
  if (!const3 && !joint) {
    update_const();   // (synchronises: models without a solve plan -- small ones)
    if (h_sol_const) ctx.d2h(hs_const, sol_const.p, (size_t)it * d);
  }
  if (joint) update_const_pre();
  HYP_CHECK(hipEventRecord(ctx.ev[5], ctx.stream));   // (end of the update_lhs part: timed on the device, the host does not wait here)
  Scal rs[MR], dsc[MR];
  double rn[MR];
  int ns = 0;
  build_rhs_pair(0, m_rhs.d(), s_point.d(), mu, tau, kap, tau_residual, nullptr, nullptr, reinterpret_cast<double*>(rs), true);
  pair_enqueue(m_rhs.d(), rs, mu, tau, max_ref_steps, const3, joint, true, dsc);
  wait_scalars();
  if (dir_poll_on() && ctx.h_sc_dev != nullptr) ctx.h_info[Ctx::H_INFO_FACT] = (int)ctx.h_sc()[SC_INFO];   // (the info word came with the block)
  // (the device is idle from here until the second pair's first launches arrive: nothing that can wait is done before them --
  //  the phases' event times are read at the end of the call)
  const auto tf0 = std::chrono::steady_clock::now();
  factor_lhs_end(info, used_fallback, true);
  const double fallback_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
  auto read_times = [&] {
    factor_lhs_times();
    float ms = 0;
    HYP_CHECK(hipEventElapsedTime(&ms, ctx.ev[0], ctx.ev[5]));
    last_update_lhs_s = 1e-3 * ms;
  };
  if (*info != 0 || use_bk) read_times();
  if (*info != 0) return;
  if (use_bk) {   // the Cholesky failed, a fall-back factorization stands: everything queued behind the attempt is void
    const double dev_s = last_update_lhs_s;
    step_directions_rest(false, tau, kap, tau_residual, mu, max_ref_steps, res_norm_cutoff, min_impr_tol, h_dirs, res_norms, n_solves,
                         h_sol_const, hs_const, hs_dirs, false, nullptr, nullptr);
    last_update_lhs_s = dev_s + fallback_s + last_rest_update_lhs_s;
    return;
  }
  pair_finish(m_rhs.d(), rs, mu, tau, max_ref_steps, res_norm_cutoff, min_impr_tol, const3, joint, true, dsc, rn, &ns);
  if ((const3 || joint) && h_sol_const) ctx.d2h(hs_const, sol_const.p, (size_t)it * d);   // (host mirror of sys.sol_const)
  *n_solves += ns;
  step_directions_rest(true, tau, kap, tau_residual, mu, max_ref_steps, res_norm_cutoff, min_impr_tol, h_dirs, res_norms, n_solves, h_sol_const,
                       hs_const, hs_dirs, true, dsc, rn);
  read_times();
}

static bool env_default_on(const char* name) {
  const char* e = getenv(name);
  return !(e && e[0] == '0');
}
bool SysSolver::const3_on() { static const bool on = env_default_on("HYP_CONST_COL3"); return on; }
bool SysSolver::tri3_on() { static const bool on = env_default_on("HYP_CONST_TRI3"); return on; }
bool SysSolver::getenv_on(const char* name) {
  const char* e = getenv(name);
  return e && e[0] && e[0] != '0';
}

// the part of step_directions behind a standing factorization (first_pair_done: the resident form has solved the first pair already)
void SysSolver::step_directions_rest(bool resident, double tau, double kap, double tau_residual, double mu, int max_ref_steps,
                                     double res_norm_cutoff, double min_impr_tol, double* h_dirs, double* res_norms, int* n_solves,
                                     double* h_sol_const, double* hs_const, double* hs_dirs, bool first_pair_done, Scal* d01_in, double* rn01) {
  const size_t d = sizeof(double);
  const int dv = dimv(), it = n + p + q, ik = dv - 1;
  t_rest0 = std::chrono::steady_clock::now();
  last_rest_update_lhs_s = 0.0;
  Scal rs[MR], dsc[MR];
  double rn[MR];
  int ns = 0;
  if (!first_pair_done) {
    // The constant column of update_lhs (qrchol.jl:191-197: rhs_const = (-c, H h), one solve_subsystem3 per iteration) rides along
    // as a THIRD column of the first paired solve: its right-hand side, both passes over G, the cone products and the triangular
    // solves come out of the pair's launches (two passes over G, ~70 launches and a host synchronisation fewer per iteration).
    // Every column of every kernel on that path is computed with exactly the sums it gets when computed alone (the multi-column
    // G x kernel has the one-column kernel's four partial sums since round 3 -- that was the one kernel that differed, and the
    // reason this was off until then: tools/diag_const3.py), so the iterates are bitwise those of the separate solve
    // (tests/test_hip_switches.py).  Not with a Bunch-Kaufman factor (gather / scatter of two columns) and only for cones whose
    // multi-column products are column-wise the one-column ones.  HYP_CONST_COL3=0: off.
    bool cones_ok = true;   // (see Cone::products_columnwise; config 5's WSOS cone: with the switch forced on, 0.48 instead of 0.13
                            //  Bunch-Kaufman fall-backs per iteration and 10.3 instead of 6.6 ms in the directions)
    for (const Cone* ck : cones) cones_ok = cones_ok && ck->products_columnwise();
    const bool const3 = const3_on() && !use_bk && cones_ok;
    // With HYP_CONST_COL3=0: the constant column keeps its own right-hand side, its own passes over G and cone products (the numbers of
    // update_const()), but its two triangular solves -- 60 launches of ~5 us -- ride along with the first pair's as a third column
    // of the same launches (coldot3: per column the very sums of the separate kernels).  HYP_CONST_TRI3=0: solved on its own first.
    const bool joint = !const3 && tri3_on() && !dist() && !use_bk && nmp > 0 && tri.ready(nmp) && tri.sb > 0 && tri.sb <= 1024;
    if (!const3 && !joint) {
      update_const();
      if (h_sol_const) ctx.d2h(hs_const, sol_const.p, (size_t)it * d);   // (host mirror of sys.sol_const)
    }
    if (joint) update_const_pre();
    last_rest_update_lhs_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_rest0).count();
    // (cent, pred)
    build_rhs_pair(0, m_rhs.d(), s_point.d(), mu, tau, kap, tau_residual, nullptr, nullptr, reinterpret_cast<double*>(rs), resident);
    pair_enqueue(m_rhs.d(), rs, mu, tau, max_ref_steps, const3, joint, resident, dsc);
    ctx.sync();
    pair_finish(m_rhs.d(), rs, mu, tau, max_ref_steps, res_norm_cutoff, min_impr_tol, const3, joint, resident, dsc, rn, &ns);
    if ((const3 || joint) && h_sol_const) ctx.d2h(hs_const, sol_const.p, (size_t)it * d);   // (host mirror of sys.sol_const)
    *n_solves += ns;
  } else {
    dsc[0] = d01_in[0]; dsc[1] = d01_in[1];
    rn[0] = rn01[0]; rn[1] = rn01[1];
  }
  res_norms[0] = rn[0];
  res_norms[1] = rn[1];
  s_dirs.ensure((size_t)2 * MR * dv * d);   // (all four directions stay here for search_alpha(..., resident))
  const double dtau[MR] = {dsc[0].tau, dsc[1].tau};
  const Scal d01[MR] = {dsc[0], dsc[1]};
  // (centadj, predadj) -- right-hand sides from the first pair's directions where they lie; their copies (to the host's staging and
  // to the resident block of the line search) are queued behind the right-hand sides' launches: the device is waiting for work here
  build_rhs_pair(1, m_rhs.d(), s_point.d(), mu, tau, kap, tau_residual, m_dir.d(), dtau, reinterpret_cast<double*>(rs), resident);
  // (dirs_x_only: only the x rows travel; columns stay dv apart in the staging block, the z / s rows of the caller's block are not touched)
  const size_t rows_down = dirs_x_only ? (size_t)n : (size_t)dv;
  auto dirs_d2h = [&](int half) {
    double* dst = hs_dirs + (long)half * MR * dv;
    if (!dirs_x_only) { ctx.d2h(dst, m_dir.d(), (size_t)MR * dv * d); return; }
    HYP_CHECK(hipMemcpy2DAsync(dst, (size_t)dv * d, m_dir.d(), (size_t)dv * d, rows_down * d, MR, hipMemcpyDeviceToHost, ctx.stream));
  };
  auto dirs_out = [&](int half) {
    for (int r = 0; r < MR; ++r)
      std::memcpy(h_dirs + ((long)half * MR + r) * dv, hs_dirs + ((long)half * MR + r) * dv, rows_down * d);
  };
  dirs_d2h(0);
  ctx.d2d(s_dirs.p, m_dir.p, (size_t)MR * dv * d);
  if (!dirs_copied_ev) HYP_CHECK(hipEventCreateWithFlags(&dirs_copied_ev, hipEventDisableTiming));
  HYP_CHECK(hipEventRecord(dirs_copied_ev, ctx.stream));
  ns = 0;
  pair_enqueue(m_rhs.d(), rs, mu, tau, max_ref_steps, false, false, resident, dsc, false);
  // (the raw directions travel to the host and to the resident block IN FRONT of the scalars: when those have landed, so have they;
  //  a refined pair is copied again below)
  dirs_d2h(1);
  ctx.d2d(s_dirs.d() + (long)MR * dv, m_dir.p, (size_t)MR * dv * d);
  cols_read_scalars(resident);
  // the first pair's directions reach the caller's (pageable) block while the device works on the second pair
  HYP_CHECK(hipEventSynchronize(dirs_copied_ev));
  dirs_out(0);
  if (resident) wait_scalars();
  else ctx.sync();
  const int ns_before = ns;
  pair_finish(m_rhs.d(), rs, mu, tau, max_ref_steps, res_norm_cutoff, min_impr_tol, false, false, resident, dsc, rn, &ns);
  *n_solves += ns;
  res_norms[2] = rn[0];
  res_norms[3] = rn[1];
  if (ns > ns_before + MR) {   // refinement moved a direction of the second pair
    dirs_d2h(1);
    ctx.d2d(s_dirs.d() + (long)MR * dv, m_dir.p, (size_t)MR * dv * d);
    ctx.sync();
  }
  dirs_out(1);
  if (h_sol_const) std::memcpy(h_sol_const, hs_const, (size_t)it * d);
  for (int r = 0; r < MR; ++r) {
    h_dirs[(long)r * dv + it] = d01[r].tau;
    h_dirs[(long)r * dv + ik] = d01[r].kap;
    h_dirs[(long)(MR + r) * dv + it] = dsc[r].tau;
    h_dirs[(long)(MR + r) * dv + ik] = dsc[r].kap;
    s_tk[1 + r][0] = d01[r].tau;
    s_tk[1 + r][1] = d01[r].kap;
    s_tk[1 + MR + r][0] = dsc[r].tau;
    s_tk[1 + MR + r][1] = dsc[r].kap;
  }
  s_tk[0][0] = tau;
  s_tk[0][1] = kap;
  s_resident_q = q;
  s_resident = (MR == 2);   // the point (s_point) and the four directions (s_dirs) are on the device until the next call
}

}  // namespace hyp
