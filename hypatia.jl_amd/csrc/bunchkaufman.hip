// Symmetric indefinite factorization with rook pivoting, on the device.
//
// The reference falls back to `bunchkaufman!(Symmetric(A, :U), true, check = false)` whenever a Cholesky
// factorization fails (symm_fact!, dense.jl:164-165, reached from posdef_fact_copy!, dense.jl:194-215): the
// Schur matrix of the QRChol solver (qrchol.jl:249-250) and the explicit Hessian of the generic cones
// (Cones.jl:239-251).  In Julia that is LAPACK dsytrf_rook / dsytrs_rook.  This file restates the same
// pivoting rule (Ashcraft-Grimes-Lewis "rook" search, alpha = (1 + sqrt(17)) / 8, 1x1 and 2x2 pivot blocks,
// symmetric interchanges applied to the whole factor) in the storage the rest of the library solves with:
//
//     P A P' = U' D U,   U unit upper triangular,  D block diagonal,
//
// eliminating forwards (k = 0, 1, ...) on the upper triangle in place, so that the existing blocked
// triangular solves (trsv_upper / trsm_upper_left / TriSolvePlan: U'^-1 then U^-1) are reused unchanged,
// with a gather by P before, a block-diagonal solve between and a scatter after.
//
// Shape of the computation.  A pivot step is inherently sequential (search -> interchange -> eliminate), so
// each step is two launches with no host involvement: a one-workgroup kernel that does the whole rook
// search, the interchanges and the scaling of the pivot row(s) (all decisions on the device; the step's
// column index lives in device memory because a 2x2 pivot advances it by two), and a many-workgroup
// rank-1 / rank-2 update of the trailing upper triangle, which is the HBM-bound part:
// sum_k (n-k)^2/2 * 16 B = 8 n^3 / 3 B (3.3e11 B at n = 5000).  This is the fallback of a failed Cholesky,
// reached on a few late iterations of badly conditioned instances, not the steady-state factorization.
#include "hyp_internal.hpp"
#include <limits.h>

namespace hyp {
namespace {

constexpr int BK_T = 1024;   // threads of the pivot kernel
constexpr int BK_TILE = 64;  // trailing update tile

struct BkState {
  int knext;   // first column not yet eliminated
  int k;       // column of the step just prepared
  int kstep;   // 1 or 2
  int skip;    // 1: exactly singular column, nothing to eliminate (LAPACK sets info and moves on)
  int info;    // 0 or 1-based index of the first exactly singular pivot
  int n2x2;    // number of 2x2 pivots (diagnostics)
};

// max |.| with the SMALLEST index among equal maxima (idamax returns the first one); every thread returns the result
__device__ __forceinline__ void bk_argmax(double& v, int& ix, double* s_v, int* s_i) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_down(v, off);
    const int oi = __shfl_down(ix, off);
    if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { s_v[w] = v; s_i[w] = ix; }
  __syncthreads();
  if (threadIdx.x < 64) {
    v = (lane < BK_T / 64) ? s_v[lane] : -1.0;
    ix = (lane < BK_T / 64) ? s_i[lane] : INT_MAX;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      const double ov = __shfl_down(v, off);
      const int oi = __shfl_down(ix, off);
      if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
    }
    if (lane == 0) { s_v[16] = v; s_i[16] = ix; }
  }
  __syncthreads();
  v = s_v[16];
  ix = s_i[16];
}

// symmetric interchange of indices a < b on the upper-stored matrix, all rows (the rows above the active
// block hold U, which the rook variant permutes too, so that ONE permutation describes the factorization)
__device__ __forceinline__ void bk_swap(int n, double* __restrict__ A, long lda, int a, int b, int* __restrict__ perm) {
  const int t = threadIdx.x;
  double* ca = A + (long)a * lda;
  double* cb = A + (long)b * lda;
  for (int i = t; i < a; i += BK_T) { const double x = ca[i]; ca[i] = cb[i]; cb[i] = x; }
  for (int i = a + 1 + t; i < b; i += BK_T) {   // row a right of the diagonal <-> column b above it
    double* pr = A + (long)i * lda + a;
    const double x = *pr; *pr = cb[i]; cb[i] = x;
  }
  for (int i = b + 1 + t; i < n; i += BK_T) {
    double* pa = A + (long)i * lda + a;
    double* pb = A + (long)i * lda + b;
    const double x = *pa; *pa = *pb; *pb = x;
  }
  if (t == 0) {
    const double x = ca[a]; ca[a] = cb[b]; cb[b] = x;
    const int pi = perm[a]; perm[a] = perm[b]; perm[b] = pi;
  }
  __syncthreads();
}

// One pivot step: rook search (dsytf2_rook's rule), interchanges, pivot block into (dd, de, blk), pivot
// row(s) scaled in place to rows of U, the unscaled / scaled rows saved contiguously for the update kernel.
__global__ __launch_bounds__(BK_T) void bk_pivot_kernel(int n, double* __restrict__ A, long lda, BkState* __restrict__ st,
                                                        double* __restrict__ dd, double* __restrict__ de, int* __restrict__ blk,
                                                        int* __restrict__ perm, double* __restrict__ wl) {
  __shared__ double s_v[17];
  __shared__ int s_i[17];
  const int t = threadIdx.x;
  const int k = st->knext;
  if (k >= n) {
    if (t == 0) st->k = n;
    return;
  }
  const double alpha = 0.6403882032022076;   // (1 + sqrt(17)) / 8
  double* w1 = wl;
  double* l1 = wl + n;
  double* w2 = wl + 2L * n;
  double* l2 = wl + 3L * n;

  const double absakk = fabs(A[(long)k * lda + k]);
  double colmax = -1.0;
  int imax = INT_MAX;
  for (int j = k + 1 + t; j < n; j += BK_T) {
    const double a = fabs(A[(long)j * lda + k]);
    if (a > colmax) { colmax = a; imax = j; }
  }
  bk_argmax(colmax, imax, s_v, s_i);
  if (colmax < 0.0) colmax = 0.0;

  int kstep = 1, kp = k, p = k;
  bool skip = false;
  if (fmax(absakk, colmax) == 0.0 || absakk != absakk) {
    skip = true;
  } else if (!(absakk < alpha * colmax)) {
    kp = k;
  } else {
    for (;;) {
      // largest off-diagonal of row/column imax inside the active block: (j, imax) for k <= j < imax down
      // the stored column, (imax, j) for j > imax along the stored row
      double rowmax = -1.0;
      int jmax = INT_MAX;
      const double* ci = A + (long)imax * lda;
      for (int j = k + t; j < imax; j += BK_T) {
        const double a = fabs(ci[j]);
        if (a > rowmax) { rowmax = a; jmax = j; }
      }
      for (int j = imax + 1 + t; j < n; j += BK_T) {
        const double a = fabs(A[(long)j * lda + imax]);
        if (a > rowmax) { rowmax = a; jmax = j; }
      }
      bk_argmax(rowmax, jmax, s_v, s_i);
      if (rowmax < 0.0) rowmax = 0.0;
      if (!(fabs(ci[imax]) < alpha * rowmax)) {
        kp = imax; kstep = 1;
        break;
      } else if (p == jmax || rowmax <= colmax) {
        kp = imax; kstep = 2;
        break;
      } else {
        p = imax; colmax = rowmax; imax = jmax;
      }
    }
  }
  __syncthreads();
  if (kstep == 2 && p != k) bk_swap(n, A, lda, k, p, perm);
  const int kk = k + kstep - 1;
  if (kp != kk) bk_swap(n, A, lda, kk, kp, perm);

  if (skip) {
    if (t == 0) {
      dd[k] = A[(long)k * lda + k];
      de[k] = 0.0;
      blk[k] = 0;
      A[(long)k * lda + k] = 1.0;
      if (st->info == 0) st->info = k + 1;
    }
  } else if (kstep == 1) {
    const double d = A[(long)k * lda + k];
    __syncthreads();
    for (int j = k + 1 + t; j < n; j += BK_T) {
      double* pe = A + (long)j * lda + k;
      const double w = *pe;
      const double l = w / d;
      w1[j] = w; l1[j] = l;
      *pe = l;
    }
    if (t == 0) {
      dd[k] = d; de[k] = 0.0; blk[k] = 0;
      A[(long)k * lda + k] = 1.0;
    }
  } else {
    const double d11 = A[(long)k * lda + k];
    const double d12 = A[(long)(k + 1) * lda + k];
    const double d22 = A[(long)(k + 1) * lda + k + 1];
    __syncthreads();
    // [l1 l2] = [w1 w2] D^-1 with the scaling of dsytf2_rook (everything divided by the large off-diagonal)
    const double D11 = d22 / d12, D22 = d11 / d12;
    const double T = 1.0 / (D11 * D22 - 1.0);
    for (int j = k + 2 + t; j < n; j += BK_T) {
      double* pe = A + (long)j * lda + k;
      const double a = pe[0], b = pe[1];
      const double la = T * (D11 * a - b) / d12;
      const double lb = T * (D22 * b - a) / d12;
      w1[j] = a; w2[j] = b; l1[j] = la; l2[j] = lb;
      pe[0] = la; pe[1] = lb;
    }
    if (t == 0) {
      dd[k] = d11; dd[k + 1] = d22; de[k] = d12; de[k + 1] = 0.0;
      blk[k] = 1; blk[k + 1] = 2;
      A[(long)k * lda + k] = 1.0;
      A[(long)(k + 1) * lda + k] = 0.0;
      A[(long)(k + 1) * lda + k + 1] = 1.0;
      st->n2x2 += 1;
    }
  }
  if (t == 0) {
    st->k = k;
    st->kstep = kstep;
    st->skip = skip ? 1 : 0;
    st->knext = k + kstep;
  }
}

// trailing update A[i, j] -= l1[i] w1[j] (+ l2[i] w2[j]) on the upper triangle i <= j of the block that starts
// after the pivot; tiles are numbered along the upper triangle of the tile grid (host launches for the
// largest block the step can have, surplus workgroups leave)
__global__ __launch_bounds__(256) void bk_update_kernel(int n, double* __restrict__ A, long lda, const BkState* __restrict__ st,
                                                        const double* __restrict__ wl) {
  const int k = st->k;
  if (k >= n || st->skip) return;
  const int ks = st->kstep;
  const int base = k + ks;
  const int rem = n - base;
  if (rem <= 0) return;
  const int nt = (rem + BK_TILE - 1) / BK_TILE;
  const long idx = blockIdx.x;
  if (idx >= (long)nt * (nt + 1) / 2) return;
  int bj = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
  while ((long)bj * (bj + 1) / 2 > idx) --bj;
  while ((long)(bj + 1) * (bj + 2) / 2 <= idx) ++bj;
  const int bi = (int)(idx - (long)bj * (bj + 1) / 2);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = base + bi * BK_TILE + tx;
  const int j0 = base + bj * BK_TILE;
  if (i >= n) return;
  const double* w1 = wl;
  const double* l1 = wl + n;
  const double* w2 = wl + 2L * n;
  const double* l2 = wl + 3L * n;
  const double a1 = l1[i];
  const double a2 = (ks == 2) ? l2[i] : 0.0;
#pragma unroll 4
  for (int c = ty; c < BK_TILE; c += 4) {
    const int j = j0 + c;
    if (j < n && i <= j) {
      double* pe = A + (long)j * lda + i;
      double v = *pe - a1 * w1[j];
      if (ks == 2) v -= a2 * w2[j];
      *pe = v;
    }
  }
}

__global__ void bk_init_kernel(int n, BkState* st, int* perm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) perm[i] = i;
  if (i == 0) { st->knext = 0; st->k = 0; st->kstep = 1; st->skip = 0; st->info = 0; st->n2x2 = 0; }
}

__global__ void bk_gather_kernel(int n, int nr, const int* __restrict__ perm, const double* __restrict__ x, long ldx,
                                 double* __restrict__ y, long ldy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int pi = perm[i];
  for (int r = blockIdx.y; r < nr; r += gridDim.y) y[(long)r * ldy + i] = x[(long)r * ldx + pi];
}
__global__ void bk_scatter_kernel(int n, int nr, const int* __restrict__ perm, const double* __restrict__ y, long ldy,
                                  double* __restrict__ x, long ldx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int pi = perm[i];
  for (int r = blockIdx.y; r < nr; r += gridDim.y) x[(long)r * ldx + pi] = y[(long)r * ldy + i];
}
// y <- D^-1 y, block by block (the 2x2 solve scaled like dsytrs_rook: by the off-diagonal)
__global__ void bk_dsolve_kernel(int n, int nr, const double* __restrict__ dd, const double* __restrict__ de, const int* __restrict__ blk,
                                 double* __restrict__ y, long ldy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = blk[i];
  if (b == 2) return;
  for (int r = blockIdx.y; r < nr; r += gridDim.y) {
    double* yr = y + (long)r * ldy;
    if (b == 0) {
      yr[i] = yr[i] / dd[i];
    } else {
      const double e = de[i];
      const double akm1 = dd[i] / e, ak = dd[i + 1] / e;
      const double den = akm1 * ak - 1.0;
      const double bkm1 = yr[i] / e, bk = yr[i + 1] / e;
      yr[i] = (ak * bkm1 - bk) / den;
      yr[i + 1] = (akm1 * bk - bkm1) / den;
    }
  }
}

}  // namespace

int BKFact::factor(Ctx& c, int n_, double* A, long lda, double* dinv) {
  n = n_;
  if (n <= 0) return 0;
  const size_t d = sizeof(double);
  dd.ensure((size_t)n * d);
  de.ensure((size_t)n * d);
  blk.ensure((size_t)n * sizeof(int));
  perm.ensure((size_t)n * sizeof(int));
  wl.ensure((size_t)4 * n * d);
  state.ensure(64);
  BkState* st = (BkState*)state.p;
  hipLaunchKernelGGL(bk_init_kernel, dim3((n + 255) / 256), dim3(256), 0, c.stream, n, st, perm.i());
  for (int s = 0; s < n; ++s) {
    hipLaunchKernelGGL(bk_pivot_kernel, dim3(1), dim3(BK_T), 0, c.stream, n, A, lda, st, dd.d(), de.d(), blk.i(), perm.i(), wl.d());
    const int rem = n - s - 1;   // the step's column is >= s, so its trailing block has at most n - s - 1 rows
    if (rem > 0) {
      const long nt = (rem + BK_TILE - 1) / BK_TILE;
      hipLaunchKernelGGL(bk_update_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, c.stream, n, A, lda, st, wl.d());
    }
  }
  if (dinv) potrf_invert_diag_blocks(c, n, A, lda, 0, 1, dinv);
  c.d2h(c.h_info, st, sizeof(BkState));
  c.sync();
  const BkState* hs = (const BkState*)c.h_info;
  n_2x2 = hs->n2x2;
  return hs->info;
}

double* BKFact::gather(Ctx& c, const double* x, long ldx, int nr) {
  tmp.ensure((size_t)n * std::max(nr, 1) * sizeof(double));
  hipLaunchKernelGGL(bk_gather_kernel, dim3((n + 255) / 256, std::min(nr, 64)), dim3(256), 0, c.stream, n, nr, perm.i(), x, ldx, tmp.d(),
                     (long)n);
  return tmp.d();
}
void BKFact::dsolve(Ctx& c, double* y, long ldy, int nr) {
  hipLaunchKernelGGL(bk_dsolve_kernel, dim3((n + 255) / 256, std::min(nr, 64)), dim3(256), 0, c.stream, n, nr, dd.d(), de.d(), blk.i(), y,
                     ldy);
}
void BKFact::scatter(Ctx& c, const double* y, double* x, long ldx, int nr) {
  hipLaunchKernelGGL(bk_scatter_kernel, dim3((n + 255) / 256, std::min(nr, 64)), dim3(256), 0, c.stream, n, nr, perm.i(), y, (long)n, x,
                     ldx);
}

// x <- A^-1 x for nr right-hand sides through the factorization (dsytrs_rook's role)
void BKFact::solve(Ctx& c, const double* U, long ldu, const double* dinv, double* x, long ldx, int nr, DBuf& trsm_work) {
  if (n <= 0 || nr <= 0) return;
  double* y = gather(c, x, ldx, nr);
  if (nr == 1) {
    trsv_upper(c, n, U, ldu, dinv, true, y);
    dsolve(c, y, n, 1);
    trsv_upper(c, n, U, ldu, dinv, false, y);
  } else {
    trsm_work.ensure((size_t)NB * nr * sizeof(double));
    trsm_upper_left(c, n, nr, U, ldu, dinv, true, y, n, trsm_work.d());
    dsolve(c, y, n, nr);
    trsm_upper_left(c, n, nr, U, ldu, dinv, false, y, n, trsm_work.d());
  }
  scatter(c, y, x, ldx, nr);
}

}  // namespace hyp
