// Diagonal-block Cholesky + triangular inverse kernel (one workgroup per <=128 x 128 block).
// See dense.hip for the blocked algorithm that calls it (dpotrf 'U' of the reference:
// /root/reference/src/linearalgebra/dense.jl:189-200, src/Cones/possemideftri.jl:85,94).
//
// The block lives in registers: 256 threads form a 16 x 16 grid, thread (tr, tc) owns the elements
// (16 r + tr, 16 c + tc), r <= c (upper block-triangle only: 36 doubles for U, 36 for its inverse).
// The 2-D cyclic distribution keeps every thread busy as the trailing matrix shrinks.  Each column
// step publishes one pivot row (and, for the inverse, one column of U) through LDS and costs one
// barrier.
#include "hyp_internal.hpp"

namespace hyp {

// index of (r, c), r <= c, in the packed per-thread upper block array (row-major over r)
#define UIDX(r, c) ((r) * 8 - (r) * ((r) - 1) / 2 + ((c) - (r)))

// 1 / x on the critical path of every column step: v_rcp_f64 + two Newton steps (the IEEE division
// expansion is a ~30-instruction dependent chain); dpotf2 itself scales by the reciprocal of the pivot
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void potrf_diag_kernel(double* __restrict__ A, long lda, long strideA, int n, int k0, double* __restrict__ dinv, long strideD,
                       int* __restrict__ info) {
  __shared__ double rowbuf[2][NB];
  __shared__ double colbuf[2][NB];
  __shared__ double dsq[NB];
  __shared__ double rdsq[NB];
  const int tid = threadIdx.x;
  const int tr = tid & 15, tc = tid >> 4;
  const int nb = min(NB, n - k0);
  double* Ab = A + (long)blockIdx.x * strideA + (long)k0 * lda + k0;
  double* Db = dinv + (long)blockIdx.x * strideD + (long)(k0 / NB) * DINV_BLK;

  double a[36];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = r; c < 8; ++c) {
      const int i = 16 * r + tr, l = 16 * c + tc;
      double v = (i == l) ? 1.0 : 0.0;
      if (i < nb && l < nb && i <= l) v = Ab[(long)l * lda + i];
      a[UIDX(r, c)] = v;
    }

  const double dmask = (tc >= tr) ? 1.0 : 0.0;
  int fail = 0;
  if (tr == 0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) rowbuf[0][16 * c + tc] = a[UIDX(0, c)];
  }
  __syncthreads();

  // ---- factorization: right-looking, one barrier per column
#pragma unroll
  for (int r0 = 0; r0 < 8; ++r0) {
#pragma unroll 1
    for (int jj = 0; jj < 16; ++jj) {
      const int j = 16 * r0 + jj;
      const int cur = j & 1;
      double piv = rowbuf[cur][j];
      if (!(piv > 0.0)) {
        if (!fail) fail = j + 1;
        piv = 1.0;
      }
      const double rpiv = fast_rcp(piv);
      // branch-free rank-1 update: rows i <= j contribute a zero multiplier; within the diagonal
      // register block (c == r) only columns l >= i are touched (dmask), for c > r always l > i
      double rl[8];
#pragma unroll
      for (int c = r0; c < 8; ++c) rl[c] = rowbuf[cur][16 * c + tc];
#pragma unroll
      for (int r = r0; r < 8; ++r) {
        const int i = 16 * r + tr;
        double f = rowbuf[cur][i] * rpiv;
        f = (i > j) ? f : 0.0;
        a[UIDX(r, r)] -= (f * dmask) * rl[r];
#pragma unroll
        for (int c = r + 1; c < 8; ++c) a[UIDX(r, c)] -= f * rl[c];
      }
      if (tr == jj && tc == jj) dsq[j] = piv;   // pivot kept; row scaling U[j,:] = A[j,:]/sqrt(piv) is deferred
      // publish row j + 1 (current values, before its own scaling); columns < 16 r' of that row are
      // never read (they are left of the diagonal), so only c >= r' is stored
      if (jj < 15) {
        if (tr == jj + 1) {
#pragma unroll
          for (int c = r0; c < 8; ++c) rowbuf[cur ^ 1][16 * c + tc] = a[UIDX(r0, c)];
        }
      } else if (r0 < 7) {
        if (tr == 0) {
#pragma unroll
          for (int c = (r0 + 1) & 7; c < 8; ++c) rowbuf[cur ^ 1][16 * c + tc] = a[UIDX((r0 + 1) & 7, c)];
        }
      }
      __syncthreads();
    }
  }

  // deferred row scaling, all rows at once (keeps sqrt / division off the per-column critical path):
  // dsq <- sqrt(pivot), rdsq <- 1 / sqrt(pivot)
  if (tid < NB) {
    const double sq = sqrt(dsq[tid]);
    dsq[tid] = sq;
    rdsq[tid] = 1.0 / sq;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const double rs = rdsq[16 * r + tr];
#pragma unroll
    for (int c = r; c < 8; ++c) a[UIDX(r, c)] *= rs;
  }

  // write U back (upper triangle of the nb x nb block)
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = r; c < 8; ++c) {
      const int i = 16 * r + tr, l = 16 * c + tc;
      if (i < nb && l < nb && i <= l) Ab[(long)l * lda + i] = a[UIDX(r, c)];
    }
  if (tid == 0 && fail && fail <= nb) atomicCAS(&info[blockIdx.x], 0, k0 + fail);

  // ---- inverse of U: Gauss-Jordan on [U | I], columns in descending order (row j of the U part is
  //      already reduced to its diagonal when it is used), one barrier per column
  double x[36];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = r; c < 8; ++c) x[UIDX(r, c)] = (16 * r + tr == 16 * c + tc) ? 1.0 : 0.0;

  // publish row 127 of X (scaled) and column 127 of U
  if (tr == 15) {
    const double rd = rdsq[NB - 1];
    x[UIDX(7, 7)] *= rd;
    rowbuf[1][16 * 7 + tc] = x[UIDX(7, 7)];
  }
  if (tc == 15) {
#pragma unroll
    for (int r = 0; r < 8; ++r) colbuf[1][16 * r + tr] = a[UIDX(r, 7)];
  }
  __syncthreads();

#pragma unroll
  for (int r0 = 7; r0 >= 0; --r0) {
#pragma unroll 1
    for (int jj = 15; jj >= 0; --jj) {
      const int j = 16 * r0 + jj;
      const int cur = j & 1;
      // branch-free: rows i >= j get a zero multiplier; in register column block r0 only l >= j
      double rl[8];
#pragma unroll
      for (int c = r0; c < 8; ++c) rl[c] = rowbuf[cur][16 * c + tc];
      rl[r0] = (tc >= jj) ? rl[r0] : 0.0;
#pragma unroll
      for (int r = 0; r <= r0; ++r) {
        const int i = 16 * r + tr;
        double f = colbuf[cur][i];
        f = (i < j) ? f : 0.0;
#pragma unroll
        for (int c = r0; c < 8; ++c) x[UIDX(r, c)] -= f * rl[c];
      }
      // publish row j - 1 of X (scaled by 1 / U[j-1, j-1]) and column j - 1 of U
      if (jj > 0) {
        if (tr == jj - 1) {
          const double rd = rdsq[j - 1];
#pragma unroll
          for (int c = r0; c < 8; ++c) {
            x[UIDX(r0, c)] *= rd;
            rowbuf[cur ^ 1][16 * c + tc] = x[UIDX(r0, c)];
          }
        }
        if (tc == jj - 1) {
#pragma unroll
          for (int r = 0; r <= r0; ++r) colbuf[cur ^ 1][16 * r + tr] = a[UIDX(r, r0)];
        }
      } else if (r0 > 0) {
        if (tr == 15) {
          const double rd = rdsq[j - 1];
#pragma unroll
          for (int c = (r0 + 7) & 7; c < 8; ++c) {
            x[UIDX((r0 + 7) & 7, c)] *= rd;
            rowbuf[cur ^ 1][16 * c + tc] = x[UIDX((r0 + 7) & 7, c)];
          }
        }
        if (tc == 15) {
#pragma unroll
          for (int r = 0; r <= ((r0 + 7) & 7); ++r) colbuf[cur ^ 1][16 * r + tr] = a[UIDX(r, (r0 + 7) & 7)];
        }
      }
      __syncthreads();
    }
  }
  // write the inverse block (full NB x NB storage, zeros below the diagonal and in the padding)
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = 16 * r + tr, l = 16 * c + tc;
      double v = 0.0;
      if (c >= r) v = (i < nb && l < nb && i <= l) ? x[UIDX(r, c < r ? r : c)] : 0.0;
      Db[(long)l * NB + i] = v;
    }
  // and its transpose inv(U)' (lower triangular): every thread stores its own elements at the
  // mirrored position; the parallel diagonal solve of the forward substitution reads it row-major
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = 16 * r + tr, l = 16 * c + tc;
      double v = 0.0;
      if (c >= r) v = (i < nb && l < nb && i <= l) ? x[UIDX(r, c < r ? r : c)] : 0.0;
      Db[NB * NB + (long)i * NB + l] = v;
    }
}

}  // namespace hyp
