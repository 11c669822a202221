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
#include <utility>

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

// FACTOR: factor the block in place;  INVERT: emit inv(U_kk) and its transpose.  <true,false> is the
// critical-path kernel of the blocked Cholesky; <false,true> runs once at the end for ALL diagonal
// blocks at once (grid.y = block index), off the critical path.
template <bool FACTOR, bool INVERT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void potrf_diag_kernel_t(double* __restrict__ A, long lda, long strideA, int n, int k0_, double* __restrict__ dinv, long strideD,
                         int* __restrict__ info) {
  const int k0 = k0_ + blockIdx.y * NB;
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

  if (FACTOR) {
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
    if (16 * r0 >= nb) break;   // (a short last block: the identity padding needs no elimination steps)
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
    if (tid >= ((nb + 15) & ~15)) dsq[tid] = 1.0;   // rows whose elimination steps were skipped (identity padding)
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
  } else {   // inverse only: the factored block was just loaded; 1 / U_jj from its diagonal
    if (tr == tc) {
#pragma unroll
      for (int r = 0; r < 8; ++r) rdsq[16 * r + tr] = 1.0 / a[UIDX(r, r)];
    }
    __syncthreads();
  }
  if (!INVERT) return;

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

template __global__ void potrf_diag_kernel_t<true, true>(double*, long, long, int, int, double*, long, int*);
template __global__ void potrf_diag_kernel_t<true, false>(double*, long, long, int, int, double*, long, int*);
template __global__ void potrf_diag_kernel_t<false, true>(double*, long, long, int, int, double*, long, int*);

// own_cu_lds > 0 (factor-only launches on the critical path of the look-ahead Cholesky): the single workgroup asks for
// that much dynamic LDS on top of its own, so that no GEMM workgroup of the concurrent trailing update (36 / 72 KB of
// LDS each) fits beside it and the CU's four SIMDs are its own -- 47 us instead of 62-79 us per diagonal block when the
// helper stream's GEMM covers the chip (profiles/r01_cholesky_timeline.txt).  The other 255 CUs keep the GEMM.
void potrf_diag_launch(hipStream_t st, bool factor, bool invert, int batch, int nblocks, double* A, long lda, long strideA, int n, int k0,
                       double* dinv, long strideD, int* info, int own_cu_lds) {
  const dim3 grid(batch, nblocks), blk(256);
  if (factor && invert) hipLaunchKernelGGL((potrf_diag_kernel_t<true, true>), grid, blk, 0, st, A, lda, strideA, n, k0, dinv, strideD, info);
  else if (factor) hipLaunchKernelGGL((potrf_diag_kernel_t<true, false>), grid, blk, own_cu_lds > 0 ? own_cu_lds : 0, st, A, lda, strideA, n, k0, dinv, strideD, info);
  else hipLaunchKernelGGL((potrf_diag_kernel_t<false, true>), grid, blk, 0, st, A, lda, strideA, n, k0, dinv, strideD, info);
  HYP_CHECK(hipGetLastError());
}
// dynamic LDS (bytes) that leaves less than the smallest GEMM workgroup's 36 KB free on a 160 KB CU; 0 when the device
// refuses the opt-in
int potrf_diag_own_cu_lds() {
  const int want = 124 * 1024;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_diag_kernel_t<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return want;
}

// =============================================================================================
// Panel solve of the blocked Cholesky: X = U11^-T A12 in place (dtrsm 'L','U','T','N'), by forward
// substitution -- no inverse of U11 on the critical path, and dtrsm's own rounding behaviour.
// Every column of A12 is independent.  16 lanes (one DPP row) share a column: lane t owns rows
// i = 16 k + t (8 values per column, PS_NC columns per thread).  Step j: the owner of row j scales
// it by 1 / U_jj, the value is broadcast inside the DPP row (row_newbcast), and every lane applies
// a_i -= U[j, i] x_j to its rows i > j.  U11 sits in LDS packed by rows (66 KB), so a step reads 16
// consecutive doubles per k (conflict-free, broadcast across the four columns of a wavefront).  No
// barrier inside the 128 steps.
// =============================================================================================
constexpr int PS_NC = 2;                 // columns per thread
constexpr int PS_COLS = 16 * PS_NC;      // columns per workgroup (256 threads = 16 column groups x 16 lanes)
#define PS_OFF(j) ((j) * NB - (j) * ((j) - 1) / 2)

template <int L>
__device__ __forceinline__ double row_bcast(double v) {   // value of lane L of each 16-lane row
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + L, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + L, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// one substitution step for row J = 16 kj + TJ of the current 16-row block.  The register rows are
// ROTATED after every 16 steps so that the current block is always a[0] (static register indices with a
// rolled outer loop: the fully unrolled 128-step body was instruction-fetch bound); KMAX = how many of
// the 8 register rows can still be live (rows past the end hold zeros and meet finite garbage of U).
template <int TJ, int KMAX>
__device__ __forceinline__ void panel_step(double (&a)[8][PS_NC], const double* __restrict__ Urow, double rj, int t) {
  double xj[PS_NC];
#pragma unroll
  for (int cc = 0; cc < PS_NC; ++cc) {
    const double own = a[0][cc] * rj;
    xj[cc] = row_bcast<TJ>(own);
    a[0][cc] = (t == TJ) ? own : a[0][cc];
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    double u = Urow[16 * k + t - TJ];        // U[J, 16 (kj + k) + t]   (k == 0, t < TJ: in-bounds read, masked)
    if (k == 0) u = (t > TJ) ? u : 0.0;
#pragma unroll
    for (int cc = 0; cc < PS_NC; ++cc) a[k][cc] -= u * xj[cc];
  }
}
template <int KMAX, int... TJs>
__device__ __forceinline__ void panel_block(std::integer_sequence<int, TJs...>, double (&a)[8][PS_NC], const double* Us, const double* rinv,
                                            int kj, int t) {
  (panel_step<TJs, KMAX>(a, Us + PS_OFF(16 * kj + TJs), rinv[16 * kj + TJs], t), ...);
}

__global__ __launch_bounds__(256) void potrf_panel_solve_kernel(double* __restrict__ A, long lda, long strideA, int k0, int mcols) {
  extern __shared__ double ps_lds[];
  double* Us = ps_lds;                         // packed rows of U11: NB (NB + 1) / 2 doubles
  double* rinv = ps_lds + NB * (NB + 1) / 2;   // 1 / U_jj  (also the finite landing zone of reads past the last rows)
  const int tid = threadIdx.x, t = tid & 15, cg = tid >> 4;
  double* Ab = A + (long)blockIdx.y * strideA;
  const double* U11 = Ab + (long)k0 * lda + k0;
  double* A12 = Ab + (long)(k0 + NB) * lda + k0;
  {   // U11 -> LDS.  Loads are unconditional and issued 16 at a time (a predicated load makes the compiler
      // wait for each one: 64 serial round trips); the strictly-lower part of the block is valid memory.
    const int j = tid & (NB - 1), ih = tid >> 7;    // row j, columns ih + 2 u (coalesced down the columns)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = U11[(long)(ih + 2 * (16 * b + u)) * lda + j];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = ih + 2 * (16 * b + u);
        if (j <= i) Us[PS_OFF(j) + i - j] = v[u];
      }
    }
  }
  const int c0 = blockIdx.x * PS_COLS + cg * PS_NC;
  double a[8][PS_NC];
#pragma unroll
  for (int cc = 0; cc < PS_NC; ++cc) {
    const int col = min(c0 + cc, mcols - 1);        // (clamped: out-of-range columns are computed and dropped)
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k][cc] = A12[(long)col * lda + 16 * k + t];
  }
  __syncthreads();
  if (tid < NB) rinv[tid] = 1.0 / Us[PS_OFF(tid)];
  __syncthreads();
#pragma unroll 1
  for (int kj = 0; kj < 8; ++kj) {
    if (kj < 4) panel_block<8>(std::make_integer_sequence<int, 16>{}, a, Us, rinv, kj, t);
    else panel_block<4>(std::make_integer_sequence<int, 16>{}, a, Us, rinv, kj, t);
#pragma unroll
    for (int cc = 0; cc < PS_NC; ++cc) {
      if (c0 + cc < mcols) A12[(long)(c0 + cc) * lda + 16 * kj + t] = a[0][cc];
#pragma unroll
      for (int k = 0; k < 7; ++k) a[k][cc] = a[k + 1][cc];
      a[7][cc] = 0.0;
    }
  }
}

void potrf_panel_solve_launch(hipStream_t st, int batch, double* A, long lda, long strideA, int k0, int mcols) {
  if (mcols <= 0) return;
  const size_t lds = (size_t)(NB * (NB + 1) / 2 + NB) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    HYP_CHECK(hipFuncSetAttribute((const void*)potrf_panel_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(potrf_panel_solve_kernel, dim3((mcols + PS_COLS - 1) / PS_COLS, batch), dim3(256), lds, st, A, lda, strideA, k0, mcols);
  HYP_CHECK(hipGetLastError());
}

}  // namespace hyp
