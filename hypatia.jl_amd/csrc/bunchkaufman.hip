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
// eliminating forwards (k = 0, 1, ...) and leaving U in the upper triangle, so that the existing blocked
// triangular solves (trsv_upper / trsm_upper_left / TriSolvePlan: U'^-1 then U^-1) are reused unchanged,
// with a gather by P before, a block-diagonal solve between and a scatter after.
//
// Storage during the factorization: the TRANSPOSE of the upper triangle (element (i, j), i <= j, at B[i * ld + j]), so that
// "row k right of the diagonal" -- the pivot column of the symmetric matrix, read by the search, scaled by the
// elimination -- is contiguous.  On the upper triangle itself those were 5000 accesses 40 KB apart per pass (one cache line
// and one page each): 30 us per pivot step against 8 us now.  factor() transposes in and out.
//
// Shape of the computation.  A pivot step is inherently sequential (search -> interchange -> eliminate), so the host
// enqueues pairs of launches with no involvement of its own: a one-workgroup kernel that does the whole rook search,
// the interchanges and the scaling of the pivot row(s) -- all decisions on the device; the step's column index lives in
// device memory because a 2x2 pivot advances it by two -- and a many-workgroup update of the trailing triangle, the
// HBM-bound part (16 B per element and pass).  Updates are DELAYED: a step that takes its diagonal entry as a 1x1 pivot
// without interchange (the usual case for the nearly positive definite matrices a failed Cholesky hands over) needs only
// its own row up to date, which the pivot kernel does itself from the pending (l, w) vectors; the trailing matrix is
// updated once per BK_M such steps (rank-BK_M, one pass) instead of once per step, and the update launches in between
// return at once.  A step that needs the rook search first has the pending eliminations applied (one extra launch pair),
// then runs on current data and is applied immediately, as in the unblocked algorithm.
#include "hyp_internal.hpp"
#include <limits.h>

namespace hyp {
namespace {

constexpr int BK_T = 1024;   // threads of the pivot kernel
constexpr int BK_TILE = 64;  // trailing update tile
constexpr int BK_M = 8;      // pending eliminations (pairs of vectors l, w) a trailing update applies at once
constexpr double BK_SFMIN = 2.2250738585072014e-308;   // dlamch('S')
constexpr int BK_NI = 5;     // columns per thread in the register-resident fast path of the pivot kernel (n <= BK_T * BK_NI)

struct BkState {
  int knext;   // first column not yet eliminated
  int k;       // column of the step just prepared
  int kstep;   // 1 or 2
  int skip;    // 1: exactly singular column, nothing to eliminate (LAPACK sets info and moves on)
  int info;    // 0 or 1-based index of the first exactly singular pivot
  int n2x2;    // number of 2x2 pivots (diagnostics)
  int npend;   // eliminations whose trailing update is still pending (their l / w vectors sit in slots 0 .. npend - 1)
  int flush;   // 1: the update launch that follows applies the pending eliminations to rows >= base (and the next pivot launch starts with none)
  int base;    // first row the flush applies to
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

// Barrier of the register-resident fast path of the pivot kernel, where the only data the workgroup exchanges are in LDS:
// its barriers order LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access of the wavefront
// (s_waitcnt vmcnt(0)) -- here the prefetch of the next row and the stores of the step just finished, i.e. a full memory
// round trip in front of each barrier of a step.
__device__ __forceinline__ void bk_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// symmetric interchange of indices a < b, all rows (the rows above the active block hold U, which the rook variant
// permutes too, so that ONE permutation describes the factorization).  E(i, j), i <= j, is B[i * ld + j].
__device__ __forceinline__ void bk_swap(int n, double* __restrict__ B, long ld, int a, int b, int* __restrict__ perm) {
  const int t = threadIdx.x;
  double* ra = B + (long)a * ld;   // E(a, .)
  double* rb = B + (long)b * ld;   // E(b, .)
  for (int i = t; i < a; i += BK_T) {   // E(i, a) <-> E(i, b)
    double* pi = B + (long)i * ld;
    const double x = pi[a]; pi[a] = pi[b]; pi[b] = x;
  }
  for (int i = a + 1 + t; i < b; i += BK_T) {   // E(a, i) <-> E(i, b)
    double* pc = B + (long)i * ld + b;
    const double x = ra[i]; ra[i] = *pc; *pc = x;
  }
  for (int i = b + 1 + t; i < n; i += BK_T) {   // E(a, i) <-> E(b, i)
    const double x = ra[i]; ra[i] = rb[i]; rb[i] = x;
  }
  if (t == 0) {
    const double x = ra[a]; ra[a] = rb[b]; rb[b] = x;
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
  __shared__ double s_lk[BK_M];
  const int t = threadIdx.x;
  const double alpha = 0.6403882032022076;   // (1 + sqrt(17)) / 8
  double* PW = wl;                     // slot p: w_p at PW + p n
  double* PL = wl + (long)BK_M * n;    //         l_p at PL + p n
  int k = st->knext;
  int m = st->flush ? 0 : st->npend;   // (a flush ran since the last pivot launch: nothing is pending any more)
  int n2x2 = st->n2x2, info = st->info;
  if (k >= n) {
    if (t == 0) { st->k = n; st->flush = 0; }
    return;
  }
  // Steps that take their diagonal entry as a 1x1 pivot without interchange follow one another INSIDE this launch (each
  // needs only its own row up to date, which the loop does from the pending vectors -- written by this same workgroup a
  // moment ago); the launch ends when the pending slots are full, a step needs the rook search, or the matrix is done.
  //
  // Fast path for the usual case (nothing pending at the start, n <= BK_T * BK_NI): a thread keeps its BK_NI columns of
  // the pending w vectors in REGISTERS and the few entries of the l vectors the coming rows need in LDS, and the next row is
  // fetched while the current one is searched -- a step then costs the two reductions and no dependent memory round trip
  // (through global memory it was three: pending vectors, row, re-read for the scaling; 6 us per step, now ~2).  Same
  // operations in the same order as the generic loop below, which takes over (at the same k, with the same pending
  // vectors in memory) as soon as a step is not a plain 1x1 pivot.
  if (m == 0 && n <= BK_T * BK_NI) {
    __shared__ double lwin[BK_M - 1][BK_M];   // lwin[p][r] = l_p[k0 + r]
    __shared__ double s_piv;
    __shared__ int s_flag;
    const int k0 = k;
    double wreg[BK_M - 1][BK_NI], row[BK_NI];
#pragma unroll
    for (int i = 0; i < BK_NI; ++i) row[i] = A[(long)k0 * lda + min(t + BK_T * i, n - 1)];
    bool generic = false;
    int steps = 0;
#pragma unroll
    for (int sidx = 0; sidx < BK_M - 1; ++sidx) {
      const int kk = k0 + sidx;
      if (kk >= n) break;
      double v[BK_NI], nxt[BK_NI];
#pragma unroll
      for (int i = 0; i < BK_NI; ++i) {
        v[i] = row[i];
#pragma unroll
        for (int pp = 0; pp < sidx; ++pp) v[i] -= lwin[pp][sidx] * wreg[pp][i];
      }
      const int kn = min(kk + 1, n - 1);
#pragma unroll
      for (int i = 0; i < BK_NI; ++i) nxt[i] = A[(long)kn * lda + min(t + BK_T * i, n - 1)];
      // dsytf2_rook keeps the diagonal entry as a 1x1 pivot without interchange iff NOT (|a_kk| < alpha colmax).  Only that
      // decision is needed here (a step that fails it is redone by the generic loop, which finds the index), and
      // |a_kk| < alpha max_j |v_j|  <=>  some j has alpha |v_j| > |a_kk|  (rounded multiplication by alpha > 0 is monotone):
      // the block-wide argmax -- two levels of cross-lane shuffles through the LDS crossbar -- becomes a broadcast of a_kk
      // and a block-wide OR (a ballot per wavefront, one LDS word).  |a_kk| = 0 or NaN also leaves the fast path.
      if (t == 0) s_flag = 0;
#pragma unroll
      for (int i = 0; i < BK_NI; ++i)
        if (t + BK_T * i == kk) s_piv = v[i];
      bk_lds_barrier();
      const double d = s_piv;
      const double absakk = fabs(d);
      bool larger = false;
#pragma unroll
      for (int i = 0; i < BK_NI; ++i) {
        const int j = t + BK_T * i;
        larger |= (j > kk && j < n && alpha * fabs(v[i]) > absakk);
      }
      if (__ballot(larger) != 0 && (t & 63) == 0) s_flag = 1;
      bk_lds_barrier();
      if (s_flag != 0 || !(absakk > 0.0)) {   // not a plain 1x1 pivot
        generic = true;
        break;
      }
      double* w1 = PW + (long)sidx * n;
      double* l1 = PL + (long)sidx * n;
      double* rkk = A + (long)kk * lda;
      // dsytf2_rook / dlasyf_rook scale the column by the reciprocal of the pivot (R1 = ONE / A(K,K), DSCAL) when
      // |pivot| >= sfmin and divide entry by entry otherwise: one division per step instead of BK_NI per thread
      const bool recip = absakk >= BK_SFMIN;
      const double r1 = 1.0 / d;
#pragma unroll
      for (int i = 0; i < BK_NI; ++i) {
        const int j = t + BK_T * i;
        const double w = v[i];
        const double l = recip ? w * r1 : w / d;
        if (j > kk && j < n) {
          w1[j] = w; l1[j] = l;
          rkk[j] = l;
          if (j - k0 < BK_M) lwin[sidx][j - k0] = l;
        }
        wreg[sidx][i] = (j > kk && j < n) ? w : 0.0;
        row[i] = nxt[i];
      }
      if (t == 0) {
        dd[kk] = d; de[kk] = 0.0; blk[kk] = 0;
        rkk[kk] = 1.0;
      }
      steps = sidx + 1;
      bk_lds_barrier();
    }
    k = k0 + steps;
    m = steps;
    if (!generic) {   // the slots are full or the matrix is done
      if (t == 0) {
        st->k = k - 1;
        st->kstep = 1;
        st->skip = 0;
        st->knext = k;
        st->npend = m;
        st->base = k;
        st->flush = (m > BK_M - 2) ? 1 : 0;
      }
      return;
    }
    __syncthreads();
  }
  for (;;) {
    double* rk = A + (long)k * lda;   // E(k, .): the pivot column of the symmetric matrix, contiguous
    if (m > 0) {   // bring row k up to date: E(k, j) -= sum_p l_p[k] w_p[j]
      __syncthreads();
      if (t < m) s_lk[t] = PL[(long)t * n + k];
      __syncthreads();
      for (int j = k + t; j < n; j += BK_T) {
        double v = rk[j];
        for (int p = 0; p < m; ++p) v -= s_lk[p] * PW[(long)p * n + j];
        rk[j] = v;
      }
      __syncthreads();
    }
    const double absakk = fabs(rk[k]);
    double colmax = -1.0;
    int imax = INT_MAX;
    for (int j = k + 1 + t; j < n; j += BK_T) {
      const double a = fabs(rk[j]);
      if (a > colmax) { colmax = a; imax = j; }
    }
    bk_argmax(colmax, imax, s_v, s_i);
    if (colmax < 0.0) colmax = 0.0;

    int kstep = 1, kp = k, p = k;
    bool skip = false, searched = false;
    if (fmax(absakk, colmax) == 0.0 || absakk != absakk) {
      skip = true;
    } else if (!(absakk < alpha * colmax)) {
      kp = k;
    } else {
      if (m > 0) {   // the rook search reads other rows: have the pending eliminations applied first, then come back to step k
        if (t == 0) {   // (row k itself is already up to date: the flush starts below it)
          st->k = n;
          st->knext = k;
          st->npend = m;
          st->flush = 1;
          st->base = k + 1;
          st->n2x2 = n2x2;
          st->info = info;
        }
        return;
      }
      searched = true;
      for (;;) {
        // largest off-diagonal of row/column imax inside the active block: E(j, imax) for k <= j < imax (strided),
        // E(imax, j) for j > imax (contiguous)
        double rowmax = -1.0;
        int jmax = INT_MAX;
        const double* ri = A + (long)imax * lda;
        for (int j = k + t; j < imax; j += BK_T) {
          const double a = fabs(A[(long)j * lda + imax]);
          if (a > rowmax) { rowmax = a; jmax = j; }
        }
        for (int j = imax + 1 + t; j < n; j += BK_T) {
          const double a = fabs(ri[j]);
          if (a > rowmax) { rowmax = a; jmax = j; }
        }
        bk_argmax(rowmax, jmax, s_v, s_i);
        if (rowmax < 0.0) rowmax = 0.0;
        if (!(fabs(ri[imax]) < alpha * rowmax)) {
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

    double* w1 = PW + (long)m * n;
    double* l1 = PL + (long)m * n;
    if (skip) {
      if (t == 0) {
        dd[k] = rk[k];
        de[k] = 0.0;
        blk[k] = 0;
        rk[k] = 1.0;
      }
      if (info == 0) info = k + 1;
    } else if (kstep == 1) {
      const double d = rk[k];
      const bool recip = fabs(d) >= BK_SFMIN;   // (LAPACK's rule, as in the fast path above)
      const double r1 = 1.0 / d;
      __syncthreads();
      for (int j = k + 1 + t; j < n; j += BK_T) {
        const double w = rk[j];
        const double l = recip ? w * r1 : w / d;
        w1[j] = w; l1[j] = l;
        rk[j] = l;
      }
      if (t == 0) {
        dd[k] = d; de[k] = 0.0; blk[k] = 0;
        rk[k] = 1.0;
      }
    } else {
      double* w2 = PW + (long)(m + 1) * n;
      double* l2 = PL + (long)(m + 1) * n;
      const double d11 = rk[k];
      double* rk1 = A + (long)(k + 1) * lda;   // E(k + 1, .)
      const double d12 = rk[k + 1];
      const double d22 = rk1[k + 1];
      __syncthreads();
      // [l1 l2] = [w1 w2] D^-1 with the scaling of dsytf2_rook (everything divided by the large off-diagonal)
      const double D11 = d22 / d12, D22 = d11 / d12;
      const double T = 1.0 / (D11 * D22 - 1.0);
      for (int j = k + 2 + t; j < n; j += BK_T) {
        const double a = rk[j], b = rk1[j];
        const double la = T * (D11 * a - b) / d12;
        const double lb = T * (D22 * b - a) / d12;
        w1[j] = a; w2[j] = b; l1[j] = la; l2[j] = lb;
        rk[j] = la; rk1[j] = lb;
      }
      if (t == 0) {
        dd[k] = d11; dd[k + 1] = d22; de[k] = d12; de[k + 1] = 0.0;
        blk[k] = 1; blk[k + 1] = 2;
        rk[k] = 1.0;
        rk[k + 1] = 0.0;
        rk1[k + 1] = 1.0;
      }
      n2x2 += 1;
    }
    const int np = m + (skip ? 0 : kstep);
    const int knext = k + kstep;
    // apply at once after a searched step (the next one most likely searches too and must see current data), or when
    // the slots could not take a 2x2 pivot any more
    const bool flush = np > 0 && (searched || np > BK_M - 2);
    if (flush || knext >= n) {
      if (t == 0) {
        st->k = k;
        st->kstep = kstep;
        st->skip = skip ? 1 : 0;
        st->knext = knext;
        st->npend = np;
        st->base = knext;
        st->flush = flush ? 1 : 0;
        st->n2x2 = n2x2;
        st->info = info;
      }
      return;
    }
    k = knext;
    m = np;
  }
}

// flush: E(i, j) -= sum_p l_p[i] w_p[j] for base <= i <= j over the npend pending eliminations (E(i, j) at B[i * ld + j]:
// lanes run along j); tiles are numbered along the triangle of the tile grid (the host launches for the largest block the
// flush can have, surplus workgroups leave; without the flush flag every workgroup leaves at once)
__global__ __launch_bounds__(256) void bk_update_kernel(int n, double* __restrict__ A, long lda, const BkState* __restrict__ st,
                                                        const double* __restrict__ wl) {
  if (!st->flush) return;
  const int m = st->npend;
  const int base = st->base;
  const int rem = n - base;
  if (rem <= 0 || m <= 0) return;
  const int nt = (rem + BK_TILE - 1) / BK_TILE;
  const long idx = blockIdx.x;
  if (idx >= (long)nt * (nt + 1) / 2) return;
  int bj = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
  while ((long)bj * (bj + 1) / 2 > idx) --bj;
  while ((long)(bj + 1) * (bj + 2) / 2 <= idx) ++bj;
  const int bi = (int)(idx - (long)bj * (bj + 1) / 2);   // bi <= bj: tile rows i, tile columns j
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = base + bj * BK_TILE + tx;
  const int i0 = base + bi * BK_TILE;
  const double* PW = wl;
  const double* PL = wl + (long)BK_M * n;
  __shared__ double s_l[BK_M][BK_TILE];
  for (int e = threadIdx.x; e < BK_M * BK_TILE; e += 256) {
    const int p = e / BK_TILE, c = e % BK_TILE;
    s_l[p][c] = (p < m && i0 + c < n) ? PL[(long)p * n + i0 + c] : 0.0;
  }
  __syncthreads();
  if (j >= n) return;
  double a[BK_M];
#pragma unroll
  for (int p = 0; p < BK_M; ++p) a[p] = (p < m) ? PW[(long)p * n + j] : 0.0;
#pragma unroll 2
  for (int c = ty; c < BK_TILE; c += 4) {
    const int i = i0 + c;
    if (i < n && i <= j) {
      double* pe = A + (long)i * lda + j;
      double v = *pe;
#pragma unroll
      for (int p = 0; p < BK_M; ++p) v -= s_l[p][c] * a[p];
      *pe = v;
    }
  }
}

__global__ void bk_init_kernel(int n, BkState* st, int* perm, int k0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) perm[i] = i;
  if (i == 0) { st->knext = k0; st->k = k0; st->kstep = 1; st->skip = 0; st->info = 0; st->n2x2 = 0; st->npend = 0; st->flush = 0; st->base = k0; }
}
// factor_from: row i < k0 of a Cholesky factor (E(i, j) at B[i * ld + j], j >= i) -> row of the unit factor, its pivot into D
__global__ __launch_bounds__(256) void bk_unit_rows_kernel(int n, double* __restrict__ B, long ld, double* __restrict__ dd, double* __restrict__ de,
                                                           int* __restrict__ blk) {
  const int i = blockIdx.x;
  double* r = B + (long)i * ld;
  const double u = r[i];
  const double inv = 1.0 / u;
  __syncthreads();
  for (int j = i + 1 + threadIdx.x; j < n; j += 256) r[j] *= inv;
  if (threadIdx.x == 0) { r[i] = 1.0; dd[i] = u * u; de[i] = 0.0; blk[i] = 0; }
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

int BKFact::factor(Ctx& c, int n_, double* A, long lda, double* dinv) { return factor_from(c, n_, A, lda, dinv, 0); }

int BKFact::factor_from(Ctx& c, int n_, double* A, long lda, double* dinv, int k0) {
  c.kstat[6] += 1;   // (Bunch-Kaufman factorizations: the fallback of a failed Cholesky, cone Hessian or Schur matrix)
  n = n_;
  if (n <= 0) return 0;
  HYP_REQUIRE(k0 >= 0 && k0 < n, "Bunch-Kaufman: start column");
  const size_t d = sizeof(double);
  dd.ensure((size_t)n * d);
  de.ensure((size_t)n * d);
  blk.ensure((size_t)n * sizeof(int));
  perm.ensure((size_t)n * sizeof(int));
  wl.ensure((size_t)2 * BK_M * n * d);
  state.ensure(64);
  tr.ensure((size_t)n * n * d);
  double* B = tr.d();
  dev_transpose(c, n, n, A, lda, B, n, 1, 0, 0);   // B[i * n + j] = A(i, j): the upper triangle, transposed
  BkState* st = (BkState*)state.p;
  hipLaunchKernelGGL(bk_init_kernel, dim3((n + 255) / 256), dim3(256), 0, c.stream, n, st, perm.i(), k0);
  if (k0 > 0) hipLaunchKernelGGL(bk_unit_rows_kernel, dim3(k0), dim3(256), 0, c.stream, n, B, (long)n, dd.d(), de.d(), blk.i());
  // Launch pairs are enqueued in chunks; after each chunk the host reads how far the device got.  A pair either completes
  // a step or only requests a flush (always followed by a pair that completes one), so after q further pairs the next
  // column is at least known + q / 2: that bounds the trailing block an update launch can meet.
  const int CH = std::min(128, 2 * (n - k0) + 2);
  int known = k0;
  for (int chunk = 0; chunk < 4 * (n - k0) / CH + 8 && known < n; ++chunk) {
    for (int q = 0; q < CH; ++q) {
      hipLaunchKernelGGL(bk_pivot_kernel, dim3(1), dim3(BK_T), 0, c.stream, n, B, (long)n, st, dd.d(), de.d(), blk.i(), perm.i(), wl.d());
      const int rem = n - 1 - (known + q / 2);
      if (rem > 0) {
        const long nt = (rem + BK_TILE - 1) / BK_TILE;
        hipLaunchKernelGGL(bk_update_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, c.stream, n, B, (long)n, st, wl.d());
      }
    }
    c.d2h(c.h_info, st, sizeof(BkState));
    c.sync();
    known = ((const BkState*)c.h_info)->knext;
  }
  HYP_REQUIRE(known >= n, "Bunch-Kaufman: the factorization did not finish (internal)");
  dev_transpose(c, n, n, B, n, A, lda, 1, 0, 0);   // U back into the upper triangle of A (its lower triangle gets its old content back)
  if (dinv) potrf_invert_diag_blocks(c, n, A, lda, 0, 1, dinv);
  c.d2h(c.h_info, st, sizeof(BkState));
  c.sync();
  const BkState* hs = (const BkState*)c.h_info;
  n_2x2 = hs->n2x2;
  return hs->info;
}

// ---- growth guard of the hybrid factorization ----------------------------------------------------------------------------------
// The block steps kept from the Cholesky are UNPIVOTED eliminations.  dsytrf_rook (the reference's symm_fact!, dense.jl:164-165) is the
// fall-back because its element growth is bounded; a kept step is as good only while (a) its pivots are not positive by rounding
// alone, U_ii^2 >= n eps max|a_ii|, and (b) its rows show no growth, U_ij^2 <= 16 max|a_ii| (a positive definite matrix has
// U_ij^2 <= a_jj).  out[2 b] = min_i U_ii^2, out[2 b + 1] = max_{j >= i} U_ij^2 over the rows i of block step b; out[2 kb] = max|a_ii|
// of the matrix itself (its diagonal is read from the caller's copy).
__global__ __launch_bounds__(256) void bk_guard_kernel(int n, int kb, const double* __restrict__ F, long ldf, const double* __restrict__ Asrc,
                                                       long lds, double* __restrict__ out) {
  __shared__ double red[2][256];
  const int b = blockIdx.x, t = threadIdx.x;
  double mn = INFINITY, mx = 0.0;
  if (b < kb) {
    const int i = b * NB + (t & (NB - 1));
    for (int j = b * NB + (t >> 7); j < n; j += 2) {
      if (j < i) continue;
      const double u = F[(long)j * ldf + i], u2 = u * u;
      if (!(u2 <= mx)) mx = u2;                  // (a NaN sticks)
      if (j == i && !(u2 >= mn)) mn = u2;
    }
  } else {
    for (int i = t; i < n; i += 256) {
      const double a = fabs(Asrc[(long)i * lds + i]);
      if (!(a <= mx)) mx = a;
    }
    mn = 0.0;
  }
  red[0][t] = mn;
  red[1][t] = mx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) {
      const double m2 = red[0][t + off], x2 = red[1][t + off];
      if (!(m2 >= red[0][t])) red[0][t] = m2;
      if (!(x2 <= red[1][t])) red[1][t] = x2;
    }
    __syncthreads();
  }
  if (t == 0) {
    if (b < kb) { out[2 * b] = red[0][0]; out[2 * b + 1] = red[1][0]; }
    else out[2 * kb] = red[1][0];
  }
}

// number of leading block steps (<= kb) of the partly factored F that pass the guard
static int bk_guard_steps(Ctx& c, BKFact& bk, int n, int kb, const double* F, long ldf, const double* Asrc, long lds) {
  static const bool on = [] { const char* e = getenv("HYP_BK_GUARD"); return !(e && e[0] == '0'); }();
  if (!on) return kb;
  bk.guard.ensure((size_t)(2 * kb + 1) * sizeof(double));
  hipLaunchKernelGGL(bk_guard_kernel, dim3(kb + 1), dim3(256), 0, c.stream, n, kb, F, ldf, Asrc, lds, bk.guard.d());
  std::vector<double> h(2 * kb + 1);
  c.d2h(h.data(), bk.guard.p, h.size() * sizeof(double));
  c.sync();
  const double amax = h[2 * kb];
  if (!(amax > 0.0) || !std::isfinite(amax)) return 0;
  const double piv_floor = (double)n * 2.220446049250313e-16 * amax, row_cap = 16.0 * amax;
  int ok = 0;
  while (ok < kb && h[2 * ok] >= piv_floor && h[2 * ok + 1] <= row_cap) ++ok;
  return ok;
}

int bk_after_failed_cholesky(Ctx& c, BKFact& bk, int n, double* A, long lda, double* dinv, int* d_info_scratch, int chol_info,
                             const double* A_src, long ld_src) {
  static const bool hybrid = [] { const char* e = getenv("HYP_BK_HYBRID"); return !(e && e[0] == '0'); }();
  int kb = (chol_info > 0) ? (chol_info - 1) / NB : 0;   // block step of the failing pivot: the steps before it succeeded
  bk.k0_used = 0;
  bk.guard_trimmed = 0;
  for (int attempt = 0; hybrid && kb >= 1 && kb * NB < n && attempt < 2; ++attempt) {
    potrf_upper_batched(c, n, A, lda, 0, 1, nullptr, d_info_scratch, kb);
    c.d2h(c.h_info + 41, d_info_scratch, sizeof(int));
    c.sync();
    // (a failure here cannot happen -- the same kernels on the same data succeeded a moment ago --; the matrix is then half eliminated,
    //  so start again from the caller's copy with the plain factorization)
    const int ok = (c.h_info[41] == 0) ? bk_guard_steps(c, bk, n, kb, A, lda, A_src, ld_src) : 0;
    if (ok == kb) {
      bk.k0_used = kb * NB;
      ++c.bk_hybrid_count;
      return bk.factor_from(c, n, A, lda, dinv, kb * NB);
    }
    // some kept step fails the guard: the matrix again, and only the steps in front of the first offender (none: plain rook pivoting)
    bk.guard_trimmed = kb - ok;
    ++c.bk_guard_trims;
    HYP_CHECK(hipMemcpy2DAsync(A, (size_t)lda * sizeof(double), A_src, (size_t)ld_src * sizeof(double), (size_t)n * sizeof(double), n,
                               hipMemcpyDeviceToDevice, c.stream));
    kb = ok;
  }
  ++c.bk_plain_count;
  return bk.factor(c, n, A, lda, dinv);
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
