// Column-pivoted Householder QR on the device with LAPACK dgeqp3's semantics (dgeqp3 -> dlaqps: BLAS-3 blocked, norm
// downdating with the lsticc recomputation rule), for the preprocessing of Hypatia's solve:
//   find_initial_x   /root/reference/src/Solvers/process.jl:64-178   qr!(AG, ColumnNorm()) of [A; G], rank decision :373-382
// Julia's qr(., ColumnNorm()) IS dgeqp3, so pivots, R and the rank decision follow the reference up to rounding.
// Layout: the m x n matrix (plus, optionally, one extra column b that rides along: after the factorization it holds Q'b, so
// the least-squares right-hand side needs no separate pass over the reflectors) lives in HBM column-major; R ends up in the
// upper triangle, the Householder vectors below the diagonal (unit leading entry implied), as LAPACK stores them.
// Per column (dlaqps): pivot search on the partial norms, column swap, update of the column with the block's pending
// reflectors, dlarfg, ONE pass over the trailing matrix (F(:, k) = tau A' v: the HBM-bound part, 8 m n^2 / 2 bytes in total),
// small updates of F and of the pivot row, norm downdate.  Per block of <= 64 columns: one MFMA GEMM A -= V F'.  Every
// decision stays on the device (a `stop` flag ends a block early when a norm needs recomputing, as dlaqps does); the host
// synchronises once per block.  Deterministic (fixed reduction order).
#include "hyp_internal.hpp"
#include <cmath>

namespace hyp {

namespace {

constexpr int QNB = 64;          // block size (columns per dlaqps call)
constexpr int QT = 256;

struct QrState {                 // device-resident control words of one factorization
  int pvt;                       // pivot column of the current step
  int stop;                      // lsticc != 0: the block ends after the current column (set between columns, from `pending`)
  int kcount;                    // columns completed in the current block
  int pending;                   // raised by the row kernel of the current column when a norm needs recomputing
  double tau, scale, akk, alpha; // of the current reflector
};

__global__ void qr_colnorm_kernel(int m, int n, const double* __restrict__ A, long lda, int row0, double* __restrict__ vn1, double* __restrict__ vn2,
                                  const int* __restrict__ flags) {
  const int c = blockIdx.x;
  if (c >= n || (flags && !flags[c])) return;
  __shared__ double red[QT];
  double s = 0.0;
  for (int i = row0 + threadIdx.x; i < m; i += QT) {
    const double v = A[(long)c * lda + i];
    s += v * v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = QT / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double nr = sqrt(red[0]);
    vn1[c] = nr;
    vn2[c] = nr;
  }
}

// idamax(vn1[k : n]) (lowest index on ties), then the bookkeeping swaps of dlaqps (jpvt, vn1, vn2)
__global__ void qr_pivot_kernel(int n, int k, double* __restrict__ vn1, double* __restrict__ vn2, int* __restrict__ jpvt, QrState* st) {
  if (st->stop) return;
  __shared__ double bv[QT];
  __shared__ int bi[QT];
  double best = -1.0;
  int idx = k;
  for (int c = k + threadIdx.x; c < n; c += QT) {
    const double v = fabs(vn1[c]);
    if (v > best) { best = v; idx = c; }
  }
  bv[threadIdx.x] = best;
  bi[threadIdx.x] = idx;
  __syncthreads();
  for (int w = QT / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      const double ov = bv[threadIdx.x + w];
      const int oi = bi[threadIdx.x + w];
      if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int p = bi[0];
    st->pvt = p;
    if (p != k) {
      const int t = jpvt[p]; jpvt[p] = jpvt[k]; jpvt[k] = t;
      vn1[p] = vn1[k];
      vn2[p] = vn2[k];
    }
  }
}

// swap columns k <-> pvt of A (all m rows) and the matching rows of F (first kk entries)
__global__ void qr_swap_kernel(int m, int k, int kk, int j0, double* __restrict__ A, long lda, double* __restrict__ Ft, const QrState* st) {
  if (st->stop) return;
  const int p = st->pvt;
  if (p == k) return;
  const int i = blockIdx.x * QT + threadIdx.x;
  if (i < m) {
    const double a = A[(long)k * lda + i], b = A[(long)p * lda + i];
    A[(long)k * lda + i] = b;
    A[(long)p * lda + i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < kk) {
    const int l = threadIdx.x;
    const double a = Ft[l + (long)(k - j0) * QNB], b = Ft[l + (long)(p - j0) * QNB];
    Ft[l + (long)(k - j0) * QNB] = b;
    Ft[l + (long)(p - j0) * QNB] = a;
  }
}

// A(rk : m, k) -= A(rk : m, j0 : k) F(k, 0 : kk)'  (the block's pending reflectors), then partial sums of squares of A(rk+1 : m, k)
__global__ void qr_colupd_kernel(int m, int k, int kk, int j0, double* __restrict__ A, long lda, const double* __restrict__ Ft, double* __restrict__ part,
                                 const QrState* st) {
  if (st->stop) return;
  __shared__ double f[QNB];
  __shared__ double red[QT];
  if (threadIdx.x < kk) f[threadIdx.x] = Ft[threadIdx.x + (long)(k - j0) * QNB];
  __syncthreads();
  const int rk = k;
  const int i = rk + blockIdx.x * QT + threadIdx.x;
  double sq = 0.0;
  if (i < m) {
    double a = A[(long)k * lda + i];
    for (int l = 0; l < kk; ++l) a -= A[(long)(j0 + l) * lda + i] * f[l];
    A[(long)k * lda + i] = a;
    if (i > rk) sq = a * a;
  }
  red[threadIdx.x] = sq;
  __syncthreads();
  for (int w = QT / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// dlarfg on A(rk : m, k): beta = -sign(alpha) hypot(alpha, xnorm); tau = (beta - alpha) / beta; v = x / (alpha - beta)
__global__ void qr_larfg_kernel(int m, int k, int nparts, const double* __restrict__ A, long lda, const double* __restrict__ part, double* __restrict__ tau,
                                QrState* st) {
  if (st->stop) return;
  if (threadIdx.x != 0) return;
  double s = 0.0;
  for (int b = 0; b < nparts; ++b) s += part[b];
  double xnorm = sqrt(s);
  if (!(s > 1e-280 && s < 1e280)) {   // squares near the ends of the exponent range: dnrm2's scaled sum over the tail (rare, serial)
    double scl = 0.0, ssq = 1.0;
    for (int i = k + 1; i < m; ++i) {
      const double a = fabs(A[(long)k * lda + i]);
      if (a > 0.0) {
        if (scl < a) { ssq = 1.0 + ssq * (scl / a) * (scl / a); scl = a; }
        else ssq += (a / scl) * (a / scl);
      }
    }
    xnorm = scl * sqrt(ssq);
  }
  const double alpha = A[(long)k * lda + k];
  double t = 0.0, scale = 0.0, beta = alpha;
  if (xnorm != 0.0) {
    beta = -copysign(hypot(alpha, xnorm), alpha);
    // dlarfg's guard for tiny columns: while |beta| < safmin = dlamch('S') / dlamch('E'), x, alpha and beta are scaled up by
    // 1 / safmin (at most 20 times), tau and the scaling of x are formed there, beta is scaled back
    const double safmin = 2.2250738585072014e-308 / 1.1102230246251565e-16, rsafmn = 1.0 / safmin;
    double a_s = alpha, xs = 1.0;
    int knt = 0;
    if (fabs(beta) < safmin) {
      do {
        ++knt;
        xs *= rsafmn;
        beta *= rsafmn;
        a_s *= rsafmn;
      } while (fabs(beta) < safmin && knt < 20);
      beta = -copysign(hypot(a_s, xnorm * xs), a_s);
    }
    t = (beta - a_s) / beta;
    scale = xs / (a_s - beta);
    for (int j = 0; j < knt; ++j) beta *= safmin;
  }
  tau[k] = t;
  st->tau = t;
  st->scale = scale;
  st->akk = beta;
  st->alpha = alpha;
}

__global__ void qr_scale_kernel(int m, int k, double* __restrict__ A, long lda, const QrState* st) {
  if (st->stop) return;
  const int i = k + blockIdx.x * QT + threadIdx.x;
  if (i >= m) return;
  if (i == k) A[(long)k * lda + i] = 1.0;
  else if (st->scale != 0.0) A[(long)k * lda + i] *= st->scale;
  // (xnorm == 0: H = I, tau = 0; the tail of the column is already zero)
}

// F(c, kk) = tau * A(rk : m, c)' A(rk : m, k) for the trailing columns c = k + 1 .. nt - 1: the pass over the trailing matrix
__global__ __launch_bounds__(QT) void qr_gemv_kernel(int m, int k, int kk, int j0, const double* __restrict__ A, long lda, double* __restrict__ Ft,
                                                     const QrState* st) {
  if (st->stop) return;
  const int c = k + 1 + blockIdx.x;
  const double* __restrict__ col = A + (long)c * lda;
  const double* __restrict__ v = A + (long)k * lda;
  double s0 = 0.0, s1 = 0.0;
  int i = k + threadIdx.x;
  for (; i + QT < m; i += 2 * QT) {
    s0 += col[i] * v[i];
    s1 += col[i + QT] * v[i + QT];
  }
  if (i < m) s0 += col[i] * v[i];
  __shared__ double red[QT];
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  for (int w = QT / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) Ft[kk + (long)(c - j0) * QNB] = st->tau * red[0];
}

// auxv(l) = -tau A(rk : m, j0 + l)' A(rk : m, k), l < kk
__global__ void qr_auxv_kernel(int m, int k, int j0, const double* __restrict__ A, long lda, double* __restrict__ auxv, const QrState* st) {
  if (st->stop) return;
  const int l = blockIdx.x;
  const double* __restrict__ col = A + (long)(j0 + l) * lda;
  const double* __restrict__ v = A + (long)k * lda;
  double s = 0.0;
  for (int i = k + threadIdx.x; i < m; i += QT) s += col[i] * v[i];
  __shared__ double red[QT];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = QT / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) auxv[l] = -st->tau * red[0];
}

// F(0 : kk + 1, kk) = 0; F(:, kk) += F(:, 0 : kk) auxv     (all rows c = 0 .. ncols - 1 of the block's F)
__global__ void qr_fadd_kernel(int ncols, int kk, double* __restrict__ Ft, const double* __restrict__ auxv, const QrState* st) {
  if (st->stop) return;
  const int c = blockIdx.x * QT + threadIdx.x;
  if (c >= ncols) return;
  double s = (c <= kk) ? 0.0 : Ft[kk + (long)c * QNB];
  for (int l = 0; l < kk; ++l) s += Ft[l + (long)c * QNB] * auxv[l];
  Ft[kk + (long)c * QNB] = s;
}

// pivot row: A(rk, c) -= A(rk, j0 : k + 1) F(c, 0 : kk + 1)' for c > k; norm downdate (dlaqps) for the columns of the matrix proper;
// the last thread of the step restores A(rk, k) = beta and counts the column
__global__ void qr_row_kernel(int n, int nt, int k, int kk, int j0, double* __restrict__ A, long lda, const double* __restrict__ Ft,
                              double* __restrict__ vn1, double* __restrict__ vn2, int* __restrict__ flags, double tol3z, QrState* st) {
  if (st->stop) return;
  __shared__ double arow[QNB];
  if (threadIdx.x <= kk) arow[threadIdx.x] = A[(long)(j0 + threadIdx.x) * lda + k];
  __syncthreads();
  const int c = k + 1 + blockIdx.x * QT + threadIdx.x;
  if (c < nt) {
    double a = A[(long)c * lda + k];
    for (int l = 0; l <= kk; ++l) a -= arow[l] * Ft[l + (long)(c - j0) * QNB];
    A[(long)c * lda + k] = a;
    if (c < n && vn1[c] != 0.0) {
      double temp = fabs(a) / vn1[c];
      temp = fmax(0.0, (1.0 + temp) * (1.0 - temp));
      const double r = vn1[c] / vn2[c];
      const double temp2 = temp * r * r;
      if (temp2 <= tol3z) {
        flags[c] = 1;          // norm to be recomputed after the block update (lsticc list)
        atomicOr(&st->pending, 1);   // (not `stop`: the other workgroups of THIS launch must still do their part of the row)
      } else {
        vn1[c] *= sqrt(temp);
      }
    }
  }
}
__global__ void qr_finish_col_kernel(int k, int kk, double* __restrict__ A, long lda, QrState* st) {
  // (the row kernel of THIS column may have raised `pending`: the column itself is complete; `stop` is set here, for the
  //  columns after it, whose kernels -- this one included -- then do nothing)
  if (st->stop || st->kcount != kk) return;
  A[(long)k * lda + k] = st->akk;
  st->kcount = kk + 1;
  if (st->pending) st->stop = 1;
}

// apply H_k (trans or not, a single reflector is symmetric) to a vector: two phases, fixed order
__global__ void qr_apply_dot_kernel(int m, int k, const double* __restrict__ A, long lda, const double* __restrict__ x, double* __restrict__ part) {
  __shared__ double red[QT];
  double s = 0.0;
  for (int i = k + blockIdx.x * QT + threadIdx.x; i < m; i += gridDim.x * QT) s += ((i == k) ? 1.0 : A[(long)k * lda + i]) * x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = QT / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void qr_apply_axpy_kernel(int m, int k, int nparts, const double* __restrict__ A, long lda, const double* __restrict__ tau,
                                     const double* __restrict__ part, double* __restrict__ x) {
  double s = 0.0;
  for (int b = 0; b < nparts; ++b) s += part[b];
  const double w = tau[k] * s;
  for (int i = k + blockIdx.x * QT + threadIdx.x; i < m; i += gridDim.x * QT) x[i] -= w * ((i == k) ? 1.0 : A[(long)k * lda + i]);
}

}  // namespace

struct QrcpFact {
  Ctx& c;
  int m, n, nt;       // rows, columns of the matrix, columns including the rider
  DBuf A, tau, vn1, vn2, jpvt, flags, Ft, auxv, part, state;
  QrcpFact(Ctx& ctx, int m_, int n_, bool rider) : c(ctx), m(m_), n(n_), nt(n_ + (rider ? 1 : 0)) {}
  double* a() const { return A.d(); }
  void factor();
  void apply_q(bool trans, double* d_x);
};

void QrcpFact::factor() {
  const size_t d = sizeof(double);
  const long lda = m;
  tau.alloc((size_t)n * d); vn1.alloc((size_t)n * d); vn2.alloc((size_t)n * d);
  jpvt.alloc((size_t)n * sizeof(int)); flags.alloc((size_t)n * sizeof(int));
  Ft.alloc((size_t)QNB * nt * d); auxv.alloc(QNB * d);
  const int nparts_max = (m + QT - 1) / QT;
  part.alloc((size_t)std::max(nparts_max, 64) * d);
  state.alloc(sizeof(QrState));
  hipStream_t st = c.stream;
  {
    std::vector<int> id(n);
    for (int i = 0; i < n; ++i) id[i] = i;
    HYP_CHECK(hipMemcpyAsync(jpvt.p, id.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    c.sync();
  }
  c.zero(flags.p, (size_t)n * sizeof(int));
  c.zero(tau.p, (size_t)n * d);
  hipLaunchKernelGGL(qr_colnorm_kernel, dim3(n), dim3(QT), 0, st, m, n, a(), lda, 0, vn1.d(), vn2.d(), (const int*)nullptr);
  QrState* S = (QrState*)state.p;
  const double tol3z = std::sqrt(1.1102230246251565e-16);   // sqrt(dlamch('Epsilon'))
  const int minmn = std::min(m, n);
  int j = 0;
  while (j < minmn) {
    const int jb = std::min(QNB, minmn - j);
    c.zero(state.p, sizeof(QrState));
    c.zero(Ft.p, (size_t)QNB * (nt - j) * d);
    for (int kk = 0; kk < jb; ++kk) {
      const int k = j + kk;
      hipLaunchKernelGGL(qr_pivot_kernel, dim3(1), dim3(QT), 0, st, n, k, vn1.d(), vn2.d(), jpvt.i(), S);
      hipLaunchKernelGGL(qr_swap_kernel, dim3((m + QT - 1) / QT), dim3(QT), 0, st, m, k, kk, j, a(), lda, Ft.d(), S);
      const int nparts = (m - k + QT - 1) / QT;
      hipLaunchKernelGGL(qr_colupd_kernel, dim3(nparts), dim3(QT), 0, st, m, k, kk, j, a(), lda, Ft.d(), part.d(), S);
      hipLaunchKernelGGL(qr_larfg_kernel, dim3(1), dim3(64), 0, st, m, k, nparts, a(), lda, part.d(), tau.d(), S);
      hipLaunchKernelGGL(qr_scale_kernel, dim3(nparts), dim3(QT), 0, st, m, k, a(), lda, S);
      if (nt - k - 1 > 0) hipLaunchKernelGGL(qr_gemv_kernel, dim3(nt - k - 1), dim3(QT), 0, st, m, k, kk, j, a(), lda, Ft.d(), S);
      if (kk > 0) hipLaunchKernelGGL(qr_auxv_kernel, dim3(kk), dim3(QT), 0, st, m, k, j, a(), lda, auxv.d(), S);
      hipLaunchKernelGGL(qr_fadd_kernel, dim3((nt - j + QT - 1) / QT), dim3(QT), 0, st, nt - j, kk, Ft.d(), auxv.d(), S);
      if (nt - k - 1 > 0)
        hipLaunchKernelGGL(qr_row_kernel, dim3((nt - k - 1 + QT - 1) / QT), dim3(QT), 0, st, n, nt, k, kk, j, a(), lda, Ft.d(), vn1.d(), vn2.d(), flags.i(),
                           tol3z, S);
      hipLaunchKernelGGL(qr_finish_col_kernel, dim3(1), dim3(1), 0, st, k, kk, a(), lda, S);
    }
    HYP_CHECK(hipGetLastError());
    QrState hs;
    HYP_CHECK(hipMemcpyAsync(&hs, state.p, sizeof(QrState), hipMemcpyDeviceToHost, st));
    c.sync();
    const int kb = hs.kcount;
    HYP_REQUIRE(kb >= 1 && kb <= jb, "qrcp: block made no progress");
    // block update: A(j + kb : m, j + kb : nt) -= A(j + kb : m, j : j + kb) F(kb :, 0 : kb)'
    const int mr = m - j - kb, nr = nt - j - kb;
    if (mr > 0 && nr > 0) {
      GemmArgs g{};
      g.M = mr; g.N = nr; g.K = kb;
      g.A = a() + (long)j * lda + (j + kb); g.lda = lda;             // V panel, M x K (NN form)
      g.B = Ft.d() + (long)kb * QNB; g.ldb = QNB;                    // F' : K x N, K contiguous
      g.C = a() + (long)(j + kb) * lda + (j + kb); g.ldc = lda;
      g.alpha = -1.0; g.beta = 1.0; g.tri = GEMM_FULL; g.krange = KR_ALL; g.batch = 1;
      gemm(c, false, g);
    }
    if (hs.stop) {   // recompute the flagged norms on the updated matrix (rows j + kb .. m - 1), clear the list
      hipLaunchKernelGGL(qr_colnorm_kernel, dim3(n), dim3(QT), 0, st, m, n, a(), lda, j + kb, vn1.d(), vn2.d(), (const int*)flags.i());
      c.zero(flags.p, (size_t)n * sizeof(int));
    }
    j += kb;
  }
  c.sync();
}

void QrcpFact::apply_q(bool trans, double* d_x) {   // x <- Q' x (trans) or Q x: Q = H_0 H_1 ... H_{k-1}
  const int kmax = std::min(m, n);
  const int g = std::min(64, (m + QT - 1) / QT);
  for (int t = 0; t < kmax; ++t) {
    const int k = trans ? t : kmax - 1 - t;
    hipLaunchKernelGGL(qr_apply_dot_kernel, dim3(g), dim3(QT), 0, c.stream, m, k, a(), (long)m, d_x, part.d());
    hipLaunchKernelGGL(qr_apply_axpy_kernel, dim3(g), dim3(QT), 0, c.stream, m, k, g, a(), (long)m, tau.d(), part.d(), d_x);
  }
  HYP_CHECK(hipGetLastError());
}

// ---- C-ABI helpers (capi.hip) ---------------------------------------------------------------------
QrcpFact* qrcp_create(Ctx& c, int m, int n, const double* hA, int lda, const double* hb) {
  HYP_REQUIRE(m >= 1 && n >= 1 && lda >= m, "qrcp: sizes");
  QrcpFact* f = new QrcpFact(c, m, n, hb != nullptr);
  try {
    const size_t d = sizeof(double);
    f->A.alloc((size_t)m * f->nt * d);
    if (lda == m) c.h2d(f->A.p, hA, (size_t)m * n * d);
    else HYP_CHECK(hipMemcpy2DAsync(f->A.p, (size_t)m * d, hA, (size_t)lda * d, (size_t)m * d, n, hipMemcpyHostToDevice, c.stream));
    if (hb) c.h2d(f->A.d() + (long)n * m, hb, (size_t)m * d);
    c.sync();
    f->factor();
  } catch (...) {
    delete f;
    throw;
  }
  return f;
}
void qrcp_get(QrcpFact* f, int* jpvt, double* R, double* rdiag, double* qtb) {
  Ctx& c = f->c;
  const int m = f->m, n = f->n, r = std::min(m, n);
  if (jpvt) HYP_CHECK(hipMemcpyAsync(jpvt, f->jpvt.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c.stream));
  if (R) {   // min(m, n) x n, column-major with leading dimension min(m, n); the caller reads the upper triangle
    HYP_CHECK(hipMemcpy2DAsync(R, (size_t)r * sizeof(double), f->A.p, (size_t)m * sizeof(double), (size_t)r * sizeof(double), n, hipMemcpyDeviceToHost,
                               c.stream));
  }
  if (rdiag) HYP_CHECK(hipMemcpy2DAsync(rdiag, sizeof(double), f->A.p, (size_t)(m + 1) * sizeof(double), sizeof(double), r, hipMemcpyDeviceToHost, c.stream));
  if (qtb) {
    HYP_REQUIRE(f->nt == n + 1, "qrcp_get: no right-hand side rode along");
    c.d2h(qtb, f->A.d() + (long)n * m, (size_t)m * sizeof(double));
  }
  c.sync();
}
void qrcp_apply_q(QrcpFact* f, bool trans, double* hx) {
  Ctx& c = f->c;
  DBuf x((size_t)f->m * sizeof(double));
  c.h2d(x.p, hx, (size_t)f->m * sizeof(double));
  f->apply_q(trans, x.d());
  c.d2h(hx, x.p, (size_t)f->m * sizeof(double));
  c.sync();
}
void qrcp_destroy(QrcpFact* f) { delete f; }

}  // namespace hyp
