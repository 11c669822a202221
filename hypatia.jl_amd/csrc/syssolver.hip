// QRChol system solver on the device: Schur-complement assembly (batched sqrt-Hessian products +
// FP64-MFMA syrk), blocked Cholesky, and the 3x3 solve.
// Reference: /root/reference/src/Solvers/systemsolvers/qrchol.jl (line ranges inline).
#include "syssolver.hpp"

namespace hyp {

__global__ void increase_diag_kernel(int n, double* A, long lda) {   // dense.jl:106-113
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    const double d = A[(long)j * lda + j];
    A[(long)j * lda + j] = (1.0 + 1e-5) * fmax(d, 1000 * 2.220446049250313e-16);
  }
}

SysSolver::SysSolver(Ctx& c, int n_, int p_, int q_, const std::vector<Cone*>& cs) : ctx(c), n(n_), p(p_), q(q_), nmp(n_ - p_), cones(cs) {
  HYP_REQUIRE(n >= 0 && p >= 0 && q >= 0 && p <= n, "sys: sizes");
  offs.assign(cones.size() + 1, 0);
  for (size_t k = 0; k < cones.size(); ++k) offs[k + 1] = offs[k] + cones[k]->dim;
  HYP_REQUIRE(offs.back() == q, "sys: cone dimensions do not sum to q");
  use_sqrt.assign(cones.size(), 0);
  const size_t d = sizeof(double);
  G.alloc((size_t)q * n * d);
  if (p > 0) {
    GQ1.alloc((size_t)q * p * d);
    GQ2s.alloc((size_t)q * nmp * d);
    Qm.alloc((size_t)n * n * d);
    Rinv.alloc((size_t)p * p * d);
    GQ1x.alloc((size_t)q * d);
    HGQ1x.alloc((size_t)q * d);
  }
  HGQ2.alloc((size_t)q * std::max(nmp, 1) * d);
  lhs.alloc((size_t)std::max(nmp, 1) * std::max(nmp, 1) * d);
  lhs_fact.alloc((size_t)std::max(nmp, 1) * std::max(nmp, 1) * d);
  dinv.alloc(dinv_elems(std::max(nmp, 1)) * d);
  d_info.alloc(64);
  QpbxGHbz.alloc((size_t)std::max(n, 1) * d);
  tmpn.alloc((size_t)std::max(n, 1) * d);
  Gx.alloc((size_t)std::max(q, 1) * d);
  HGx.alloc((size_t)std::max(q, 1) * d);
  tmpq.alloc((size_t)std::max(q, 1) * d);
  sol.alloc((size_t)(n + p + q + 1) * d);
  rhs.alloc((size_t)(n + p + q + 1) * d);
}

void SysSolver::load(const double* hG, const double* hGQ1, const double* hGQ2, const double* hQ, const double* hR) {
  const size_t d = sizeof(double);
  ctx.h2d(G.p, hG, (size_t)q * n * d);
  if (p > 0) {
    HYP_REQUIRE(hGQ1 && hGQ2 && hQ && hR, "sys: GQ1, GQ2, Q, R are required when p > 0");
    ctx.h2d(GQ1.p, hGQ1, (size_t)q * p * d);
    ctx.h2d(GQ2s.p, hGQ2, (size_t)q * nmp * d);
    ctx.h2d(Qm.p, hQ, (size_t)n * n * d);
    // inverse of the p x p upper triangular Ap_R, once per solve (setup, not on the iteration path)
    std::vector<double> ri((size_t)p * p, 0.0);
    for (int j = 0; j < p; ++j) {
      ri[(size_t)j * p + j] = 1.0 / hR[(size_t)j * p + j];
      for (int i = j - 1; i >= 0; --i) {
        double s = 0.0;
        for (int k = i + 1; k <= j; ++k) s += hR[(size_t)k * p + i] * ri[(size_t)j * p + k];
        ri[(size_t)j * p + i] = -s / hR[(size_t)i * p + i];
      }
    }
    ctx.h2d(Rinv.p, ri.data(), (size_t)p * p * d);
    ctx.sync();
  }
  ctx.sync();
}

void SysSolver::block_hess_prod_vec(double* d_out, const double* d_in) {   // qrchol.jl:87-98
  for (size_t k = 0; k < cones.size(); ++k) {
    Cone* ck = cones[k];
    if (ck->use_dual_barrier) ck->inv_hess_prod(d_out + offs[k], q, d_in + offs[k], q, 1);
    else ck->hess_prod(d_out + offs[k], q, d_in + offs[k], q, 1);
  }
}

void SysSolver::update_lhs_fact(int* info, int* used_fallback) {   // qrchol.jl:201-257
  *info = 0;
  *used_fallback = 0;
  if (nmp == 0) return;
  assemble_lhs();
  factor_lhs(info, used_fallback);
}

void SysSolver::assemble_lhs() {   // qrchol.jl:214-246 (this process's cones only; the multi-GPU glue sums the result)
  if (nmp == 0) return;
  const double* gq2 = GQ2();
  for (size_t k = 0; k < cones.size(); ++k) use_sqrt[k] = cones[k]->use_sqrt_hess_oracles(nmp) ? 1 : 0;   // :214-216
  bool any_sqrt = false;
  for (int v : use_sqrt) any_sqrt |= (v != 0);
  HYP_CHECK(hipEventRecord(ctx.ev[0], ctx.stream));
  HYP_CHECK(hipEventRecord(ctx.ev[1], ctx.stream));
  HYP_CHECK(hipEventRecord(ctx.ev[2], ctx.stream));
  if (any_sqrt) {   // :219-234
    int idx = 0;
    for (size_t k = 0; k < cones.size(); ++k) {
      if (!use_sqrt[k]) continue;
      Cone* ck = cones[k];
      if (ck->use_dual_barrier) ck->inv_sqrt_hess_prod(HGQ2.d() + idx, q, gq2 + offs[k], q, nmp);
      else ck->sqrt_hess_prod(HGQ2.d() + idx, q, gq2 + offs[k], q, nmp);
      idx += ck->dim;
    }
    HYP_CHECK(hipEventRecord(ctx.ev[1], ctx.stream));
    GemmArgs s{};   // lhs = HGQ2[1:idx, :]' HGQ2[1:idx, :]  (outer_prod!, dense.jl:80-86)
    s.M = nmp; s.N = nmp; s.K = idx; s.A = HGQ2.d(); s.lda = q; s.B = HGQ2.d(); s.ldb = q; s.C = lhs.d(); s.ldc = nmp;
    s.alpha = 1; s.beta = 0; s.tri = GEMM_UPPER; s.krange = KR_ALL; s.batch = 1; s.tag = 1;
    gemm(ctx, true, s);
    HYP_CHECK(hipEventRecord(ctx.ev[2], ctx.stream));
    ctx.kstat[4] += 1;
  } else {
    ctx.zero(lhs.p, (size_t)nmp * nmp * sizeof(double));
  }
  for (size_t k = 0; k < cones.size(); ++k) {   // :240-246
    if (use_sqrt[k]) continue;
    Cone* ck = cones[k];
    double* prod_k = HGQ2.d() + offs[k];
    if (ck->use_dual_barrier) ck->inv_hess_prod(prod_k, q, gq2 + offs[k], q, nmp);
    else ck->hess_prod(prod_k, q, gq2 + offs[k], q, nmp);
    GemmArgs g{};
    g.M = nmp; g.N = nmp; g.K = ck->dim; g.A = gq2 + offs[k]; g.lda = q; g.B = prod_k; g.ldb = q; g.C = lhs.d(); g.ldc = nmp;
    g.alpha = 1; g.beta = 1; g.tri = GEMM_UPPER; g.krange = KR_ALL; g.batch = 1;
    gemm(ctx, true, g);
  }
}

void SysSolver::factor_lhs(int* info, int* used_fallback) {   // qrchol.jl:249-250
  *info = 0;
  *used_fallback = 0;
  if (nmp == 0) return;
  // posdef_fact_copy! (dense.jl:194-215).  Cholesky; on failure the reference tries Bunch-Kaufman and
  // then a diagonal shift + Bunch-Kaufman.  Device Bunch-Kaufman is not built yet: the
  // fallback here is diagonal shift + Cholesky, reported through used_fallback.
  ctx.d2d(lhs_fact.p, lhs.p, (size_t)nmp * nmp * sizeof(double));
  HYP_CHECK(hipEventRecord(ctx.ev[3], ctx.stream));
  potrf_upper_batched(ctx, nmp, lhs_fact.d(), nmp, 0, 1, dinv.d(), d_info.i());
  HYP_CHECK(hipEventRecord(ctx.ev[4], ctx.stream));
  ctx.d2h(ctx.h_info, d_info.p, sizeof(int));
  ctx.sync();
  *info = ctx.h_info[0];
  {
    float ms = 0;
    HYP_CHECK(hipEventElapsedTime(&ms, ctx.ev[0], ctx.ev[1])); ctx.kstat[0] += ms;
    HYP_CHECK(hipEventElapsedTime(&ms, ctx.ev[1], ctx.ev[2])); ctx.kstat[1] += ms;
    HYP_CHECK(hipEventElapsedTime(&ms, ctx.ev[3], ctx.ev[4])); ctx.kstat[2] += ms;
    ctx.kstat[3] += 1;
  }
  if (*info != 0) {
    *used_fallback = 1;
    ctx.d2d(lhs_fact.p, lhs.p, (size_t)nmp * nmp * sizeof(double));
    hipLaunchKernelGGL(increase_diag_kernel, dim3((nmp + 255) / 256), dim3(256), 0, ctx.stream, nmp, lhs_fact.d(), (long)nmp);
    potrf_upper_batched(ctx, nmp, lhs_fact.d(), nmp, 0, 1, dinv.d(), d_info.i());
    ctx.d2h(ctx.h_info, d_info.p, sizeof(int));
    ctx.sync();
    *info = ctx.h_info[0];
  }
  fact_ok = (*info == 0);
  tri.invalidate();
  if (fact_ok && ctx.trsv_sb > 0 && nmp >= 2 * ctx.trsv_sb) tri.build(ctx, nmp, lhs_fact.d(), nmp, dinv.d());
}

void SysSolver::tri_solves(double* d_x) {
  if (tri.ready(nmp)) {
    tri.solve(ctx, lhs_fact.d(), nmp, true, d_x);
    tri.solve(ctx, lhs_fact.d(), nmp, false, d_x);
  } else {
    trsv_upper(ctx, nmp, lhs_fact.d(), nmp, dinv.d(), true, d_x);
    trsv_upper(ctx, nmp, lhs_fact.d(), nmp, dinv.d(), false, d_x);
  }
}

void SysSolver::potrs(double* d_x) { tri_solves(d_x); }

void SysSolver::solve3(double* d_sol, const double* d_rhs) {   // qrchol.jl:39-85
  const size_t d = sizeof(double);
  if (d_sol != d_rhs) ctx.d2d(d_sol, d_rhs, (size_t)(n + p + q) * d);
  double* x = d_sol;
  double* y = d_sol + n;
  double* z = d_sol + n + p;
  double* t = QpbxGHbz.d();
  // t = Q' (x + G' z)                                                   :51-53
  ctx.d2d(t, x, (size_t)n * d);
  gemv(ctx, true, q, n, 1.0, G.d(), q, z, 1.0, t);
  if (p > 0) {
    gemv(ctx, true, n, n, 1.0, Qm.d(), n, t, 0.0, tmpn.d());
    ctx.d2d(t, tmpn.p, (size_t)n * d);
    // y <- R'^-1 y ; sol.vec[1:p] = y                                   :55-57
    gemv(ctx, true, p, p, 1.0, Rinv.d(), p, y, 0.0, tmpn.d());
    ctx.d2d(y, tmpn.p, (size_t)p * d);
    ctx.d2d(x, y, (size_t)p * d);
    if (nmp > 0) {                                                       // :59-63
      gemv(ctx, false, q, p, 1.0, GQ1.d(), q, y, 0.0, GQ1x.d());
      block_hess_prod_vec(HGQ1x.d(), GQ1x.d());
      gemv(ctx, true, q, nmp, -1.0, GQ2s.d(), q, HGQ1x.d(), 1.0, t + p);
    }
  }
  if (nmp > 0) {                                                         // :66-69
    ctx.d2d(x + p, t + p, (size_t)nmp * d);
    tri_solves(x + p);
  }
  if (p > 0) {                                                           // :71  x = Q x
    gemv(ctx, false, n, n, 1.0, Qm.d(), n, x, 0.0, tmpn.d());
    ctx.d2d(x, tmpn.p, (size_t)n * d);
  }
  gemv(ctx, false, q, n, 1.0, G.d(), q, x, 0.0, Gx.d());                 // :73
  block_hess_prod_vec(HGx.d(), Gx.d());                                  // :74
  dev_axpby(ctx, q, 1.0, HGx.d(), -1.0, z);                              // :76  z = HGx - z
  if (p > 0) {                                                           // :78-82
    ctx.d2d(tmpn.p, t, (size_t)p * d);
    gemv(ctx, true, q, p, -1.0, GQ1.d(), q, HGx.d(), 1.0, tmpn.d());
    gemv(ctx, false, p, p, 1.0, Rinv.d(), p, tmpn.d(), 0.0, y);
  }
}

}  // namespace hyp
