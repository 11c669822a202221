// SymIndefDenseSystemSolver on the device (/root/reference/src/Solvers/systemsolvers/symindef.jl:1-56, 203-271): the
// 3x3 symmetric indefinite form of the Newton system,
//
//     [ 0   A'   G' ] [x]
//     [ A   0    0  ] [y]  ,   M_k = (mu H_k)^-1 for a primal-barrier cone, mu H_k for a dual-barrier one,
//     [ G   0   -M  ] [z]
//
// factored by Bunch-Kaufman with rook pivoting (symm_fact_copy!, src/linearalgebra/dense.jl:170-184: on an exactly
// singular pivot, increase_diag! and again) and solved with the same triangular sweeps as the Cholesky path.  The
// reference keeps the lower triangle; the upper one is kept here (the rest of the library factors upper triangles).
// An alternative to QRChol behind the same boundary and an independent check of it: no preprocessing of A is needed.
#include "syssolver.hpp"

namespace hyp {

__global__ void increase_diag_kernel(int n, double* A, long lda);   // syssolver.hip (dense.jl:106-113)

__global__ void negate_block_kernel(int m, double* __restrict__ B, long ld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  for (int j = blockIdx.y; j < m; j += gridDim.y) B[(long)j * ld + i] = -B[(long)j * ld + i];
}

SymIndefSys::SymIndefSys(Ctx& c, int n_, int p_, int q_, const std::vector<Cone*>& cs) : ctx(c), n(n_), p(p_), q(q_), npq(n_ + p_ + q_), cones(cs) {
  HYP_REQUIRE(n >= 0 && p >= 0 && q >= 0 && npq >= 1, "symindef: sizes");
  offs.assign(cones.size() + 1, 0);
  for (size_t k = 0; k < cones.size(); ++k) offs[k + 1] = offs[k] + cones[k]->dim;
  HYP_REQUIRE(offs.back() == q, "symindef: cone dimensions do not sum to q");
  const size_t mb = (size_t)npq * npq * sizeof(double);
  lhs.alloc(mb);
  fact.alloc(mb);
  dinv.alloc(dinv_elems(npq) * sizeof(double));
  xb.alloc((size_t)npq * sizeof(double));
}

void SymIndefSys::load(const double* hA, const double* hG) {   // symindef.jl:222-240
  const size_t d = sizeof(double);
  ctx.zero(lhs.p, (size_t)npq * npq * d);
  if (p > 0 && n > 0) {
    DBuf t((size_t)p * n * d);
    ctx.h2d(t.p, hA, (size_t)p * n * d);
    dev_transpose(ctx, p, n, t.d(), p, lhs.d() + (long)n * npq, npq, 1, 0, 0);          // lhs[0:n, n:n+p] = A'
    ctx.sync();
  }
  if (q > 0 && n > 0) {
    DBuf t((size_t)q * n * d);
    ctx.h2d(t.p, hG, (size_t)q * n * d);
    dev_transpose(ctx, q, n, t.d(), q, lhs.d() + (long)(n + p) * npq, npq, 1, 0, 0);    // lhs[0:n, n+p:] = G'
    ctx.sync();
  }
  fact_ok = false;
}

void SymIndefSys::update_lhs(int* info, int* used_fallback) {   // symindef.jl:242-262
  const long z0 = n + p;
  for (size_t k = 0; k < cones.size(); ++k) {
    Cone* ck = cones[k];
    double* blk = lhs.d() + (z0 + offs[k]) * npq + (z0 + offs[k]);
    if (ck->use_dual_barrier) ck->hess_explicit(blk, npq);
    else ck->inv_hess_explicit(blk, npq);
    hipLaunchKernelGGL(negate_block_kernel, dim3((ck->dim + 255) / 256, std::min(ck->dim, 1024)), dim3(256), 0, ctx.stream, ck->dim, blk, (long)npq);
    HYP_CHECK(hipGetLastError());
  }
  // symm_fact_copy! (dense.jl:170-184)
  *used_fallback = 0;
  ctx.d2d(fact.p, lhs.p, (size_t)npq * npq * sizeof(double));
  *info = bk.factor(ctx, npq, fact.d(), npq, dinv.d());
  if (*info != 0) {
    *used_fallback = 1;
    ctx.d2d(fact.p, lhs.p, (size_t)npq * npq * sizeof(double));
    hipLaunchKernelGGL(increase_diag_kernel, dim3((npq + 255) / 256), dim3(256), 0, ctx.stream, npq, fact.d(), (long)npq);
    *info = bk.factor(ctx, npq, fact.d(), npq, dinv.d());
  }
  fact_ok = (*info == 0);
  tri.invalidate();
  if (fact_ok && ctx.trsv_plan_sb(npq) > 0) tri.build(ctx, npq, fact.d(), npq, dinv.d());
}

void SymIndefSys::solve3(double* h_sol, const double* h_rhs) {   // symindef.jl:264-271: ldiv!(sol.vec, fact, rhs.vec)
  HYP_REQUIRE(fact_ok, "symindef solve3: no valid factorization (call update_lhs)");
  ctx.h2d(xb.p, h_rhs, (size_t)npq * sizeof(double));
  double* y = bk.gather(ctx, xb.d(), npq, 1);
  for (int pass = 0; pass < 2; ++pass) {
    const bool trans = (pass == 0);
    if (tri.ready(npq)) tri.solve(ctx, fact.d(), npq, trans, y);
    else trsv_upper(ctx, npq, fact.d(), npq, dinv.d(), trans, y);
    if (pass == 0) bk.dsolve(ctx, y, npq, 1);
  }
  bk.scatter(ctx, y, xb.d(), npq, 1);
  ctx.d2h(h_sol, xb.p, (size_t)npq * sizeof(double));
  ctx.sync();
}

}  // namespace hyp
