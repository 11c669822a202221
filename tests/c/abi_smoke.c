/* A host in plain C driving libhypatia_hip.so through include/hypatia_hip.h only (no Python, no torch): the calls a Julia
 * `ccall` glue makes.  PosSemidefTri(side 3) at a perturbed interior point: feasibility, gradient,
 * H^-1 H v = v, <g, point> = -nu; then a QRChol system solver over a random G: Schur assembly + factorization, one
 * solve_subsystem3, and the identity  lhs x = rhs_x + G'(H rhs_z-part ...)  checked through lhs itself.
 * Then the two other cones of the north star, created and driven from C as well: EpiNormSpectral(2, 3) and
 * WSOSInterpNonnegative (U = 3, two basis matrices passed as an array of pointers), their use_dual_barrier flags
 * (WSOS inverts use_dual: wsosinterpnonnegative.jl:58), identities, and a system solver over both.
 * Exit code 0 on success; prints the failing check otherwise.  Built and run by tests/test_c_abi.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "hypatia_hip.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != 0) {                                                                   \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, hyp_last_error(ctx));       \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)
#define REQUIRE(cond)                                        \
  do {                                                       \
    if (!(cond)) {                                           \
      fprintf(stderr, "check failed: %s\n", #cond);          \
      return 1;                                              \
    }                                                        \
  } while (0)

int main(void) {
  hyp_ctx* ctx = NULL;
  int ndev = 0;
  if (hyp_device_count(&ndev) != 0 || ndev <= 0) {
    fprintf(stderr, "no HIP device\n");
    return 77;
  }
  CHECK(hyp_ctx_create(0, &ctx));
  enum { SIDE = 3, DIM = 6, N = 4 };
  hyp_cone* cone = NULL;
  CHECK(hyp_cone_create_possemideftri(ctx, DIM, &cone));
  double pt[DIM], g[DIM], v[DIM] = {0.3, -0.2, 0.5, 0.1, -0.4, 0.7}, hv[DIM], w[DIM], nu = 0;
  int feas = 0, dim = 0;
  CHECK(hyp_cone_dimension(cone, &dim));
  CHECK(hyp_cone_get_nu(cone, &nu));
  REQUIRE(dim == DIM && nu == SIDE);
  CHECK(hyp_cone_set_initial_point(cone, pt));
  pt[1] += 0.1; pt[3] -= 0.05; pt[0] += 0.2;          /* still positive definite */
  CHECK(hyp_cone_load_point(cone, pt, 1.0));
  CHECK(hyp_cone_reset_data(cone));
  CHECK(hyp_cone_is_feas(cone, &feas));
  REQUIRE(feas == 1);
  CHECK(hyp_cone_grad(cone, g));
  double gp = 0;
  for (int i = 0; i < DIM; ++i) gp += g[i] * pt[i];
  REQUIRE(fabs(gp + nu) < 1e-12);                        /* <grad, point> = -nu */
  CHECK(hyp_cone_hess_prod(cone, hv, DIM, v, DIM, 1));
  CHECK(hyp_cone_inv_hess_prod(cone, w, DIM, hv, DIM, 1));
  for (int i = 0; i < DIM; ++i) REQUIRE(fabs(w[i] - v[i]) < 1e-12);

  /* system solver over this cone: n = 4, p = 0, q = 6 */
  double G[DIM * N];
  unsigned s = 12345u;
  for (int i = 0; i < DIM * N; ++i) { s = s * 1664525u + 1013904223u; G[i] = ((double)(s >> 8) / 16777216.0) - 0.5; }
  hyp_sys* sys = NULL;
  hyp_cone* cones[1] = {cone};
  CHECK(hyp_sys_create(ctx, N, 0, DIM, cones, 1, &sys));
  CHECK(hyp_sys_load(sys, G, NULL, NULL, NULL, NULL));
  int use_sqrt[1] = {0}, info = -1, fb = -1;
  CHECK(hyp_sys_update_lhs_fact(sys, use_sqrt, &info, &fb));
  REQUIRE(info == 0 && fb == 0);
  double lhs[N * N];
  CHECK(hyp_sys_get_lhs(sys, lhs));
  /* lhs = G' H G: compare column 0 with G' (H G[:, 0]) */
  double Hg0[DIM];
  CHECK(hyp_cone_hess_prod(cone, Hg0, DIM, G, DIM, 1));
  for (int j = 0; j < N; ++j) {
    double r = 0;
    for (int i = 0; i < DIM; ++i) r += G[j * DIM + i] * Hg0[i];
    REQUIRE(fabs(r - lhs[j * N + 0]) < 1e-11 * (1 + fabs(r)));     /* (upper triangle: row 0) */
  }
  /* solve_subsystem3 with rhs = (x; z = 0): x_sol = lhs^-1 x, z_sol = H G x_sol */
  double rhs[N + DIM] = {1.0, -2.0, 0.5, 3.0, 0, 0, 0, 0, 0, 0}, sol[N + DIM];
  CHECK(hyp_sys_solve3(sys, sol, rhs));
  for (int i = 0; i < N; ++i) {
    double r = 0;
    for (int j = 0; j < N; ++j) r += (i <= j ? lhs[j * N + i] : lhs[i * N + j]) * sol[j];
    REQUIRE(fabs(r - rhs[i]) < 1e-10);
  }
  CHECK(hyp_sys_destroy(sys));
  CHECK(hyp_cone_destroy(cone));

  /* ---- EpiNormSpectral(d1 = 2, d2 = 3): dim 7, nu = d1 + 1 (epinormspectral.jl:53-66, 97) */
  enum { D1 = 2, D2 = 3, EDIM = 1 + D1 * D2, U = 3, K = 2, Q2 = EDIM + U, N2 = 3 };
  hyp_cone* ens = NULL;
  CHECK(hyp_cone_create_epinormspectral(ctx, D1, D2, 0, &ens));
  int udb = -1;
  CHECK(hyp_cone_dimension(ens, &dim));
  CHECK(hyp_cone_get_nu(ens, &nu));
  CHECK(hyp_cone_use_dual_barrier(ens, &udb));
  REQUIRE(dim == EDIM && nu == D1 + 1 && udb == 0);
  double ept[EDIM], eg[EDIM], ev[EDIM] = {0.2, -0.1, 0.3, 0.05, -0.2, 0.15, 0.1}, ehv[EDIM], ew[EDIM];
  CHECK(hyp_cone_set_initial_point(ens, ept));            /* (sqrt(nu), 0) */
  REQUIRE(fabs(ept[0] - sqrt(3.0)) < 1e-15);
  ept[1] += 0.3; ept[4] -= 0.2; ept[6] += 0.1;            /* sigma_1(W) well below u */
  CHECK(hyp_cone_load_point(ens, ept, 1.0));
  CHECK(hyp_cone_reset_data(ens));
  CHECK(hyp_cone_is_feas(ens, &feas));
  REQUIRE(feas == 1);
  CHECK(hyp_cone_grad(ens, eg));
  gp = 0;
  for (int i = 0; i < EDIM; ++i) gp += eg[i] * ept[i];
  REQUIRE(fabs(gp + nu) < 1e-12);
  CHECK(hyp_cone_hess_prod(ens, ehv, EDIM, ev, EDIM, 1));
  CHECK(hyp_cone_inv_hess_prod(ens, ew, EDIM, ehv, EDIM, 1));
  for (int i = 0; i < EDIM; ++i) REQUIRE(fabs(ew[i] - ev[i]) < 1e-10);
  /* dual feasibility = nuclear norm test (epinormspectral.jl:125-132): u = 10 dominates any sum of singular values here */
  double edual[EDIM] = {10.0, 0.5, -0.5, 0.25, 1.0, -1.0, 0.75};
  CHECK(hyp_cone_load_dual_point(ens, edual));
  CHECK(hyp_cone_is_dual_feas(ens, &feas));
  REQUIRE(feas == 1);
  edual[0] = 0.1;
  CHECK(hyp_cone_load_dual_point(ens, edual));
  CHECK(hyp_cone_is_dual_feas(ens, &feas));
  REQUIRE(feas == 0);

  /* ---- WSOSInterpNonnegative(U = 3, Ps = [P0 (3 x 2), P1 (3 x 1)]): univariate, half-degree 1 on [-1, 1] at the points
   *      -1, 0, 1 with the monomial basis (wsosinterpnonnegative.jl:49-63); Ps as an array of K column-major matrices */
  static const double P0[U * 2] = {1.0, 1.0, 1.0, -1.0, 0.0, 1.0};   /* columns: 1, x */
  static const double P1[U * 1] = {0.0, 1.0, 0.0};                  /* sqrt(1 - x^2) * 1 */
  const double* Ps[K] = {P0, P1};
  const int Ls[K] = {2, 1};
  hyp_cone* wsos = NULL;
  CHECK(hyp_cone_create_wsosinterpnonnegative(ctx, U, K, Ls, Ps, 0, &wsos));
  CHECK(hyp_cone_dimension(wsos, &dim));
  CHECK(hyp_cone_get_nu(wsos, &nu));
  CHECK(hyp_cone_use_dual_barrier(wsos, &udb));
  REQUIRE(dim == U && nu == 3.0 && udb == 1);            /* use_dual = false -> the barrier is for the dual cone (:58) */
  double wpt[U], wg[U], wv[U] = {0.3, -0.2, 0.1}, whv[U], ww[U];
  CHECK(hyp_cone_set_initial_point(wsos, wpt));
  for (int i = 0; i < U; ++i) REQUIRE(wpt[i] == 1.0);
  wpt[0] = 0.8; wpt[1] = 1.3; wpt[2] = 1.1;
  CHECK(hyp_cone_load_point(wsos, wpt, 1.0));
  CHECK(hyp_cone_reset_data(wsos));
  CHECK(hyp_cone_is_feas(wsos, &feas));
  REQUIRE(feas == 1);
  CHECK(hyp_cone_grad(wsos, wg));
  gp = 0;
  for (int i = 0; i < U; ++i) gp += wg[i] * wpt[i];
  REQUIRE(fabs(gp + nu) < 1e-12);
  CHECK(hyp_cone_hess_prod(wsos, whv, U, wv, U, 1));
  CHECK(hyp_cone_inv_hess_prod(wsos, ww, U, whv, U, 1));
  for (int i = 0; i < U; ++i) REQUIRE(fabs(ww[i] - wv[i]) < 1e-10);
  double wneg[U] = {1.0, -0.5, 1.0};                      /* Lambda_1 = P1' diag(pt) P1 = pt[1] < 0 */
  CHECK(hyp_cone_load_point(wsos, wneg, 1.0));
  CHECK(hyp_cone_reset_data(wsos));
  CHECK(hyp_cone_is_feas(wsos, &feas));
  REQUIRE(feas == 0);
  CHECK(hyp_cone_load_point(wsos, wpt, 1.0));
  CHECK(hyp_cone_reset_data(wsos));
  CHECK(hyp_cone_is_feas(wsos, &feas));
  REQUIRE(feas == 1);

  /* ---- a system solver over both: lhs = G1' H1 G1 + G2' H2^-1 G2 (primal- / dual-barrier cone: qrchol.jl:219-246) */
  double G2[Q2 * N2];
  for (int i = 0; i < Q2 * N2; ++i) { s = s * 1664525u + 1013904223u; G2[i] = ((double)(s >> 8) / 16777216.0) - 0.5; }
  hyp_cone* cones2[2] = {ens, wsos};
  CHECK(hyp_sys_create(ctx, N2, 0, Q2, cones2, 2, &sys));
  CHECK(hyp_sys_load(sys, G2, NULL, NULL, NULL, NULL));
  int use_sqrt2[2] = {0, 0};
  CHECK(hyp_sys_update_lhs_fact(sys, use_sqrt2, &info, &fb));
  REQUIRE(info == 0);
  double lhs2[N2 * N2], col1[EDIM], col2[U], Hc1[EDIM], Hc2[U];
  CHECK(hyp_sys_get_lhs(sys, lhs2));
  for (int i = 0; i < EDIM; ++i) col1[i] = G2[i];        /* column 0 of G, rows of each cone */
  for (int i = 0; i < U; ++i) col2[i] = G2[EDIM + i];
  CHECK(hyp_cone_hess_prod(ens, Hc1, EDIM, col1, EDIM, 1));
  CHECK(hyp_cone_inv_hess_prod(wsos, Hc2, U, col2, U, 1));
  for (int j = 0; j < N2; ++j) {
    double r = 0;
    for (int i = 0; i < EDIM; ++i) r += G2[j * Q2 + i] * Hc1[i];
    for (int i = 0; i < U; ++i) r += G2[j * Q2 + EDIM + i] * Hc2[i];
    REQUIRE(fabs(r - lhs2[j * N2 + 0]) < 1e-10 * (1 + fabs(r)));
  }
  double rhs2[N2 + Q2] = {1.0, -2.0, 0.5}, sol2[N2 + Q2];
  CHECK(hyp_sys_solve3(sys, sol2, rhs2));
  for (int i = 0; i < N2; ++i) {
    double r = 0;
    for (int j = 0; j < N2; ++j) r += (i <= j ? lhs2[j * N2 + i] : lhs2[i * N2 + j]) * sol2[j];
    REQUIRE(fabs(r - rhs2[i]) < 1e-9);
  }
  /* ---- the residual products of Solvers.jl:418-483 in one call (single GPU: the sums over ranks are the local values) */
  {
    double cvec[N2] = {0.3, -0.1, 0.2}, hvec[Q2], bdummy[1] = {0.0};
    for (int i = 0; i < Q2; ++i) hvec[i] = 0.1 * (i + 1);
    CHECK(hyp_sys_load_model(sys, cvec, bdummy, hvec, NULL));
    double xx[N2] = {0.5, -1.0, 0.25}, zz[Q2], ss[Q2], gtz[N2], gxs[Q2], dots[2];
    for (int i = 0; i < Q2; ++i) { zz[i] = 0.01 * (i + 1); ss[i] = 1.0 - 0.02 * i; }
    CHECK(hyp_sys_residual_products(sys, xx, zz, ss, gtz, gxs, dots));
    double hz = 0, zs = 0;
    for (int i = 0; i < Q2; ++i) { hz += hvec[i] * zz[i]; zs += zz[i] * ss[i]; }
    REQUIRE(fabs(dots[0] - hz) < 1e-12 && fabs(dots[1] - zs) < 1e-12);
    for (int j = 0; j < N2; ++j) {
      double r = 0;
      for (int i = 0; i < Q2; ++i) r += G2[j * Q2 + i] * zz[i];
      REQUIRE(fabs(gtz[j] - r) < 1e-12);
    }
    for (int i = 0; i < Q2; ++i) {
      double r = ss[i];
      for (int j = 0; j < N2; ++j) r += G2[j * Q2 + i] * xx[j];
      REQUIRE(fabs(gxs[i] - r) < 1e-12);
    }
    double two[2] = {1.5, -2.5};
    CHECK(hyp_sys_allreduce_host(sys, two, 2, 1));           /* no communicator: a no-op */
    REQUIRE(two[0] == 1.5 && two[1] == -2.5);
  }

  /* ---- WSOSInterpNonnegative with COMPLEX bases (wsosinterpnonnegative.jl:15 with R = Complex{T}): Ps[k] = U x L_k complex
   *      numbers, (re, im) interleaved; the cone vector stays real.  U = 4, one basis of two columns [1, z] at four points of
   *      the unit disc: Lambda(pt) = P' diag(pt) P is Hermitian 2 x 2, nu = 2 */
  {
    enum { CU = 4 };
    static const double zr[CU] = {0.5, -0.3, 0.1, -0.6}, zi[CU] = {0.2, 0.7, -0.8, -0.1};
    double CP0[2 * CU * 2];
    for (int i = 0; i < CU; ++i) { CP0[2 * i] = 1.0; CP0[2 * i + 1] = 0.0; CP0[2 * (CU + i)] = zr[i]; CP0[2 * (CU + i) + 1] = zi[i]; }
    const double* CPs[1] = {CP0};
    const int CLs[1] = {2};
    hyp_cone* cw = NULL;
    CHECK(hyp_cone_create_wsosinterpnonnegative_complex(ctx, CU, 1, CLs, CPs, 0, &cw));
    CHECK(hyp_cone_dimension(cw, &dim));
    CHECK(hyp_cone_get_nu(cw, &nu));
    CHECK(hyp_cone_use_dual_barrier(cw, &udb));
    REQUIRE(dim == CU && nu == 2.0 && udb == 1);
    double cpt[CU] = {1.2, 0.7, 0.9, 1.4}, cg[CU], cv[CU] = {0.3, -0.2, 0.1, 0.25}, chv[CU], cww[CU];
    CHECK(hyp_cone_load_point(cw, cpt, 1.0));
    CHECK(hyp_cone_reset_data(cw));
    CHECK(hyp_cone_is_feas(cw, &feas));
    REQUIRE(feas == 1);
    CHECK(hyp_cone_grad(cw, cg));
    gp = 0;
    for (int i = 0; i < CU; ++i) gp += cg[i] * cpt[i];
    REQUIRE(fabs(gp + nu) < 1e-12);                          /* <grad, point> = -nu */
    /* the gradient from the definition: -|e_i' P Lambda^-1 P' e_i| with Lambda = [[a, b], [conj b, d]] */
    double a = 0, d = 0, br = 0, bi = 0;
    for (int i = 0; i < CU; ++i) { a += cpt[i]; d += cpt[i] * (zr[i] * zr[i] + zi[i] * zi[i]); br += cpt[i] * zr[i]; bi += cpt[i] * zi[i]; }
    const double det = a * d - (br * br + bi * bi);
    for (int i = 0; i < CU; ++i) {
      const double m2 = zr[i] * zr[i] + zi[i] * zi[i];
      /* [1, conj z] Lambda^-1 [1; z] = (d - 2 Re(b z) ... ) / det with Lambda^-1 = [[d, -b], [-conj b, a]] / det */
      const double quad = (d - 2.0 * (br * zr[i] + bi * zi[i]) + a * m2) / det;
      REQUIRE(fabs(cg[i] + quad) < 1e-11);
    }
    CHECK(hyp_cone_hess_prod(cw, chv, CU, cv, CU, 1));
    CHECK(hyp_cone_inv_hess_prod(cw, cww, CU, chv, CU, 1));
    for (int i = 0; i < CU; ++i) REQUIRE(fabs(cww[i] - cv[i]) < 1e-10);
    double cneg[CU] = {-1.0, -1.0, -1.0, -1.0};
    CHECK(hyp_cone_load_point(cw, cneg, 1.0));
    CHECK(hyp_cone_reset_data(cw));
    CHECK(hyp_cone_is_feas(cw, &feas));
    REQUIRE(feas == 0);
    CHECK(hyp_cone_destroy(cw));
  }

  CHECK(hyp_sys_destroy(sys));
  CHECK(hyp_cone_destroy(ens));
  CHECK(hyp_cone_destroy(wsos));
  CHECK(hyp_ctx_destroy(ctx));
  printf("c abi smoke ok\n");
  return 0;
}
