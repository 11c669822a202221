/* A host in plain C driving libhypatia_hip.so through include/hypatia_hip.h only (no Python, no torch): the calls a Julia
 * `ccall` glue makes.  PosSemidefTri(side 3) at a perturbed interior point: feasibility, gradient,
 * H^-1 H v = v, <g, point> = -nu; then a QRChol system solver over a random G: Schur assembly + factorization, one
 * solve_subsystem3, and the identity  lhs x = rhs_x + G'(H rhs_z-part ...)  checked through lhs itself.
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
  CHECK(hyp_ctx_destroy(ctx));
  printf("c abi smoke ok\n");
  return 0;
}
