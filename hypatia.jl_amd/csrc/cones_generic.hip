// Cones with an explicitly formed and factored Hessian: the generic fallbacks of
// /root/reference/src/Cones/Cones.jl:101-118, 189-259, and WSOSInterpNonnegative
// (/root/reference/src/Cones/wsosinterpnonnegative.jl).
#include "cones.hpp"

namespace hyp {

static const double EPS = 2.220446049250313e-16;

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------
// out[i, j] = s[i] * in[i, j]   (row scaling, m x n col-major)
__global__ void row_scale_kernel(int m, int n, const double* __restrict__ s, const double* __restrict__ in, long ldi,
                                 double* __restrict__ out, long ldo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const double si = s[i];
  for (int j = blockIdx.y; j < n; j += gridDim.y) out[(long)j * ldo + i] = si * in[(long)j * ldi + i];
}
// out[j] (+)= sign * sum_i A[i, j] * B[i, j]  (column dots of two m x n matrices); one wavefront per column
__global__ __launch_bounds__(256) void col_dot_kernel(int m, int n, const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                      long ldb, double sign, int accumulate, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= n) return;
  const double* a = A + (long)col * lda;
  const double* b = B + (long)col * ldb;
  double s = 0.0;
  for (int i = lane; i < m; i += 64) s += a[i] * b[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) out[col] = (accumulate ? out[col] : 0.0) + sign * s;
}

static void row_scale(Ctx& c, int m, int n, const double* s, const double* in, long ldi, double* out, long ldo) {
  hipLaunchKernelGGL(row_scale_kernel, dim3((m + 255) / 256, std::min(n, 2048)), dim3(256), 0, c.stream, m, n, s, in, ldi, out, ldo);
  HYP_CHECK(hipGetLastError());
}
static void col_dot(Ctx& c, int m, int n, const double* A, long lda, const double* B, long ldb, double sign, bool acc, double* out) {
  hipLaunchKernelGGL(col_dot_kernel, dim3((n + 3) / 4), dim3(256), 0, c.stream, m, n, A, lda, B, ldb, sign, acc ? 1 : 0, out);
  HYP_CHECK(hipGetLastError());
}
static int read_info(Ctx& ctx, const int* d_info) {
  ctx.d2h(ctx.h_info, d_info, sizeof(int));
  ctx.sync();
  return ctx.h_info[0];
}

// ---------------------------------------------------------------------------------------------
// GenericHessCone
// ---------------------------------------------------------------------------------------------
// The dim x dim explicit Hessian and its factor are allocated on first use: the closed-form oracles of a large cone
// (EpiNormSpectral 500 x 500: dim = 250 001, an explicit Hessian of 500 GB) work without them, exactly as in the
// reference, where only `inv_hess_prod!` / `check_numerics` / `get_proxsqr` reach the generic explicit path.
void GenericHessCone::ensure_hess_storage(bool with_fact) {
  const size_t mb = (size_t)dim * dim * sizeof(double);
  H.ensure(mb);
  if (with_fact) {
    Hfact.ensure(mb);
    Hdinv.ensure(dinv_elems(dim) * sizeof(double));
  }
}

void GenericHessCone::alloc_generic() {
  Hinfo.alloc(64);
  tmpd.alloc((size_t)dim * sizeof(double));
  tmpd2.alloc((size_t)dim * sizeof(double));
}

bool GenericHessCone::update_hess_fact() {   // Cones.jl:239-251: posdef_fact_copy!(hess_fact_mat, hess, false)
  if (hess_fact_updated) return hess_fact_ok;
  if (!hess_updated) update_hess();
  ensure_hess_storage(true);
  // posdef_fact_copy!(hess_fact_mat, hess, false), dense.jl:194-215: Cholesky, else Bunch-Kaufman (rook), no shift
  const char* fb = getenv("HYP_FORCE_BK");   // tests: treat the Cholesky as failed
  const bool force_bk = fb && fb[0] && fb[0] != '0';
  hess_fact_bk = false;
  hess_fact_ok = false;
  int hinfo_fail = 0;
  ctx.kstat[5] += 1;   // (cone Hessian factorizations: bench.py's executed-work count)
  if (!force_bk) {
    ctx.d2d(Hfact.p, H.p, (size_t)dim * dim * sizeof(double));
    potrf_upper_batched(ctx, dim, Hfact.d(), dim, 0, 1, Hdinv.d(), Hinfo.i());
    const int hinfo = read_info(ctx, Hinfo.i());
    hess_fact_ok = (hinfo == 0);
    hinfo_fail = hinfo;
    static const bool tdbg = [] { const char* e = getenv("HYP_TRIAL_DBG"); return e && e[0] == '1'; }();
    if (tdbg && hinfo != 0) fprintf(stderr, "[hess] Cholesky of the %d x %d Hessian failed at pivot %d\n", dim, dim, hinfo);
  }
  if (hess_fact_ok) {
    dev_zero_strict_lower(ctx, dim, Hfact.d(), dim, 1, 0);
  } else {
    hess_fact_bk = true;
    ctx.d2d(Hfact.p, H.p, (size_t)dim * dim * sizeof(double));
    hess_fact_ok = (bk_after_failed_cholesky(ctx, Hbk, dim, Hfact.d(), dim, Hdinv.d(), Hinfo.i(), force_bk ? 0 : hinfo_fail, H.d(), dim) == 0);
  }
  hess_fact_updated = true;
  Hplan.invalidate();
  return hess_fact_ok;
}

void GenericHessCone::update_use_hess_prod_slow() {   // Cones.jl:222-231
  if (!hess_updated) update_hess();
  gemv(ctx, false, dim, dim, 1.0, H.d(), dim, point.d(), 0.0, tmpd.d());
  const double rel_viol = fabs(1.0 - dot_host(dim, point.d(), tmpd.d()) / nu);
  use_hess_prod_slow = (rel_viol > dim * sqrt(EPS));
  use_hess_prod_slow_updated = true;
}

void GenericHessCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :101-105
  if (!hess_updated) update_hess();
  if (ncols <= 0) return;
  if (ncols == 1) {
    gemv(ctx, false, dim, dim, 1.0, H.d(), dim, arr, 0.0, prod);
    return;
  }
  GemmArgs g{};   // prod = H * arr (H symmetric, both triangles stored: the K-contiguous "TN" form)
  g.M = dim; g.N = ncols; g.K = dim; g.A = H.d(); g.lda = dim; g.B = arr; g.ldb = lda; g.C = prod; g.ldc = ldp;
  g.alpha = 1; g.beta = 0; g.batch = 1;
  gemm(ctx, true, g);
}

void GenericHessCone::inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :113-118
  update_hess_fact();
  HYP_REQUIRE(hess_fact_ok, "inv_hess_prod: the cone Hessian has no factorization (singular)");
  if (ncols <= 0) return;
  if (prod != arr) HYP_CHECK(hipMemcpy2DAsync(prod, ldp * sizeof(double), arr, lda * sizeof(double), (size_t)dim * sizeof(double), ncols,
                                              hipMemcpyDeviceToDevice, ctx.stream));
  if (ncols == 1 && ctx.trsv_plan_sb(dim) > 0) {
    // one vector against a large factor (check_numerics / get_proxsqr of every line-search trial): the super-block solves
    // of the system solver (76 dependent block steps otherwise); Bunch-Kaufman factor: P before, D^-1 between, P' after
    if (!Hplan.ready(dim)) Hplan.build(ctx, dim, Hfact.d(), dim, Hdinv.d());
    if (!hess_fact_bk) {
      Hplan.solve_both(ctx, Hfact.d(), dim, prod, ldp, 1);
    } else {
      double* y = Hbk.gather(ctx, prod, ldp, 1);
      Hplan.solve(ctx, Hfact.d(), dim, true, y);
      Hbk.dsolve(ctx, y, dim, 1);
      Hplan.solve(ctx, Hfact.d(), dim, false, y);
      Hbk.scatter(ctx, y, prod, ldp, 1);
    }
  } else if (ncols == 2 && ctx.trsv_plan_sb(dim) > 0) {   // the pair of the proximity test (Cone::prox_launch): the same sweeps on two columns
    if (!Hplan.ready(dim)) Hplan.build(ctx, dim, Hfact.d(), dim, Hdinv.d());
    if (!hess_fact_bk) {
      Hplan.solve_both(ctx, Hfact.d(), dim, prod, ldp, 2);
    } else {
      double* y = Hbk.gather(ctx, prod, ldp, 2);
      Hplan.solve_multi(ctx, Hfact.d(), dim, true, y, dim, 2);
      Hbk.dsolve(ctx, y, dim, 2);
      Hplan.solve_multi(ctx, Hfact.d(), dim, false, y, dim, 2);
      Hbk.scatter(ctx, y, prod, ldp, 2);
    }
  } else if (hess_fact_bk) {   // ldiv!(::BunchKaufman, .)
    Hbk.solve(ctx, Hfact.d(), dim, Hdinv.d(), prod, ldp, ncols, trsm_work);
  } else if (ncols == 1) {
    trsv_upper(ctx, dim, Hfact.d(), dim, Hdinv.d(), true, prod);
    trsv_upper(ctx, dim, Hfact.d(), dim, Hdinv.d(), false, prod);
  } else {
    trsm_work.ensure((size_t)NB * ncols * sizeof(double));
    trsm_upper_left(ctx, dim, ncols, Hfact.d(), dim, Hdinv.d(), true, prod, ldp, trsm_work.d());
    trsm_upper_left(ctx, dim, ncols, Hfact.d(), dim, Hdinv.d(), false, prod, ldp, trsm_work.d());
  }
}

bool GenericHessCone::use_sqrt_hess_oracles(int arr_dim) {   // :189-195
  if (!hess_fact_updated) {
    if (arr_dim < dim) return false;
    if (!update_hess_fact()) return false;
  }
  return hess_fact_ok && !hess_fact_bk;   // (hess_fact isa Cholesky)
}

void GenericHessCone::sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :198-206  U * arr
  HYP_REQUIRE(hess_fact_updated && hess_fact_ok && !hess_fact_bk, "sqrt_hess_prod: no Cholesky factor");
  GemmArgs g{};
  g.M = dim; g.N = ncols; g.K = dim; g.A = Hfact.d(); g.lda = dim; g.B = arr; g.ldb = lda; g.C = prod; g.ldc = ldp;
  g.alpha = 1; g.beta = 0; g.krange = KR_GE_M; g.batch = 1;
  gemm(ctx, false, g);
}

void GenericHessCone::inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :209-218  U'^-1 arr
  HYP_REQUIRE(hess_fact_updated && hess_fact_ok && !hess_fact_bk, "inv_sqrt_hess_prod: no Cholesky factor");
  if (prod != arr) HYP_CHECK(hipMemcpy2DAsync(prod, ldp * sizeof(double), arr, lda * sizeof(double), (size_t)dim * sizeof(double), ncols,
                                              hipMemcpyDeviceToDevice, ctx.stream));
  if (ncols >= 1 && ncols <= 2 && ctx.trsv_plan_sb(dim) > 0) {
    // one or two columns against a large factor (polymin in primal form: n = 1, so update_lhs applies U'^-1 to ONE column of length
    // U = 4845 -- 38 block steps of the multi-column solve, 1.3 ms of launches, against 0.15 ms through the super-block plan the
    // proximity test builds for this factor anyway; profiles/r04_cfg5p_trial_timeline.txt)
    if (!Hplan.ready(dim)) Hplan.build(ctx, dim, Hfact.d(), dim, Hdinv.d());
    if (ncols == 1) Hplan.solve(ctx, Hfact.d(), dim, true, prod);
    else Hplan.solve_multi(ctx, Hfact.d(), dim, true, prod, ldp, 2);
    return;
  }
  trsm_work.ensure((size_t)NB * std::max(ncols, 1) * sizeof(double));
  trsm_upper_left(ctx, dim, ncols, Hfact.d(), dim, Hdinv.d(), true, prod, ldp, trsm_work.d());
}

void GenericHessCone::hess_explicit(double* d_out, long ld) {
  if (!hess_updated) update_hess();
  HYP_CHECK(hipMemcpy2DAsync(d_out, ld * sizeof(double), H.p, (size_t)dim * sizeof(double), (size_t)dim * sizeof(double), dim,
                             hipMemcpyDeviceToDevice, ctx.stream));
  ctx.sync();
}

// ---------------------------------------------------------------------------------------------
// WSOSInterpNonnegative (wsosinterpnonnegative.jl:16-200).  The barrier is for the DUAL cone:
// use_dual_barrier = !use_dual (:58).  Lambda_k = P_k' diag(pt) P_k = U_k' U_k (upper factor; the
// reference keeps the lower one, L_k = U_k').
// ---------------------------------------------------------------------------------------------
WsosCone::WsosCone(Ctx& c, int U_, int K_, const int* Ls_, const double* const* hPs, bool use_dual) : GenericHessCone(c, CONE_WSOS) {
  HYP_REQUIRE(U_ >= 1 && K_ >= 1, "WSOS: sizes");
  U = U_; K = K_;
  dim = U;
  use_dual_barrier = !use_dual;
  nu = 0;
  for (int k = 0; k < K; ++k) {
    HYP_REQUIRE(Ls_[k] >= 1 && Ls_[k] <= U, "WSOS: 1 <= L_k <= U");
    Ls.push_back(Ls_[k]);
    nu += Ls_[k];
  }
  alloc_common();
  alloc_generic();
  infos.alloc(64 * sizeof(int));
  size_t lam_total = 0;
  for (int k = 0; k < K; ++k) lam_total += (((size_t)Ls[k] * Ls[k] + 1) & ~(size_t)1);   // (every matrix 16-byte aligned)
  LamArena.alloc(lam_total * sizeof(double));   // the Lambda_k back to back: runs of equal L_k factor as ONE batched Cholesky
  size_t lam_off = 0;
  for (int k = 0; k < K; ++k) {
    const int Lk = Ls[k];
    const size_t pb = (size_t)U * Lk * sizeof(double);
    P.emplace_back(pb); PT.emplace_back(pb); SP.emplace_back(pb); LFLP.emplace_back(pb); LFLPT.emplace_back(pb); LU.emplace_back(pb);
    Lam.emplace_back();
    Lam.back().view(LamArena.d() + lam_off, (size_t)Lk * Lk * sizeof(double));
    lam_off += (((size_t)Lk * Lk + 1) & ~(size_t)1);
    LL.emplace_back((size_t)Lk * Lk * sizeof(double));
    LamDinv.emplace_back(dinv_elems(Lk) * sizeof(double));
    ctx.h2d(P[k].p, hPs[k], pb);
    dev_transpose(ctx, U, Lk, P[k].d(), U, PT[k].d(), Lk, 1, 0, 0);
  }
  ctx.sync();
}

void WsosCone::set_initial_point(double* h) {   // :87
  for (int i = 0; i < dim; ++i) h[i] = 1.0;
}

bool WsosCone::update_feas() {   // :89-117
  // The K chains (scale, Lambda_k = P_k' diag(pt) P_k, its Cholesky) are independent and each is bound by the latency of
  // the small factorization (0.2 - 0.3 ms; 1.6 ms one after the other at U = 4845, K = 5, with a host round trip per k for
  // the reference's early exit).  They are dealt out to the two streams by accumulated block steps and all flags come back with
  // one synchronisation; the point is feasible iff every factorization succeeded, as in the reference's sweep.
  static const bool par = [] { const char* e = getenv("HYP_WSOS_PAR"); return !(e && e[0] == '0'); }();
  int Lmax = 0;
  for (int k = 0; k < K; ++k) Lmax = std::max(Lmax, Ls[k]);
  // (from 6 block steps on the factorization itself runs on both streams -- and is no longer bound by latency alone)
  if (par && K >= 2 && K <= 64 && Lmax < 6 * NB) {
    hipEvent_t e0 = ctx.aux_event(2);
    HYP_CHECK(hipEventRecord(e0, ctx.stream));                 // (the point was loaded on the main stream)
    HYP_CHECK(hipStreamWaitEvent(ctx.stream2, e0, 0));
    // runs of equal L_k (the weighted bases of a box domain all have the same size) factor as one BATCHED Cholesky: the
    // latency of one chain for the whole run.  Groups are dealt out to the two streams by accumulated latency.
    int load[2] = {0, 0};
    for (int k0 = 0; k0 < K;) {
      const int Lk = Ls[k0];
      int cnt = 1;
      while (k0 + cnt < K && Ls[k0 + cnt] == Lk) ++cnt;
      const int side = (load[1] < load[0]) ? 1 : 0;
      load[side] += (Lk + NB - 1) / NB + cnt;
      auto chain = [&] {
        for (int k = k0; k < k0 + cnt; ++k) {
          row_scale(ctx, U, Lk, point.d(), P[k].d(), U, SP[k].d(), U);
          GemmArgs g{};
          g.M = Lk; g.N = Lk; g.K = U; g.A = SP[k].d(); g.lda = U; g.B = P[k].d(); g.ldb = U; g.C = Lam[k].d(); g.ldc = Lk;
          g.alpha = 1; g.beta = 0; g.tri = GEMM_UPPER; g.batch = 1;
          gemm(ctx, true, g);
        }
        // (factor only: the inverted diagonal blocks the gradient's triangular solves use are formed there -- a candidate that
        //  fails this test, the usual fate of the first steps of the schedule, never needs them)
        potrf_upper_batched(ctx, Lk, Lam[k0].d(), Lk, (long)(((size_t)Lk * Lk + 1) & ~(size_t)1), cnt, nullptr, infos.i() + k0);
      };
      if (side == 1) {
        StreamSwap on_helper(ctx);
        chain();
      } else {
        chain();
      }
      k0 += cnt;
    }
    hipEvent_t e1 = ctx.aux_event(3);
    HYP_CHECK(hipEventRecord(e1, ctx.stream2));
    HYP_CHECK(hipStreamWaitEvent(ctx.stream, e1, 0));
    ctx.d2h(ctx.h_info + 64, infos.p, (size_t)K * sizeof(int));
    ctx.sync();
    is_feas_ = true;
    for (int k = 0; k < K; ++k)
      if (ctx.h_info[64 + k] != 0) is_feas_ = false;
    feas_updated = true;
    lam_dinv_ready = false;
    return is_feas_;
  }
  is_feas_ = true;
  lam_dinv_ready = true;
  for (int k = 0; k < K && is_feas_; ++k) {
    const int Lk = Ls[k];
    row_scale(ctx, U, Lk, point.d(), P[k].d(), U, SP[k].d(), U);           // diag(pt) P_k
    GemmArgs g{};   // Lambda = (diag(pt) P_k)' P_k, upper triangle
    g.M = Lk; g.N = Lk; g.K = U; g.A = SP[k].d(); g.lda = U; g.B = P[k].d(); g.ldb = U; g.C = Lam[k].d(); g.ldc = Lk;
    g.alpha = 1; g.beta = 0; g.tri = GEMM_UPPER; g.batch = 1;
    gemm(ctx, true, g);
    potrf_upper_batched(ctx, Lk, Lam[k].d(), Lk, 0, 1, LamDinv[k].d(), infos.i());
    if (read_info(ctx, infos.i()) != 0) is_feas_ = false;
  }
  feas_updated = true;
  return is_feas_;
}

// out[j] = sum_k parts[k * n + j], k ascending (the order in which the one-stream form accumulates)
__global__ void sum_parts_kernel(int n, int K, const double* __restrict__ parts, double* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double s = parts[j];
  for (int k = 1; k < K; ++k) s += parts[(long)k * n + j];
  out[j] = s;
}

void WsosCone::update_grad() {   // :119-133
  ctx.kstat[7] += 1;   // (gradients of generic-Hessian cones)
  // the K chains (triangular solve of P_k' against Lambda_k's factor, column norms, transpose) are independent and short: on the
  // two streams like the feasibility chains, each with a work space and a partial gradient of its own; the partial gradients
  // are then added in the order of k, which is the order the one-stream form accumulates in (same bits)
  static const bool par = [] { const char* e = getenv("HYP_WSOS_PAR"); return !(e && e[0] == '0'); }();
  if (par && K >= 2 && K <= 64) {
    // (round 4: the chains are dealt out to Ctx::max_lanes() streams -- three by default, see there; a chain's kernels and their
    //  order do not depend on the lane, so neither do its bits)
    const int nl = std::min(K, Ctx::max_lanes());
    gparts.ensure((size_t)K * U * sizeof(double));
    trsm_work.ensure((size_t)NB * U * sizeof(double));
    trsm_work2.ensure((size_t)NB * U * sizeof(double) * (size_t)std::max(1, nl - 1));
    fork_lanes(ctx, nl);
    std::vector<int> load(nl, 0);
    for (int k = 0; k < K; ++k) {
      const int Lk = Ls[k];
      int side = 0;
      for (int i = 1; i < nl; ++i)
        if (load[i] < load[side]) side = i;
      load[side] += (Lk + NB - 1) / NB;
      double* work = (side == 0) ? trsm_work.d() : trsm_work2.d() + (size_t)(side - 1) * NB * U;
      LaneSwitch on_lane(ctx, side);
      if (!lam_dinv_ready) potrf_invert_diag_blocks(ctx, Lk, Lam[k].d(), Lk, 0, 1, LamDinv[k].d());
      ctx.d2d(LFLP[k].p, PT[k].p, (size_t)U * Lk * sizeof(double));
      trsm_upper_left(ctx, Lk, U, Lam[k].d(), Lk, LamDinv[k].d(), true, LFLP[k].d(), Lk, work);
      col_dot(ctx, Lk, U, LFLP[k].d(), Lk, LFLP[k].d(), Lk, -1.0, false, gparts.d() + (long)k * U);
      dev_transpose(ctx, Lk, U, LFLP[k].d(), Lk, LFLPT[k].d(), U, 1, 0, 0);
    }
    join_lanes(ctx, nl);
    hipLaunchKernelGGL(sum_parts_kernel, dim3((U + 255) / 256), dim3(256), 0, ctx.stream, U, K, gparts.d(), grad.d());
    HYP_CHECK(hipGetLastError());
    lam_dinv_ready = true;
    grad_updated = true;
    return;
  }
  for (int k = 0; k < K; ++k) {
    if (!lam_dinv_ready) potrf_invert_diag_blocks(ctx, Ls[k], Lam[k].d(), Ls[k], 0, 1, LamDinv[k].d());
    const int Lk = Ls[k];
    // LFLP_k = L_k^-1 P_k' = U_k'^-1 P_k'   (L_k x U)
    ctx.d2d(LFLP[k].p, PT[k].p, (size_t)U * Lk * sizeof(double));
    trsm_work.ensure((size_t)NB * U * sizeof(double));
    trsm_upper_left(ctx, Lk, U, Lam[k].d(), Lk, LamDinv[k].d(), true, LFLP[k].d(), Lk, trsm_work.d());
    col_dot(ctx, Lk, U, LFLP[k].d(), Lk, LFLP[k].d(), Lk, -1.0, k > 0, grad.d());     // grad_j -= ||LFLP_k[:, j]||^2
    dev_transpose(ctx, Lk, U, LFLP[k].d(), Lk, LFLPT[k].d(), U, 1, 0, 0);
  }
  lam_dinv_ready = true;
  grad_updated = true;
}

void WsosCone::update_hess() {   // :135-150: H = sum_k (LFLP_k' LFLP_k) .^ 2
  ensure_hess_storage(false);
  get_grad();
  for (int k = 0; k < K; ++k) {
    GemmArgs g{};
    g.M = U; g.N = U; g.K = Ls[k]; g.A = LFLP[k].d(); g.lda = Ls[k]; g.B = LFLP[k].d(); g.ldb = Ls[k]; g.C = H.d(); g.ldc = U;
    g.alpha = 1; g.beta = (k > 0) ? 1.0 : 0.0; g.tri = GEMM_UPPER; g.epi = 1; g.batch = 1;
    gemm(ctx, true, g);
  }
  dev_symmetrize_from_upper(ctx, U, H.d(), U, 1, 0);
  hess_updated = true;
}

void WsosCone::lambda_of(int k, const double* d_dir) {   // LL_k = LFLP_k diag(dir) LFLP_k'  (L x L, both triangles)
  const int Lk = Ls[k];
  // LU' = diag(dir) LFLP'  (U x L) ; LL = LU LFLP' = (diag(dir) LFLP')' LFLP'
  row_scale(ctx, U, Lk, d_dir, LFLPT[k].d(), U, SP[k].d(), U);
  GemmArgs a{};
  a.M = Lk; a.N = Lk; a.K = U; a.A = SP[k].d(); a.lda = U; a.B = LFLPT[k].d(); a.ldb = U; a.C = LL[k].d(); a.ldc = Lk;
  a.alpha = 1; a.beta = 0; a.tri = GEMM_UPPER; a.batch = 1;
  gemm(ctx, true, a);
  dev_symmetrize_from_upper(ctx, Lk, LL[k].d(), Lk, 1, 0);   // Hermitian(LLk)
}

// <v, H^-1 v> >= <v, w>^2 / <w, H w> with w = M v, M = the inverse of the Hessian at an EARLIER point -- whatever Cholesky factor
// the cone still holds from its last factorization (the iterate the line search started from, or the previous candidate): the
// closer that point, the tighter the bound, and any w gives a valid one.  <w, H w> = sum_k || LFLP_k diag(w) LFLP_k' ||_F^2 needs
// the K Gram products of the gradient's LFLP_k (0.3 ms at U = 4845) -- not the U x U Hessian (1.2 ms), its Cholesky (3.4 ms),
// the solve plan (0.65 ms) and the solves (0.4 ms) a candidate costs that far outside the neighbourhood.
void WsosCone::gram_norms(const double* d_dir, double* d_out) {
  const int nl = std::min(K, Ctx::max_lanes());
  fork_lanes(ctx, nl);
  for (int k = 0; k < K; ++k) {
    LaneSwitch on_lane(ctx, k % nl);
    lambda_of(k, d_dir);
    dev_dot(ctx, Ls[k] * Ls[k], LL[k].d(), LL[k].d(), d_out + k);
  }
  join_lanes(ctx, nl);
}

void WsosCone::hess_vec_from_LL(double* d_out) {   // :152-175 (the matrix-free Hessian product), partial sums per k as in update_grad
  gparts.ensure((size_t)K * U * sizeof(double));
  const int nl = std::min(K, Ctx::max_lanes());
  fork_lanes(ctx, nl);
  for (int k = 0; k < K; ++k) {
    const int Lk = Ls[k];
    LaneSwitch on_lane(ctx, k % nl);
    GemmArgs b{};   // LU = LL * LFLP  (L x U)
    b.M = Lk; b.N = U; b.K = Lk; b.A = LL[k].d(); b.lda = Lk; b.B = LFLP[k].d(); b.ldb = Lk; b.C = LU[k].d(); b.ldc = Lk;
    b.alpha = 1; b.beta = 0; b.batch = 1;
    gemm(ctx, true, b);
    col_dot(ctx, Lk, U, LFLP[k].d(), Lk, LU[k].d(), Lk, 1.0, false, gparts.d() + (long)k * U);
  }
  join_lanes(ctx, nl);
  hipLaunchKernelGGL(sum_parts_kernel, dim3((U + 255) / 256), dim3(256), 0, ctx.stream, U, K, gparts.d(), d_out);
  HYP_CHECK(hipGetLastError());
}

// For ANY x, <v, H^-1 v> >= 2 <v, x> - <x, H x> (it is || H^-1 v - x ||_H^2 >= 0 written out).  x runs over the span of up to
// three directions p_0 = M v, p_j = M (v - H x_{j-1}) -- preconditioned Krylov directions, M = the inverse of the Hessian at an
// EARLIER point, whose Cholesky factor the cone still holds -- and the bound is the maximum over the span, b' G^-1 b with
// G_ij = <p_i, H p_j>, b_i = <v, p_i>: every entry an explicit scalar product, so the bound is rigorous whatever the rounding
// did to the directions.  <p, H p> = sum_k || LFLP_k diag(p) LFLP_k' ||_F^2 costs the K Gram products of the gradient's
// LFLP_k (0.3 ms at U = 4845), a vector H p two more small GEMMs per k -- against the U x U Hessian (1.2 ms), its Cholesky
// (3.4 ms), the solve plan (0.65 ms) and the solves (0.4 ms) of the exact value.  One direction settles the candidates far
// outside the neighbourhood; the second and third are only asked for when the bound is inconclusive but no longer small.
bool WsosCone::prox_lower_bound(double irtmu, double limit, double* lb) {
  static const bool on = [] { const char* e = getenv("HYP_PROX_LB"); return !(e && e[0] == '0'); }();
  static const int max_dirs = [] { const char* e = getenv("HYP_PROX_LB_DIRS"); const int v = e ? atoi(e) : 3; return std::min(3, std::max(1, v)); }();
  if (!on || hess_fact_updated || !hess_fact_ok || hess_fact_bk || !Hfact.p || dim < 512 || K > 16) return false;
  if (ctx.trsv_plan_sb(dim) <= 0) return false;
  const size_t vb = (size_t)dim * sizeof(double);
  const double* g = get_grad();
  ctx.d2d(vec1.p, g, vb);
  dev_axpby(ctx, dim, irtmu, dual_point.d(), 1.0, vec1.d());        // v
  if (!Hplan.ready(dim)) Hplan.build(ctx, dim, Hfact.d(), dim, Hdinv.d());
  lbP.ensure(3 * vb); lbHP.ensure(2 * vb); lbR.ensure(vb);
  double* P0 = lbP.d();
  double* ds = ctx.dscal.d() + 44;            // [0] <v, p_j>, [1..2] <H p_i, p_j>, [3 .. 3 + K) the Gram norms
  double* hp = ctx.h_pinned + 44;
  double G[3][3], bv[3], c[3] = {0, 0, 0};
  double bound = 0.0;
  for (int j = 0; j < max_dirs; ++j) {
    double* pj = P0 + (long)j * dim;
    if (j == 0) {
      ctx.d2d(pj, vec1.p, vb);
    } else {
      // H p_{j-1} from the Gram matrices of the previous direction, then r = v - sum_i c_i H p_i
      hess_vec_from_LL(lbHP.d() + (long)(j - 1) * dim);
      ctx.d2d(lbR.p, vec1.p, vb);
      for (int i = 0; i < j; ++i) dev_axpby(ctx, dim, -c[i], lbHP.d() + (long)i * dim, 1.0, lbR.d());
      ctx.d2d(pj, lbR.p, vb);
    }
    Hplan.solve(ctx, Hfact.d(), dim, true, pj);
    Hplan.solve(ctx, Hfact.d(), dim, false, pj);                     // p_j = M (.)
    dev_dot(ctx, dim, vec1.d(), pj, ds);
    for (int i = 0; i < j; ++i) dev_dot(ctx, dim, lbHP.d() + (long)i * dim, pj, ds + 1 + i);
    gram_norms(pj, ds + 3);
    ctx.d2h(hp, ds, (size_t)(3 + K) * sizeof(double));
    ctx.sync();
    bv[j] = hp[0];
    double q = 0.0;
    for (int k = 0; k < K; ++k) q += hp[3 + k];
    G[j][j] = q;
    for (int i = 0; i < j; ++i) G[i][j] = G[j][i] = hp[1 + i];
    if (!(q > 0.0) || !(q < INFINITY) || !(bv[j] == bv[j])) break;
    // maximum of 2 <v, x> - <x, H x> over the span: solve G c = b (Cholesky of the (j + 1) x (j + 1) Gram matrix)
    const int m = j + 1;
    double Lm[3][3] = {{0}};
    bool pd = true;
    for (int r = 0; r < m && pd; ++r)
      for (int cc = 0; cc <= r; ++cc) {
        double sum = G[r][cc];
        for (int t = 0; t < cc; ++t) sum -= Lm[r][t] * Lm[cc][t];
        if (r == cc) { if (!(sum > 0.0)) { pd = false; break; } Lm[r][r] = std::sqrt(sum); }
        else Lm[r][cc] = sum / Lm[cc][cc];
      }
    if (!pd) break;
    double y[3];
    for (int r = 0; r < m; ++r) { double sum = bv[r]; for (int t = 0; t < r; ++t) sum -= Lm[r][t] * y[t]; y[r] = sum / Lm[r][r]; }
    for (int r = m - 1; r >= 0; --r) { double sum = y[r]; for (int t = r + 1; t < m; ++t) sum -= Lm[t][r] * c[t]; c[r] = sum / Lm[r][r]; }
    double val = 0.0;
    for (int r = 0; r < m; ++r) val += bv[r] * c[r];
    // (the value actually attained by this x, 2 b'c - c'G c, equals b'c for the exact solution; evaluated as written it stays a
    //  valid bound for the computed c)
    double cGc = 0.0;
    for (int r = 0; r < m; ++r) for (int t = 0; t < m; ++t) cGc += c[r] * G[r][t] * c[t];
    val = 2.0 * val - cGc;
    if (val > bound) bound = val;
    if (bound > limit || bound < 0.3 * limit) break;   // settled, or so small that the candidate is probably inside: the exact test decides
  }
  if (!(bound > 0.0)) return false;
  *lb = bound;
  return true;
}

void WsosCone::partial_lambda(int k, const double* d_dir) {   // :190-200
  const int Lk = Ls[k];
  lambda_of(k, d_dir);
  GemmArgs b{};   // LU = LL * LFLP  (L x U)
  b.M = Lk; b.N = U; b.K = Lk; b.A = LL[k].d(); b.lda = Lk; b.B = LFLP[k].d(); b.ldb = Lk; b.C = LU[k].d(); b.ldc = Lk;
  b.alpha = 1; b.beta = 0; b.batch = 1;
  gemm(ctx, true, b);
}

void WsosCone::hess_prod_slow(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :152-175
  if (!use_hess_prod_slow_updated) update_use_hess_prod_slow();
  if (!use_hess_prod_slow) {
    hess_prod(prod, ldp, arr, lda, ncols);
    return;
  }
  // the K chains of a column (two Gram-type products, a symmetrization, the column sums: ~95-130 us each at U = 4845, eight and
  // more columns per iteration through apply_lhs and the right-hand sides) are independent: on the lanes like dder3's, partial
  // sums per k added in the order of k (the one-stream form's bits)
  static const bool par = [] { const char* e = getenv("HYP_WSOS_PAR"); return !(e && e[0] == '0'); }();
  if (par && K >= 2 && K <= 64) {
    const int nl = std::min(K, Ctx::max_lanes());
    gparts.ensure((size_t)K * U * sizeof(double));
    for (int j = 0; j < ncols; ++j) {
      fork_lanes(ctx, nl);
      for (int k = 0; k < K; ++k) {
        LaneSwitch on_lane(ctx, k % nl);
        partial_lambda(k, arr + (long)j * lda);
        col_dot(ctx, Ls[k], U, LFLP[k].d(), Ls[k], LU[k].d(), Ls[k], 1.0, false, gparts.d() + (long)k * U);
      }
      join_lanes(ctx, nl);
      hipLaunchKernelGGL(sum_parts_kernel, dim3((U + 255) / 256), dim3(256), 0, ctx.stream, U, K, gparts.d(), prod + (long)j * ldp);
    }
    HYP_CHECK(hipGetLastError());
    return;
  }
  for (int j = 0; j < ncols; ++j) {
    for (int k = 0; k < K; ++k) {
      partial_lambda(k, arr + (long)j * lda);
      col_dot(ctx, Ls[k], U, LFLP[k].d(), Ls[k], LU[k].d(), Ls[k], 1.0, k > 0, prod + (long)j * ldp);
    }
  }
}

const double* WsosCone::dder3(const double* d_dir) {   // :177-188
  // the K chains (two Gram-type products, a symmetrization, the column sums) are independent and short (~95 us each at U = 4845, twice
  // per iteration): on the lanes like the gradient's chains, partial sums per k added in the order of k (the one-stream form's bits)
  static const bool par = [] { const char* e = getenv("HYP_WSOS_PAR"); return !(e && e[0] == '0'); }();
  if (par && K >= 2 && K <= 64) {
    const int nl = std::min(K, Ctx::max_lanes());
    gparts.ensure((size_t)K * U * sizeof(double));
    fork_lanes(ctx, nl);
    for (int k = 0; k < K; ++k) {
      LaneSwitch on_lane(ctx, k % nl);
      partial_lambda(k, d_dir);
      col_dot(ctx, Ls[k], U, LU[k].d(), Ls[k], LU[k].d(), Ls[k], 1.0, false, gparts.d() + (long)k * U);
    }
    join_lanes(ctx, nl);
    hipLaunchKernelGGL(sum_parts_kernel, dim3((U + 255) / 256), dim3(256), 0, ctx.stream, U, K, gparts.d(), dder3v.d());
    HYP_CHECK(hipGetLastError());
    return dder3v.d();
  }
  for (int k = 0; k < K; ++k) {
    partial_lambda(k, d_dir);
    col_dot(ctx, Ls[k], U, LU[k].d(), Ls[k], LU[k].d(), Ls[k], 1.0, k > 0, dder3v.d());
  }
  return dder3v.d();
}

// ---------------------------------------------------------------------------------------------
// LinMatrixIneq (linmatrixineq.jl:9-159), real dense symmetric members: barrier -logdet(sum_i w_i A_i).
// With sumA = U'U (the reference keeps L = U'), M_i = L^-1 A_i L^-T = R' A_i R for R = U^-1: the two-sided
// product of the PSD cone, batched over the members; everything else is a product with the
// side^2 x dim matrix Mmat = [vec M_1 ... vec M_dim]: gradient = -traces, Hessian = Mmat' Mmat (a syrk),
// the slow Hessian product and the third-order term = Mmat' (.) of a side x side matrix.
// ---------------------------------------------------------------------------------------------
__global__ void lmi_neg_trace_kernel(int side, int dim, const double* __restrict__ M, double* __restrict__ grad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dim) return;
  const double* m = M + (long)i * side * side;
  double t = 0.0;
  for (int k = 0; k < side; ++k) t += m[(long)k * side + k];
  grad[i] = -t;
}

LmiCone::LmiCone(Ctx& c, int dim_, int side_, const double* hAs, bool use_dual, bool complex_members) : GenericHessCone(c, CONE_LMI) {
  // (side_ is the side of the members as the caller sees them: complex side s -> embedded side 2 s; the bound of
  //  linmatrixineq.jl:56, svec_length(side) >= dim, is the same for real and complex members)
  HYP_REQUIRE(dim_ > 1 && side_ >= 1 && (long)side_ * (side_ + 1) / 2 >= dim_, "LinMatrixIneq: 1 < dim <= side (side + 1) / 2");   // :42, :56
  dim = dim_; side = complex_members ? 2 * side_ : side_;
  use_dual_barrier = use_dual;
  nu = side_;                                                                                         // :72
  bscale = complex_members ? 0.5 : 1.0;
  alloc_common();
  alloc_generic();
  const size_t s2 = (size_t)side * side * sizeof(double);
  Amat.alloc(s2 * dim); Mmat.alloc(s2 * dim); T.alloc(s2 * dim);
  sumA.alloc(s2); fact.alloc(s2); Rinv.alloc(s2); dirmat.alloc(s2); Zm.alloc(s2);
  fdinv.alloc(dinv_elems(side) * sizeof(double));
  infos.alloc(64);
  if (complex_members) {   // hAs: dim matrices of side_ x side_ complex numbers (re, im interleaved), column-major
    std::vector<double> emb((size_t)side * side * dim);
    const long cs = side_;
    for (long m = 0; m < dim; ++m) {
      const double* a = hAs + 2 * cs * cs * m;
      double* e = emb.data() + (size_t)side * side * m;
      for (long j = 0; j < cs; ++j)
        for (long i = 0; i < cs; ++i) {
          const double re = a[2 * (j * cs + i)], im = a[2 * (j * cs + i) + 1];
          e[(2 * j) * side + 2 * i] = re;          e[(2 * j + 1) * side + 2 * i + 1] = re;
          e[(2 * j + 1) * side + 2 * i] = -im;     e[(2 * j) * side + 2 * i + 1] = im;      // [[a, -b], [b, a]]
        }
    }
    ctx.h2d(Amat.p, emb.data(), s2 * dim);
    ctx.sync();
  } else {
    ctx.h2d(Amat.p, hAs, s2 * dim);
    ctx.sync();
  }
}

void LmiCone::set_initial_point(double* h) {   // :74-81
  for (int i = 0; i < dim; ++i) h[i] = 0.0;
  h[0] = 1.0;
}

bool LmiCone::update_feas() {   // :87-96
  const int s2 = side * side;
  gemv(ctx, false, s2, dim, 1.0, Amat.d(), s2, point.d(), 0.0, sumA.d());        // sumA = sum_i w_i A_i
  ctx.d2d(fact.p, sumA.p, (size_t)s2 * sizeof(double));
  potrf_upper_batched(ctx, side, fact.d(), side, 0, 1, fdinv.d(), infos.i());
  is_feas_ = (read_info(ctx, infos.i()) == 0);
  feas_updated = true;
  return is_feas_;
}

void LmiCone::update_grad() {   // :98-109
  const long s2 = (long)side * side;
  trtri_upper_batched(ctx, side, fact.d(), side, 0, fdinv.d(), 0, Rinv.d(), side, 0, 1);   // R = U^-1
  GemmArgs a{};   // T_i = A_i R
  a.M = side; a.N = side; a.K = side; a.A = Amat.d(); a.lda = side; a.strideA = s2; a.B = Rinv.d(); a.ldb = side; a.strideB = 0;
  a.C = T.d(); a.ldc = side; a.strideC = s2; a.alpha = 1; a.beta = 0; a.batch = dim;
  gemm(ctx, false, a);
  GemmArgs b{};   // M_i = R' T_i
  b.M = side; b.N = side; b.K = side; b.A = Rinv.d(); b.lda = side; b.strideA = 0; b.B = T.d(); b.ldb = side; b.strideB = s2;
  b.C = Mmat.d(); b.ldc = side; b.strideC = s2; b.alpha = 1; b.beta = 0; b.batch = dim;
  gemm(ctx, true, b);
  dev_symmetrize_from_upper(ctx, side, Mmat.d(), side, dim, s2);                             // Hermitian(., :U)
  hipLaunchKernelGGL(lmi_neg_trace_kernel, dim3((dim + 255) / 256), dim3(256), 0, ctx.stream, side, dim, Mmat.d(), grad.d());
  HYP_CHECK(hipGetLastError());
  if (bscale != 1.0) dev_scale_copy(ctx, dim, bscale, grad.d(), grad.d());
  grad_updated = true;
}

void LmiCone::update_hess() {   // :111-123: H[i, j] = <M_i, M_j>
  ensure_hess_storage(false);
  get_grad();
  const int s2 = side * side;
  GemmArgs g{};
  g.M = dim; g.N = dim; g.K = s2; g.A = Mmat.d(); g.lda = s2; g.B = Mmat.d(); g.ldb = s2; g.C = H.d(); g.ldc = dim;
  g.alpha = bscale; g.beta = 0; g.tri = GEMM_UPPER; g.batch = 1;
  gemm(ctx, true, g);
  dev_symmetrize_from_upper(ctx, dim, H.d(), dim, 1, 0);
  hess_updated = true;
}

void LmiCone::hess_prod_slow(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :125-144
  if (!use_hess_prod_slow_updated) update_use_hess_prod_slow();
  if (!use_hess_prod_slow) {
    hess_prod(prod, ldp, arr, lda, ncols);
    return;
  }
  if (ncols <= 0) return;
  const int s2 = side * side;
  Jm.ensure((size_t)s2 * ncols * sizeof(double));
  GemmArgs a{};   // j_mat = sum_i arr[i, j] M_i, all columns j at once
  a.M = s2; a.N = ncols; a.K = dim; a.A = Mmat.d(); a.lda = s2; a.B = arr; a.ldb = lda; a.C = Jm.d(); a.ldc = s2;
  a.alpha = 1; a.beta = 0; a.batch = 1;
  gemm(ctx, false, a);
  GemmArgs b{};   // prod[i, j] = <j_mat, M_i>
  b.M = dim; b.N = ncols; b.K = s2; b.A = Mmat.d(); b.lda = s2; b.B = Jm.d(); b.ldb = s2; b.C = prod; b.ldc = ldp;
  b.alpha = bscale; b.beta = 0; b.batch = 1;
  gemm(ctx, true, b);
}

const double* LmiCone::dder3(const double* d_dir) {   // :146-159
  const int s2 = side * side;
  gemv(ctx, false, s2, dim, 1.0, Mmat.d(), s2, d_dir, 0.0, dirmat.d());           // dir_mat = sum_i d_i M_i (symmetric)
  GemmArgs z{};   // Z = dir_mat dir_mat'
  z.M = side; z.N = side; z.K = side; z.A = dirmat.d(); z.lda = side; z.B = dirmat.d(); z.ldb = side; z.C = Zm.d(); z.ldc = side;
  z.alpha = 1; z.beta = 0; z.batch = 1;
  gemm(ctx, true, z);
  gemv(ctx, true, s2, dim, bscale, Mmat.d(), s2, Zm.d(), 0.0, dder3v.d());        // dder3_i = <Z, M_i>
  return dder3v.d();
}

// ---------------------------------------------------------------------------------------------
// DoublyNonnegativeTri (doublynonnegativetri.jl:9-205): barrier -logdet(smat(s)) - sum over the off-diagonal svec
// entries of log(s_k).  The -logdet part IS the PosSemidefTri cone at the same point (its kernels are reused through
// an inner PsdCone); the entrywise part adds a diagonal term to the gradient, Hessian and third-order oracle.  The
// inverse Hessian has no closed form: explicit Hessian + factorization, the generic path of Cones.jl:101-118, 239-259.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool svec_is_diag(long k) {   // svec position k holds (i, j) with j(j+1)/2 + i = k, i <= j
  long j = (long)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
  while (j * (j + 1) / 2 > k) --j;
  while ((j + 1) * (j + 2) / 2 <= k) ++j;
  return k == j * (j + 1) / 2 + j;
}
__global__ void dnn_count_small_kernel(int dim, const double* __restrict__ pt, double thresh, int* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < dim && !(pt[k] > thresh)) atomicAdd(out, 1);
}
// mode 0: out[k] -= 1 / p_k;  mode 1: out[k, c] += arr[k, c] / p_k / p_k;  mode 2: out[k] += (arr[k] / p_k)^2 / p_k   (off-diagonal k only)
__global__ void dnn_offdiag_kernel(int dim, int ncols, int mode, const double* __restrict__ pt, const double* __restrict__ arr, long lda,
                                   double* __restrict__ out, long ldo) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= dim || svec_is_diag(k)) return;
  const double p = pt[k];
  if (mode == 0) {
    out[k] -= 1.0 / p;
  } else if (mode == 1) {
    for (int c = blockIdx.y; c < ncols; c += gridDim.y) out[(long)c * ldo + k] += arr[(long)c * lda + k] / p / p;
  } else {
    const double t = arr[k] / p;
    out[k] += t * t / p;
  }
}

DnnCone::DnnCone(Ctx& c, int dim_, bool use_dual) : GenericHessCone(c, CONE_DNN), psd(c, dim_) {
  dim = dim_;
  use_dual_barrier = use_dual;
  nu = dim;                                                                                           // :69
  alloc_common();
  alloc_generic();
  cnt.alloc(64);
}

// real roots of a y^3 + b y^2 + c y + d (a != 0), polished by Newton steps
static int cubic_real_roots(double a, double b, double c, double d, double* r) {
  const double PI = 3.14159265358979323846;
  const double p = (3 * a * c - b * b) / (3 * a * a), q = (2 * b * b * b - 9 * a * b * c + 27 * a * a * d) / (27 * a * a * a);
  const double sh = -b / (3 * a);
  int n = 0;
  const double disc = q * q / 4 + p * p * p / 27;
  if (disc > 0) {
    const double sq = sqrt(disc);
    r[n++] = cbrt(-q / 2 + sq) + cbrt(-q / 2 - sq) + sh;
  } else if (p == 0) {
    r[n++] = sh;
  } else {
    const double m = 2 * sqrt(-p / 3);
    double arg = 3 * q / (p * m);
    arg = arg > 1 ? 1 : (arg < -1 ? -1 : arg);
    const double th = acos(arg) / 3;
    for (int k = 0; k < 3; ++k) r[n++] = m * cos(th - 2 * PI * k / 3) + sh;
  }
  for (int i = 0; i < n; ++i)
    for (int it = 0; it < 4; ++it) {
      const double y = r[i], f = ((a * y + b) * y + c) * y + d, fp = (3 * a * y + 2 * b) * y + c;
      if (fp != 0) r[i] = y - f / fp;
    }
  return n;
}

void DnnCone::set_initial_point(double* h) {   // :71-128
  const int side = psd.side;
  const double rt2 = sqrt(2.0), n = side, d = dim, tol = sqrt(EPS);
  auto approx = [&](double x, double y) { return fabs(x - y) <= tol * fmax(fabs(x), fabs(y)); };   // Julia's isapprox default
  double on_diag, off_diag;
  if (side == 1) {
    on_diag = off_diag = 1.0;
  } else if (side == 2) {
    on_diag = sqrt(5.0) / 2; off_diag = 1.0 / rt2;
  } else {
    // the off-diagonal is a root of -n-1 + (n^2+n+7) x^2 - (2n^2+8) x^4 + n^2 x^6: a cubic in y = x^2
    on_diag = n + 1; off_diag = 1.0;
    double r[3];
    const int nr = cubic_real_roots(n * n, -(2 * n * n + 8), n * n + n + 7, -(n + 1), r);
    for (int i = 0; i < nr; ++i) {
      if (!(r[i] > 0)) continue;
      const double offd = sqrt(r[i]);
      const double temp = d - (d - n) * offd * offd;
      if (!(temp > tol)) continue;
      const double ond = sqrt(temp / n);
      const double denom = ond * ond + (n - 2) / rt2 * ond * offd - (n - 1) * offd * offd / 2;
      if (approx(ond * rt2 + (n - 2) * offd, ond * denom * rt2) && approx(denom, offd * offd * (denom + 1))) {   // s = -g(s)
        on_diag = ond; off_diag = offd;
        break;
      }
    }
  }
  for (int k = 0; k < dim; ++k) h[k] = off_diag;
  long k = 0;
  for (int i = 1; i <= side; ++i) { h[k] = on_diag; k += i + 1; }
}

bool DnnCone::update_feas() {   // :130-143: every svec entry > eps, then the Cholesky of smat(point)
  ctx.zero(cnt.p, sizeof(int));
  hipLaunchKernelGGL(dnn_count_small_kernel, dim3((dim + 255) / 256), dim3(256), 0, ctx.stream, dim, point.d(), EPS, cnt.i());
  HYP_CHECK(hipGetLastError());
  is_feas_ = (read_info(ctx, cnt.i()) == 0);
  if (is_feas_) {
    psd.load_point(point.d(), 1.0);
    psd.reset_data();
    is_feas_ = psd.is_feas();
  }
  feas_updated = true;
  return is_feas_;
}

void DnnCone::update_grad() {   // :145-156
  ctx.d2d(grad.p, psd.get_grad(), (size_t)dim * sizeof(double));
  hipLaunchKernelGGL(dnn_offdiag_kernel, dim3((dim + 255) / 256), dim3(256), 0, ctx.stream, dim, 1, 0, point.d(), nullptr, 0L, grad.d(), 0L);
  HYP_CHECK(hipGetLastError());
  grad_updated = true;
}

void DnnCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :173-192
  HYP_REQUIRE(prod != arr, "DoublyNonnegativeTri hess_prod: in-place call");
  if (ncols <= 0) return;
  if (!feas_updated) update_feas();
  psd.hess_prod(prod, ldp, arr, lda, ncols);
  hipLaunchKernelGGL(dnn_offdiag_kernel, dim3((dim + 255) / 256, std::min(ncols, 1024)), dim3(256), 0, ctx.stream, dim, ncols, 1, point.d(), arr,
                     lda, prod, ldp);
  HYP_CHECK(hipGetLastError());
}

void DnnCone::update_hess() {   // :158-171: symm_kron(inv(X)) + diag(1 / s_off^2) = hess_prod applied to the identity
  ensure_hess_storage(false);
  get_grad();
  DBuf eye((size_t)dim * dim * sizeof(double));
  dev_fill_identity(ctx, dim, eye.d(), dim);
  hess_prod(H.d(), dim, eye.d(), dim, dim);
  dev_symmetrize_from_upper(ctx, dim, H.d(), dim, 1, 0);
  ctx.sync();   // (eye is released on return)
  hess_updated = true;
}

const double* DnnCone::dder3(const double* d_dir) {   // :194-205
  ctx.d2d(dder3v.p, psd.dder3(d_dir), (size_t)dim * sizeof(double));
  hipLaunchKernelGGL(dnn_offdiag_kernel, dim3((dim + 255) / 256), dim3(256), 0, ctx.stream, dim, 1, 2, point.d(), d_dir, 0L, dder3v.d(), 0L);
  HYP_CHECK(hipGetLastError());
  return dder3v.d();
}

// ---------------------------------------------------------------------------------------------
// HypoRootdetTri (hyporootdettri.jl:9-324, real): barrier -log(rootdet(W) - u) - logdet(W), W = smat(w).
// Every W-block of the reference's oracles is U^-1 (a S^2 + b S + c I) U^-T with S = U^-T R U^-1, i.e. a combination of
// the PosSemidefTri oracles at W -- W^-1 R W^-1 R W^-1 (dder3), W^-1 R W^-1 (hess_prod), W R W (inv_hess_prod) -- and of
// svec(W^-1) or w, with coefficients that depend on the column only through p = arr[1] and one inner product
// (tr S = <svec W^-1, r>, or <w, r>).  So: one batched PSD product through the inner cone, one pass of column inner
// products, one combine kernel.  The explicit Hessian (only the generic sqrt oracles need it) = hess_prod of I.
// ---------------------------------------------------------------------------------------------
__global__ void logdet_diag_kernel(int side, const double* __restrict__ U, long ld, double* __restrict__ out) {   // log det(U'U)
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < side; i += 256) s += log(U[(long)i * ld + i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = 2.0 * ((red[0] + red[1]) + (red[2] + red[3]));
}
struct RootdetScal { double zeta, phi, di, phizidi; };
// mode 0 hess_prod (:172-203): vec = svec(W^-1), t = <vec, r>;  mode 1 inv_hess_prod (:235-272): vec = w, t = <w, r>.
// out[0, j] and out[1:, j] = a P[:, j] + b_j vec, with P already in out[1:, j]
__global__ void rootdet_combine_kernel(int dw, int ncols, int mode, RootdetScal sc, const double* __restrict__ arr, long lda,
                                       const double* __restrict__ dots, const double* __restrict__ vec, double* __restrict__ out, long ldo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
    const double p = arr[(long)j * lda], t = dots[j];
    double a, b, o0;
    if (mode == 0) {
      const double c0 = sc.phizidi * t;
      const double c1 = c0 - p / sc.zeta;
      b = sc.phizidi * c1 - sc.di * c0;
      a = sc.phizidi + 1.0;
      o0 = c1 / -sc.zeta;
    } else {
      const double phidi = sc.phi * sc.di;
      a = 1.0 / (sc.phizidi + 1.0);
      const double c3 = a / sc.zeta * sc.di;
      const double c4 = sc.zeta * sc.zeta + phidi * sc.phi;
      b = phidi * (c3 * t + p);
      o0 = phidi * t + c4 * p;
    }
    if (i < dw) {
      double* o = out + (long)j * ldo + 1 + i;
      *o = a * (*o) + b * vec[i];
    }
    if (i == 0) out[(long)j * ldo] = o0;
  }
}
// out[0] = o0; out[1 + i] = c9 D3[i] + c8 P1[i] + c7 vec[i]
__global__ void rootdet_dder3_kernel(int dw, double o0, double c9, double c8, double c7, const double* __restrict__ D3, const double* __restrict__ P1,
                                     const double* __restrict__ vec, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < dw) out[1 + i] = c9 * D3[i] + c8 * P1[i] + c7 * vec[i];
  if (i == 0) out[0] = o0;
}
__global__ void scale_into_kernel(int n, double a, const double* __restrict__ x, double* __restrict__ y, double y0, int write_y0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[1 + i] = a * x[i];
  if (i == 0 && write_y0) y[0] = y0;
}

HypoRootdetTriCone::HypoRootdetTriCone(Ctx& c, int dim_, bool use_dual)
    : GenericHessCone(c, CONE_HYPOROOTDET), psd(c, dim_ - 1), psdd(c, dim_ - 1) {
  HYP_REQUIRE(dim_ >= 2, "HypoRootdetTri: dim >= 2");
  dim = dim_;
  d = psd.side;
  di = 1.0 / d;
  use_dual_barrier = use_dual;
  nu = 1 + d;                                                                                         // :80
  alloc_common();
  alloc_generic();
  Wi_vec.alloc((size_t)(dim - 1) * sizeof(double));
  tmpw.alloc((size_t)(dim - 1) * sizeof(double));
  ld.alloc(64);
}

void HypoRootdetTriCone::set_initial_point(double* h) {   // :82-99
  for (int i = 0; i < dim; ++i) h[i] = 0.0;
  const double dd = d;
  const double c1 = sqrt(5 * dd * dd + 2 * dd + 1);
  const double c2 = h[0] = -sqrt((3 * dd + 1 - c1) / (2 * dd + 2));
  const double c3 = -c2 * (dd + 1 + c1) / (2 * dd);
  long k = 1;
  for (int i = 1; i <= d; ++i) { h[k] = c3; k += i + 1; }
}

double HypoRootdetTriCone::logdet_of(PsdCone& k) {
  hipLaunchKernelGGL(logdet_diag_kernel, dim3(1), dim3(256), 0, ctx.stream, k.side, k.U.d(), (long)k.side, ld.d());
  HYP_CHECK(hipGetLastError());
  ctx.d2h(ctx.h_pinned, ld.p, sizeof(double));
  ctx.sync();
  return ctx.h_pinned[0];
}

bool HypoRootdetTriCone::update_feas() {   // :101-115
  ctx.d2h(ctx.h_pinned + 1, point.p, sizeof(double));
  psd.load_point(point.d() + 1, 1.0);
  psd.reset_data();
  is_feas_ = psd.is_feas();        // (synchronises: the u read above has arrived)
  u = ctx.h_pinned[1];
  if (is_feas_) {
    phi = exp(logdet_of(psd) / d);
    zeta = phi - u;
    is_feas_ = (zeta > EPS);
  }
  feas_updated = true;
  return is_feas_;
}

bool HypoRootdetTriCone::is_dual_feas() {   // :117-127
  ctx.d2h(ctx.h_pinned + 1, dual_point.p, sizeof(double));
  ctx.sync();
  const double ud = ctx.h_pinned[1];
  if (!(ud < -EPS)) return false;
  psdd.load_point(dual_point.d() + 1, 1.0);
  psdd.reset_data();
  if (!psdd.is_feas()) return false;
  return logdet_of(psdd) - d * log(-ud / d) > EPS;
}

void HypoRootdetTriCone::update_grad() {   // :129-141
  phizidi = phi / zeta * di;
  const int dw = dim - 1;
  dev_scale_copy(ctx, dw, -1.0, psd.get_grad(), Wi_vec.d());                // svec(W^-1) = -grad of the PSD barrier
  hipLaunchKernelGGL(scale_into_kernel, dim3((dw + 255) / 256), dim3(256), 0, ctx.stream, dw, -phizidi - 1.0, Wi_vec.d(), grad.d(), 1.0 / zeta, 1);
  HYP_CHECK(hipGetLastError());
  grad_updated = true;
}

void HypoRootdetTriCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :172-203
  HYP_REQUIRE(prod != arr, "HypoRootdetTri hess_prod: in-place call");
  if (ncols <= 0) return;
  get_grad();
  const int dw = dim - 1;
  dots.ensure((size_t)ncols * sizeof(double));
  gemv(ctx, true, dw, ncols, 1.0, arr + 1, lda, Wi_vec.d(), 0.0, dots.d());      // tr(U^-T R U^-1) = <svec W^-1, r>
  psd.hess_prod(prod + 1, ldp, arr + 1, lda, ncols);                              // svec(W^-1 R W^-1)
  hipLaunchKernelGGL(rootdet_combine_kernel, dim3((dw + 255) / 256, std::min(ncols, 1024)), dim3(256), 0, ctx.stream, dw, ncols, 0,
                     RootdetScal{zeta, phi, di, phizidi}, arr, lda, dots.d(), Wi_vec.d(), prod, ldp);
  HYP_CHECK(hipGetLastError());
}

void HypoRootdetTriCone::inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :235-272
  HYP_REQUIRE(prod != arr, "HypoRootdetTri inv_hess_prod: in-place call");
  if (ncols <= 0) return;
  get_grad();
  const int dw = dim - 1;
  dots.ensure((size_t)ncols * sizeof(double));
  gemv(ctx, true, dw, ncols, 1.0, arr + 1, lda, point.d() + 1, 0.0, dots.d());   // <w, r>
  psd.inv_hess_prod(prod + 1, ldp, arr + 1, lda, ncols);                          // svec(W R W)
  hipLaunchKernelGGL(rootdet_combine_kernel, dim3((dw + 255) / 256, std::min(ncols, 1024)), dim3(256), 0, ctx.stream, dw, ncols, 1,
                     RootdetScal{zeta, phi, di, phizidi}, arr, lda, dots.d(), point.d() + 1, prod, ldp);
  HYP_CHECK(hipGetLastError());
}

void HypoRootdetTriCone::update_hess() {   // :143-170
  ensure_hess_storage(false);
  get_grad();
  DBuf eye((size_t)dim * dim * sizeof(double));
  dev_fill_identity(ctx, dim, eye.d(), dim);
  hess_prod(H.d(), dim, eye.d(), dim, dim);
  dev_symmetrize_from_upper(ctx, dim, H.d(), dim, 1, 0);
  ctx.sync();
  hess_updated = true;
}

const double* HypoRootdetTriCone::dder3(const double* d_dir) {   // :274-324
  get_grad();
  const int dw = dim - 1;
  const double* r = d_dir + 1;
  ctx.d2h(ctx.h_pinned + 2, d_dir, sizeof(double));
  psd.hess_prod(tmpw.d(), dw, r, dw, 1);                       // P1 = svec(W^-1 R W^-1)
  const double* D3 = psd.dder3(r);                             // svec(W^-1 R W^-1 R W^-1)
  const double trS = dot_host(dw, Wi_vec.d(), r);              // tr S, S = U^-T R U^-1
  const double frS = dot_host(dw, r, tmpw.d());                // ||S||_F^2 = <R, W^-1 R W^-1>
  const double p = ctx.h_pinned[2];
  const double c0 = trS * di, c6 = frS * di;
  const double zichi = (p - phi * c0) / zeta;
  const double c1 = zichi * zichi + phi / zeta * (c6 - c0 * c0) / 2;
  const double c7 = phizidi * (c1 - c6 / 2 + c0 * (zichi + c0 / 2));
  const double c8 = -phizidi * (zichi + c0);
  const double c9 = phizidi + 1;
  hipLaunchKernelGGL(rootdet_dder3_kernel, dim3((dw + 255) / 256), dim3(256), 0, ctx.stream, dw, c1 / -zeta, c9, c8, c7, D3, tmpw.d(), Wi_vec.d(),
                     dder3v.d());
  HYP_CHECK(hipGetLastError());
  return dder3v.d();
}

// ---------------------------------------------------------------------------------------------
// HypoPerLogdetTri (hypoperlogdettri.jl:9-368, real): barrier -log(v logdet(W / v) - u) - log(v) - logdet(W).  Same
// structure as HypoRootdetTri above with two leading scalars (u, v): per column p = arr[1], q = arr[2] and one inner
// product; the W-blocks are combinations of the PSD oracles at W and of svec(W^-1) or w.
// ---------------------------------------------------------------------------------------------
struct PerLogdetScal { double v, d, zeta, phi; };
__global__ void perlogdet_combine_kernel(int dw, int ncols, int mode, PerLogdetScal sc, const double* __restrict__ arr, long lda,
                                         const double* __restrict__ dots, const double* __restrict__ vec, double* __restrict__ out, long ldo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double v = sc.v, d = sc.d, zeta = sc.zeta, phi = sc.phi;
  for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
    const double p = arr[(long)j * lda], q = arr[(long)j * lda + 1], t = dots[j];
    double a, b, o0, o1;
    if (mode == 0) {   // hess_prod (:195-236): t = <svec W^-1, r>
      const double sigma = phi - d;
      const double qzi = q / zeta;
      const double c0 = t / zeta;
      const double c1 = (v * c0 - p / zeta + sigma * qzi) / zeta;
      b = c1 * v - qzi;
      a = v / zeta + 1.0;
      o0 = -c1;
      o1 = c1 * sigma - c0 + (qzi * d + q / v) / v;
    } else {           // inv_hess_prod (:271-316): t = <w, r>
      const double zv = zeta + v;
      const double zzvi = zeta / zv;
      const double c3 = v / (zv + d * v);
      const double c0 = phi - d * zzvi;
      const double c4 = v * c3 * zv;
      const double vphi = v * phi, zvp = zeta + v * phi;
      const double c6 = vphi * vphi + zeta * (zeta + d * v) - d * zvp * zvp * c3;
      const double c7 = c4 * c0;
      const double c8 = c7 + v * zeta;
      const double c1 = t / zv;
      const double c5 = c0 * p + q + c1;
      b = v * (zzvi * p + c3 * c5);
      a = zzvi;
      o0 = c6 * p + c7 * q + c8 * c1;
      o1 = c4 * c5;
    }
    if (i < dw) {
      double* o = out + (long)j * ldo + 2 + i;
      *o = a * (*o) + b * vec[i];
    }
    if (i == 0) {
      out[(long)j * ldo] = o0;
      out[(long)j * ldo + 1] = o1;
    }
  }
}
__global__ void perlogdet_dder3_kernel(int dw, double o0, double o1, double ca, double cb, double cc, const double* __restrict__ D3,
                                       const double* __restrict__ P1, const double* __restrict__ vec, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < dw) out[2 + i] = ca * D3[i] + cb * P1[i] + cc * vec[i];
  if (i == 0) { out[0] = o0; out[1] = o1; }
}
__global__ void perlogdet_grad_kernel(int dw, double g0, double g1, double a, const double* __restrict__ x, double* __restrict__ g) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < dw) g[2 + i] = a * x[i];
  if (i == 0) { g[0] = g0; g[1] = g1; }
}

HypoPerLogdetTriCone::HypoPerLogdetTriCone(Ctx& c, int dim_, bool use_dual)
    : GenericHessCone(c, CONE_HYPOPERLOGDET), psd(c, dim_ - 2), psdd(c, dim_ - 2) {
  HYP_REQUIRE(dim_ >= 3, "HypoPerLogdetTri: dim >= 3");
  dim = dim_;
  d = psd.side;
  use_dual_barrier = use_dual;
  nu = 2 + d;                                                                                         // :78
  alloc_common();
  alloc_generic();
  Wi_vec.alloc((size_t)(dim - 2) * sizeof(double));
  tmpw.alloc((size_t)(dim - 2) * sizeof(double));
  ld.alloc(64);
}

// hypoperlog.jl:289-319: central ray of the hypograph-of-perspective-of-sum-log cone (looked up for d <= 10, fitted beyond)
void central_ray_hypoperlog(int d, double* uvw) {
  static const double tab[10][3] = {
      {-0.827838387, 0.805102007, 1.290927686}, {-0.689607388, 0.724605082, 1.224617936}, {-0.584372665, 0.68128058, 1.182421942},
      {-0.503499342, 0.65448622, 1.153053152},  {-0.440285893, 0.636444224, 1.131466926}, {-0.389979809, 0.623569352, 1.114979519},
      {-0.349255921, 0.613978276, 1.102013921}, {-0.315769104, 0.606589839, 1.091577908}, {-0.287837744, 0.600745284, 1.083013},
      {-0.264242734, 0.596019009, 1.075868782}};
  if (d <= 10) {
    for (int k = 0; k < 3; ++k) uvw[k] = tab[d - 1][k];
    return;
  }
  const double x = 1.0 / d;
  if (d <= 70) {
    uvw[0] = 4.657876 * x * x - 3.116192 * x + 0.000647; uvw[1] = 0.424682 * x + 0.553392; uvw[2] = 0.760412 * x + 1.001795;
  } else {
    uvw[0] = -3.011166 * x - 0.000122; uvw[1] = 0.395308 * x + 0.553955; uvw[2] = 0.837545 * x + 1.000024;
  }
}

void HypoPerLogdetTriCone::set_initial_point(double* h) {   // :80-95
  for (int i = 0; i < dim; ++i) h[i] = 0.0;
  double uvw[3];
  central_ray_hypoperlog(d, uvw);
  h[0] = uvw[0]; h[1] = uvw[1];
  long k = 2;
  for (int i = 1; i <= d; ++i) { h[k] = uvw[2]; k += i + 1; }
}

double HypoPerLogdetTriCone::logdet_of(PsdCone& k) {
  hipLaunchKernelGGL(logdet_diag_kernel, dim3(1), dim3(256), 0, ctx.stream, k.side, k.U.d(), (long)k.side, ld.d());
  HYP_CHECK(hipGetLastError());
  ctx.d2h(ctx.h_pinned, ld.p, sizeof(double));
  ctx.sync();
  return ctx.h_pinned[0];
}

bool HypoPerLogdetTriCone::update_feas() {   // :97-118
  ctx.d2h(ctx.h_pinned + 1, point.p, 2 * sizeof(double));
  ctx.sync();
  u = ctx.h_pinned[1];
  v = ctx.h_pinned[2];
  is_feas_ = false;
  if (v > EPS) {
    psd.load_point(point.d() + 2, 1.0);
    psd.reset_data();
    if (psd.is_feas()) {
      phi = logdet_of(psd) - d * log(v);
      zeta = v * phi - u;
      is_feas_ = (zeta > EPS);
    }
  }
  feas_updated = true;
  return is_feas_;
}

bool HypoPerLogdetTriCone::is_dual_feas() {   // :120-131
  ctx.d2h(ctx.h_pinned + 1, dual_point.p, 2 * sizeof(double));
  ctx.sync();
  const double ud = ctx.h_pinned[1], vd = ctx.h_pinned[2];
  if (!(ud < -EPS)) return false;
  psdd.load_point(dual_point.d() + 2, 1.0);
  psdd.reset_data();
  if (!psdd.is_feas()) return false;
  return vd - ud * (logdet_of(psdd) + d * (1.0 - log(-ud))) > EPS;
}

void HypoPerLogdetTriCone::update_grad() {   // :133-150
  const int dw = dim - 2;
  dev_scale_copy(ctx, dw, -1.0, psd.get_grad(), Wi_vec.d());                // svec(W^-1)
  hipLaunchKernelGGL(perlogdet_grad_kernel, dim3((dw + 255) / 256), dim3(256), 0, ctx.stream, dw, 1.0 / zeta, -1.0 / v - (phi - d) / zeta,
                     -1.0 - v / zeta, Wi_vec.d(), grad.d());
  HYP_CHECK(hipGetLastError());
  grad_updated = true;
}

void HypoPerLogdetTriCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :195-236
  HYP_REQUIRE(prod != arr, "HypoPerLogdetTri hess_prod: in-place call");
  if (ncols <= 0) return;
  get_grad();
  const int dw = dim - 2;
  dots.ensure((size_t)ncols * sizeof(double));
  gemv(ctx, true, dw, ncols, 1.0, arr + 2, lda, Wi_vec.d(), 0.0, dots.d());      // tr(U^-T R U^-1)
  psd.hess_prod(prod + 2, ldp, arr + 2, lda, ncols);                              // svec(W^-1 R W^-1)
  hipLaunchKernelGGL(perlogdet_combine_kernel, dim3((dw + 255) / 256, std::min(ncols, 1024)), dim3(256), 0, ctx.stream, dw, ncols, 0,
                     PerLogdetScal{v, (double)d, zeta, phi}, arr, lda, dots.d(), Wi_vec.d(), prod, ldp);
  HYP_CHECK(hipGetLastError());
}

void HypoPerLogdetTriCone::inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :271-316
  HYP_REQUIRE(prod != arr, "HypoPerLogdetTri inv_hess_prod: in-place call");
  if (ncols <= 0) return;
  get_grad();
  const int dw = dim - 2;
  dots.ensure((size_t)ncols * sizeof(double));
  gemv(ctx, true, dw, ncols, 1.0, arr + 2, lda, point.d() + 2, 0.0, dots.d());   // <w, r>
  psd.inv_hess_prod(prod + 2, ldp, arr + 2, lda, ncols);                          // svec(W R W)
  hipLaunchKernelGGL(perlogdet_combine_kernel, dim3((dw + 255) / 256, std::min(ncols, 1024)), dim3(256), 0, ctx.stream, dw, ncols, 1,
                     PerLogdetScal{v, (double)d, zeta, phi}, arr, lda, dots.d(), point.d() + 2, prod, ldp);
  HYP_CHECK(hipGetLastError());
}

void HypoPerLogdetTriCone::update_hess() {   // :152-193
  ensure_hess_storage(false);
  get_grad();
  DBuf eye((size_t)dim * dim * sizeof(double));
  dev_fill_identity(ctx, dim, eye.d(), dim);
  hess_prod(H.d(), dim, eye.d(), dim, dim);
  dev_symmetrize_from_upper(ctx, dim, H.d(), dim, 1, 0);
  ctx.sync();
  hess_updated = true;
}

const double* HypoPerLogdetTriCone::dder3(const double* d_dir) {   // :318-368
  get_grad();
  const int dw = dim - 2;
  const double* r = d_dir + 2;
  ctx.d2h(ctx.h_pinned + 2, d_dir, 2 * sizeof(double));
  psd.hess_prod(tmpw.d(), dw, r, dw, 1);                       // P1 = svec(W^-1 R W^-1)
  const double* D3 = psd.dder3(r);                             // svec(W^-1 R W^-1 R W^-1)
  const double c0 = dot_host(dw, Wi_vec.d(), r);               // tr S
  const double c7 = dot_host(dw, r, tmpw.d());                 // ||S||_F^2
  const double p = ctx.h_pinned[2], q = ctx.h_pinned[3];
  const double sigma = phi - d;
  const double viq = q / v, viq2 = viq * viq, vzi = v / zeta, vzi1 = vzi + 1;
  const double zichi = (-p + sigma * q + c0 * v) / zeta;
  const double c4 = (viq * (-viq * d + 2 * c0) - c7) / zeta / 2;
  const double c1 = (zichi * zichi - v * c4) / zeta;
  const double c3 = -(zichi + viq) / zeta;
  const double c5 = c3 * q + vzi * viq2;
  const double c6 = -2 * vzi * viq - c3 * v;
  const double c8 = c5 + c1 * v;
  hipLaunchKernelGGL(perlogdet_dder3_kernel, dim3((dw + 255) / 256), dim3(256), 0, ctx.stream, dw, -c1,
                     c1 * sigma + (viq2 - (d * c5 + c6 * c0 + vzi * c7)) / v - c4, vzi1, c6, c8, D3, tmpw.d(), Wi_vec.d(), dder3v.d());
  HYP_CHECK(hipGetLastError());
  return dder3v.d();
}

// ---------------------------------------------------------------------------------------------
// WSOSInterpPosSemidefTri (wsosinterppossemideftri.jl:9-321).  The barrier is for the DUAL cone (use_dual_barrier =
// !use_dual, :61): -sum_k logdet Lambda_k with Lambda_k the (L_k R) x (L_k R) matrix of blocks P_k' diag(s_pq) P_k
// (off-diagonal blocks scaled by 1 / sqrt 2).  Lambda_k = U_k' U_k (the reference keeps L_k = U_k');
// FLP_k = L_k^-1 kron(I_R, P_k') by ONE triangular solve with R U right-hand sides (the reference substitutes block by
// block; the result is the same block lower triangular matrix).  Gradient, slow Hessian product and dder3 are diagonals
// of U x U blocks of products with FLP_k; the explicit Hessian combines U x U blocks of FLP_k' FLP_k elementwise.
// ---------------------------------------------------------------------------------------------
// out[b(j, i) U + u] (+)= sign * <m1[:, i U + u], m2[:, j U + u]> * (i == j ? 1 : sqrt 2), i <= j   (:264-286); one wavefront per entry
__global__ __launch_bounds__(256) void wsospsd_bdp_kernel(int rows, int U, int R, const double* __restrict__ m1, long ld1,
                                                          const double* __restrict__ m2, long ld2, double sign, int accumulate,
                                                          double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long total = (long)U * (R * (R + 1) / 2);
  if (e >= total) return;
  const int b = (int)(e / U), u = (int)(e % U);
  int j = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
  while (j * (j + 1) / 2 > b) --j;
  while ((j + 1) * (j + 2) / 2 <= b) ++j;
  const int i = b - j * (j + 1) / 2;
  const double* a = m1 + ((long)i * U + u) * ld1;
  const double* c = m2 + ((long)j * U + u) * ld2;
  double s = 0.0;
  for (int r = lane; r < rows; r += 64) s += a[r] * c[r];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) out[e] = (accumulate ? out[e] : 0.0) + sign * s * (i == j ? 1.0 : 1.4142135623730951);
}
// H[b1 U + u, b2 U + v] (+)= PLiP[p, p2] .* PLiP[q, q2] * scal (+ PLiP[p, q2] .* PLiP[q, p2] when p != q and p2 != q2), b1 <= b2   (:207-232)
__global__ void wsospsd_hess_kernel(int U, int R, const double* __restrict__ G, long ldg, int accumulate, double* __restrict__ H, long ldh) {
  const int nb = R * (R + 1) / 2;
  int pr = blockIdx.z;   // pair index over b1 <= b2
  int b2 = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
  while (b2 * (b2 + 1) / 2 > pr) --b2;
  while ((b2 + 1) * (b2 + 2) / 2 <= pr) ++b2;
  const int b1 = pr - b2 * (b2 + 1) / 2;
  if (b2 >= nb) return;
  auto rc = [](int b, int& row, int& col) {
    row = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
    while (row * (row + 1) / 2 > b) --row;
    while ((row + 1) * (row + 2) / 2 <= b) ++row;
    col = b - row * (row + 1) / 2;
  };
  int p, q, p2, q2;
  rc(b1, p, q);
  rc(b2, p2, q2);
  const int u = blockIdx.x * 16 + (threadIdx.x & 15), v = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (u >= U || v >= U) return;
  auto Gb = [&](int a, int b) { return G[((long)b * U + v) * ldg + (long)a * U + u]; };
  const double scal = ((p == q) != (p2 == q2)) ? 1.4142135623730951 : 1.0;
  double val = Gb(p, p2) * Gb(q, q2) * scal;
  if (p != q && p2 != q2) val += Gb(p, q2) * Gb(q, p2);
  double* h = H + ((long)b2 * U + v) * ldh + (long)b1 * U + u;
  *h = (accumulate ? *h : 0.0) + val;
}

WsosPsdCone::WsosPsdCone(Ctx& c, int R_, int U_, int K_, const int* Ls_, const double* const* hPs, bool use_dual)
    : GenericHessCone(c, CONE_WSOSPSD) {
  HYP_REQUIRE(R_ >= 1 && U_ >= 1 && K_ >= 1, "WSOSInterpPosSemidefTri: sizes");
  R = R_; U = U_; K = K_;
  nblk = R * (R + 1) / 2;
  dim = U * nblk;
  use_dual_barrier = !use_dual;                                                                       // :61
  nu = 0;
  for (int k = 0; k < K; ++k) {
    HYP_REQUIRE(Ls_[k] >= 1 && Ls_[k] <= U, "WSOSInterpPosSemidefTri: 1 <= L_k <= U");
    Ls.push_back(Ls_[k]);
    nu += (double)R * Ls_[k];                                                                         // :66
  }
  alloc_common();
  alloc_generic();
  infos.alloc(64);
  tU.alloc((size_t)U * sizeof(double));
  PLiP.alloc((size_t)R * U * R * U * sizeof(double));
  for (int k = 0; k < K; ++k) {
    const long Lk = Ls[k], LR = Lk * R;
    const size_t pb = (size_t)U * Lk * sizeof(double);
    P.emplace_back(pb); SP.emplace_back(pb);
    Lam.emplace_back((size_t)LR * LR * sizeof(double));
    Mk.emplace_back((size_t)LR * LR * sizeof(double));
    Tk.emplace_back((size_t)LR * LR * sizeof(double));
    LamDinv.emplace_back(dinv_elems((int)LR) * sizeof(double));
    FLP.emplace_back((size_t)LR * R * U * sizeof(double));
    LRUR.emplace_back((size_t)LR * R * U * sizeof(double));
    ctx.h2d(P[k].p, hPs[k], pb);
  }
  ctx.sync();
}

void WsosPsdCone::set_initial_point(double* h) {   // :100-108
  for (int i = 0; i < dim; ++i) h[i] = 0.0;
  for (int r = 0; r < R; ++r) {
    const long b = (long)r * (r + 1) / 2 + r;
    for (int u = 0; u < U; ++u) h[b * U + u] = 1.0;
  }
}

// M (L R x L R, ld L R): upper blocks (q, p), q <= p, = P_k' diag(vec_pq * (p == q ? 1 : 1 / sqrt 2)) P_k
void WsosPsdCone::block_matrix(int k, const double* d_vec, double* M) {
  const int Lk = Ls[k];
  const long LR = (long)Lk * R;
  for (int p = 0; p < R; ++p)
    for (int q = 0; q <= p; ++q) {
      const long b = (long)p * (p + 1) / 2 + q;
      dev_scale_copy(ctx, U, p == q ? 1.0 : 0.7071067811865476, d_vec + b * U, tU.d());
      row_scale(ctx, U, Lk, tU.d(), P[k].d(), U, SP[k].d(), U);
      GemmArgs g{};
      g.M = Lk; g.N = Lk; g.K = U; g.A = SP[k].d(); g.lda = U; g.B = P[k].d(); g.ldb = U;
      g.C = M + (long)q * Lk + (long)p * Lk * LR; g.ldc = LR;
      g.alpha = 1; g.beta = 0; g.tri = (p == q) ? GEMM_UPPER : GEMM_FULL; g.batch = 1;
      gemm(ctx, true, g);
    }
}

bool WsosPsdCone::update_feas() {   // :110-140
  is_feas_ = true;
  for (int k = 0; k < K && is_feas_; ++k) {
    const int LR = Ls[k] * R;
    block_matrix(k, point.d(), Lam[k].d());
    potrf_upper_batched(ctx, LR, Lam[k].d(), LR, 0, 1, LamDinv[k].d(), infos.i());
    if (read_info(ctx, infos.i()) != 0) is_feas_ = false;
  }
  feas_updated = true;
  return is_feas_;
}

void WsosPsdCone::update_grad() {   // :142-186
  for (int k = 0; k < K; ++k) {
    const int Lk = Ls[k];
    const long LR = (long)Lk * R, RU = (long)R * U;
    // kron(I_R, P_k'): block (r, r) = P_k' (L x U)
    ctx.zero(FLP[k].p, (size_t)LR * RU * sizeof(double));
    for (int r = 0; r < R; ++r) dev_transpose(ctx, U, Lk, P[k].d(), U, FLP[k].d() + (long)r * Lk + (long)r * U * LR, LR, 1, 0, 0);
    trsm_work.ensure((size_t)NB * RU * sizeof(double));
    trsm_upper_left(ctx, (int)LR, (int)RU, Lam[k].d(), LR, LamDinv[k].d(), true, FLP[k].d(), LR, trsm_work.d());   // U'^-1 (.)
    const long total = (long)U * nblk;
    hipLaunchKernelGGL(wsospsd_bdp_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, ctx.stream, (int)LR, U, R, FLP[k].d(), LR, FLP[k].d(), LR,
                       -1.0, k > 0 ? 1 : 0, grad.d());
    HYP_CHECK(hipGetLastError());
  }
  grad_updated = true;
}

void WsosPsdCone::update_hess() {   // :188-236
  ensure_hess_storage(false);
  get_grad();
  const long RU = (long)R * U;
  const int npairs = nblk * (nblk + 1) / 2;
  for (int k = 0; k < K; ++k) {
    const long LR = (long)Ls[k] * R;
    GemmArgs g{};   // P Lambda^-1 P = FLP' FLP (R U x R U)
    g.M = (int)RU; g.N = (int)RU; g.K = (int)LR; g.A = FLP[k].d(); g.lda = LR; g.B = FLP[k].d(); g.ldb = LR; g.C = PLiP.d(); g.ldc = RU;
    g.alpha = 1; g.beta = 0; g.tri = GEMM_UPPER; g.batch = 1;
    gemm(ctx, true, g);
    dev_symmetrize_from_upper(ctx, (int)RU, PLiP.d(), RU, 1, 0);
    hipLaunchKernelGGL(wsospsd_hess_kernel, dim3((U + 15) / 16, (U + 15) / 16, npairs), dim3(256), 0, ctx.stream, U, R, PLiP.d(), RU, k > 0 ? 1 : 0,
                       H.d(), (long)dim);
    HYP_CHECK(hipGetLastError());
  }
  dev_symmetrize_from_upper(ctx, dim, H.d(), dim, 1, 0);
  hess_updated = true;
}

void WsosPsdCone::partial_prod(double* prod, long ldp, const double* arr, long lda, int ncols, bool use_symm_prod) {   // :288-321
  get_grad();
  const long RU = (long)R * U;
  const long total = (long)U * nblk;
  for (int j = 0; j < ncols; ++j) {
    for (int k = 0; k < K; ++k) {
      const int LR = Ls[k] * R;
      double* M = Mk[k].d();
      double* T = Tk[k].d();
      block_matrix(k, arr + (long)j * lda, M);
      dev_symmetrize_from_upper(ctx, LR, M, LR, 1, 0);
      // W = L^-1 M L^-T = U^-T M U^-1: left solve, transpose, left solve (W is symmetric)
      trsm_work.ensure((size_t)NB * std::max<long>(LR, RU) * sizeof(double));
      trsm_upper_left(ctx, LR, LR, Lam[k].d(), LR, LamDinv[k].d(), true, M, LR, trsm_work.d());
      dev_transpose(ctx, LR, LR, M, LR, T, LR, 1, 0, 0);
      trsm_upper_left(ctx, LR, LR, Lam[k].d(), LR, LamDinv[k].d(), true, T, LR, trsm_work.d());
      dev_symmetrize_from_upper(ctx, LR, T, LR, 1, 0);                                    // Symmetric(., :U)
      GemmArgs g{};   // LRUR = W FLP
      g.M = LR; g.N = (int)RU; g.K = LR; g.A = T; g.lda = LR; g.B = FLP[k].d(); g.ldb = LR; g.C = LRUR[k].d(); g.ldc = LR;
      g.alpha = 1; g.beta = 0; g.batch = 1;
      gemm(ctx, true, g);
      hipLaunchKernelGGL(wsospsd_bdp_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, ctx.stream, LR, U, R,
                         use_symm_prod ? LRUR[k].d() : FLP[k].d(), (long)LR, LRUR[k].d(), (long)LR, 1.0, k > 0 ? 1 : 0, prod + (long)j * ldp);
      HYP_CHECK(hipGetLastError());
    }
  }
}

void WsosPsdCone::hess_prod_slow(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :238-247
  if (!use_hess_prod_slow_updated) update_use_hess_prod_slow();
  if (!use_hess_prod_slow) {
    hess_prod(prod, ldp, arr, lda, ncols);
    return;
  }
  partial_prod(prod, ldp, arr, lda, ncols, false);
}

const double* WsosPsdCone::dder3(const double* d_dir) {   // :249-252
  partial_prod(dder3v.d(), dim, d_dir, dim, 1, true);
  return dder3v.d();
}

}  // namespace hyp
