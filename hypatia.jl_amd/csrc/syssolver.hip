// QRChol system solver on the device: Schur-complement assembly (batched sqrt-Hessian products +
// FP64-MFMA syrk), blocked Cholesky, and the 3x3 solve.
// Reference: /root/reference/src/Solvers/systemsolvers/qrchol.jl (line ranges inline).
#include <cstring>
#include "syssolver.hpp"
#include <chrono>

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
  make_psd_runs();
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

// Runs of equal PSD cones: one arena per run, the members' matrices become views into it (see PsdCone::group_arena).
SysSolver::~SysSolver() {
  if (const char* e = getenv("HYP_RP_PREFETCH_STATS"))
    if (e[0] == '1') fprintf(stderr, "[residual_products prefetch] handed out %ld, recomputed %ld\n", rp_pre_hits, rp_pre_misses);
  if (const char* e = getenv("HYP_SHP_PRELAUNCH_STATS"))
    if (e[0] == '1') fprintf(stderr, "[sqrt_hess_prod prelaunch] used %ld, not used %ld\n", shp_used, shp_unused);
  if (const char* e = getenv("HYP_CHOL_SPLIT_STATS"))
    if (e[0] == '1') fprintf(stderr, "[chol split] %ld factorizations in two column groups\n", chol_split_count);
  if (rp_pre_host) (void)hipHostFree(rp_pre_host);
  for (hipEvent_t e : {rp_pre_ev, plan_ev_fork, plan_ev_done, dirs_copied_ev, up_ev0, up_ev1, split_ev_ready, split_ev_done, ov_ev_fork})
    if (e) (void)hipEventDestroy(e);
}

void SysSolver::make_psd_runs() {
  static const bool on = [] { const char* e = getenv("HYP_PSD_GROUP"); return !(e && e[0] == '0'); }();
  if (!on) return;
  size_t k = 0;
  while (k < cones.size()) {
    PsdCone* pk = (cones[k]->kind == CONE_PSD) ? static_cast<PsdCone*>(cones[k]) : nullptr;
    size_t e = k + 1;
    if (pk)
      while (e < cones.size() && cones[e]->kind == CONE_PSD && static_cast<PsdCone*>(cones[e])->side == pk->side) ++e;
    const int cnt = (int)(e - k);
    if (pk && cnt >= 4) {
      const long s2 = (long)pk->side * pk->side;
      const long dv = 2 * (long)dinv_elems(pk->side);
      const long dm = pk->dim;
      auto arena = std::make_shared<DBuf>((size_t)cnt * (6 * s2 + dv + 2 * dm) * sizeof(double));
      ctx.zero(arena->p, arena->bytes);
      double* base = arena->d();
      PsdRun r{(int)k, cnt, pk->side, base, base + cnt * s2, base + 2 * cnt * s2, base + 3 * cnt * s2, base + 4 * cnt * s2, base + 5 * cnt * s2,
               base + 6 * cnt * s2, base + 6 * cnt * s2 + cnt * dv, base + 6 * cnt * s2 + cnt * dv + cnt * dm};
      for (int g = 0; g < cnt; ++g) {   // the members' matrices move into the arena with their contents (cached factor, inverses)
        PsdCone* c = static_cast<PsdCone*>(cones[k + g]);
        auto move_in = [&](DBuf& b, double* dst, long count) {
          ctx.d2d(dst, b.p, (size_t)count * sizeof(double));
          ctx.sync();   // (the old allocation is released right below)
          b.view(dst, (size_t)count * sizeof(double));
        };
        move_in(c->X, r.X + g * s2, s2);
        move_in(c->U, r.U + g * s2, s2);
        move_in(c->UT, r.UT + g * s2, s2);
        move_in(c->Uinv, r.Uinv + g * s2, s2);
        move_in(c->UinvT, r.UinvT + g * s2, s2);
        move_in(c->Xinv, r.Xinv + g * s2, s2);
        move_in(c->dinvb, r.dinvb + g * dv, dv);
        move_in(c->point, r.point + g * dm, dm);
        move_in(c->dual_point, r.dual + g * dm, dm);
        c->group_arena = arena;   // (a previous arena, if any, is released with its last member)
      }
      psd_runs.push_back(r);
    }
    k = e;
  }
}

// PsdCone::ensure_inverses (possemideftri.jl:97-107, 126-177 need U^-1, X^-1) for whole runs at once: the same seven launches
// as for one cone, with batch = members.  Runs with an infeasible, not yet factored or already inverted member are left
// to the per-cone path.
void SysSolver::group_inverses() {
  for (const PsdRun& r : psd_runs) {
    bool all = true;
    for (int g = 0; g < r.count && all; ++g) {
      PsdCone* c = static_cast<PsdCone*>(cones[r.k0 + g]);
      all = c->feas_updated && c->is_feas_ && !c->inv_ready && c->U.p == (void*)(r.U + (long)g * r.side * r.side);
    }
    if (!all) continue;
    const int s = r.side, B = r.count;
    const long s2 = (long)s * s, dv = 2 * (long)dinv_elems(s);
    dev_zero_strict_lower(ctx, s, r.U, s, B, s2);
    potrf_invert_diag_blocks(ctx, s, r.U, s, s2, B, r.dinvb, dv);
    trtri_upper_batched(ctx, s, r.U, s, s2, r.dinvb, dv, r.Uinv, s, s2, B);
    dev_transpose(ctx, s, s, r.Uinv, s, r.UinvT, s, B, s2, s2);
    dev_transpose(ctx, s, s, r.U, s, r.UT, s, B, s2, s2);
    GemmArgs g{};   // Xinv = Uinv Uinv'
    g.M = s; g.N = s; g.K = s; g.A = r.Uinv; g.lda = s; g.strideA = s2; g.B = r.UinvT; g.ldb = s; g.strideB = s2;
    g.C = r.Xinv; g.ldc = s; g.strideC = s2; g.alpha = 1; g.beta = 0; g.tri = GEMM_FULL; g.krange = KR_GE_M; g.batch = B;
    gemm(ctx, false, g);
    for (int gi = 0; gi < B; ++gi) static_cast<PsdCone*>(cones[r.k0 + gi])->inv_ready = true;
  }
}

// out[3 g + off] = <a_g, b_g> over the members' consecutive segments of length len; one wavefront per member
__global__ __launch_bounds__(256) void seg_dot3_kernel(int B, int len, const double* __restrict__ a, const double* __restrict__ b, int off,
                                                       double* __restrict__ out, int stride = 3) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= B) return;
  const double* x = a + (long)g * len;
  const double* y = b + (long)g * len;
  double s0 = 0.0, s1 = 0.0;
  int i = lane;
  for (; i + 64 < len; i += 128) { s0 += x[i] * y[i]; s1 += x[i + 64] * y[i + 64]; }
  if (i < len) s0 += x[i] * y[i];
  double s = s0 + s1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (lane == 0) out[stride * g + off] = s;
}
void seg_dots(Ctx& c, int B, int len, const double* a, const double* b, int off, int stride, double* out) {
  hipLaunchKernelGGL(seg_dot3_kernel, dim3((B + 3) / 4), dim3(256), 0, c.stream, B, len, a, b, off, out, stride);
}

void SysSolver::run_prox_launch(const PsdRun& r, double irtmu, double* d_out) {
  const int s = r.side, B = r.count, dm = cones[r.k0]->dim;
  const long s2 = (long)s * s, tot = (long)B * dm;
  for (DBuf* b : {&run_ws1, &run_ws2}) b->ensure((size_t)B * s2 * sizeof(double));
  for (DBuf* b : {&run_g, &run_v, &run_h}) b->ensure((size_t)tot * sizeof(double));
  double* W1 = run_ws1.d();
  double* W2 = run_ws2.d();
  auto xvx = [&](const double* v, double* out) {   // PsdCone::inv_hess_prod (possemideftri.jl:144-159) for every member: svec(X smat(v) X)
    svec_unpack(ctx, s, B, v, dm, W1);
    GemmArgs a{};
    a.M = s; a.N = s; a.K = s; a.A = W1; a.lda = s; a.strideA = s2; a.B = r.X; a.ldb = s; a.strideB = s2; a.C = W2; a.ldc = s; a.strideC = s2;
    a.alpha = 1; a.beta = 0; a.tri = GEMM_FULL; a.krange = KR_ALL; a.batch = B;
    gemm(ctx, true, a);
    GemmArgs b{};
    b.M = s; b.N = s; b.K = s; b.A = r.X; b.lda = s; b.strideA = s2; b.B = W2; b.ldb = s; b.strideB = s2; b.C = W1; b.ldc = s; b.strideC = s2;
    b.alpha = 1; b.beta = 0; b.tri = GEMM_FULL; b.krange = KR_ALL; b.batch = B;
    gemm(ctx, true, b);
    svec_pack(ctx, s, B, W1, out, dm, 1.0);
  };
  const dim3 grid((B + 3) / 4), blk(256);
  svec_pack(ctx, s, B, r.Xinv, run_g.d(), dm, -1.0);                                   // g = -svec(X^-1)  (possemideftri.jl:97-107)
  hipLaunchKernelGGL(seg_dot3_kernel, grid, blk, 0, ctx.stream, B, dm, run_g.d(), r.point, 0, d_out);      // <g, point>
  xvx(run_g.d(), run_h.d());
  hipLaunchKernelGGL(seg_dot3_kernel, grid, blk, 0, ctx.stream, B, dm, run_h.d(), run_g.d(), 1, d_out);    // <H^-1 g, g>
  ctx.d2d(run_v.p, run_g.p, (size_t)tot * sizeof(double));                             // v = irtmu dual + g
  dev_axpby(ctx, (int)tot, irtmu, r.dual, 1.0, run_v.d());
  xvx(run_v.d(), run_h.d());
  hipLaunchKernelGGL(seg_dot3_kernel, grid, blk, 0, ctx.stream, B, dm, run_h.d(), run_v.d(), 2, d_out);    // <H^-1 v, v>
  HYP_CHECK(hipGetLastError());
}

const SysSolver::PsdRun* SysSolver::whole_model_run() {
  if (psd_runs.size() != 1 || psd_runs[0].k0 != 0 || (size_t)psd_runs[0].count != cones.size()) return nullptr;
  const PsdRun& r = psd_runs[0];
  for (size_t k = 0; k < cones.size(); ++k) {
    PsdCone* c = static_cast<PsdCone*>(cones[k]);
    if (c->use_dual_barrier || c->U.p != (void*)(r.U + (long)k * r.side * r.side) || c->point.p != (void*)(r.point + (long)k * c->dim)) return nullptr;
  }
  return &r;
}

void SysSolver::run_grad(const PsdRun& r, double* d_out) {
  group_inverses();
  for (int g = 0; g < r.count; ++g) static_cast<PsdCone*>(cones[r.k0 + g])->ensure_inverses();
  svec_pack(ctx, r.side, r.count, r.Xinv, d_out, cones[r.k0]->dim, -1.0);
}

void SysSolver::run_dder3(const PsdRun& r, const double* d_dir, double* d_out) {   // PsdCone::dder3 with batch = members
  group_inverses();
  const int s = r.side, B = r.count, dm = cones[r.k0]->dim;
  const long s2 = (long)s * s;
  for (int g = 0; g < B; ++g) static_cast<PsdCone*>(cones[r.k0 + g])->ensure_inverses();
  for (DBuf* b : {&run_ws1, &run_ws2, &run_ws3}) b->ensure((size_t)B * s2 * sizeof(double));
  double* W1 = run_ws1.d();
  double* W2 = run_ws2.d();
  double* W3 = run_ws3.d();
  auto bgemm = [&](const double* A, const double* Bm, double* C, int kr) {   // C_g = A_g' B_g
    GemmArgs a{};
    a.M = s; a.N = s; a.K = s; a.A = A; a.lda = s; a.strideA = s2; a.B = Bm; a.ldb = s; a.strideB = s2; a.C = C; a.ldc = s; a.strideC = s2;
    a.alpha = 1; a.beta = 0; a.tri = GEMM_FULL; a.krange = kr; a.batch = B;
    gemm(ctx, true, a);
  };
  svec_unpack(ctx, s, B, d_dir, dm, W1);
  bgemm(W1, r.Uinv, W2, KR_LE_N);        // P = U^-T D U^-1
  bgemm(r.Uinv, W2, W1, KR_LE_M);
  bgemm(W1, W1, W3, KR_ALL);             // Q = P' P
  bgemm(W3, r.UinvT, W2, KR_GE_N);       // U^-1 Q U^-T
  bgemm(r.UinvT, W2, W1, KR_GE_M);
  svec_pack(ctx, s, B, W1, d_out, dm, 1.0);
}

int SysSolver::run_hess_prod(size_t k, double* prod, long ldp, const double* arr, long lda, int ncols) {
  if (psd_runs.empty() || ncols < 1 || ncols > 3) return 0;
  for (const PsdRun& r : psd_runs) {
    if ((size_t)r.k0 != k) continue;
    const int s = r.side, B = r.count, dimk = cones[k]->dim;
    const long s2 = (long)s * s;
    group_inverses();
    for (int g = 0; g < B; ++g) {
      PsdCone* c = static_cast<PsdCone*>(cones[k + g]);
      if (c->use_dual_barrier || c->U.p != (void*)(r.U + g * s2)) return 0;
      c->ensure_inverses();   // (no-op after the batched pass; covers a member that was handled alone)
    }
    run_ws1.ensure((size_t)B * s2 * sizeof(double));
    run_ws2.ensure((size_t)B * s2 * sizeof(double));
    double* W1 = run_ws1.d();
    double* W2 = run_ws2.d();
    auto two_sided = [&](const double* R, int kr2, int kr3) {   // W1_g <- R_g' W1_g R_g for every member g
      GemmArgs a{};
      a.M = s; a.N = s; a.K = s; a.A = W1; a.lda = s; a.strideA = s2; a.B = R; a.ldb = s; a.strideB = s2; a.C = W2; a.ldc = s; a.strideC = s2;
      a.alpha = 1; a.beta = 0; a.tri = GEMM_FULL; a.krange = kr2; a.batch = B;
      gemm(ctx, true, a);
      GemmArgs b{};
      b.M = s; b.N = s; b.K = s; b.A = R; b.lda = s; b.strideA = s2; b.B = W2; b.ldb = s; b.strideB = s2; b.C = W1; b.ldc = s; b.strideC = s2;
      b.alpha = 1; b.beta = 0; b.tri = GEMM_FULL; b.krange = kr3; b.batch = B;
      gemm(ctx, true, b);
    };
    for (int col = 0; col < ncols; ++col) {
      svec_unpack(ctx, s, B, arr + (long)col * lda, dimk, W1);        // the members' slices are consecutive columns of length dim
      two_sided(r.Uinv, KR_LE_N, KR_LE_M);                            // U^-T V U^-1   (possemideftri.jl:126-142, as PsdCone::hess_prod)
      two_sided(r.UinvT, KR_GE_N, KR_GE_M);                           // U^-1 (.) U^-T
      svec_pack(ctx, s, B, W1, prod + (long)col * ldp, dimk, 1.0);
    }
    return B;
  }
  return 0;
}

void SysSolver::load(const double* hG, const double* hGQ1, const double* hGQ2, const double* hQ, const double* hR) {
  const size_t d = sizeof(double);
  s_resident = false;   // (resident directions belong to the G they were computed with)
  shp_prelaunched = false; shp_wasted = 0;   // (so do prelaunched cone products)
  rp_pre_valid = false;
  ctx.h2d(G.p, hG, (size_t)q * n * d);
  if (p > 0) {
    HYP_REQUIRE(hQ && hR, "sys: Q, R are required when p > 0");
    HYP_REQUIRE((hGQ1 == nullptr) == (hGQ2 == nullptr), "sys: pass both GQ1 and GQ2, or neither");
    ctx.h2d(Qm.p, hQ, (size_t)n * n * d);
    if (hGQ1) {
      ctx.h2d(GQ1.p, hGQ1, (size_t)q * p * d);
      ctx.h2d(GQ2s.p, hGQ2, (size_t)q * nmp * d);
    } else {   // GQ = G * Ap_Q on the device (qrchol.jl:154), split into its first p and last n - p columns
      GemmArgs g1{};
      g1.M = q; g1.N = p; g1.K = n; g1.A = G.d(); g1.lda = q; g1.B = Qm.d(); g1.ldb = n; g1.C = GQ1.d(); g1.ldc = q;
      g1.alpha = 1; g1.beta = 0; g1.batch = 1;
      gemm(ctx, false, g1);
      if (nmp > 0) {
        GemmArgs g2 = g1;
        g2.N = nmp; g2.B = Qm.d() + (long)p * n; g2.C = GQ2s.d();
        gemm(ctx, false, g2);
      }
    }
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

// rccl_comm.hip
void rccl_allreduce_inplace(void* comm, double* d_buf, long count, int op, hipStream_t st);

// upper triangle of an n x n column-major matrix <-> packed columns (column j at j (j + 1) / 2): what the ranks exchange
__global__ void tri_pack_kernel(int n, const double* __restrict__ A, long lda, double* __restrict__ P, int unpack) {
  const int j = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > j) return;
  if (unpack) const_cast<double*>(A)[(long)j * lda + i] = P[(long)j * (j + 1) / 2 + i];
  else P[(long)j * (j + 1) / 2 + i] = A[(long)j * lda + i];
}

hipEvent_t SysSolver::comm_event() {
  if (!comm_ev_free.empty()) { hipEvent_t e = comm_ev_free.back(); comm_ev_free.pop_back(); return e; }
  hipEvent_t e;
  HYP_CHECK(hipEventCreate(&e));
  return e;
}
void SysSolver::comm_time_begin(int site, hipEvent_t* a, hipStream_t st) {
  (void)site;
  if (comm_ev_pending.size() >= 2048) comm_times_flush();   // (bounded: a flush waits for the newest pending pair)
  *a = comm_event();
  HYP_CHECK(hipEventRecord(*a, st ? st : ctx.stream));
}
void SysSolver::comm_time_end(int site, hipEvent_t a, hipStream_t st) {
  hipEvent_t b = comm_event();
  HYP_CHECK(hipEventRecord(b, st ? st : ctx.stream));
  comm_ev_pending.push_back(CommEv{a, b, site & 15});
}

// rows [r0, r1) of the upper triangle <-> a contiguous segment: column j >= r0 contributes rows r0 .. min(r1 - 1, j)
__global__ void tri_pack_rows_kernel(int n, int r0, int r1, const double* __restrict__ A, long lda, double* __restrict__ P, int unpack) {
  const int j = r0 + blockIdx.y;
  const int i = r0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n || i >= r1 || i > j) return;
  const long h = r1 - r0, dj = j - r0;
  const long off = (j < r1) ? dj * (dj + 1) / 2 : h * (h + 1) / 2 + (dj - h) * h;
  if (unpack) const_cast<double*>(A)[(long)j * lda + i] = P[off + (i - r0)];
  else P[off + (i - r0)] = A[(long)j * lda + i];
}

bool SysSolver::assemble_lhs_overlapped(long kr0, long kr1, int groups) {
  // (every condition here is the same on every rank: the ranks must issue the same collectives -- ADVICE r05; a rank with few or no
  //  rows of its own still takes part, its share of a group is a zero block)
  if (!rccl_comm || groups < 2 || nmp < 1024) return false;
  const int T = (nmp + 127) / 128;
  groups = std::min(groups, T);
  // tile-row boundaries of groups of (nearly) equal area; tile row i has T - i tiles
  const long total = (long)T * (T + 1) / 2;
  std::vector<int> bound(1, 0);
  long acc = 0;
  for (int i = 0, g = 1; i < T && g < groups; ++i) {
    acc += T - i;
    if (acc * groups >= total * g) { bound.push_back(i + 1); ++g; }
  }
  if (bound.back() != T) bound.push_back(T);
  const int ng = (int)bound.size() - 1;
  const long cnt = (long)nmp * (nmp + 1) / 2;
  ov_tri.ensure((size_t)cnt * sizeof(double));
  while ((int)ov_events.size() < 2 * ng) { hipEvent_t e; HYP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ov_events.push_back(e); }
  hipEvent_t ev_whole = nullptr;
  comm_time_begin(14, &ev_whole);
  long seg = 0;
  // (round 6) the products of consecutive row groups run on TWO lanes in turn, each with its own split-K scratch: a launch ends in a
  // round of workgroups that drains for milliseconds (a K slice of a tile lives 6 ms at config 4) -- one after the other the four
  // products took 92.6 ms where the one-piece product takes 80 (profiles/r06_dist_overlap_timeline.txt); side by side the next
  // group's workgroups fill the compute units the previous one frees.  The thin last columns (n mod 128) come first, from the skinny
  // kernel of the one-piece product, instead of as a last tile column that is 6 % of every group's tiles.  HYP_DIST_OVERLAP_OLD=1:
  // one queue, the round-5 slice counts, no tile order.
  static const bool old_launch = [] { const char* e = getenv("HYP_DIST_OVERLAP_OLD"); return e && atoi(e) == 1; }();
  int N0 = nmp;
  if (!old_launch && kr1 > kr0) {
    GemmArgs w{};
    w.M = nmp; w.N = nmp; w.K = (int)(kr1 - kr0); w.A = HGQ2.d() + kr0; w.lda = q; w.B = w.A; w.ldb = q; w.C = lhs.d(); w.ldc = nmp;
    w.alpha = 1; w.beta = 0; w.tri = GEMM_UPPER; w.krange = KR_ALL; w.batch = 1; w.tag = 1;
    HYP_CHECK(schur_syrk_edge(ctx.stream, w, &N0));
  }
  if (!ov_ev_fork) HYP_CHECK(hipEventCreateWithFlags(&ov_ev_fork, hipEventDisableTiming));
  HYP_CHECK(hipEventRecord(ov_ev_fork, ctx.stream));
  for (int g = 0; g < ng; ++g) {
    const int r0 = bound[g] * 128, r1 = std::min(nmp, bound[g + 1] * 128);
    const int rows = r1 - r0, cols = nmp - r0;
    long nblk = 0;
    for (int i = bound[g]; i < bound[g + 1]; ++i) nblk += T - i;
    GemmArgs s{};   // lhs[r0:r1, r0:] = HGQ2[K range, r0:r1]' HGQ2[K range, r0:]  (upper trapezoid)
    s.M = rows; s.N = cols; s.K = (int)(kr1 - kr0);
    s.A = HGQ2.d() + kr0 + (long)r0 * q; s.lda = q;
    s.B = s.A; s.ldb = q;
    s.C = lhs.d() + (long)r0 * nmp + r0; s.ldc = nmp;
    s.alpha = 1; s.beta = 0; s.tri = GEMM_UPPER_RECT; s.krange = KR_ALL; s.batch = 1; s.tile_hint = 128;
    if (old_launch) {   // K slices so that the group is about two rounds of 512 resident workgroups (for 5 slices of 205 - 225 tiles: three)
      int S = (int)std::max<long>(1, std::min<long>(8, (1024 + nblk / 2) / std::max<long>(nblk, 1)));
      while (S > 1 && s.K / S < 1024) --S;
      s.splitk_req = S;
      if (s.K > 0) gemm(ctx, true, s);
      else HYP_CHECK(hipMemset2DAsync(s.C, (size_t)nmp * sizeof(double), 0, (size_t)rows * sizeof(double), (size_t)cols, ctx.stream));
      HYP_CHECK(hipEventRecord(ov_events[2 * g], ctx.stream));
    } else {
      // an instance of the Schur product (tag 1): only the tiles that do work, in the XCD-aware order, slice count and cut last round
      // chosen as for the one-piece product (gemm_f64_kernel.hpp: trap_tile_map); the edge columns are done
      s.tag = 1;
      s.M = std::min(r1, N0) - r0; s.N = N0 - r0;
      LaneSwitch on_lane(ctx, 2 + (g & 1));
      HYP_CHECK(hipStreamWaitEvent(ctx.stream, ov_ev_fork, 0));
      if (s.K > 0 && s.M > 0 && s.N > 0) gemm(ctx, true, s);
      else if (s.K <= 0) HYP_CHECK(hipMemset2DAsync(lhs.d() + (long)r0 * nmp + r0, (size_t)nmp * sizeof(double), 0, (size_t)rows * sizeof(double), (size_t)cols, ctx.stream));
      HYP_CHECK(hipEventRecord(ov_events[2 * g], ctx.stream));
    }
    HYP_CHECK(hipStreamWaitEvent(ctx.stream2, ov_events[2 * g], 0));
    const long h = rows;
    const long segcnt = h * (h + 1) / 2 + (long)(cols - rows) * h;
    double* P = ov_tri.d() + seg;
    const dim3 grid((rows + 255) / 256, cols);
    hipLaunchKernelGGL(tri_pack_rows_kernel, grid, dim3(256), 0, ctx.stream2, nmp, r0, r1, lhs.d(), (long)nmp, P, 0);
    comm_calls += 1;
    comm_doubles += (double)segcnt;
    comm_hist[0] += 1;
    hipEvent_t ea = nullptr;
    comm_time_begin(0, &ea, ctx.stream2);
    rccl_allreduce_inplace(rccl_comm, P, segcnt, 0, ctx.stream2);
    comm_time_end(0, ea, ctx.stream2);
    hipLaunchKernelGGL(tri_pack_rows_kernel, grid, dim3(256), 0, ctx.stream2, nmp, r0, r1, lhs.d(), (long)nmp, P, 1);
    HYP_CHECK(hipEventRecord(ov_events[2 * g + 1], ctx.stream2));
    seg += segcnt;
  }
  HYP_CHECK(hipGetLastError());
  for (int g = 0; g < ng; ++g) HYP_CHECK(hipStreamWaitEvent(ctx.stream, ov_events[2 * g + 1], 0));
  comm_time_end(14, ev_whole);
  return true;
}
void SysSolver::comm_times_flush() {
  for (const CommEv& e : comm_ev_pending) {
    HYP_CHECK(hipEventSynchronize(e.b));
    float ms = 0;
    HYP_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
    comm_ms[e.site] += ms;
    comm_ev_free.push_back(e.a);
    comm_ev_free.push_back(e.b);
  }
  comm_ev_pending.clear();
}

void SysSolver::allreduce_lhs() {
  const bool have = (comm_fn != nullptr || rccl_comm != nullptr);
  if (!have || (ks_world <= 1 && !dist())) return;
  const int kw = ks_world;
  ks_world = 1;   // (allreduce_dev is the cone-sharded mode's entry point: borrow it)
  hipEvent_t ev_whole = nullptr;
  comm_time_begin(14, &ev_whole);
  try {
    // only the upper triangle is meaningful (syrk 'U'): the ranks exchange its n (n + 1) / 2 entries, not the n^2 of the
    // square buffer -- half the bytes over xGMI for two passes over the matrix in HBM
    const long cnt = (long)nmp * (nmp + 1) / 2;
    lhs_tri.ensure((size_t)cnt * sizeof(double));
    const dim3 grid((nmp + 255) / 256, nmp);
    hipLaunchKernelGGL(tri_pack_kernel, grid, dim3(256), 0, ctx.stream, nmp, lhs.d(), (long)nmp, lhs_tri.d(), 0);
    allreduce_dev(lhs_tri.d(), cnt, 0, 0);
    hipLaunchKernelGGL(tri_pack_kernel, grid, dim3(256), 0, ctx.stream, nmp, lhs.d(), (long)nmp, lhs_tri.d(), 1);
    HYP_CHECK(hipGetLastError());
  } catch (...) {
    ks_world = kw;
    throw;
  }
  ks_world = kw;
  comm_time_end(14, ev_whole);
}

void SysSolver::allreduce_dev(double* d_buf, long count, int op, int site) {
  if (!dist() || count <= 0) return;
  comm_calls += 1;
  comm_doubles += (double)count;
  comm_hist[site & 15] += 1;
  if (rccl_comm) {   // in place, on the library stream: the consumers of d_buf are queued behind it
    hipEvent_t ea = nullptr;
    comm_time_begin(site, &ea);
    rccl_allreduce_inplace(rccl_comm, d_buf, count, op, ctx.stream);
    comm_time_end(site, ea);
    return;
  }
  HYP_REQUIRE(count <= comm_cap, "sys: all-reduce payload exceeds the registered staging buffer");
  ctx.d2d(comm_stage, d_buf, (size_t)count * sizeof(double));
  ctx.sync();
  const auto t0 = std::chrono::steady_clock::now();
  HYP_REQUIRE(comm_fn(comm_user, count, op) == 0, "sys: all-reduce callback failed");
  comm_ms[site & 15] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx.d2d(d_buf, comm_stage, (size_t)count * sizeof(double));
}
// the tail of a fused exchange: sums in place, this rank's maxima in its slots, zeros in the other ranks' slots
struct FusedPtrs { const double* sum_src[8]; const double* max_src[8]; };
__global__ void fused_tail_kernel(double* __restrict__ tail, FusedPtrs p, int nsum, int nmax, int rank, int world) {
  const int i = threadIdx.x;
  if (i < nsum) tail[i] = *p.sum_src[i];
  for (int e = i; e < world * nmax; e += blockDim.x) {
    const int r = e / nmax, j = e - r * nmax;
    tail[nsum + e] = (r == rank) ? *p.max_src[j] : 0.0;
  }
}
void SysSolver::allreduce_fused(double* d_buf, long npay, const FusedTail& t, double* h_out, int site) {
  HYP_REQUIRE(dist() && comm_world_ > 0 && t.nsum <= 8 && t.nmax <= 8, "allreduce_fused: communicator layout");
  const int ntail = t.nsum + comm_world_ * t.nmax;
  HYP_REQUIRE((size_t)ntail + 32 <= ctx.h_pinned_n - 256, "allreduce_fused: too many ranks for the pinned block");
  FusedPtrs fp{};
  for (int i = 0; i < t.nsum; ++i) fp.sum_src[i] = t.sum_src[i];
  for (int j = 0; j < t.nmax; ++j) fp.max_src[j] = t.max_src[j];
  hipLaunchKernelGGL(fused_tail_kernel, dim3(1), dim3(64), 0, ctx.stream, d_buf + npay, fp, t.nsum, t.nmax, comm_rank_, comm_world_);
  HYP_CHECK(hipGetLastError());
  allreduce_dev(d_buf, npay + ntail, 0, site);
  double* hp = ctx.h_pinned + 256;
  ctx.d2h(hp, d_buf + npay, (size_t)ntail * sizeof(double));
  ctx.sync();
  for (int i = 0; i < t.nsum; ++i) h_out[i] = hp[i];
  for (int j = 0; j < t.nmax; ++j) {
    double m = hp[t.nsum + j];
    bool nan = (m != m);
    for (int r = 1; r < comm_world_; ++r) {
      const double v = hp[t.nsum + r * t.nmax + j];
      nan = nan || (v != v);
      if (v > m) m = v;
    }
    h_out[t.nsum + j] = nan ? __builtin_nan("") : m;
  }
}
void SysSolver::allreduce_fused_dev(double* d_buf, long npay, const FusedTail& t, int site) {
  HYP_REQUIRE(dist() && comm_world_ > 0 && t.nsum <= 8 && t.nmax <= 8, "allreduce_fused_dev: communicator layout");
  const int ntail = t.nsum + comm_world_ * t.nmax;
  FusedPtrs fp{};
  for (int i = 0; i < t.nsum; ++i) fp.sum_src[i] = t.sum_src[i];
  for (int j = 0; j < t.nmax; ++j) fp.max_src[j] = t.max_src[j];
  hipLaunchKernelGGL(fused_tail_kernel, dim3(1), dim3(64), 0, ctx.stream, d_buf + npay, fp, t.nsum, t.nmax, comm_rank_, comm_world_);
  HYP_CHECK(hipGetLastError());
  allreduce_dev(d_buf, npay + ntail, 0, site);
}
void SysSolver::allreduce_host(double* h_buf, int count, int op, int site) {
  if (!dist() || count <= 0) return;
  comm_calls += 1;
  comm_doubles += (double)count;
  comm_hist[site & 15] += 1;
  if (rccl_comm) {   // scalars: through the context's device scalar buffer (64 doubles)
    double* d = ctx.dscal.d() + 32;
    if (count > 32) {   // (the candidate screen's vectors: a few numbers per candidate)
      ar_dev.ensure((size_t)count * sizeof(double));
      d = ar_dev.d();
    }
    ctx.h2d(d, h_buf, (size_t)count * sizeof(double));
    rccl_allreduce_inplace(rccl_comm, d, count, op, ctx.stream);
    ctx.d2h(h_buf, d, (size_t)count * sizeof(double));
    ctx.sync();
    return;
  }
  HYP_REQUIRE(count <= comm_cap, "sys: all-reduce payload exceeds the registered staging buffer");
  ctx.h2d(comm_stage, h_buf, (size_t)count * sizeof(double));
  ctx.sync();
  HYP_REQUIRE(comm_fn(comm_user, count, op) == 0, "sys: all-reduce callback failed");
  ctx.d2h(h_buf, comm_stage, (size_t)count * sizeof(double));
  ctx.sync();
}

void SysSolver::residual_products(const double* h_x, const double* h_z, const double* h_s, double* h_Gtz, double* h_Gx_s, double* h_dots) {
  residual_products2(h_x, h_z, h_s, 0.0, h_Gtz, h_Gx_s, h_dots, nullptr);
}

// max_i |a_i - tau b_i| (b may be null: max |a_i|) into out[0]; one workgroup
__global__ __launch_bounds__(1024) void absmax_diff_kernel(int m, const double* __restrict__ a, const double* __restrict__ b, double tau, double* __restrict__ out) {
  __shared__ double red[1024];
  __shared__ int anynan;
  if (threadIdx.x == 0) anynan = 0;
  __syncthreads();
  double v = 0.0;
  for (int i = threadIdx.x; i < m; i += 1024) {
    const double e = std::fabs(b ? a[i] - tau * b[i] : a[i]);
    if (e != e) anynan = 1;
    v = (e > v) ? e : v;
  }
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] = (red[threadIdx.x + off] > red[threadIdx.x]) ? red[threadIdx.x + off] : red[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = anynan ? __builtin_nan("") : red[0];
}

// h_norms (may be null) = {max |G x + s|, max |G x + s - h tau|} over ALL ranks' rows (Solvers.jl:447-457): with them the exchange of
// the sums carries the two maxima as well (allreduce_fused), and the host needs no collective of its own per iteration
void SysSolver::residual_products2(const double* h_x, const double* h_z, const double* h_s, double tau, double* h_Gtz, double* h_Gx_s, double* h_dots,
                                   double* h_norms) {
  HYP_REQUIRE(model_loaded, "residual_products: load_model first");
  const size_t d = sizeof(double);
  if (rp_pre_valid) {   // queued at accept time (prefetch_residual_products): valid only for exactly the point they were formed from
    rp_pre_valid = false;
    if (!h_norms && !dist() && q > 0 && rp_pre_cand.size() == 2 * (size_t)q + 2) {
      HYP_CHECK(hipEventSynchronize(rp_pre_ev));
      const double* pre = rp_pre_host;
      if (std::memcmp(h_x, pre + n + 2 + q, (size_t)n * d) == 0 && std::memcmp(h_z, rp_pre_cand.data(), (size_t)q * d) == 0 &&
          std::memcmp(h_s, rp_pre_cand.data() + q + 1, (size_t)q * d) == 0) {
        std::memcpy(h_Gtz, pre, (size_t)n * d);
        h_dots[0] = pre[n];
        h_dots[1] = pre[n + 1];
        std::memcpy(h_Gx_s, pre + n + 2, (size_t)q * d);
        ++rp_pre_hits;
        return;
      }
      ++rp_pre_misses;
    }
  }
  rp_x.ensure(std::max<size_t>(n, 1) * d);
  rp_t.ensure(((size_t)n + 2 + 2 + 2 * (size_t)std::max(comm_world_, 1)) * d);   // [G' z (n); h' z; z' s | the two local maxima | their slots]: ONE sum over the ranks
  for (DBuf* b : {&rp_z, &rp_s, &rp_g}) b->ensure(std::max<size_t>(q, 1) * d);
  double* hs = ctx.stage_host((size_t)n + 2 * (size_t)q + 2);
  memcpy(hs, h_x, (size_t)n * d);
  memcpy(hs + n, h_z, (size_t)q * d);
  memcpy(hs + n + q, h_s, (size_t)q * d);
  ctx.h2d(rp_x.p, hs, (size_t)n * d);
  ctx.zero(rp_t.p, ((size_t)n + 2) * d);
  if (q > 0) {
    ctx.h2d(rp_z.p, hs + n, (size_t)q * d);
    ctx.h2d(rp_s.p, hs + n + q, (size_t)q * d);
    ctx.d2d(rp_g.p, rp_s.p, (size_t)q * d);
    if (n > 0 && gemv_both_ok(q, n, G.d(), q)) {                             // G' z (these rows) and G x + s in one pass over G
      gemv_both(ctx, q, n, 1, G.d(), q, rp_x.d(), n, 1.0, rp_g.d(), q, rp_z.d(), q, 0.0, rp_t.d(), n);
    } else {
      sgemv(true, q, n, 1.0, G.d(), q, rp_z.d(), 0.0, rp_t.d());
      sgemv(false, q, n, 1.0, G.d(), q, rp_x.d(), 1.0, rp_g.d());
    }
    dev_dot(ctx, q, mh.d(), rp_z.d(), rp_t.d() + n);
    dev_dot(ctx, q, rp_z.d(), rp_s.d(), rp_t.d() + n + 1);
  }
  double nrm[2] = {0.0, 0.0};
  if (h_norms) {
    double* loc = rp_t.d() + n + 2;   // (behind the payload; the slots follow)
    if (q > 0) {
      hipLaunchKernelGGL(absmax_diff_kernel, dim3(1), dim3(1024), 0, ctx.stream, q, rp_g.d(), (const double*)nullptr, 0.0, loc);
      hipLaunchKernelGGL(absmax_diff_kernel, dim3(1), dim3(1024), 0, ctx.stream, q, rp_g.d(), mh.d(), tau, loc + 1);
      HYP_CHECK(hipGetLastError());
    } else {
      ctx.zero(loc, 2 * d);
    }
    if (dist() && fused_ok()) {
      rp_loc.ensure(2 * d);
      ctx.d2d(rp_loc.p, loc, 2 * d);   // (the tail is written over the scratch position)
      FusedTail t;
      t.nmax = 2; t.max_src[0] = rp_loc.d(); t.max_src[1] = rp_loc.d() + 1;
      allreduce_fused(rp_t.d(), (long)n + 2, t, nrm, 11);
    } else {
      allreduce_dev(rp_t.d(), (long)n + 2, 0, 11);
      ctx.d2h(ctx.h_pinned + 24, loc, 2 * d);
      ctx.sync();
      double v[4] = {(ctx.h_pinned[24] != ctx.h_pinned[24]) ? 1.0 : 0.0, (ctx.h_pinned[25] != ctx.h_pinned[25]) ? 1.0 : 0.0, 0.0, 0.0};
      v[2] = v[0] > 0.5 ? 0.0 : ctx.h_pinned[24];
      v[3] = v[1] > 0.5 ? 0.0 : ctx.h_pinned[25];
      if (dist()) allreduce_host(v, 4, 1, 12);
      nrm[0] = v[0] > 0.5 ? __builtin_nan("") : v[2];
      nrm[1] = v[1] > 0.5 ? __builtin_nan("") : v[3];
    }
    h_norms[0] = nrm[0];
    h_norms[1] = nrm[1];
  } else {
    allreduce_dev(rp_t.d(), (long)n + 2, 0, 11);                            // sum over ranks (in place, stream order)
  }
  ctx.d2h(hs, rp_t.p, ((size_t)n + 2) * d);
  if (q > 0) ctx.d2h(hs + n + 2, rp_g.p, (size_t)q * d);
  ctx.sync();
  memcpy(h_Gtz, hs, (size_t)n * d);
  h_dots[0] = hs[n];
  h_dots[1] = hs[n + 1];
  if (q > 0) memcpy(h_Gx_s, hs + n + 2, (size_t)q * d);
}

void SysSolver::block_hess_prod_vec(double* d_out, const double* d_in) {   // qrchol.jl:87-98
  for (size_t k = 0; k < cones.size(); ++k) {
    Cone* ck = cones[k];
    if (const int used = run_hess_prod(k, d_out + offs[k], q, d_in + offs[k], q, 1)) { k += used - 1; continue; }
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

static bool force_bk_env();

// HYP_SHP_PRELAUNCH=1 (round 6; default off -- measured: 17.41 / 17.65 / 17.46 ms per iteration with it against 17.38 / 17.42 / 17.33
// without, alternating on one box: outside the profiler the gap it closes is smaller than what the extra launches inside the search
// call cost, EXPERIMENTS.md r06-24): when the line search accepts a candidate, every cone holds the state the NEXT update_lhs
// starts from, and nothing the host does until then (convergence check, stepper bookkeeping, the call into step_directions: ~0.2 ms,
// the last gap above 100 us of an iteration) feeds the cones' square-root Hessian products (qrchol.jl:219-233).  They are queued right
// there, behind the prefetched residual products; assemble_lhs finds HGQ2 filled and continues with the Schur product -- provided no
// cone has taken a point or been reset since (Ctx::cone_epoch).  If the solve ends instead (converged), 2 ms of device time were spent
// for nothing, once; a host that reloads the cones every iteration makes every prelaunch useless: after two in a row it stays off.
void SysSolver::prelaunch_sqrt_hess() {
  static const bool on = [] { const char* e = getenv("HYP_SHP_PRELAUNCH"); return e && e[0] == '1'; }();
  shp_prelaunched = false;
  if (!on || shp_wasted >= 2 || nmp <= 0 || p != 0 || dist() || ks_world > 1 || !dirs_resident() || force_bk_env() ||
      ctx.stream != ctx.stream_primary)
    return;
  {
    const char* ff = getenv("HYP_FORCE_FACT_FAIL");
    if (ff && ff[0] && ff[0] != '0') return;
  }
  group_inverses();
  bool all_sqrt = true;
  for (size_t k = 0; k < cones.size(); ++k) {
    use_sqrt[k] = cones[k]->use_sqrt_hess_oracles(nmp) ? 1 : 0;
    all_sqrt &= (use_sqrt[k] != 0);
  }
  if (!all_sqrt) return;   // (a cone that adds G' (H G) instead: assemble_lhs as a whole, later)
  const double* gq2 = GQ2();
  HYP_CHECK(hipEventRecord(ctx.ev[0], ctx.stream));
  int idx = 0;
  for (size_t k = 0; k < cones.size(); ++k) {
    Cone* ck = cones[k];
    if (ck->use_dual_barrier) ck->inv_sqrt_hess_prod(HGQ2.d() + idx, q, gq2 + offs[k], q, nmp);
    else ck->sqrt_hess_prod(HGQ2.d() + idx, q, gq2 + offs[k], q, nmp);
    idx += ck->dim;
  }
  HYP_CHECK(hipEventRecord(ctx.ev[1], ctx.stream));
  shp_prelaunched = true;
  shp_epoch = ctx.cone_epoch;
}

void SysSolver::assemble_lhs() {
  if (split_unjoined) {   // (a split factorization nobody finished: its lane must not run into this assembly's copies)
    HYP_CHECK(hipStreamWaitEvent(ctx.stream, split_ev_done, 0));
    split_unjoined = false;
  }
  chol_split_n1 = 0;
  // (the cones' products of this assembly may be in HGQ2 already: prelaunch_sqrt_hess)
  const bool pre = shp_prelaunched && shp_epoch == ctx.cone_epoch && nmp > 0 && !dist() && ks_world == 1;
  if (shp_prelaunched) {
    if (pre) { shp_wasted = 0; ++shp_used; }
    else { ++shp_wasted; ++shp_unused; }
  }
  shp_prelaunched = false;
  if (!pre) group_inverses();   // qrchol.jl:214-246 (this process's cones only; the multi-GPU glue sums the result)
  if (nmp == 0) return;
  const double* gq2 = GQ2();
  if (!pre)
    for (size_t k = 0; k < cones.size(); ++k) use_sqrt[k] = cones[k]->use_sqrt_hess_oracles(nmp) ? 1 : 0;   // :214-216
  bool any_sqrt = false;
  for (int v : use_sqrt) any_sqrt |= (v != 0);
  // (round 5, HYP_DIST_OVERLAP=G: row groups, each exchanged on the helper stream under the next one's product)
  // The choice between the grouped exchange and the single one must be the SAME on every rank (they issue different collectives --
  // ADVICE r05): with the cones sharded, "every cone of mine went through its square root" is a per-rank fact (a cone loses its
  // square-root oracle behind a Bunch-Kaufman fall-back of its Hessian), so the ranks agree on the minimum first -- one scalar
  // exchange, only with the switch on, in front of everything a rank may or may not do.  K-panel sharding replicates the model: the
  // flag is the same everywhere by construction.  A rank without rows of its own still takes part (zero blocks).
  static const int ov_groups = [] { const char* e = getenv("HYP_DIST_OVERLAP"); return e ? atoi(e) : 0; }();
  bool use_ov = ov_groups >= 2 && (dist() || ks_world > 1) && rccl_comm != nullptr && nmp >= 1024;
  if (use_ov) {
    bool all_sqrt = true;
    for (int v : use_sqrt) all_sqrt &= (v != 0);
    if (dist()) {
      double f = all_sqrt ? 1.0 : 0.0;
      allreduce_host(&f, 1, 2, 15);
      all_sqrt = f > 0.5;
    }
    use_ov = all_sqrt;
  }
  if (!pre) {
    HYP_CHECK(hipEventRecord(ctx.ev[0], ctx.stream));
    HYP_CHECK(hipEventRecord(ctx.ev[1], ctx.stream));
  }
  HYP_CHECK(hipEventRecord(ctx.ev[2], ctx.stream));
  if (any_sqrt || use_ov) {   // :219-234
    int idx = 0;
    for (size_t k = 0; k < cones.size(); ++k) {
      if (!use_sqrt[k]) continue;
      Cone* ck = cones[k];
      if (!pre) {
        if (ck->use_dual_barrier) ck->inv_sqrt_hess_prod(HGQ2.d() + idx, q, gq2 + offs[k], q, nmp);
        else ck->sqrt_hess_prod(HGQ2.d() + idx, q, gq2 + offs[k], q, nmp);
      }
      idx += ck->dim;
    }
    if (!pre) HYP_CHECK(hipEventRecord(ctx.ev[1], ctx.stream));
    GemmArgs s{};   // lhs = HGQ2[1:idx, :]' HGQ2[1:idx, :]  (outer_prod!, dense.jl:80-86)
    long r0 = 0, r1 = idx;
    if (ks_world > 1) {   // this rank's K panel (16-row granularity keeps the aligned fast loader); the sum over ranks follows
      const long per = (((long)idx + ks_world - 1) / ks_world + 15) / 16 * 16;
      r0 = std::min<long>((long)idx, per * ks_rank);
      r1 = std::min<long>((long)idx, r0 + per);
    }
    if (use_ov && assemble_lhs_overlapped(r0, r1, ov_groups)) {
      HYP_CHECK(hipEventRecord(ctx.ev[2], ctx.stream));
      ctx.kstat[4] += 1;
      return;   // (the exchange is done)
    }
    s.M = nmp; s.N = nmp; s.K = (int)(r1 - r0); s.A = HGQ2.d() + r0; s.lda = q; s.B = HGQ2.d() + r0; s.ldb = q; s.C = lhs.d(); s.ldc = nmp;
    s.alpha = 1; s.beta = 0; s.tri = GEMM_UPPER; s.krange = KR_ALL; s.batch = 1; s.tag = 1;
    // HYP_CHOL_SPLIT (round 6): product and factorization in two column groups, the leading block of the Schur complement factored
    // underneath the product of the rest (dense.hip: schur_split_begin / _finish).  Only when this product IS the Schur complement
    // (every cone through its square root, nothing to add or to exchange afterwards) and the Cholesky will be attempted.
    {
      bool all_sqrt = true;
      for (int v : use_sqrt) all_sqrt &= (v != 0);
      const int n1 = chol_split_point(nmp, s.K);
      if (n1 > 0 && all_sqrt && !dist() && ks_world == 1 && !force_bk_env() && ctx.stream == ctx.stream_primary) {
        if (!split_ev_ready) {
          HYP_CHECK(hipEventCreateWithFlags(&split_ev_ready, hipEventDisableTiming));
          HYP_CHECK(hipEventCreateWithFlags(&split_ev_done, hipEventDisableTiming));
        }
        schur_split_begin(ctx, nmp, s.K, s.A, q, lhs.d(), lhs_fact.d(), n1, d_info.i(), ctx.lane(2).s, split_ev_ready, split_ev_done);
        chol_split_n1 = n1;
        split_unjoined = true;
        HYP_CHECK(hipEventRecord(ctx.ev[2], ctx.stream));
        ctx.kstat[4] += 1;
        return;   // (all_sqrt, one process: nothing below applies)
      }
    }
    // (the thin last columns on a lane beside the product instead of behind it: 8.66 against 8.32 - 8.38 ms, EXPERIMENTS.md r06-22 -- as in r01-4)
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
    // K-panel sharding: a cone that has lost its square root mid-solve (a Hessian whose Cholesky fell through to Bunch-Kaufman:
    // use_sqrt_hess_oracles answers false, Cones.jl:189-195) takes this branch on every rank at once -- the product prod_k is
    // replicated, and each rank contracts only ITS rows of the cone, so that the all-reduce below sums to the whole term
    // exactly as it does for the square-root branch (16-row granularity as above)
    long r0 = 0, r1 = ck->dim;
    if (ks_world > 1) {
      const long per = (((long)ck->dim + ks_world - 1) / ks_world + 15) / 16 * 16;
      r0 = std::min<long>((long)ck->dim, per * ks_rank);
      r1 = std::min<long>((long)ck->dim, r0 + per);
    }
    if (r1 <= r0) continue;
    GemmArgs g{};
    g.M = nmp; g.N = nmp; g.K = (int)(r1 - r0); g.A = gq2 + offs[k] + r0; g.lda = q; g.B = prod_k + r0; g.ldb = q; g.C = lhs.d(); g.ldc = nmp;
    g.alpha = 1; g.beta = 1; g.tri = GEMM_UPPER; g.krange = KR_ALL; g.batch = 1;
    gemm(ctx, true, g);
  }
  allreduce_lhs();   // the one large exchange: sum of the ranks' Schur contributions (cones or K panels)
}

static bool force_bk_env() {
  const char* fb = getenv("HYP_FORCE_BK");
  return fb && fb[0] && fb[0] != '0';
}

// posdef_fact_copy! (dense.jl:194-215) in two halves.  begin: the Cholesky attempt, the read-back of its info word and the solve
// plan of the (almost always successful) factor are QUEUED; nothing waits.  end (after the caller's synchronisation): info is
// read, and behind a failed Cholesky the fall-back chain runs.  factor_lhs() is begin + synchronise + end; step_directions
// (round 6) puts the first pair of direction solves between the two, so that the host learns info together with that pair's
// scalars instead of stopping the device for a round trip of its own.
// Where the Schur complement is cut for HYP_CHOL_SPLIT (0: not at all).  The leading block's factorization -- ~60 us per block step,
// up to twice that beside the product -- has to fit under the product of the remaining columns: with T = n / 128 tile columns the
// leading t hold t (t + 1) / 2 of the T (T + 1) / 2 upper tiles.  HYP_CHOL_SPLIT=<fraction of n, in percent>; default 0 = off:
// measured at config 2 (EXPERIMENTS.md r06-19) the iteration gets 1.8 - 2.8 ms LONGER -- the product in two launches costs 1.5 ms
// (the trapezoid of the late columns has neither the upper-tile order nor the cut last round of the one-piece product), and the
// factorization's kernels find no compute unit beside it (profiles/r06_chol_split_timeline_*.txt).
int SysSolver::chol_split_point(int n, int K) {
  static const int pct = [] { const char* e = getenv("HYP_CHOL_SPLIT"); return e ? atoi(e) : 0; }();
  static const int min_n = [] { const char* e = getenv("HYP_CHOL_SPLIT_MIN_N"); return e ? atoi(e) : 3072; }();
  if (pct <= 0 || n < min_n || K < 4096) return 0;
  int n1 = (int)((long)n * std::min(pct, 95) / 100) / 128 * 128;
  return potrf_split_ok(n, n1) ? n1 : 0;
}

void SysSolver::factor_lhs_begin() {
  if (nmp == 0 || force_bk_env()) return;
  use_bk = false;
  if (chol_split_n1 > 0) {   // (assemble_lhs has factored the leading block)
    HYP_CHECK(hipEventRecord(ctx.ev[3], ctx.stream));
    schur_split_finish(ctx, nmp, lhs.d(), lhs_fact.d(), chol_split_n1, dinv.d(), d_info.i(), split_ev_done);
    chol_split_n1 = 0;
    split_unjoined = false;
    ++chol_split_count;
  } else {
    ctx.d2d(lhs_fact.p, lhs.p, (size_t)nmp * nmp * sizeof(double));
    HYP_CHECK(hipEventRecord(ctx.ev[3], ctx.stream));
    potrf_upper_batched(ctx, nmp, lhs_fact.d(), nmp, 0, 1, dinv.d(), d_info.i());
  }
  HYP_CHECK(hipEventRecord(ctx.ev[4], ctx.stream));
  ctx.d2h(ctx.h_info + Ctx::H_INFO_FACT, d_info.p, sizeof(int));
  // the solve plan is queued BEFORE the host learns info: reading info first left the device idle for the round trip (~0.2 ms
  // per iteration, profiles/r02_iteration_timeline.txt); after a failed Cholesky the plan's kernels ran on meaningless numbers
  // and the plan is discarded in factor_lhs_end
  tri.invalidate();
  if (ctx.trsv_plan_sb(nmp) <= 0) return;
  // HYP_PLAN_LANE=1 (round 6): the plan's launches -- inversions of the diagonal super-blocks, transposes: ~0.3 ms of latency-bound
  // chains at n = 5000 -- go to a lane of their own, so that whatever the caller queues next on the main stream and does not need
  // the plan (the first pair's right-hand sides, cone products and G' z) runs beside them; the first user of the plan joins (join_plan)
  static const bool plan_lane = [] { const char* e = getenv("HYP_PLAN_LANE"); return e && e[0] == '1'; }();
  if (plan_lane && Ctx::max_lanes() >= 3 && ctx.stream == ctx.stream_primary) {
    if (!plan_ev_fork) {
      HYP_CHECK(hipEventCreateWithFlags(&plan_ev_fork, hipEventDisableTiming));
      HYP_CHECK(hipEventCreateWithFlags(&plan_ev_done, hipEventDisableTiming));
    }
    HYP_CHECK(hipEventRecord(plan_ev_fork, ctx.stream));
    {
      LaneSwitch on_lane(ctx, 2);
      HYP_CHECK(hipStreamWaitEvent(ctx.stream, plan_ev_fork, 0));
      tri.build(ctx, nmp, lhs_fact.d(), nmp, dinv.d());
      HYP_CHECK(hipEventRecord(plan_ev_done, ctx.stream));
    }
    plan_join_pending = true;
    return;
  }
  tri.build(ctx, nmp, lhs_fact.d(), nmp, dinv.d());
}

void SysSolver::join_plan() {
  if (!plan_join_pending) return;
  HYP_CHECK(hipStreamWaitEvent(ctx.stream_primary, plan_ev_done, 0));
  plan_join_pending = false;
}

void SysSolver::factor_lhs_times() {   // HIP-event times of the update_lhs phases (every event has completed)
  float ms = 0;
  HYP_CHECK(hipEventElapsedTime(&ms, ctx.ev[0], ctx.ev[1])); ctx.kstat[0] += ms;
  HYP_CHECK(hipEventElapsedTime(&ms, ctx.ev[1], ctx.ev[2])); ctx.kstat[1] += ms;
  HYP_CHECK(hipEventElapsedTime(&ms, ctx.ev[3], ctx.ev[4])); ctx.kstat[2] += ms;
  ctx.kstat[3] += 1;
}

void SysSolver::factor_lhs_end(int* info, int* used_fallback, bool times_later) {
  *info = 0;
  *used_fallback = 0;
  if (nmp == 0) return;
  // Cholesky; on failure Bunch-Kaufman with rook pivoting (symm_fact!, dense.jl:164-165); if that finds an exactly singular
  // pivot, increase_diag! (dense.jl:106-113) and Bunch-Kaufman again.  used_fallback: 0 Cholesky, 1 Bunch-Kaufman, 2 diagonal
  // shift + Bunch-Kaufman.  HYP_FORCE_BK=1 (tests) treats the Cholesky as failed.
  const bool force_bk = force_bk_env();
  use_bk = false;
  join_plan();   // (nothing below may touch the factor under a plan build that still reads it)
  if (!force_bk) {
    *info = ctx.h_info[Ctx::H_INFO_FACT];
    if (!times_later) factor_lhs_times();
  }
  if (force_bk || *info != 0) {
    use_bk = true;
    *used_fallback = 1;
    const int chol_info = force_bk ? 0 : *info;
    ctx.d2d(lhs_fact.p, lhs.p, (size_t)nmp * nmp * sizeof(double));
    *info = bk_after_failed_cholesky(ctx, bk, nmp, lhs_fact.d(), nmp, dinv.d(), d_info.i(), chol_info, lhs.d(), nmp);
    if (*info != 0) {
      *used_fallback = 2;
      ctx.d2d(lhs_fact.p, lhs.p, (size_t)nmp * nmp * sizeof(double));
      hipLaunchKernelGGL(increase_diag_kernel, dim3((nmp + 255) / 256), dim3(256), 0, ctx.stream, nmp, lhs_fact.d(), (long)nmp);
      *info = bk.factor(ctx, nmp, lhs_fact.d(), nmp, dinv.d());
    }
  }
  {   // HYP_FORCE_FACT_FAIL=1 (tests): report every link of the chain as failed, so that the hosts' NumericalFailure path runs
    const char* ff = getenv("HYP_FORCE_FACT_FAIL");
    if (ff && ff[0] && ff[0] != '0') *info = 1;
  }
  fact_ok = (*info == 0);
  if (use_bk || !fact_ok) tri.invalidate();
  if (fact_ok && !tri.ready(nmp) && ctx.trsv_plan_sb(nmp) > 0) tri.build(ctx, nmp, lhs_fact.d(), nmp, dinv.d());
}

void SysSolver::factor_lhs(int* info, int* used_fallback) {   // qrchol.jl:249-250
  *info = 0;
  *used_fallback = 0;
  if (nmp == 0) return;
  factor_lhs_begin();
  if (!force_bk_env()) ctx.sync();
  factor_lhs_end(info, used_fallback);
}

// x <- lhs^-1 x.  Cholesky: U'^-1 then U^-1.  Bunch-Kaufman: the same two sweeps with the unit factor, between a
// gather by P, the block-diagonal solve and a scatter.
void SysSolver::tri_solves(double* d_x) {
  join_plan();
  if (!use_bk && tri.ready(nmp)) { tri.solve_both(ctx, lhs_fact.d(), nmp, d_x, nmp, 1); return; }
  double* y = use_bk ? bk.gather(ctx, d_x, nmp, 1) : d_x;
  for (int pass = 0; pass < 2; ++pass) {
    const bool trans = (pass == 0);
    if (tri.ready(nmp)) tri.solve(ctx, lhs_fact.d(), nmp, trans, y);
    else trsv_upper(ctx, nmp, lhs_fact.d(), nmp, dinv.d(), trans, y);
    if (use_bk && pass == 0) bk.dsolve(ctx, y, nmp, 1);
  }
  if (use_bk) bk.scatter(ctx, y, d_x, nmp, 1);
}

void SysSolver::potrs(double* d_x) { tri_solves(d_x); }

// One-right-hand-side products with G (and A, Q): for models whose cones are column-wise (Cone::products_columnwise) through the
// multi-column kernels with one column, so that a column has the same sums wherever it is formed (step_directions); for the others
// through the one-column kernels of rounds 1-2 -- their constant column keeps products of its own, and with them the arithmetic
// their trajectories were pinned with (tests/test_hip_trajectory.py: mixed_dual_barriers ended in SlowProgress otherwise).
void SysSolver::sgemv(bool trans, int m, int n_, double alpha, const double* A, long lda, const double* x, double beta, double* y) {
  bool cw = true;
  for (const Cone* ck : cones) cw = cw && ck->products_columnwise();
  const bool keep = ctx.gemv_one;
  ctx.gemv_one = cw;
  try {
    gemv(ctx, trans, m, n_, alpha, A, lda, x, beta, y);
  } catch (...) {
    ctx.gemv_one = keep;
    throw;
  }
  ctx.gemv_one = keep;
}

void SysSolver::solve3(double* d_sol, const double* d_rhs) {   // qrchol.jl:39-85
  const size_t d = sizeof(double);
  if (d_sol != d_rhs) ctx.d2d(d_sol, d_rhs, (size_t)(n + p + q) * d);
  double* x = d_sol;
  double* y = d_sol + n;
  double* z = d_sol + n + p;
  double* t = QpbxGHbz.d();
  // t = Q' (x + G' z)                                                   :51-53
  if (dist()) {   // G = this rank's rows: sum the partial G' z over the ranks, then add the (replicated) x once
    HYP_REQUIRE(p == 0, "sys: the sharded path assumes the reduced model (p = 0)");
    sgemv(true, q, n, 1.0, G.d(), q, z, 0.0, t);
    allreduce_dev(t, n, 0, 1);
    dev_axpby(ctx, n, 1.0, x, 1.0, t);
  } else {
    ctx.d2d(t, x, (size_t)n * d);
    sgemv(true, q, n, 1.0, G.d(), q, z, 1.0, t);
  }
  if (p > 0) {
    sgemv(true, n, n, 1.0, Qm.d(), n, t, 0.0, tmpn.d());
    ctx.d2d(t, tmpn.p, (size_t)n * d);
    // y <- R'^-1 y ; sol.vec[1:p] = y                                   :55-57
    sgemv(true, p, p, 1.0, Rinv.d(), p, y, 0.0, tmpn.d());
    ctx.d2d(y, tmpn.p, (size_t)p * d);
    ctx.d2d(x, y, (size_t)p * d);
    if (nmp > 0) {                                                       // :59-63
      sgemv(false, q, p, 1.0, GQ1.d(), q, y, 0.0, GQ1x.d());
      block_hess_prod_vec(HGQ1x.d(), GQ1x.d());
      sgemv(true, q, nmp, -1.0, GQ2s.d(), q, HGQ1x.d(), 1.0, t + p);
    }
  }
  if (nmp > 0) {                                                         // :66-69
    ctx.d2d(x + p, t + p, (size_t)nmp * d);
    tri_solves(x + p);
  }
  if (p > 0) {                                                           // :71  x = Q x
    sgemv(false, n, n, 1.0, Qm.d(), n, x, 0.0, tmpn.d());
    ctx.d2d(x, tmpn.p, (size_t)n * d);
  }
  sgemv(false, q, n, 1.0, G.d(), q, x, 0.0, Gx.d());                 // :73
  block_hess_prod_vec(HGx.d(), Gx.d());                                  // :74
  dev_axpby(ctx, q, 1.0, HGx.d(), -1.0, z);                              // :76  z = HGx - z
  if (p > 0) {                                                           // :78-82
    ctx.d2d(tmpn.p, t, (size_t)p * d);
    sgemv(true, q, p, -1.0, GQ1.d(), q, HGx.d(), 1.0, tmpn.d());
    sgemv(false, p, p, 1.0, Rinv.d(), p, tmpn.d(), 0.0, y);
  }
}

// =============================================================================================
// device-resident get_directions (systemsolvers/common.jl:15-182)
// =============================================================================================
// out[0] = max_i |a_i - b_i| over n entries (NaN if any entry is NaN); a <- a - b in place.  SA_WGS workgroups; the last one to
// finish (a ticket in device memory, reset for the next call) folds the partial maxima -- one launch; with a single workgroup
// the ~45 dependent trips per thread took 26 us at config 2, four times per iteration.
constexpr int SA_WGS = 32;
__device__ __forceinline__ double nanmax(double m, double o) { return (m != m || o != o) ? __builtin_nan("") : fmax(m, o); }
__global__ __launch_bounds__(256) void sub_absmax_kernel(int n, double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out,
                                                         double* __restrict__ part, unsigned* __restrict__ ticket) {
  __shared__ double red[4];
  __shared__ bool last;
  double m = 0.0;
  bool bad = false;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += SA_WGS * 256) {
    const double v = a[i] - b[i];
    a[i] = v;
    bad |= (v != v);
    m = fmax(m, fabs(v));
  }
  if (bad) m = __builtin_nan("");
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = nanmax(m, __shfl_down(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3]));
    __threadfence();
    last = (atomicAdd(ticket, 1u) == SA_WGS - 1);
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    __threadfence();
    double r = (threadIdx.x < SA_WGS) ? reinterpret_cast<volatile double*>(part)[threadIdx.x] : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) r = nanmax(r, __shfl_down(r, off));
    if (threadIdx.x == 0) {
      out[0] = r;
      *ticket = 0u;
    }
  }
}

// the same for nr columns (a, b: columns ld apart; out[r]) in ONE launch: column blockIdx.y with its own partial maxima and ticket
// (work: nr x (SA_WGS doubles + one ticket word), zeroed at allocation, reset by every call)
__global__ __launch_bounds__(256) void sub_absmax_cols_kernel(int n, double* __restrict__ a, const double* __restrict__ b, long ld,
                                                              double* __restrict__ out, double* __restrict__ work) {
  __shared__ double red[4];
  __shared__ bool last;
  const int r = blockIdx.y;
  a += (long)r * ld;
  b += (long)r * ld;
  double* part = work + (long)r * (SA_WGS + 1);
  unsigned* ticket = reinterpret_cast<unsigned*>(part + SA_WGS);
  double m = 0.0;
  bool bad = false;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += SA_WGS * 256) {
    const double v = a[i] - b[i];
    a[i] = v;
    bad |= (v != v);
    m = fmax(m, fabs(v));
  }
  if (bad) m = __builtin_nan("");
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = nanmax(m, __shfl_down(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = nanmax(nanmax(red[0], red[1]), nanmax(red[2], red[3]));
    __threadfence();
    last = (atomicAdd(ticket, 1u) == SA_WGS - 1);
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    __threadfence();
    double v = (threadIdx.x < SA_WGS) ? reinterpret_cast<volatile double*>(part)[threadIdx.x] : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = nanmax(v, __shfl_down(v, off));
    if (threadIdx.x == 0) {
      out[r] = v;
      *ticket = 0u;
    }
  }
}
void dev_sub_absmax_cols(Ctx& c, int n, int nr, double* a, const double* b, long ld, double* d_out, double* work) {
  if (nr <= 0) return;
  hipLaunchKernelGGL(sub_absmax_cols_kernel, dim3(SA_WGS, nr), dim3(256), 0, c.stream, n, a, b, ld, d_out, work);
  HYP_CHECK(hipGetLastError());
}

void dev_sub_absmax(Ctx& c, int n, double* a, const double* b, double* d_out) {
  // partial maxima and the ticket live behind the 64 scalar slots of the context (zeroed at creation, reset by every call)
  double* part = c.dscal.d() + 64;
  unsigned* ticket = reinterpret_cast<unsigned*>(c.dscal.d() + 64 + SA_WGS);
  hipLaunchKernelGGL(sub_absmax_kernel, dim3(SA_WGS), dim3(256), 0, c.stream, n, a, b, d_out, part, ticket);
  HYP_CHECK(hipGetLastError());
}

void SysSolver::load_model(const double* hc, const double* hb, const double* hh, const double* hA) {
  const size_t d = sizeof(double);
  mc.ensure(std::max(n, 1) * d); mb.ensure(std::max(p, 1) * d); mh.ensure(std::max(q, 1) * d);
  ctx.h2d(mc.p, hc, (size_t)n * d);
  ctx.h2d(mb.p, hb, (size_t)p * d);
  ctx.h2d(mh.p, hh, (size_t)q * d);
  if (p > 0) {
    HYP_REQUIRE(hA != nullptr, "sys: model.A is required when p > 0");
    mA.ensure((size_t)p * n * d);
    ctx.h2d(mA.p, hA, (size_t)p * n * d);
  }
  for (DBuf* b : {&v_rhs, &v_dir, &v_res, &v_tmp}) {
    b->ensure((size_t)dimv() * d);
    ctx.zero(b->p, (size_t)dimv() * d);
  }
  for (DBuf* b : {&sub_rhs, &sub_sol, &sol_const}) b->ensure((size_t)(n + p + q) * d);
  Gx_dir.ensure(std::max(q, 1) * d);
  ctx.sync();
  model_loaded = true;
  screen_agreed = -1;
  gprev_acc_ = -1;
  s_resident = false;   // (the point and directions of an earlier model are not this model's)
  shp_prelaunched = false; shp_wasted = 0;
  rp_pre_valid = false;
}

// rhs_const = [-c; b; H h], sol_const = solve_subsystem3(rhs_const)   (qrchol.jl:191-197)
void SysSolver::update_const() {
  HYP_REQUIRE(model_loaded, "sys: load_model first");
  double* r = sub_rhs.d();
  dev_scale_copy(ctx, n, -1.0, mc.d(), r);
  if (p > 0) ctx.d2d(r + n, mb.p, (size_t)p * sizeof(double));
  block_hess_prod_vec(r + n + p, mh.d());
  solve3(sol_const.d(), r);
  double* ds = ctx.dscal.d();
  {
    DotSpecs sp;
    sp.add(n, mc.d(), sol_const.d(), ds);
    sp.add(p, mb.d(), sol_const.d() + n, ds + 1);
    sp.add(q, mh.d(), sol_const.d() + n + p, ds + 2);
    dev_dots(ctx, sp);
  }
  ctx.d2h(ctx.h_pinned, ds, 3 * sizeof(double));
  ctx.sync();
  double hz = ctx.h_pinned[2];
  if (dist()) allreduce_host(&hz, 1, 0, 6);   // h' z runs over all ranks' rows; c' x is replicated
  dot_const = ctx.h_pinned[0] + (p > 0 ? ctx.h_pinned[1] : 0.0) + hz;
}

// update_const() in two halves around the triangular solves (p = 0, one process, Cholesky factor): pre leaves lhs-space right-hand
// side x + G' z in sol_const's x part, post continues solve3 from there (z = H (G x) - z); the dot products and dot_const are the
// caller's (pair_solve_device)
void SysSolver::update_const_pre() {
  HYP_REQUIRE(model_loaded && p == 0 && !dist(), "update_const_pre: single process, p = 0");
  const size_t d = sizeof(double);
  double* r = sub_rhs.d();
  dev_scale_copy(ctx, n, -1.0, mc.d(), r);
  block_hess_prod_vec(r + n, mh.d());
  double* sol = sol_const.d();
  ctx.d2d(sol, r, (size_t)(n + q) * d);
  double* t = QpbxGHbz.d();
  ctx.d2d(t, sol, (size_t)n * d);
  sgemv(true, q, n, 1.0, G.d(), q, sol + n, 1.0, t);     // :51-53
  ctx.d2d(sol, t, (size_t)n * d);                            // :66-69 (the solves follow, with the first pair's)
}
void SysSolver::update_const_post() {
  double* sol = sol_const.d();
  sgemv(false, q, n, 1.0, G.d(), q, sol, 0.0, Gx.d());   // :73
  block_hess_prod_vec(HGx.d(), Gx.d());                      // :74
  dev_axpby(ctx, q, 1.0, HGx.d(), -1.0, sol + n);            // :76  z = HGx - z
}

SysSolver::Scal SysSolver::solve_system(double* sol, const double* rhs, Scal rs, double mu, double taubar) {
  const size_t d = sizeof(double);
  const int oz = n + p, os = n + p + q + 1;
  double* sr = sub_rhs.d();
  double* ss = sub_sol.d();
  // solve_subsystem4 (common.jl:146-182): rhs_sub.x = rhs.x; rhs_sub.y = -rhs.y; setup_rhs3 (qrchol.jl:16-37)
  ctx.d2d(sr, rhs, (size_t)n * d);
  if (p > 0) dev_scale_copy(ctx, p, -1.0, rhs + n, sr + n);
  for (size_t k = 0; k < cones.size(); ++k) {
    Cone* ck = cones[k];
    const int o = offs[k], dk = ck->dim;
    if (ck->use_dual_barrier) {
      double* tmp = ss + oz + o;   // (z_temp_k = sol.z_k in the reference)
      dev_scale_copy(ctx, dk, -1.0, rhs + oz + o, tmp);
      dev_axpby(ctx, dk, -1.0, rhs + os + o, 1.0, tmp);
      ck->inv_hess_prod(sr + oz + o, q, tmp, q, 1);
    } else if (const int used = run_hess_prod(k, sr + oz + o, q, rhs + oz + o, q, 1)) {
      dev_axpby(ctx, offs[k + used] - o, -1.0, rhs + os + o, -1.0, sr + oz + o);
      k += used - 1;
    } else {
      ck->hess_prod(sr + oz + o, q, rhs + oz + o, q, 1);
      dev_axpby(ctx, dk, -1.0, rhs + os + o, -1.0, sr + oz + o);
    }
  }
  solve3(ss, sr);
  double* ds = ctx.dscal.d();
  dev_dot(ctx, n, mc.d(), ss, ds);
  dev_dot(ctx, p, mb.d(), ss + n, ds + 1);
  dev_dot(ctx, q, mh.d(), ss + oz, ds + 2);
  ctx.d2h(ctx.h_pinned, ds, 3 * d);
  ctx.sync();
  double hz_sub = ctx.h_pinned[2];
  if (dist()) allreduce_host(&hz_sub, 1, 0, 2);
  const double dot_sub = ctx.h_pinned[0] + (p > 0 ? ctx.h_pinned[1] : 0.0) + hz_sub;
  const double tau_num = rs.tau + rs.kap + dot_sub;
  const double tau_denom = mu / taubar / taubar - dot_const;
  const double sol_tau = tau_num / tau_denom;
  // sol[1:dim3] = sol_sub + sol_tau * sol_const
  ctx.d2d(sol, ss, (size_t)(n + p + q) * d);
  dev_axpby(ctx, n + p + q, sol_tau, sol_const.d(), 1.0, sol);
  // sol.s = h * tau - rhs.z - G sol.x     (common.jl:139-141).  G sol.x is formed from the rounded sol.x itself
  // (NOT as G sol_sub.x + tau G sol_const.x: when the two parts cancel, that sum loses the consistency
  // between x and s that the residual check relies on); it is kept for the residual, which needs the same product.
  sgemv(false, q, n, 1.0, G.d(), q, sol, 0.0, Gx_dir.d());
  Gx_dir_valid = true;
  dev_scale_copy(ctx, q, sol_tau, mh.d(), sol + os);
  dev_axpby(ctx, q, -1.0, rhs + oz, 1.0, sol + os);
  dev_axpby(ctx, q, -1.0, Gx_dir.d(), 1.0, sol + os);
  Scal out;
  out.tau = sol_tau;
  out.kap = -mu / taubar / taubar * sol_tau + rs.kap;
  return out;
}

SysSolver::Scal SysSolver::apply_lhs(double* res, const double* dir, Scal ds_, double mu, double taubar) {
  const size_t d = sizeof(double);
  const int oz = n + p, os = n + p + q + 1;
  const double tau_dir = ds_.tau, kap_dir = ds_.kap;
  // res.x = c tau + G' z (+ A' y)
  if (dist()) {
    sgemv(true, q, n, 1.0, G.d(), q, dir + oz, 0.0, res);
    allreduce_dev(res, n, 0, 3);
    dev_axpby(ctx, n, tau_dir, mc.d(), 1.0, res);
  } else {
    dev_scale_copy(ctx, n, tau_dir, mc.d(), res);
    sgemv(true, q, n, 1.0, G.d(), q, dir + oz, 1.0, res);
  }
  // res.z = h tau - s - G x
  dev_scale_copy(ctx, q, tau_dir, mh.d(), res + oz);
  dev_axpby(ctx, q, -1.0, dir + os, 1.0, res + oz);
  if (Gx_dir_valid) dev_axpby(ctx, q, -1.0, Gx_dir.d(), 1.0, res + oz);   // G dir.x was just formed by solve_system
  else sgemv(false, q, n, -1.0, G.d(), q, dir, 1.0, res + oz);
  if (p > 0) {
    sgemv(true, p, n, 1.0, mA.d(), p, dir + n, 1.0, res);                  // res.x += A' y
    dev_scale_copy(ctx, p, tau_dir, mb.d(), res + n);                           // res.y = b tau - A x
    sgemv(false, p, n, -1.0, mA.d(), p, dir, 1.0, res + n);
  }
  for (size_t k = 0; k < cones.size(); ++k) {   // res.s_k = H_k prim_dir_k + dual_dir_k
    Cone* ck = cones[k];
    const int o = offs[k], dk = ck->dim;
    const double* prim = ck->use_dual_barrier ? dir + oz + o : dir + os + o;
    const double* dual = ck->use_dual_barrier ? dir + os + o : dir + oz + o;
    if (const int used = run_hess_prod(k, res + os + o, q, prim, q, 1)) {   // (PosSemidefTri: hess_prod_slow! = hess_prod!)
      dev_axpby(ctx, offs[k + used] - o, 1.0, dual, 1.0, res + os + o);
      k += used - 1;
      continue;
    }
    ck->hess_prod_slow(res + os + o, q, prim, q, 1);
    dev_axpby(ctx, dk, 1.0, dual, 1.0, res + os + o);
  }
  // (the dots come AFTER the cone loop: cone oracles use ctx.dscal for their own scalar read-backs)
  double* dsc = ctx.dscal.d();
  dev_dot(ctx, n, mc.d(), dir, dsc);
  dev_dot(ctx, q, mh.d(), dir + oz, dsc + 1);
  if (p > 0) dev_dot(ctx, p, mb.d(), dir + n, dsc + 2);
  ctx.d2h(ctx.h_pinned + 8, dsc, 3 * d);
  ctx.sync();
  double hz_dir = ctx.h_pinned[9];
  if (dist()) allreduce_host(&hz_dir, 1, 0, 4);
  Scal out;
  out.tau = -ctx.h_pinned[8] - hz_dir - kap_dir - (p > 0 ? ctx.h_pinned[10] : 0.0);
  out.kap = mu / taubar * tau_dir / taubar + kap_dir;
  return out;
}

bool SysSolver::check_cone_points(const double* h, double min_prox, double prox_bound, bool use_max_prox, double nup1, double* prox_out,
                                  int* n_loaded, double* irtmu_out) {
  const double EPS = 2.220446049250313e-16;
  *n_loaded = 0;
  *prox_out = 0.0;
  *irtmu_out = 0.0;
  const double* hz = h;
  const double tau = h[q];
  const double* hs = h + q + 1;
  const double kap = h[2 * q + 1];
  const double proxsqr_bound = prox_bound * prox_bound;
  const double taukap = tau * kap;
  if (std::min(std::min(tau, kap), taukap) < EPS) return false;                     // search.jl:86-88 (tau, kap: same on every rank)
  const size_t nc = cones.size();
  std::vector<double> szk(nc);
  double szsum = 0.0;
  bool ok = true;
  for (size_t k = 0; k < nc; ++k) {                                                   // :90-95
    const double* a = hz + offs[k];
    const double* b = hs + offs[k];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
    const int dk = cones[k]->dim;
    int i = 0;
    for (; i + 3 < dk; i += 4) { d0 += a[i] * b[i]; d1 += a[i + 1] * b[i + 1]; d2 += a[i + 2] * b[i + 2]; d3 += a[i + 3] * b[i + 3]; }
    for (; i < dk; ++i) d0 += a[i] * b[i];
    szk[k] = (d0 + d1) + (d2 + d3);
    if (szk[k] < EPS) ok = false;
    szsum += szk[k];
  }
  // (sharded: this rank's cones only -- every decision below is taken on all-reduced quantities, so that all
  //  ranks leave through the same exit and issue the same sequence of collectives)
  // (screen_sz_ is the screen's DEVICE tree sum of <z, s>, not the four-way host sum above, and its failure flag carries
  //  the screen's margin -- so mu, irtmu and the loaded scaled point of a survivor differ in their last bits between HYP_DIST_FUSED=1
  //  and 0 and from the single-process walk: the fused sharded path is reproducible run to run, not bitwise against the unfused one;
  //  tests/test_hip_distributed.py compares the two at 1e-9 in the objective, docs/NUMERICS.md says so)
  if (dist() && screen_pass_g_ >= 0 && fused_ok()) {   // a survivor of the candidate screen: the screen has exchanged exactly these two numbers
    szsum = screen_sz_[screen_pass_g_];
    ok = (screen_szfail_[screen_pass_g_] < 0.5);
  } else if (dist()) {
    double v[2] = {szsum, ok ? 0.0 : 1.0};
    allreduce_host(v, 2, 0, 9);
    szsum = v[0];
    ok = (v[1] < 0.5);
  }
  if (!ok) return false;
  const double mu = (szsum + taukap) / nup1;                                          // :97-100
  if (mu < EPS) return false;
  const double taukap_rel = taukap / mu;                                              // :102-108
  if (taukap_rel < min_prox) return false;
  const double taukap_proxsqr = (taukap_rel - 1.0) * (taukap_rel - 1.0);
  if (taukap_proxsqr > proxsqr_bound) return false;
  for (size_t k = 0; k < nc; ++k) {                                                   // :110-116
    const double nu_k = cones[k]->nu;
    const double rel = szk[k] / (mu * nu_k);
    if (rel < min_prox || nu_k * (rel - 1.0) * (rel - 1.0) > proxsqr_bound) ok = false;
  }
  // (sharded: a rank whose cones fail this test does not leave alone -- it skips its cone work below and reports
  //  the failure in the trial's closing all-reduce, so the ranks stay in step without a collective of their own here)
  if (!ok && !dist()) return false;
  const bool local_reject = !ok;
  const double irtmu = 1.0 / std::sqrt(mu);
  *irtmu_out = irtmu;
  cand_d.ensure((size_t)(2 * q + 2) * sizeof(double));
  ctx.h2d(cand_d.p, h, (size_t)(2 * q + 2) * sizeof(double));
  const double* dz = cand_d.d();
  const double* dsv = cand_d.d() + q + 1;
  double agg = dist() ? 0.0 : taukap_proxsqr;
  // With several cones, the feasibility work of ALL of them is queued first (two chains per cone on the two
  // streams) and one synchronisation collects every flag; the reference's sweep (:118-136) visits the cones in
  // order and stops at the first failure, which only the points loaded beyond it could tell apart -- and nothing
  // reads those before the next candidate reloads them.
  std::vector<char> launched(nc, 0);
  bool run_feas = false;
  if (nc > 1 && !local_reject && psd_runs.size() == 1 && psd_runs[0].k0 == 0 && (size_t)psd_runs[0].count == nc) {
    // one run of equal PSD cones is the whole model (config 4): load_point / load_dual_point of all members are two copies
    // over the arena, the 2 x nc feasibility Choleskys two batched factorizations, their infos one read-back
    const PsdRun& r = psd_runs[0];
    bool usable = true;
    for (size_t k = 0; k < nc && usable; ++k) {
      PsdCone* c = static_cast<PsdCone*>(cones[k]);
      usable = !c->use_dual_barrier && c->U.p == (void*)(r.U + (long)k * r.side * r.side) && c->point.p == (void*)(r.point + (long)k * c->dim);
    }
    if (usable) {
      const int sd = r.side, B = r.count, dm = cones[0]->dim;
      const long s2 = (long)sd * sd, tot = (long)B * dm;
      if (irtmu == 1.0) ctx.d2d(r.point, dsv, (size_t)tot * sizeof(double));          // Cone::load_point (Cones.jl:157-166)
      else dev_scale_copy(ctx, (int)tot, irtmu, dsv, r.point);
      ctx.d2d(r.dual, dz, (size_t)tot * sizeof(double));                               // Cone::load_dual_point
      run_ws1.ensure((size_t)B * s2 * sizeof(double));
      run_info.ensure((size_t)2 * B * sizeof(int));
      svec_unpack(ctx, sd, B, r.point, dm, r.X);                                       // PsdCone::update_feas (possemideftri.jl:80-90)
      ctx.d2d(r.U, r.X, (size_t)B * s2 * sizeof(double));
      potrf_upper_batched(ctx, sd, r.U, sd, s2, B, nullptr, run_info.i());
      svec_unpack(ctx, sd, B, r.dual, dm, run_ws1.d());                                // PsdCone::is_dual_feas (:92-95)
      potrf_upper_batched(ctx, sd, run_ws1.d(), sd, s2, B, nullptr, run_info.i() + B);
      HYP_REQUIRE(64 + 2 * (size_t)B < 8192, "check_cone_points: too many cones for the pinned info words");
      ctx.d2h(ctx.h_info + 64, run_info.p, (size_t)2 * B * sizeof(int));
      ctx.sync();
      for (int g = 0; g < B; ++g) {
        PsdCone* c = static_cast<PsdCone*>(cones[g]);
        c->reset_data();
        c->is_feas_ = (ctx.h_info[64 + g] == 0);
        c->dual_feas_ = (ctx.h_info[64 + B + g] == 0);
        c->feas_updated = true;
        c->dual_cached = true;
        c->inv_ready = false;
      }
      *n_loaded = (int)nc;
      run_feas = true;
    }
  }
  if (nc > 1 && !local_reject && !run_feas) {
    for (size_t k = 0; k < nc; ++k) {
      Cone* ck = cones[k];
      ck->load_point((ck->use_dual_barrier ? dz : dsv) + offs[k], irtmu);
      ck->load_dual_point((ck->use_dual_barrier ? dsv : dz) + offs[k]);
      ck->reset_data();
      launched[k] = ck->prefetch_launch((int)k) ? 1 : 0;
    }
    hipEvent_t e1 = ctx.aux_event(3);
    HYP_CHECK(hipEventRecord(e1, ctx.stream2));
    HYP_CHECK(hipStreamWaitEvent(ctx.stream, e1, 0));
    ctx.sync();
    for (size_t k = 0; k < nc; ++k)
      if (launched[k]) cones[k]->prefetch_finish((int)k);
    *n_loaded = (int)nc;
  }
  // several cones, all with the generic proximity test: after the batched feasibility flags, the scalar products of
  // check_numerics / get_proxsqr are queued for a chunk of cones and read back with one synchronisation (three per cone
  // otherwise); the tests are then evaluated in the reference's order.  Same decision: a trial is rejected iff some cone fails.
  bool single_loaded = false;
  if (nc == 1 && !local_reject && cones[0]->prox_batchable()) {   // one cone: load it here, so that its scalars also come back in one read
    Cone* ck = cones[0];
    ck->load_point((ck->use_dual_barrier ? dz : dsv), irtmu);
    ck->load_dual_point((ck->use_dual_barrier ? dsv : dz));
    ck->reset_data();
    *n_loaded = 1;
    if (!dist() && ck->early_reject(irtmu, proxsqr_bound)) {   // (before the cone's feasibility work is queued)
      static const bool edbg = [] { const char* e = getenv("HYP_TRIAL_DBG"); return e && e[0] == '1'; }();
      if (edbg) fprintf(stderr, "[trial] early reject (%s)\n", ck->is_feas() ? "bound" : "infeasible");
      return false;
    }
    ck->prefetch_feas();
    single_loaded = true;
  }
  bool batched_prox = ((nc > 1 || single_loaded) && !local_reject);
  for (size_t k = 0; k < nc && batched_prox; ++k) batched_prox = cones[k]->prox_batchable();
  if (batched_prox) {
    for (size_t k = 0; k < nc && ok; ++k)
      if (!(cones[k]->is_feas() && cones[k]->is_dual_feas())) ok = false;          // (answered by the prefetch above)
    static const bool tdbg = [] { const char* e = getenv("HYP_TRIAL_DBG"); return e && e[0] == '1'; }();
    if (tdbg && !ok) fprintf(stderr, "[trial] infeasible\n");
    // candidates far outside the neighbourhood: rejected on a lower bound of the proximity value, before any Hessian is
    // assembled or factored for them (Cone::prox_lower_bound; single process only -- sharded ranks leave together below)
    // (not for a survivor of the side-by-side screen: the screen has just evaluated this very bound for it, batched with the
    //  other candidates, and found it within the neighbourhood -- the proximity value itself decides below)
    if (ok && !dist() && nc <= 2 && !screen_survivor) {   // (each bound is a read-back of its own: for models of one or two large cones)
      for (size_t k = 0; k < nc && ok; ++k) {
        double lb = 0.0;
        const bool have = cones[k]->prox_lower_bound(irtmu, proxsqr_bound * (1.0 + 1e-9), &lb);
        if (have && lb > proxsqr_bound * (1.0 + 1e-9)) ok = false;
        if (tdbg) fprintf(stderr, "[trial] bound %s %.4g (limit %.4g)%s\n", have ? "=" : "n/a", lb, proxsqr_bound, ok ? "" : " -> rejected");
      }
    }
    if (ok) {
      group_inverses();   // runs of equal PSD cones: U^-1, X^-1 of all members in one batched launch sequence
      const double gtol = std::sqrt(std::sqrt(EPS)), Htol = 10 * std::sqrt(gtol), negtol = std::sqrt(EPS);
      bool run_done = false;
      if (psd_runs.size() == 1 && psd_runs[0].k0 == 0 && (size_t)psd_runs[0].count == nc) {
        // one run is the whole model (config 4): its members' scalar products in ~16 launches and one read-back
        const PsdRun& r = psd_runs[0];
        bool usable = true;
        for (size_t k = 0; k < nc && usable; ++k) {
          PsdCone* c = static_cast<PsdCone*>(cones[k]);
          usable = c->inv_ready && c->U.p == (void*)(r.U + (long)k * r.side * r.side);
        }
        if (usable) {
          prox_scal.ensure(3 * nc * sizeof(double));
          HYP_REQUIRE(3 * nc <= ctx.h_pinned_n, "check_cone_points: too many cones for the pinned staging buffer");
          run_prox_launch(r, irtmu, prox_scal.d());
          ctx.d2h(ctx.h_pinned, prox_scal.p, 3 * nc * sizeof(double));
          ctx.sync();
          for (size_t k = 0; k < nc; ++k) {
            const double dk = cones[k]->dim, nuk = cones[k]->nu;
            const double* hp = ctx.h_pinned + 3 * k;
            if (std::fabs(1 + hp[0] / nuk) > gtol * dk || std::fabs(1 - hp[1] / nuk) > Htol * dk) {   // Cones.jl:273-290
              if (tdbg) fprintf(stderr, "[trial] run: numerics of cone %zu (%.3g, %.3g)\n", k, 1 + hp[0] / nuk, 1 - hp[1] / nuk);
              ok = false;
              break;
            }
            const double pk = (hp[2] < -negtol * dk) ? INFINITY : std::fabs(hp[2]);                                        // Cones.jl:294-310
            agg = use_max_prox ? std::max(agg, pk) : agg + pk;
            if (!dist() && !(agg < proxsqr_bound)) {
              if (tdbg) fprintf(stderr, "[trial] run: proximity %.6g at cone %zu (bound %.6g)\n", agg, k, proxsqr_bound);
              ok = false;
              break;
            }
          }
          if (tdbg && ok) fprintf(stderr, "[trial] run: accepted so far, aggregate %.6g\n", agg);
          run_done = true;
        }
      }
      // in chunks of 8 cones: most rejected trials fail the proximity bound at one of the first cones, and the
      // reference's sweep stops there -- a chunk bounds the work queued beyond that point
      const size_t CHK = 8;
      prox_scal.ensure(3 * std::max(CHK, nc) * sizeof(double));
      for (size_t k0 = 0; k0 < nc && ok && !run_done; k0 += CHK) {
        const size_t k1 = std::min(nc, k0 + CHK);
        for (size_t k = k0; k < k1 && ok; ++k)
          if (!cones[k]->prox_launch(irtmu, prox_scal.d() + 3 * (k - k0))) ok = false;
        if (!ok) break;
        ctx.d2h(ctx.h_pinned, prox_scal.p, 3 * (k1 - k0) * sizeof(double));
        ctx.sync();
        for (size_t k = k0; k < k1; ++k) {
          const double dk = cones[k]->dim, nuk = cones[k]->nu;
          const double* hp = ctx.h_pinned + 3 * (k - k0);
          if (std::fabs(1 + hp[0] / nuk) > gtol * dk || std::fabs(1 - hp[1] / nuk) > Htol * dk) { ok = false; break; }   // Cones.jl:273-290
          const double pk = (hp[2] < -negtol * dk) ? INFINITY : std::fabs(hp[2]);                                        // Cones.jl:294-310
          agg = use_max_prox ? std::max(agg, pk) : agg + pk;
          if (!dist() && !(agg < proxsqr_bound)) { ok = false; break; }
        }
      }
    }
  }
  for (size_t k = 0; k < nc && !local_reject && !batched_prox; ++k) {                 // :118-136
    Cone* ck = cones[k];
    if (nc == 1 && !single_loaded) {
      ck->load_point((ck->use_dual_barrier ? dz : dsv) + offs[k], irtmu);
      ck->load_dual_point((ck->use_dual_barrier ? dsv : dz) + offs[k]);
      ck->reset_data();
      ck->prefetch_feas();
      *n_loaded = 1;
    }
    bool in_prox = false;
    if (ck->is_feas() && ck->is_dual_feas() && ck->check_numerics()) {
      const double pk = ck->get_proxsqr(irtmu, use_max_prox);
      agg = use_max_prox ? std::max(agg, pk) : agg + pk;
      in_prox = dist() ? true : (agg < proxsqr_bound);   // (sharded: the bound is applied to the all-reduced aggregate)
    }
    if (!in_prox) {
      ok = false;
      break;
    }
  }
  if (dist()) {
    double v[2] = {ok ? 0.0 : 1.0, agg};   // [any failure, proximity aggregate]
    if (use_max_prox) {
      allreduce_host(v, 2, 1, 10);             // one MAX serves both
    } else {
      allreduce_host(v, 2, 0, 10);             // SUM: failures count, proximities add
    }
    ok = (v[0] < 0.5);
    agg = use_max_prox ? std::max(taukap_proxsqr, v[1]) : taukap_proxsqr + v[1];
    if (ok) ok = (agg < proxsqr_bound);
  }
  if (!ok) return false;
  *prox_out = std::sqrt(agg);
  return true;
}

// ---------------------------------------------------------------------------------------------
// Side-by-side screening of line-search candidates (see syssolver.hpp).  The schedule walk of search.jl:46-69 visits the
// candidates one after the other, and each visit is a latency-bound chain of small kernels (two side-200 factorizations, one
// two-sided product, two read-backs: ~200 us on an otherwise idle chip) behind 20 us of host arithmetic that forms the
// candidate; config 2 walks 7.8 candidates per iteration.  The candidates are known in advance (the schedule is fixed), so the
// tests that REJECT -- search.jl:86-116 on the scalars, PosSemidefTri's update_feas / is_dual_feas (possemideftri.jl:80-95)
// and the proximity bound (Cones.jl:294-310, in the inverse-free form of PsdCone::prox_lower_bound) -- run for all remaining
// candidates of the schedule at once, batch = candidates.  The walk then skips what the screen rejected and hands the first
// survivor to check_cone_points: acceptance, the cone's state afterwards and the trial count are those of the sequential walk.
// The screen's numbers are not bitwise the sequential test's (tree-summed <z, s>, factor of the unscaled primal point, GEMM
// route for U Z U'), so it rejects only beyond a relative margin of 1e-6 on the scalar tests and 1e-5 on the proximity value.
// ---------------------------------------------------------------------------------------------
struct ScreenForm {
  double alpha[SysSolver::SCREEN_MAX], a2[SysSolver::SCREEN_MAX], am1[SysSolver::SCREEN_MAX], am1s[SysSolver::SCREEN_MAX];
  double tau[SysSolver::SCREEN_MAX], kap[SysSolver::SCREEN_MAX];   // the candidates' tau / kap (formed on the host)
  int mode;   // 0: pt + alpha dc; 1: pt + (alpha dp + am1 dc); 2: pt + (alpha dc + a2 dca); 3: all four (combined.jl:124-170)
};
// candidate g = update_stepper_points(alpha_g) over the ztsk rows, the host loop's operations in the host loop's order (no
// contraction: a survivor downloaded from here is bitwise the candidate the host forms)
__global__ __launch_bounds__(256) void screen_form_kernel(int q, const double* __restrict__ pt, const double* __restrict__ dc,
                                                          const double* __restrict__ dp, const double* __restrict__ dca,
                                                          const double* __restrict__ dpa, ScreenForm f, double* __restrict__ cands) {
#pragma clang fp contract(off)   // (device code contracts a * b + c into an fma by default, and ROCm's __dadd_rn is a plain +)
  const int len = 2 * q + 2;
  const int i = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (i >= len) return;
  double v;
  if (i == q) v = f.tau[g];
  else if (i == len - 1) v = f.kap[g];
  else {
    const double al = f.alpha[g];
    if (f.mode == 0) v = pt[i] + al * dc[i];
    else if (f.mode == 1) v = pt[i] + (al * dp[i] + f.am1[g] * dc[i]);
    else if (f.mode == 2) v = pt[i] + (al * dc[i] + f.a2[g] * dca[i]);
    else v = pt[i] + (((al * dp[i] + f.a2[g] * dpa[i]) + f.am1[g] * dc[i]) + f.am1s[g] * dca[i]);
  }
  cands[(long)g * len + i] = v;
}
struct ScreenScal { double v[SysSolver::SCREEN_MAX]; };
// sz_g = <z_g, s_g>; scal_g = 1 / mu_g with mu_g = (sz_g + taukap_g) / nup1   (search.jl:90-100); one workgroup per candidate
__global__ __launch_bounds__(1024) void screen_dot_kernel(int dm, const double* __restrict__ cands, long len, long off_s, ScreenScal taukap,
                                                          double nup1, double* __restrict__ sz, double* __restrict__ scal) {
  __shared__ double red[1024];
  const int g = blockIdx.x;
  const double* z = cands + (long)g * len;
  const double* sv = z + off_s;
  double acc = 0.0;
  for (int i = threadIdx.x; i < dm; i += 1024) acc = fma(z[i], sv[i], acc);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sz[g] = red[0];
    scal[g] = nup1 / (red[0] + taukap.v[g]);
  }
}
// out[g] = || scal_g W_g - I ||_F^2 over the full side x side matrix W_g (= the svec sum of psd_prox_direct_kernel for a symmetric W)
__global__ __launch_bounds__(1024) void screen_prox_kernel(int side, const double* __restrict__ W, long stride, const double* __restrict__ scal_d,
                                                           double* __restrict__ out) {
  __shared__ double red[1024];
  const int g = blockIdx.x;
  const double* w = W + (long)g * stride;
  const double scal = scal_d[g];
  const int tot = side * side;
  double s = 0.0;
  for (int i = threadIdx.x; i < tot; i += 1024) {
    const int c = i / side, r = i - c * side;
    const double v = scal * w[i] - (r == c ? 1.0 : 0.0);
    s = fma(v, v, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[g] = red[0];
}

int SysSolver::screen_mode() const {
  static const bool on = [] { const char* e = getenv("HYP_SEARCH_SCREEN"); return !(e && e[0] == '0'); }();
  static const bool lb_on = [] { const char* e = getenv("HYP_PROX_LB"); return !(e && e[0] == '0'); }();
  static const bool run_on = [] { const char* e = getenv("HYP_SEARCH_SCREEN_RUN"); return !(e && e[0] == '0'); }();
  // (sharded: the run form only -- every rank screens ITS cones and two all-reduces of a few numbers per candidate make the
  //  verdict the same on all ranks; HYP_SEARCH_SCREEN_DIST=0: off)
  static const bool dist_on = [] { const char* e = getenv("HYP_SEARCH_SCREEN_DIST"); return !(e && e[0] == '0'); }();
  if (!on || !lb_on || cones.empty() || (dist() && !dist_on)) return 0;
  if (cones.size() == 1 && !dist())   // (the one-cone form lays its buffers out for the whole schedule: all of it must fit)
    return (cones[0]->kind == CONE_PSD && !cones[0]->use_dual_barrier && static_cast<const PsdCone*>(cones[0])->side >= 32 &&
            screen_kmax() == SCREEN_MAX) ? 1 : 0;
  // (equal primal-barrier PosSemidefTri cones, back to back: the screen reads the candidates only, no cone state, so the cones need
  //  not form a grouped run; sharded, one cone per rank -- the weak-scaling workload -- is the same thing with one matrix per
  //  candidate)
  if (!run_on) return 0;
  const int sd0 = (cones[0]->kind == CONE_PSD) ? static_cast<const PsdCone*>(cones[0])->side : -1;
  for (const Cone* ck : cones)
    if (ck->kind != CONE_PSD || ck->use_dual_barrier || static_cast<const PsdCone*>(ck)->side != sd0) return 0;
  return screen_kmax() >= 2 ? 2 : 0;   // (too many cones / too large a side for a batch of two: the sequential walk)
}

// Candidates per batch that the screen's buffers admit for THIS model (all cones PosSemidefTri of one side): the flags and sums of the
// K x B (candidate, cone) pairs come back through the pinned blocks (64 + 2 K B of h_info's 8192 ints, 128 + 3 K B of h_pinned's
// doubles) and screen_buf holds K candidates + 6 K B side x side matrices, bounded by a byte budget (HYP_SCREEN_MB, default 2048).
// A model beyond these walks the schedule in smaller batches, or sequentially (screen_mode() = 0) when not even two candidates fit.
int SysSolver::screen_kmax() const {
  static const long budget = [] { const char* e = getenv("HYP_SCREEN_MB"); return (e ? std::max(1L, atol(e)) : 2048L) << 20; }();
  if (cones.empty() || cones[0]->kind != CONE_PSD) return 0;
  const long B = (long)cones.size(), sd = static_cast<const PsdCone*>(cones[0])->side, s2 = sd * sd, len = 2L * q + 2;
  long k = SCREEN_MAX;
  k = std::min(k, (8192L - 64) / (2 * B));
  k = std::min(k, ((long)ctx.h_pinned_n - 128) / (3 * B));
  k = std::min(k, budget / ((len + 6 * B * s2 + 3 * B) * (long)sizeof(double)));
  return (int)std::max(k, 0L);
}

// ---- the same screen for a model that is ONE RUN of equal PosSemidefTri cones (config 4: 64 cones of side 80): batch =
// candidates x cones.  Per candidate g and cone k the device returns <z, s>, both factorization flags, || W ||_F^2 and tr W for
// W = U Z U' (U' U = smat(s_k)); mu_g needs the sum over the cones, so the proximity value is finished on the host:
// || W / mu - I ||_F^2 = ||W||^2 / mu^2 - 2 tr W / mu + side.  Aggregation and bound as in check_cone_points (search.jl:118-136).
__global__ __launch_bounds__(256) void screen_dot_run_kernel(int dm, const double* __restrict__ cands, long len, long off_s, int B,
                                                             double* __restrict__ sz) {
  __shared__ double red[256];
  const int k = blockIdx.x, g = blockIdx.y;
  const double* z = cands + (long)g * len + (long)k * dm;
  const double* sv = z + off_s;
  double acc = 0.0;
  for (int i = threadIdx.x; i < dm; i += 256) acc = fma(z[i], sv[i], acc);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) sz[(long)g * B + k] = red[0];
}
// out2[2 m] = || W_m ||_F^2, out2[2 m + 1] = tr W_m
__global__ __launch_bounds__(256) void screen_norm_trace_kernel(int side, const double* __restrict__ W, long stride, double* __restrict__ out2) {
  __shared__ double r0[256], r1[256];
  const long m = blockIdx.x;
  const double* w = W + m * stride;
  const int tot = side * side;
  double s = 0.0, t = 0.0;
  for (int i = threadIdx.x; i < tot; i += 256) {
    const double v = w[i];
    s = fma(v, v, s);
    const int c = i / side;
    if (i - c * side == c) t += v;
  }
  r0[threadIdx.x] = s;
  r1[threadIdx.x] = t;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { r0[threadIdx.x] += r0[threadIdx.x + off]; r1[threadIdx.x] += r1[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out2[2 * m] = r0[0]; out2[2 * m + 1] = r1[0]; }
}

void SysSolver::screen_candidates_run(const double* cd, int K, const double* tau, const double* kap, double min_prox, double prox_bound,
                                      double nup1, bool use_max_prox, char* rej) {
  const double EPS = 2.220446049250313e-16;
  const int kcap = screen_kmax();
  HYP_REQUIRE(K >= 1 && K <= kcap, "screen_candidates_run: batch size");
  const int sd = static_cast<const PsdCone*>(cones[0])->side, B = (int)cones.size(), dm = cones[0]->dim;
  const long len = 2L * q + 2, s2 = (long)sd * sd, KB = (long)K * B, MB = (long)kcap * B;
  const double proxsqr_bound = prox_bound * prox_bound;
  double* P = screen_buf.d() + (long)kcap * len;   // layout as in screen_candidates, matrices per (candidate, cone)
  double* D = P + KB * s2;
  double* UT = D + KB * s2;
  double* Z = UT + KB * s2;
  double* T = Z + KB * s2;
  double* W = T + KB * s2;
  double* outv = screen_buf.d() + (long)kcap * len + 6L * MB * s2;   // [sz (MB) | norm, trace (2 MB)]
  screen_info.ensure((size_t)2 * MB * sizeof(int));
  hipLaunchKernelGGL(screen_dot_run_kernel, dim3(B, K), dim3(256), 0, ctx.stream, dm, cd, len, (long)q + 1, B, outv);
  svec_unpack_grouped(ctx, sd, K, B, cd + q + 1, len, dm, P);    // PsdCone::update_feas (possemideftri.jl:80-90) of the unscaled s_k
  svec_unpack_grouped(ctx, sd, K, B, cd, len, dm, D);            // PsdCone::is_dual_feas (:92-95)
  ctx.d2d(Z, D, (size_t)KB * s2 * sizeof(double));
  potrf_upper_batched(ctx, sd, P, sd, s2, (int)(2 * KB), nullptr, screen_info.i());
  dev_zero_strict_lower(ctx, sd, P, sd, (int)KB, s2);
  dev_transpose(ctx, sd, sd, P, sd, UT, sd, (int)KB, s2, s2);
  GemmArgs a{};   // T = Z U'
  a.M = sd; a.N = sd; a.K = sd; a.A = Z; a.lda = sd; a.strideA = s2; a.B = UT; a.ldb = sd; a.strideB = s2; a.C = T; a.ldc = sd; a.strideC = s2;
  a.alpha = 1; a.beta = 0; a.tri = GEMM_FULL; a.krange = KR_GE_N; a.batch = (int)KB;
  gemm(ctx, true, a);
  GemmArgs b{};   // W = U T
  b.M = sd; b.N = sd; b.K = sd; b.A = UT; b.lda = sd; b.strideA = s2; b.B = T; b.ldb = sd; b.strideB = s2; b.C = W; b.ldc = sd; b.strideC = s2;
  b.alpha = 1; b.beta = 0; b.tri = GEMM_FULL; b.krange = KR_GE_M; b.batch = (int)KB;
  gemm(ctx, true, b);
  hipLaunchKernelGGL(screen_norm_trace_kernel, dim3((unsigned)KB), dim3(256), 0, ctx.stream, sd, W, s2, outv + MB);
  HYP_CHECK(hipGetLastError());
  HYP_REQUIRE(64 + 2 * KB <= 8192 && 128 + 3 * (size_t)KB <= ctx.h_pinned_n, "screen_candidates_run: too many cones for the pinned buffers");
  int* hi = ctx.h_info + 64;
  double* hv = ctx.h_pinned + 128;
  ctx.d2h(hi, screen_info.p, (size_t)2 * KB * sizeof(int));
  ctx.d2h(hv, outv, (size_t)KB * sizeof(double));
  ctx.d2h(hv + KB, outv + MB, (size_t)2 * KB * sizeof(double));
  ctx.sync();
  ++screen_count;
  const double M = 1e-6;
  auto lt = [M](double x, double y) { return x < y - M * std::fabs(y); };
  auto gt = [M](double x, double y) { return x > y + M * std::fabs(y); };
  const double limit = proxsqr_bound * (1.0 + 1e-9);
  // first exchange (sharded): the candidates' <z, s> over all ranks' cones, and whether any cone anywhere fails the sign test
  std::vector<double> v1(2 * (size_t)K, 0.0), v2(3 * (size_t)K, 0.0);
  for (int g = 0; g < K; ++g) {
    const double* szk = hv + (long)g * B;
    for (int k = 0; k < B; ++k) {
      if (lt(szk[k], EPS)) v1[K + g] = 1.0;
      v1[g] += szk[k];
    }
  }
  if (dist()) allreduce_host(v1.data(), 2 * K, 0, 7);
  if (dist()) {   // (a survivor's sequential test takes its <z, s> over all ranks and the sign flag from here: one exchange fewer per survivor)
    for (int g = 0; g < K; ++g) { screen_sz_[g] = v1[g]; screen_szfail_[g] = v1[K + g]; }
  }
  // second exchange: [a cone fails a test | no verdict (a value that is not a number) | proximity aggregate of this rank's cones]
  std::vector<double> mus(K, 0.0), tkp(K, 0.0);
  std::vector<char> scal_rej(K, 0);
  for (int g = 0; g < K; ++g) {
    const double taukap = tau[g] * kap[g];
    const double* szk = hv + (long)g * B;
    const double* nt = hv + KB + 2L * g * B;
    const double mu = (v1[g] + taukap) / nup1, taukap_rel = taukap / mu;
    mus[g] = mu;
    tkp[g] = (taukap_rel - 1.0) * (taukap_rel - 1.0);
    if (std::min(std::min(tau[g], kap[g]), taukap) < EPS || v1[K + g] > 0.5 || lt(mu, EPS) || lt(taukap_rel, min_prox) ||
        gt(tkp[g], proxsqr_bound)) {   // (the same decision on every rank: replicated scalars and all-reduced sums)
      scal_rej[g] = 1;
      continue;
    }
    double agg = 0.0;
    for (int k = 0; k < B; ++k) {
      const double nu_k = cones[k]->nu, rel = szk[k] / (mu * nu_k);
      if (lt(rel, min_prox) || gt(nu_k * (rel - 1.0) * (rel - 1.0), proxsqr_bound)) { v2[g] = 1.0; break; }
      // A failed factorization rejects without a margin although the sequential test factors s_k / sqrt(mu), not s_k: the two can
      // disagree only where a pivot is at rounding level, i.e. smat(s_k) or smat(z_k) is numerically singular -- and then W = U Z U' / mu
      // has an eigenvalue at rounding level, || W - I ||_F^2 >= 1 - O(eps cond) > prox_bound^2 = 0.98, so the sequential walk
      // rejects the candidate on proximity (search.jl:126-136) whichever way ITS factorization falls.  Same accepted step either way.
      if (hi[(long)g * B + k] != 0 || hi[KB + (long)g * B + k] != 0) { v2[g] = 1.0; break; }
      const double v = nt[2 * k] / (mu * mu) - 2.0 * nt[2 * k + 1] / mu + (double)sd;
      if (!(v == v) || !(v < INFINITY)) { v2[K + g] = 1.0; break; }
      agg = use_max_prox ? std::max(agg, v) : agg + v;
    }
    v2[2 * K + g] = (v2[g] > 0.5 || v2[K + g] > 0.5) ? 0.0 : std::max(agg, 0.0);
  }
  if (dist()) allreduce_host(v2.data(), 3 * K, use_max_prox ? 1 : 0, 8);
  for (int g = 0; g < K; ++g) {
    bool rj = scal_rej[g] != 0 || v2[g] > 0.5;
    if (!rj && !(v2[K + g] > 0.5)) {
      const double agg = use_max_prox ? std::max(tkp[g], v2[2 * K + g]) : tkp[g] + v2[2 * K + g];
      if (agg / (1.0 + 1e-5) > limit) rj = true;
    }
    rej[g] = rj ? 1 : 0;
    screen_rejected += rej[g];
    static const bool sdbg = [] { const char* e = getenv("HYP_TRIAL_DBG"); return e && e[0] == '1'; }();
    if (sdbg) fprintf(stderr, "[screen] cand %d: scalars %d, cone test %g, no verdict %g, aggregate %.6g -> %s\n", g, (int)scal_rej[g], v2[g],
                      v2[K + g], use_max_prox ? std::max(tkp[g], v2[2 * K + g]) : tkp[g] + v2[2 * K + g], rj ? "rejected" : "passed on");
  }
}

void SysSolver::screen_candidates(const double* cd, int K, const double* tau, const double* kap, double min_prox, double prox_bound,
                                  double nup1, char* rej) {
  const double EPS = 2.220446049250313e-16;
  HYP_REQUIRE(K >= 1 && K <= SCREEN_MAX, "screen_candidates: batch size");
  const PsdCone* pc = static_cast<const PsdCone*>(cones[0]);
  const int sd = pc->side, dm = pc->dim;
  const long len = 2L * q + 2, s2 = (long)sd * sd;
  const double proxsqr_bound = prox_bound * prox_bound;
  ScreenScal tk{};
  for (int g = 0; g < K; ++g) tk.v[g] = tau[g] * kap[g];
  // device layout behind the candidates (screen_buf, laid out by search_alpha): 2 K matrices (K primal, K dual: ONE batched
  // factorization) | U' | smat(dual) | Z U' | U Z U' | K x [<z, s>, 1 / mu, proximity value]
  double* P = screen_buf.d() + (long)SCREEN_MAX * len;
  double* D = P + K * s2;
  double* UT = D + K * s2;
  double* Z = UT + K * s2;
  double* T = Z + K * s2;
  double* W = T + K * s2;
  double* outv = screen_buf.d() + (long)SCREEN_MAX * len + 6L * SCREEN_MAX * s2;
  screen_info.ensure((size_t)2 * SCREEN_MAX * sizeof(int));
  hipLaunchKernelGGL(screen_dot_kernel, dim3(K), dim3(1024), 0, ctx.stream, dm, cd, len, (long)q + 1, tk, nup1, outv, outv + SCREEN_MAX);
  svec_unpack(ctx, sd, K, cd + q + 1, len, P);              // PsdCone::update_feas (possemideftri.jl:80-90), of the UNSCALED s:
                                                            // the factor of s / sqrt(mu) is this one over mu^(1/4)
  svec_unpack(ctx, sd, K, cd, len, D);                      // PsdCone::is_dual_feas (:92-95): z sits at the head of a candidate
  ctx.d2d(Z, D, (size_t)K * s2 * sizeof(double));
  potrf_upper_batched(ctx, sd, P, sd, s2, 2 * K, nullptr, screen_info.i());
  // PsdCone::prox_lower_bound: || U Z U' / mu - I ||_F^2 with U' U = smat(s)
  dev_zero_strict_lower(ctx, sd, P, sd, K, s2);
  dev_transpose(ctx, sd, sd, P, sd, UT, sd, K, s2, s2);
  GemmArgs a{};   // T = Z U'  (Z symmetric: its transpose form; U' lower triangular)
  a.M = sd; a.N = sd; a.K = sd; a.A = Z; a.lda = sd; a.strideA = s2; a.B = UT; a.ldb = sd; a.strideB = s2; a.C = T; a.ldc = sd; a.strideC = s2;
  a.alpha = 1; a.beta = 0; a.tri = GEMM_FULL; a.krange = KR_GE_N; a.batch = K;
  gemm(ctx, true, a);
  GemmArgs b{};   // W = U T  (op(A) = (U')' = U upper triangular)
  b.M = sd; b.N = sd; b.K = sd; b.A = UT; b.lda = sd; b.strideA = s2; b.B = T; b.ldb = sd; b.strideB = s2; b.C = W; b.ldc = sd; b.strideC = s2;
  b.alpha = 1; b.beta = 0; b.tri = GEMM_FULL; b.krange = KR_GE_M; b.batch = K;
  gemm(ctx, true, b);
  hipLaunchKernelGGL(screen_prox_kernel, dim3(K), dim3(1024), 0, ctx.stream, sd, W, s2, outv + SCREEN_MAX, outv + 2 * SCREEN_MAX);
  HYP_CHECK(hipGetLastError());
  int* hi = ctx.h_info + 64;
  double* hv = ctx.h_pinned + 64;
  ctx.d2h(hi, screen_info.p, (size_t)2 * K * sizeof(int));
  ctx.d2h(hv, outv, (size_t)3 * SCREEN_MAX * sizeof(double));
  ctx.sync();
  ++screen_count;
  const double M = 1e-6;
  auto lt = [M](double x, double y) { return x < y - M * std::fabs(y); };   // x < y beyond the margin (false for NaN)
  auto gt = [M](double x, double y) { return x > y + M * std::fabs(y); };
  const double limit = proxsqr_bound * (1.0 + 1e-9);
  for (int g = 0; g < K; ++g) {   // search.jl:86-116 in the order of check_cone_points
    const double taukap = tk.v[g], sz = hv[g], v = hv[2 * SCREEN_MAX + g];
    const double mu = (sz + taukap) / nup1, taukap_rel = taukap / mu, nu_k = pc->nu, rel = sz / (mu * nu_k);
    bool r = false;
    if (std::min(std::min(tau[g], kap[g]), taukap) < EPS) r = true;   // (exact: host scalars)
    else if (lt(sz, EPS) || lt(mu, EPS)) r = true;
    else if (lt(taukap_rel, min_prox) || gt((taukap_rel - 1.0) * (taukap_rel - 1.0), proxsqr_bound)) r = true;
    else if (lt(rel, min_prox) || gt(nu_k * (rel - 1.0) * (rel - 1.0), proxsqr_bound)) r = true;
    else if (hi[g] != 0 || hi[K + g] != 0) r = true;   // (no margin needed: see screen_candidates_run)
    else if (v == v && v < INFINITY && v / (1.0 + 1e-5) > limit) r = true;
    rej[g] = r ? 1 : 0;
    screen_rejected += rej[g];
  }
}

// The scalar tests at the head of check_cone_points (search.jl:86-116) for a candidate on the host, single process: false = rejected
// there; otherwise *irtmu = 1 / sqrt(mu) exactly as check_cone_points forms it.
bool SysSolver::cand_scalars(const double* h, double min_prox, double prox_bound, double nup1, double* irtmu) const {
  const double EPS = 2.220446049250313e-16;
  const double* hz = h;
  const double tau = h[q];
  const double* hs = h + q + 1;
  const double kap = h[2 * q + 1];
  const double proxsqr_bound = prox_bound * prox_bound;
  const double taukap = tau * kap;
  if (std::min(std::min(tau, kap), taukap) < EPS) return false;
  const size_t nc = cones.size();
  std::vector<double> szk(nc);
  double szsum = 0.0;
  for (size_t k = 0; k < nc; ++k) {
    const double* a = hz + offs[k];
    const double* b = hs + offs[k];
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
    const int dk = cones[k]->dim;
    int i = 0;
    for (; i + 3 < dk; i += 4) { d0 += a[i] * b[i]; d1 += a[i + 1] * b[i + 1]; d2 += a[i + 2] * b[i + 2]; d3 += a[i + 3] * b[i + 3]; }
    for (; i < dk; ++i) d0 += a[i] * b[i];
    szk[k] = (d0 + d1) + (d2 + d3);
    if (szk[k] < EPS) return false;
    szsum += szk[k];
  }
  const double mu = (szsum + taukap) / nup1;
  if (mu < EPS) return false;
  const double taukap_rel = taukap / mu;
  if (taukap_rel < min_prox) return false;
  if ((taukap_rel - 1.0) * (taukap_rel - 1.0) > proxsqr_bound) return false;
  for (size_t k = 0; k < nc; ++k) {
    const double nu_k = cones[k]->nu;
    const double rel = szk[k] / (mu * nu_k);
    if (rel < min_prox || nu_k * (rel - 1.0) * (rel - 1.0) > proxsqr_bound) return false;
  }
  *irtmu = 1.0 / std::sqrt(mu);
  return true;
}

// the x rows of update_stepper_points (combined.jl:124-170) on the device: the host mirror's operations in the host mirror's order
__global__ __launch_bounds__(256) void point_form_kernel(int n, const double* __restrict__ pt, const double* __restrict__ dc,
                                                         const double* __restrict__ dp, const double* __restrict__ dca,
                                                         const double* __restrict__ dpa, int mode, double al, double a2, double am1, double am1s,
                                                         double* __restrict__ out) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v;
  if (mode == 0) v = pt[i] + al * dc[i];
  else if (mode == 1) v = pt[i] + (al * dp[i] + am1 * dc[i]);
  else if (mode == 2) v = pt[i] + (al * dc[i] + a2 * dca[i]);
  else v = pt[i] + (((al * dp[i] + a2 * dpa[i]) + am1 * dc[i]) + am1s * dca[i]);
  out[i] = v;
}

// Round 6: the products calc_convergence_params needs of the NEXT iterate (G' z, G x + s, h' z, z' s: residual_products) are queued
// the moment a candidate is accepted -- z / s are the accepted candidate on the device, x is formed there with the host mirror's
// roundings -- and run while the host does its end-of-iteration bookkeeping; residual_products hands them out if the point it is
// called with is, bit for bit, the one they were formed from (the x it downloads, the candidate it kept), and computes as before
// otherwise.  HYP_RP_PREFETCH=0: off.
void SysSolver::prefetch_residual_products(int mode, double alpha, const double* d_cand, const double* h_cand) {
  static const bool on = [] { const char* e = getenv("HYP_RP_PREFETCH"); return !(e && e[0] == '0'); }();
  rp_pre_valid = false;
  if (!on || dist() || ks_world > 1 || p != 0 || n <= 0 || q <= 0 || !s_resident || !gemv_both_ok(q, n, G.d(), q)) return;
  const size_t d = sizeof(double);
  const long dv = dimv();
  rp_x.ensure((size_t)n * d);
  rp_t.ensure(((size_t)n + 2 + 2 + 2 * (size_t)std::max(comm_world_, 1)) * d);
  for (DBuf* b : {&rp_z, &rp_s, &rp_g}) b->ensure((size_t)q * d);
  const size_t need = (size_t)n + 2 + (size_t)q + (size_t)n;   // [G' z (n); h' z; z' s | G x + s (q) | x (n)]
  if (rp_pre_host_n < need) {
    if (rp_pre_host) (void)hipHostFree(rp_pre_host);
    rp_pre_host = nullptr; rp_pre_host_n = 0;
    HYP_CHECK(hipHostMalloc((void**)&rp_pre_host, need * d, hipHostMallocDefault));
    rp_pre_host_n = need;
  }
  if (!rp_pre_ev) HYP_CHECK(hipEventCreateWithFlags(&rp_pre_ev, hipEventDisableTiming));
  const double a2 = alpha * alpha, am1 = 1.0 - alpha, am1s = am1 * am1;
  const double* D = s_dirs.d();
  hipLaunchKernelGGL(point_form_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx.stream, n, s_point.d(), D, D + dv, D + 2 * dv, D + 3 * dv, mode, alpha,
                     a2, am1, am1s, rp_x.d());
  HYP_CHECK(hipGetLastError());
  ctx.zero(rp_t.p, ((size_t)n + 2) * d);
  ctx.d2d(rp_z.p, d_cand, (size_t)q * d);
  ctx.d2d(rp_s.p, d_cand + q + 1, (size_t)q * d);
  ctx.d2d(rp_g.p, rp_s.p, (size_t)q * d);
  gemv_both(ctx, q, n, 1, G.d(), q, rp_x.d(), n, 1.0, rp_g.d(), q, rp_z.d(), q, 0.0, rp_t.d(), n);
  dev_dot(ctx, q, mh.d(), rp_z.d(), rp_t.d() + n);
  dev_dot(ctx, q, rp_z.d(), rp_s.d(), rp_t.d() + n + 1);
  ctx.d2h(rp_pre_host, rp_t.p, ((size_t)n + 2) * d);
  ctx.d2h(rp_pre_host + n + 2, rp_g.p, (size_t)q * d);
  ctx.d2h(rp_pre_host + n + 2 + q, rp_x.p, (size_t)n * d);
  HYP_CHECK(hipEventRecord(rp_pre_ev, ctx.stream));
  rp_pre_cand.assign(h_cand, h_cand + 2 * (size_t)q + 2);   // [z; tau; s; kap] as handed to the caller
  rp_pre_valid = true;
}

int SysSolver::search_alpha(const double* pt, const double* dc, const double* dp, const double* dca, const double* dpa, bool unadj_only,
                            bool cent_only, const double* sched, int nsched, int start, double min_prox, double prox_bound,
                            bool use_max_prox, double nup1, double* cand, double* prox_out, int* n_trials, int* n_loaded,
                            double* irtmu_out, bool resident) {
  const int len = 2 * q + 2;
  *n_trials = 0;
  *n_loaded = 0;
  int smode = screen_mode();
  if (dist()) {   // every rank must walk the same way: the screen runs only if it applies on all of them (agreed once per model)
    if (screen_agreed < 0) {
      double v = -(double)smode;
      allreduce_host(&v, 1, 1, 13);
      screen_agreed = (int)(-v);
    }
    smode = std::min(smode, screen_agreed) == 2 ? 2 : 0;
  }
  // Candidates formed on the HOST (the caller's vectors) cost ~0.6 us per 1000 entries each and an upload: worth it for the 18 x
  // 40 202 entries of config 2 (line search 1.54 -> 1.03 ms), not for the 18 x 414 722 of config 4 (sharded, one rank: 7.2 -> 8.9 ms).
  // Without the resident vectors the screen therefore runs only up to 2^20 candidate entries, and never sharded (the ranks' row
  // counts differ, the decision must not).
  if (!resident && (dist() || (long)SCREEN_MAX * len > (1L << 20))) smode = 0;
  const bool screen = smode != 0;
  const int kcap = screen ? screen_kmax() : 1;   // candidates per batch (SCREEN_MAX unless the model is too big for the buffers)
  HYP_REQUIRE(!resident || (screen && s_resident && s_resident_q == q),
              "search_alpha: no resident directions (step_directions first; a model the candidate screen applies to)");
  // candidates are formed in pinned memory and only the accepted one is copied to the caller's vector:
  // the upload inside check_cone_points is then a plain asynchronous copy -- from the caller's pageable vector it stopped the
  // host for the whole transfer, once per trial
  // (the one-cone generic screen, see the sequential branch below, stages its candidates, points and dual points behind the two
  //  candidate slots of the sequential walk -- pinned as well: an upload from pageable memory stops the host for the transfer and
  //  cost 2 ms per iteration on some boxes, profiles/r04_wsos_screen.txt)
  const int gC = (!screen && !dist() && cones.size() == 1) ? std::min(8, cones[0]->screen_max()) : 0;
  const size_t gextra = gC >= 2 ? (size_t)gC * len + 2 * (size_t)gC * cones[0]->dim : 0;
  double* const stage = ctx.stage_host((size_t)(screen && !resident ? SCREEN_MAX : 2) * len + gextra);
  double* const ghost = stage + (size_t)2 * len;
  double* const out = cand;
  const int mode = unadj_only ? (cent_only ? 0 : 1) : (cent_only ? 2 : 3);
  auto coef = [&](double alpha, double& a2, double& am1, double& am1s) { a2 = alpha * alpha; am1 = 1.0 - alpha; am1s = am1 * am1; };
  auto form1 = [&](double alpha, double p0, double c0, double p1, double ca, double pa) {   // update_stepper_points (combined.jl:124-170)
    double a2, am1, am1s;
    coef(alpha, a2, am1, am1s);
    if (mode == 0) return p0 + alpha * c0;
    if (mode == 1) return p0 + (alpha * p1 + am1 * c0);
    if (mode == 2) return p0 + (alpha * c0 + a2 * ca);
    return p0 + (((alpha * p1 + a2 * pa) + am1 * c0) + am1s * ca);
  };
  auto form = [&](double* c, double alpha) {   // same operation order as the host mirror
    double a2, am1, am1s;
    coef(alpha, a2, am1, am1s);
    if (mode == 0) {
      for (int i = 0; i < len; ++i) c[i] = pt[i] + alpha * dc[i];
    } else if (mode == 1) {
      for (int i = 0; i < len; ++i) c[i] = pt[i] + (alpha * dp[i] + am1 * dc[i]);
    } else if (mode == 2) {
      for (int i = 0; i < len; ++i) c[i] = pt[i] + (alpha * dc[i] + a2 * dca[i]);
    } else {
      for (int i = 0; i < len; ++i) c[i] = pt[i] + (((alpha * dp[i] + a2 * dpa[i]) + am1 * dc[i]) + am1s * dca[i]);
    }
  };
  int idx = start;
  static const bool skip_lb = [] { const char* e = getenv("HYP_SCREEN_SKIP_LB"); return !(e && e[0] == '0'); }();
  int gbase = -1;
  std::vector<char> gver;
  std::vector<double> gbnd;   // the screen's lower bound of a candidate's proximity value (negative: none)
  while (idx < nsched) {
    const int K = screen ? std::min(kcap, nsched - idx) : 1;
    if (K >= 2 || resident) {
      char rej[SCREEN_MAX];
      double ctau[SCREEN_MAX], ckap[SCREEN_MAX];
      const long s2 = (long)static_cast<const PsdCone*>(cones[0])->side * static_cast<const PsdCone*>(cones[0])->side;
      const long nmat = (long)kcap * (long)cones.size();   // matrices per buffer: candidates x cones
      screen_buf.ensure((size_t)((long)kcap * len + 6L * nmat * s2 + 3 * nmat) * sizeof(double));
      double* cd = screen_buf.d();
      if (resident) {   // the five vectors are on the device: so are the candidates
        ScreenForm f{};
        f.mode = mode;
        for (int g = 0; g < K; ++g) {
          const double al = sched[idx + g];
          f.alpha[g] = al;
          coef(al, f.a2[g], f.am1[g], f.am1s[g]);
          f.tau[g] = ctau[g] = form1(al, s_tk[0][0], s_tk[1][0], s_tk[2][0], s_tk[3][0], s_tk[4][0]);
          f.kap[g] = ckap[g] = form1(al, s_tk[0][1], s_tk[1][1], s_tk[2][1], s_tk[3][1], s_tk[4][1]);
        }
        const long dv = dimv(), o = n + p;
        const double* D = s_dirs.d();
        hipLaunchKernelGGL(screen_form_kernel, dim3((len + 255) / 256, K), dim3(256), 0, ctx.stream, q, s_point.d() + o, D + o, D + dv + o,
                           D + 2 * dv + o, D + 3 * dv + o, f, cd);
      } else {
        for (int g = 0; g < K; ++g) {
          double* c = stage + (size_t)g * len;
          form(c, sched[idx + g]);
          ctau[g] = c[q];
          ckap[g] = c[len - 1];
        }
        ctx.h2d(cd, stage, (size_t)K * len * sizeof(double));
      }
      if (smode == 1) screen_candidates(cd, K, ctau, ckap, min_prox, prox_bound, nup1, rej);
      else screen_candidates_run(cd, K, ctau, ckap, min_prox, prox_bound, nup1, use_max_prox, rej);
      for (int g = 0; g < K; ++g) {
        ++*n_trials;
        if (rej[g]) continue;
        if (resident) {   // a survivor comes back for the sequential test (bitwise the candidate the host would have formed)
          cand = stage;
          ctx.d2h(cand, cd + (size_t)g * len, (size_t)len * sizeof(double));
          ctx.sync();
        } else {
          cand = stage + (size_t)g * len;
        }
        screen_survivor = skip_lb;
        screen_pass_g_ = (smode == 2) ? g : -1;
        const bool acc = check_cone_points(cand, min_prox, prox_bound, use_max_prox, nup1, prox_out, n_loaded, irtmu_out);
        screen_survivor = false;
        screen_pass_g_ = -1;
        if (acc) {
          std::memcpy(out, cand, (size_t)len * sizeof(double));
          if (resident) {
            prefetch_residual_products(mode, sched[idx + g], cd + (size_t)g * len, cand);
            prelaunch_sqrt_hess();
          }
          return idx + g;
        }
      }
      idx += K;
      continue;
    }
    // One large cone whose rejected candidates are chains of short launches (WsosCone): the next few candidates of the schedule
    // go through those chains TOGETHER (Cone::screen_batch, wsos_screen.hip); the ones it reports rejected -- infeasible, or a
    // rigorous lower bound of the proximity value beyond the neighbourhood: verdicts the sequential test reaches as well -- are
    // stepped over, everything else is tested below as before.  Candidates the scalar tests of check_cone_points reject are left
    // to it (no device work there).
    if (gC >= 2 && (idx < gbase || idx >= gbase + (int)gver.size()) && nsched - idx >= 2 && cones[0]->screen_ready()) {
      Cone* ck = cones[0];
      // The batch stops in FRONT of the candidate the previous search accepted (the accepted index moves by 0 or 1 from one
      // iteration to the next): that candidate and its successor go through the sequential test alone -- a batch containing them
      // computed the gradient and the bound of candidates behind the accepted one for nothing (HYP_TRIAL_DBG on config 5: 1.3
      // of 4 per batch, the survivor not counted).  Beyond the prediction: full batches again.
      static const bool gpredict = [] { const char* e = getenv("HYP_WSOS_SCREEN_PREDICT"); return !(e && e[0] == '0'); }();
      int Cw = std::min(gC, nsched - idx);          // the window of schedule positions this decision covers
      int Cn = Cw;                                  // of which the first Cn are screened
      if (gpredict && gprev_acc_ >= 0 && idx <= gprev_acc_ + 1) {
        Cn = std::min(Cw, std::max(0, gprev_acc_ - idx));
        Cw = std::max(Cn, 1);
        if (Cn < 2) { Cn = 0; Cw = std::min(std::max(1, gprev_acc_ + 2 - idx), nsched - idx); }
      }
      const int dk = ck->dim;
      double* gp = ghost + (size_t)Cn * len;
      double* gd = gp + (size_t)Cn * dk;
      double girt[8];
      int gmap[8], nbat = 0;
      gbase = idx;
      gver.assign(Cw, 0);
      gbnd.assign(Cw, -1.0);
      for (int g = 0; g < Cn; ++g) {
        double* h = ghost + (size_t)g * len;
        form(h, sched[idx + g]);
        double irt = 0.0;
        if (!cand_scalars(h, min_prox, prox_bound, nup1, &irt)) continue;
        const double* prim = ck->use_dual_barrier ? h : h + q + 1;
        const double* dual = ck->use_dual_barrier ? h + q + 1 : h;
        double* pp = gp + (size_t)nbat * dk;
        if (irt == 1.0) std::memcpy(pp, prim + offs[0], (size_t)dk * sizeof(double));
        else for (int i = 0; i < dk; ++i) pp[i] = irt * prim[offs[0] + i];
        std::memcpy(gd + (size_t)nbat * dk, dual + offs[0], (size_t)dk * sizeof(double));
        girt[nbat] = irt;
        gmap[nbat++] = g;
      }
      if (nbat >= 2) {
        char rej[8];
        int ninf = 0;
        const double pb2 = prox_bound * prox_bound;
        double bnd[8];
        if (ck->screen_batch(nbat, gp, gd, girt, pb2 * (1.0 + 1e-9), rej, &ninf, bnd)) {
          ++screen_count;
          for (int b = 0; b < nbat; ++b) {
            gver[gmap[b]] = rej[b];
            gbnd[gmap[b]] = bnd[b];
            screen_rejected += rej[b];
          }
          static const bool gdbg = [] { const char* e = getenv("HYP_TRIAL_DBG"); return e && e[0] == '1'; }();
          if (gdbg) {
            fprintf(stderr, "[screen] %d candidates from schedule index %d:", nbat, idx);
            for (int b = 0; b < nbat; ++b) fprintf(stderr, " %d", (int)rej[b]);
            fprintf(stderr, " (%d infeasible; bounds", ninf);
            for (int b = 0; b < nbat; ++b) fprintf(stderr, " %.3g", bnd[b]);
            fprintf(stderr, ")\n");
          }
        }
      }
    }
    // HYP_WSOS_SCREEN_CHECK=1 (tests): the screen's verdicts are not used, only compared with the sequential test's
    static const bool gcheck = [] { const char* e = getenv("HYP_WSOS_SCREEN_CHECK"); return e && e[0] == '1'; }();
    const bool grej = gC >= 2 && idx >= gbase && idx < gbase + (int)gver.size() && gver[idx - gbase];
    if (grej && !gcheck) {
      ++*n_trials;
      ++idx;
      continue;
    }
    cand = stage + (size_t)((idx - start) & 1) * len;
    form(cand, sched[idx]);
    ++*n_trials;
    // (letting a survivor whose screen bound lies far inside the neighbourhood skip the sequential test's own bound was measured:
    //  no difference, 23.6 / 23.7 ms per iteration at config 5 -- not kept)
    const bool acc_seq = check_cone_points(cand, min_prox, prox_bound, use_max_prox, nup1, prox_out, n_loaded, irtmu_out);
    if (gcheck && grej) {
      ++screen_checked;
      if (acc_seq) ++screen_mismatch;
      HYP_REQUIRE(!acc_seq, "search_alpha: a candidate the one-cone screen rejected passed the sequential test (HYP_WSOS_SCREEN_CHECK)");
    }
    if (acc_seq) {
      std::memcpy(out, cand, (size_t)len * sizeof(double));
      gprev_acc_ = idx;
      return idx;
    }
    ++idx;
  }
  gprev_acc_ = -1;
  return -1;
}

// res = K dir - rhs (in place in res); returns the inf-norm including the host-side tau / kap rows
// the same on a cone-sharded solver with the communicator's layout known: apply_lhs, the subtraction and the norm with ONE exchange --
// the summed G' z carries h' z (SUM) and the norm of this rank's rows (MAX, a slot per rank) behind it (allreduce_fused)
double SysSolver::residual_fused(double* res, const double* dir, const double* rhs, Scal rs, Scal dcur, Scal& rsc, double mu, double taubar) {
  const size_t d = sizeof(double);
  const int oz = n, os = n + q + 1, dv = dimv();
  const double tau_dir = dcur.tau, kap_dir = dcur.kap;
  rf_buf.ensure(((size_t)n + 8 + 2 * (size_t)comm_world_) * d);
  sgemv(true, q, n, 1.0, G.d(), q, dir + oz, 0.0, rf_buf.d());                 // this rank's part of G' z
  dev_scale_copy(ctx, q, tau_dir, mh.d(), res + oz);                           // res.z = h tau - s - G x
  dev_axpby(ctx, q, -1.0, dir + os, 1.0, res + oz);
  if (Gx_dir_valid) dev_axpby(ctx, q, -1.0, Gx_dir.d(), 1.0, res + oz);
  else sgemv(false, q, n, -1.0, G.d(), q, dir, 1.0, res + oz);
  ctx.zero(res + n + q, d);                                                    // (tau / kap slots of the device vectors stay zero)
  ctx.zero(res + dv - 1, d);
  for (size_t k = 0; k < cones.size(); ++k) {   // res.s_k = H_k prim_dir_k + dual_dir_k
    Cone* ck = cones[k];
    const int o = offs[k], dk = ck->dim;
    const double* prim = ck->use_dual_barrier ? dir + oz + o : dir + os + o;
    const double* dual = ck->use_dual_barrier ? dir + os + o : dir + oz + o;
    if (const int used = run_hess_prod(k, res + os + o, q, prim, q, 1)) {
      dev_axpby(ctx, offs[k + used] - o, 1.0, dual, 1.0, res + os + o);
      k += used - 1;
      continue;
    }
    ck->hess_prod_slow(res + os + o, q, prim, q, 1);
    dev_axpby(ctx, dk, 1.0, dual, 1.0, res + os + o);
  }
  double* dsc = ctx.dscal.d();
  dev_dot(ctx, n, mc.d(), dir, dsc);
  dev_dot(ctx, q, mh.d(), dir + oz, dsc + 1);
  dev_sub_absmax(ctx, dv - oz, res + oz, rhs + oz, dsc + 4);                   // this rank's rows
  FusedTail t;
  t.nsum = 1; t.sum_src[0] = dsc + 1;
  t.nmax = 1; t.max_src[0] = dsc + 4;
  double ho[2];
  allreduce_fused(rf_buf.d(), n, t, ho, 3);
  ctx.d2d(res, rf_buf.p, (size_t)n * d);
  dev_axpby(ctx, n, tau_dir, mc.d(), 1.0, res);                                // res.x = c tau + G' z
  dev_sub_absmax(ctx, n, res, rhs, dsc + 5);                                   // the replicated x rows
  ctx.d2h(ctx.h_pinned + 8, dsc, 8 * d);
  ctx.sync();
  rsc.tau = -ctx.h_pinned[8] - ho[0] - kap_dir - rs.tau;
  rsc.kap = mu / taubar * tau_dir / taubar + kap_dir - rs.kap;
  const double a = ho[1], b = ctx.h_pinned[8 + 5];
  if (a != a || b != b || rsc.tau != rsc.tau || rsc.kap != rsc.kap) return __builtin_nan("");
  return std::max(std::max(a, b), std::max(std::fabs(rsc.tau), std::fabs(rsc.kap)));
}

double SysSolver::residual(double* res, const double* dir, const double* rhs, Scal rs, Scal dcur, Scal& rsc, double mu, double taubar) {
  const size_t d = sizeof(double);
  if (dist() && fused_ok() && p == 0) return residual_fused(res, dir, rhs, rs, dcur, rsc, mu, taubar);
  rsc = apply_lhs(res, dir, dcur, mu, taubar);
  rsc.tau -= rs.tau;
  rsc.kap -= rs.kap;
  // (the tau / kap slots of the device vectors are kept at zero: the scalars travel on the host)
  dev_sub_absmax(ctx, dimv(), res, rhs, ctx.dscal.d() + 8);
  ctx.d2h(ctx.h_pinned + 16, ctx.dscal.d() + 8, d);
  ctx.sync();
  double m = ctx.h_pinned[16];
  if (dist()) {   // max over the ranks' rows (NaN on any rank must win: max of a flag, then of the value)
    double v[2] = {(m != m) ? 1.0 : 0.0, (m != m) ? 0.0 : m};
    allreduce_host(v, 2, 1, 5);
    m = (v[0] > 0.5) ? __builtin_nan("") : v[1];
  }
  if (m != m || rsc.tau != rsc.tau || rsc.kap != rsc.kap) return __builtin_nan("");
  return std::max(m, std::max(std::fabs(rsc.tau), std::fabs(rsc.kap)));
}

// the refinement loop of get_directions (common.jl:38-72) on device vectors; tmp holds the best direction so far
double SysSolver::refine(double* rhs, double* dir, double* res, double* tmp, Scal rs, Scal& dsc, Scal rsc, double res_norm, double mu,
                         double taubar, int max_ref_steps, double res_norm_cutoff, double min_impr_tol, int* n_solves) {
  const size_t d = sizeof(double);
  const int dv = dimv();
  Scal tsc = dsc;
  bool is_prev_slow = false;
  double prev_res_norm = res_norm;
  for (int step = 0; step < max_ref_steps; ++step) {
    // dir = dir_temp - solve(res)
    Scal csc = solve_system(dir, res, rsc, mu, taubar);
    ++*n_solves;
    dev_axpby(ctx, dv, 1.0, tmp, -1.0, dir);
    Gx_dir_valid = false;   // dir is no longer the raw output of solve_system
    dsc.tau = tsc.tau - csc.tau;
    dsc.kap = tsc.kap - csc.kap;
    Scal rsc2{0, 0};
    const double res_norm_new = residual(res, dir, rhs, rs, dsc, rsc2, mu, taubar);
    if (!(res_norm_new < res_norm)) {   // (>= or NaN: keep the previous direction)
      ctx.d2d(dir, tmp, (size_t)dv * d);
      dsc = tsc;
      break;
    }
    ctx.d2d(tmp, dir, (size_t)dv * d);
    tsc = dsc;
    rsc = rsc2;
    res_norm = res_norm_new;
    if (res_norm < res_norm_cutoff) break;
    const bool is_curr_slow = res_norm > min_impr_tol * prev_res_norm;
    if (is_prev_slow && is_curr_slow) break;
    prev_res_norm = res_norm;
    is_prev_slow = is_curr_slow;
  }
  return res_norm;
}

double SysSolver::get_directions(double* h_dir, const double* h_rhs, double mu, double taubar, int max_ref_steps, double res_norm_cutoff,
                                 double min_impr_tol, int* n_solves) {
  HYP_REQUIRE(model_loaded, "sys: load_model first");
  const size_t d = sizeof(double);
  const int dv = dimv(), it = n + p + q, ik = dv - 1;
  double* rhs = v_rhs.d();
  double* dir = v_dir.d();
  double* res = v_res.d();
  double* tmp = v_tmp.d();
  ctx.h2d(rhs, h_rhs, (size_t)dv * d);
  ctx.zero(rhs + it, d);   // tau / kap travel as host scalars: their device slots stay zero in every work vector
  ctx.zero(rhs + ik, d);
  const Scal rs{h_rhs[it], h_rhs[ik]};
  *n_solves = 0;
  Scal dsc = solve_system(dir, rhs, rs, mu, taubar);
  ++*n_solves;
  double res_norm = 0.0;
  if (max_ref_steps > 0) {
    Scal rsc{0, 0};
    ctx.d2d(tmp, dir, (size_t)dv * d);
    res_norm = residual(res, dir, rhs, rs, dsc, rsc, mu, taubar);
    if (res_norm > res_norm_cutoff)
      res_norm = refine(rhs, dir, res, tmp, rs, dsc, rsc, res_norm, mu, taubar, max_ref_steps, res_norm_cutoff, min_impr_tol, n_solves);
  }
  ctx.d2h(h_dir, dir, (size_t)dv * d);
  ctx.sync();
  h_dir[it] = dsc.tau;
  h_dir[ik] = dsc.kap;
  return res_norm;
}

}  // namespace hyp
