// wsos_screen.hip -- side-by-side screening of line-search candidates for ONE large WSOSInterpNonnegative cone.
//
// The schedule walk of search.jl:46-69 visits the candidate step lengths one after the other; for a model of one large WSOS cone
// (config 5: U = 4845, K = 5) the candidates in front of the accepted one are each rejected either by the feasibility test
// (wsosinterpnonnegative.jl:89-117: some Lambda_k = P_k' diag(s) P_k has no Cholesky factor) or by the proximity test
// (Cones.jl:294-310), and the second kind used to cost 1.6 ms a time: K feasibility chains, K triangular solves for the gradient
// (wsosinterpnonnegative.jl:119-133), and the lower bound of WsosCone::prox_lower_bound -- all short, latency-bound chains with
// the chip mostly idle (profiles/r04_cfg5p_trial_timeline.txt).  Here the SAME chains run for several candidates at once, the
// candidate being one more batch dimension of every launch:
//   F  Lambda_{c,k} = P_k' diag(s_c) P_k, batched Cholesky over (candidate, members of a run of equal L_k)
//   G  LFLP_{c,k} = U_{c,k}'^-1 P_k' (batched blocked forward solve), g_c = - sum_k colnorm2(LFLP_{c,k})
//   B  v_c = g_c + z_c / sqrt(mu_c);  p_c = M v_c with M the inverse of the Hessian factor the cone still holds;
//      <v_c, H_c^-1 v_c> >= <v_c, p_c>^2 / <p_c, H_c p_c>,  <p, H_c p> = sum_k || LFLP_{c,k} diag(p) LFLP_{c,k}' ||_F^2
// and ONE read-back brings every flag and scalar.  A candidate is reported rejected only when a factorization failed or the bound
// exceeds the neighbourhood -- both are verdicts the sequential test reaches for the same candidate (the bound is a rigorous lower
// bound of the value the exact test compares, whatever direction p it was formed with), so the walk accepts the candidate it
// always accepted, and that candidate is then evaluated by the unchanged sequential code: the iterates do not change.
// Everything else ("not rejected") goes through check_cone_points as before.
#include "cones.hpp"
#include <algorithm>
#include <cmath>

namespace hyp {

namespace {

struct Ptr8 { const double* p[8]; };
struct Dbl8 { double v[8]; };

// out_b[i, j] = s_c[i] * in_b[i, j] for batch member b = c * cnt + t: s_c = s + c * strideS; in_b = the t-th pointer of `ins` (shared
// by all candidates) or ins.p[0] + b * strideIn (strideIn > 0); out_b = out + b * strideOut.  m x n, leading dimension m both.
__global__ void row_scale_batched_kernel(int m, int n, int cnt, const double* __restrict__ s, long strideS, Ptr8 ins, long strideIn,
                                         double* __restrict__ out, long strideOut) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int b = blockIdx.z, c = b / cnt, t = b % cnt;
  const double si = s[(long)c * strideS + i];
  const double* in = strideIn > 0 ? ins.p[0] + (long)b * strideIn : ins.p[t];
  double* o = out + (long)b * strideOut;
  for (int j = blockIdx.y; j < n; j += gridDim.y) o[(long)j * m + i] = si * in[(long)j * m + i];
}

// dst_b = src_t (elems doubles), b = c * cnt + t
__global__ void bcast_copy_kernel(long elems, int cnt, Ptr8 srcs, double* __restrict__ dst, long strideDst) {
  const int b = blockIdx.y, t = b % cnt;
  const double* s = srcs.p[t];
  double* d = dst + (long)b * strideDst;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < elems; e += (long)gridDim.x * blockDim.x) d[e] = s[e];
}

// out[(c * K + k0 + t) * n + col] = - sum_i A_b[i, col]^2, one wavefront per column
__global__ __launch_bounds__(256) void col_norm2_batched_kernel(int m, int n, int cnt, int K, int k0, const double* __restrict__ A, long strideA,
                                                                double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= n) return;
  const int b = blockIdx.y, c = b / cnt, t = b % cnt;
  const double* a = A + (long)b * strideA + (long)col * m;
  double s = 0.0;
  for (int i = lane; i < m; i += 64) s += a[i] * a[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) out[((long)c * K + k0 + t) * n + col] = -s;
}

// grad_c[j] = sum_k parts[(c K + k) n + j] (k ascending);  v_c[j] = grad_c[j] + irt_c dual_c[j];  p_c[j] = v_c[j]
__global__ void grad_v_kernel(int n, int K, const double* __restrict__ parts, const double* __restrict__ duals, Dbl8 irt, double* __restrict__ V,
                              double* __restrict__ Pm) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int c = blockIdx.y;
  const double* p = parts + (long)c * K * n;
  double s = p[j];
  for (int k = 1; k < K; ++k) s += p[(long)k * n + j];
  const double v = s + irt.v[c] * duals[(long)c * n + j];
  V[(long)c * n + j] = v;
  Pm[(long)c * n + j] = v;
}

__device__ __forceinline__ double block_sum_256(double s, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// out[c] = <x_c, y_c>
__global__ __launch_bounds__(256) void dot_batched_kernel(int n, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ out) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  const double* a = x + (long)c * n;
  const double* b = y + (long)c * n;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += a[i] * b[i];
  const double t = block_sum_256(s, red);
  if (threadIdx.x == 0) out[c] = t;
}

// out[(c * K + k0 + t) * FROB_SL + slab] = the slab's share of || sym(A_b) ||_F^2, from the upper triangle of the L x L matrix A_b
// (columns slab, slab + FROB_SL, ... : the host adds the FROB_SL shares in order)
constexpr int FROB_SL = 16;
__global__ __launch_bounds__(256) void frob_upper_batched_kernel(int L, int cnt, int K, int k0, const double* __restrict__ A, long strideA,
                                                                 double* __restrict__ out) {
  __shared__ double red[4];
  const int b = blockIdx.x, c = b / cnt, t = b % cnt;
  const double* a = A + (long)b * strideA;
  double s = 0.0;
  for (int col = blockIdx.y * 4 + (threadIdx.x >> 6); col < L; col += 4 * FROB_SL) {
    const double* ac = a + (long)col * L;
    for (int i = threadIdx.x & 63; i <= col; i += 64) {
      const double x = ac[i];
      s += (i == col) ? x * x : 2.0 * x * x;
    }
  }
  const double tot = block_sum_256(s, red);
  if (threadIdx.x == 0) out[((long)c * K + k0 + t) * FROB_SL + blockIdx.y] = tot;
}

}   // namespace

int WsosCone::screen_max() const {
  static const int cmax = [] { const char* e = getenv("HYP_WSOS_SCREEN"); const int v = e ? atoi(e) : 4; return std::min(8, std::max(0, v)); }();
  static const bool lb_on = [] { const char* e = getenv("HYP_PROX_LB"); return !(e && e[0] == '0'); }();   // (the bound it rejects on)
  if (!lb_on || cmax < 2 || dim < 512 || K > 16 || K < 1) return 0;
  // (runs of equal L_k of at most 8 members: the pointer records of the batched kernels)
  for (int k0 = 0; k0 < K;) {
    int cnt = 1;
    while (k0 + cnt < K && Ls[k0 + cnt] == Ls[k0]) ++cnt;
    if (cnt > 8) return 0;
    k0 += cnt;
  }
  return cmax;
}

bool WsosCone::screen_ready() {
  if (screen_max() < 2) return false;
  if (!hess_fact_ok || hess_fact_bk || !Hfact.p) return false;
  if (ctx.trsv_plan_sb(dim) <= 0 || ctx.trsv_plan_sb(dim) > 1024) return false;
  // byte budget of the batch buffers (three U x L_k operand copies per candidate and member; HYP_SCREEN_MB as for the PSD screen,
  // default 2048 MB per candidate slot x 8): a cone too large for it walks sequentially
  static const double budget = [] { const char* e = getenv("HYP_SCREEN_MB"); return (e ? atof(e) : 2048.0) * 8.0 * 1048576.0; }();
  double sumL = 0.0, scratch = 0.0;
  for (int k = 0; k < K; ++k) sumL += Ls[k];
  // ... plus what screen_batch asks of the launchers on BOTH streams: split-K partial sums of the largest run, the tile
  // inverses of its factorizations, and the batch's L x L matrices (Lambda, LL, the inverted diagonal blocks)
  for (int k0 = 0; k0 < K;) {
    int cnt = 1;
    while (k0 + cnt < K && Ls[k0 + cnt] == Ls[k0]) ++cnt;
    const double L = Ls[k0], nb = (double)screen_max() * cnt;
    double wsmax = 0.0;   // (a smaller batch takes more slices per product: the largest request over the batch sizes)
    for (int c = 1; c <= screen_max(); ++c) wsmax = std::max(wsmax, (double)c * cnt * screen_splitk(Ls[k0], c * cnt) * L * L);
    scratch = std::max(scratch, 2.0 * (wsmax + nb * ((Ls[k0] + NB - 1) / NB) * 2048.0));
    scratch += nb * (2.0 * L * L + (double)dinv_elems(Ls[k0]));
    k0 += cnt;
  }
  if ((3.0 * screen_max() * (double)U * sumL + scratch) * sizeof(double) > budget) return false;
  return true;
}

// K slices for a batch of L x L x U Gram products (64 x 64 tiles of the upper triangle)
int WsosCone::screen_splitk(int L, int batch) const {
  const long T = (L + 63) / 64, nb = T * (T + 1) / 2 * batch;
  long S = std::min<long>(16, std::max<long>(1, 1024 / nb));
  S = std::min<long>(S, std::max(1, U / 256));
  return (int)S;
}

// h_pts: C x U, the candidates' primal points ALREADY scaled by irtmu[c] (Cone::load_point's product); h_duals: C x U; limit: the value
// of the proximity bound beyond which a candidate is rejected.  reject[c] = 1: certainly rejected by the sequential test as well.
bool WsosCone::screen_batch(int C, const double* h_pts, const double* h_duals, const double* irtmu, double limit, char* reject, int* n_infeas, double* bounds) {
  for (int c = 0; c < C; ++c) { reject[c] = 0; bounds[c] = -1.0; }
  *n_infeas = 0;
  if (C < 1 || C > 8 || !screen_ready()) return false;
  if (!Hplan.ready(dim)) Hplan.build(ctx, dim, Hfact.d(), dim, Hdinv.d());
  struct Grp { int k0, cnt, L; size_t oSP, oLam, oDinv, oLF, oLFT, oLL; long sLam, sDinv, sUL; };
  std::vector<Grp> grps;
  size_t nSP = 0, nLam = 0, nDinv = 0;
  for (int k0 = 0; k0 < K;) {
    int cnt = 1;
    while (k0 + cnt < K && Ls[k0 + cnt] == Ls[k0]) ++cnt;
    Grp g{};
    g.k0 = k0; g.cnt = cnt; g.L = Ls[k0];
    g.sLam = (long)(((size_t)g.L * g.L + 1) & ~(size_t)1);
    g.sDinv = (long)dinv_elems(g.L);
    g.sUL = (long)U * g.L;
    const size_t nb = (size_t)C * cnt;
    g.oSP = nSP; g.oLF = nSP; g.oLFT = nSP; nSP += nb * g.sUL;
    g.oLam = nLam; g.oLL = nLam; nLam += nb * g.sLam;
    g.oDinv = nDinv; nDinv += nb * g.sDinv;
    grps.push_back(g);
    k0 += cnt;
  }
  const size_t d = sizeof(double);
  // every buffer is sized for the largest batch the walk can ask for, on the first call: a batch of 2 followed by one of 4 must not
  // re-allocate (hipFree of a GB synchronises the device and was measured to disturb the queues for milliseconds)
  const size_t Cm = (size_t)std::max(C, screen_max());
  scrSP.ensure(nSP / C * Cm * d); scrLF.ensure(nSP / C * Cm * d); scrLFT.ensure(nSP / C * Cm * d);
  scrLam.ensure(nLam / C * Cm * d); scrLL.ensure(nLam / C * Cm * d); scrDinv.ensure(nDinv / C * Cm * d);
  scrVec.ensure((size_t)(4 * Cm * U + Cm * K * U + 16 + Cm * K * FROB_SL) * d);
  scrInfo.ensure(Cm * K * sizeof(int));
  {   // the launchers' own scratch, for both streams: split-K partial sums (at most 16 slices) and the factorizations' tile inverses
    size_t ws = 0, tv = 0;
    for (const Grp& g : grps) {
      for (size_t c = 1; c <= Cm; ++c)   // (the slices a batch of c candidates really takes, not 16: a smaller batch takes more per product)
        ws = std::max(ws, c * g.cnt * (size_t)screen_splitk(g.L, (int)(c * g.cnt)) * (size_t)g.L * g.L * d);
      tv = std::max(tv, Cm * g.cnt * (size_t)((g.L + NB - 1) / NB) * 2048 * d);
    }
    for (GemmScratch* gs : {&ctx.gemm_scratch, &ctx.gemm_scratch2}) {
      if (gs->splitk_ws_bytes < ws) {
        if (gs->splitk_ws) HYP_CHECK(hipFree(gs->splitk_ws));
        gs->splitk_ws = nullptr; gs->splitk_ws_bytes = 0;
        HYP_CHECK(hipMalloc((void**)&gs->splitk_ws, ws));
        gs->splitk_ws_bytes = ws;
      }
    }
    ctx.potrf_tinv.ensure(tv);
    ctx.potrf_tinv2.ensure(tv);
    Hplan.work_n.ensure((size_t)2 * 8 * 1024 * d);
  }
  double* pts = scrVec.d();                       // C x U
  double* duals = pts + (size_t)C * U;            // C x U
  double* V = duals + (size_t)C * U;              // C x U
  double* Pm = V + (size_t)C * U;                 // C x U
  double* parts = Pm + (size_t)C * U;             // C x K x U
  double* dots = parts + (size_t)C * K * U;       // 16
  double* gram = dots + 16;                       // C x K x FROB_SL
  ctx.h2d(pts, h_pts, (size_t)C * U * d);
  ctx.h2d(duals, h_duals, (size_t)C * U * d);
  Dbl8 irt{};
  for (int c = 0; c < C; ++c) irt.v[c] = irtmu[c];

  // ---- F: per run of equal L_k, the runs dealt out to the two streams by their block steps (as update_feas does)
  auto splitk_for = [&](int L, int batch) { return screen_splitk(L, batch); };
  hipEvent_t e0 = ctx.aux_event(2), e1 = ctx.aux_event(3);
  HYP_CHECK(hipEventRecord(e0, ctx.stream));
  HYP_CHECK(hipStreamWaitEvent(ctx.stream2, e0, 0));
  int load[2] = {0, 0};
  int info_off = 0;
  std::vector<int> ginfo(grps.size()), gside(grps.size());
  for (size_t gi = 0; gi < grps.size(); ++gi) {
    const Grp& g = grps[gi];
    const int nb = C * g.cnt, L = g.L;
    ginfo[gi] = info_off;
    const int side = (load[1] < load[0]) ? 1 : 0;
    gside[gi] = side;
    load[side] += ((L + NB - 1) / NB) * (1 + g.cnt);
    auto chain = [&] {
      Ptr8 Pp{};
      for (int t = 0; t < g.cnt; ++t) Pp.p[t] = P[g.k0 + t].d();
      double* SPg = scrSP.d() + g.oSP;
      double* Lamg = scrLam.d() + g.oLam;
      double* Dinvg = scrDinv.d() + g.oDinv;
      hipLaunchKernelGGL(row_scale_batched_kernel, dim3((U + 255) / 256, std::min(L, 64), nb), dim3(256), 0, ctx.stream, U, L, g.cnt, pts, (long)U, Pp, 0L,
                         SPg, g.sUL);
      HYP_CHECK(hipGetLastError());
      for (int t = 0; t < g.cnt; ++t) {   // Lambda_{c, k0 + t} for all c: one batched product per member (its second operand P_k is shared)
        GemmArgs a{};
        a.M = L; a.N = L; a.K = U; a.A = SPg + (long)t * g.sUL; a.lda = U; a.strideA = (long)g.cnt * g.sUL;
        a.B = P[g.k0 + t].d(); a.ldb = U; a.strideB = 0;
        a.C = Lamg + (long)t * g.sLam; a.ldc = L; a.strideC = (long)g.cnt * g.sLam;
        a.alpha = 1; a.beta = 0; a.tri = GEMM_UPPER; a.batch = C; a.splitk_req = splitk_for(L, C);
        gemm(ctx, true, a);
      }
      potrf_upper_batched(ctx, L, Lamg, L, g.sLam, nb, Dinvg, scrInfo.i() + ginfo[gi]);
    };
    if (side == 1) {
      StreamSwap on_helper(ctx);
      chain();
    } else {
      chain();
    }
    info_off += nb;
  }
  HYP_CHECK(hipEventRecord(e1, ctx.stream2));
  HYP_CHECK(hipStreamWaitEvent(ctx.stream, e1, 0));
  const int nflags = C * K;
  HYP_REQUIRE(128 + (size_t)nflags <= 8192 && (size_t)(16 + nflags * FROB_SL) <= ctx.h_pinned_n, "WsosCone::screen_batch: staging too small");
  ctx.d2h(ctx.h_info + 128, scrInfo.p, (size_t)nflags * sizeof(int));
  ctx.sync();
  // the candidates behind the LAST infeasible one go on (the schedule's step lengths decrease: the infeasible candidates come
  // first, and a contiguous rest keeps every batch a plain strided one); a feasible candidate in front of it is left to the
  // sequential test
  int c0 = 0;
  for (int c = 0; c < C; ++c) {
    bool feas = true;
    for (size_t gi = 0; gi < grps.size() && feas; ++gi)
      for (int t = 0; t < grps[gi].cnt && feas; ++t)
        if (ctx.h_info[128 + ginfo[gi] + c * grps[gi].cnt + t] != 0) feas = false;
    if (!feas) {
      reject[c] = 1;
      ++*n_infeas;
      c0 = c + 1;
    }
  }
  const int Cb = C - c0;
  if (Cb <= 0) return true;
  Dbl8 irtb{};
  for (int c = 0; c < Cb; ++c) irtb.v[c] = irtmu[c0 + c];
  double* partsb = parts + (size_t)c0 * K * U;
  double* Vb = V + (size_t)c0 * U;
  double* Pmb = Pm + (size_t)c0 * U;

  // ---- G: LFLP_b = U_b'^-1 P_k' and the gradient's parts, per run on the stream that factored it
  HYP_CHECK(hipEventRecord(e0, ctx.stream));
  HYP_CHECK(hipStreamWaitEvent(ctx.stream2, e0, 0));
  for (size_t gi = 0; gi < grps.size(); ++gi) {
    const Grp& g = grps[gi];
    const int nb = Cb * g.cnt, L = g.L;
    const long b0 = (long)c0 * g.cnt;
    auto chain = [&] {
      Ptr8 PTp{};
      for (int t = 0; t < g.cnt; ++t) PTp.p[t] = PT[g.k0 + t].d();
      double* Lamg = scrLam.d() + g.oLam + b0 * g.sLam;
      double* Dinvg = scrDinv.d() + g.oDinv + b0 * g.sDinv;
      double* LFg = scrLF.d() + g.oLF + b0 * g.sUL;
      double* LFTg = scrLFT.d() + g.oLFT + b0 * g.sUL;
      hipLaunchKernelGGL(bcast_copy_kernel, dim3(256, nb), dim3(256), 0, ctx.stream, g.sUL, g.cnt, PTp, LFg, g.sUL);
      HYP_CHECK(hipGetLastError());
      trsm_upper_left_fwd_batched(ctx, L, U, Lamg, L, g.sLam, Dinvg, g.sDinv, LFg, L, g.sUL, nb);
      hipLaunchKernelGGL(col_norm2_batched_kernel, dim3((U + 3) / 4, nb), dim3(256), 0, ctx.stream, L, U, g.cnt, K, g.k0, LFg, g.sUL, partsb);
      HYP_CHECK(hipGetLastError());
      dev_transpose(ctx, L, U, LFg, L, LFTg, U, nb, g.sUL, g.sUL);
    };
    if (gside[gi] == 1) {
      StreamSwap on_helper(ctx);
      chain();
    } else {
      chain();
    }
  }
  HYP_CHECK(hipEventRecord(e1, ctx.stream2));
  HYP_CHECK(hipStreamWaitEvent(ctx.stream, e1, 0));

  // ---- B: v_c, p_c = M v_c, <v_c, p_c>, the Gram norms of p_c
  hipLaunchKernelGGL(grad_v_kernel, dim3((U + 255) / 256, Cb), dim3(256), 0, ctx.stream, U, K, partsb, duals + (size_t)c0 * U, irtb, Vb, Pmb);
  HYP_CHECK(hipGetLastError());
  Hplan.solve_n(ctx, Hfact.d(), dim, true, Pmb, U, Cb);
  Hplan.solve_n(ctx, Hfact.d(), dim, false, Pmb, U, Cb);
  hipLaunchKernelGGL(dot_batched_kernel, dim3(Cb), dim3(256), 0, ctx.stream, U, Vb, Pmb, dots + c0);
  HYP_CHECK(hipGetLastError());
  HYP_CHECK(hipEventRecord(e0, ctx.stream));
  HYP_CHECK(hipStreamWaitEvent(ctx.stream2, e0, 0));
  for (size_t gi = 0; gi < grps.size(); ++gi) {
    const Grp& g = grps[gi];
    const int nb = Cb * g.cnt, L = g.L;
    const long b0 = (long)c0 * g.cnt;
    auto chain = [&] {
      double* SPg = scrSP.d() + g.oSP + b0 * g.sUL;
      double* LFTg = scrLFT.d() + g.oLFT + b0 * g.sUL;
      double* LLg = scrLL.d() + g.oLL + b0 * g.sLam;
      Ptr8 in{};
      in.p[0] = LFTg;
      hipLaunchKernelGGL(row_scale_batched_kernel, dim3((U + 255) / 256, std::min(L, 64), nb), dim3(256), 0, ctx.stream, U, L, g.cnt, Pmb, (long)U, in,
                         g.sUL, SPg, g.sUL);
      HYP_CHECK(hipGetLastError());
      GemmArgs a{};   // LL_b = (diag(p_c) LFLP_b')' LFLP_b'  (L x L, upper)
      a.M = L; a.N = L; a.K = U; a.A = SPg; a.lda = U; a.strideA = g.sUL; a.B = LFTg; a.ldb = U; a.strideB = g.sUL;
      a.C = LLg; a.ldc = L; a.strideC = g.sLam; a.alpha = 1; a.beta = 0; a.tri = GEMM_UPPER; a.batch = nb; a.splitk_req = splitk_for(L, nb);
      gemm(ctx, true, a);
      hipLaunchKernelGGL(frob_upper_batched_kernel, dim3(nb, FROB_SL), dim3(256), 0, ctx.stream, L, g.cnt, K, g.k0, LLg, g.sLam,
                         gram + (size_t)c0 * K * FROB_SL);
      HYP_CHECK(hipGetLastError());
    };
    if (gside[gi] == 1) {
      StreamSwap on_helper(ctx);
      chain();
    } else {
      chain();
    }
  }
  HYP_CHECK(hipEventRecord(e1, ctx.stream2));
  HYP_CHECK(hipStreamWaitEvent(ctx.stream, e1, 0));
  ctx.d2h(ctx.h_pinned, dots, (size_t)(16 + nflags * FROB_SL) * d);
  ctx.sync();
  for (int c = c0; c < C; ++c) {
    const double bv = ctx.h_pinned[c];
    double qn = 0.0;
    for (int k = 0; k < K; ++k) {
      double qk = 0.0;
      for (int sl = 0; sl < FROB_SL; ++sl) qk += ctx.h_pinned[16 + (c * K + k) * FROB_SL + sl];
      qn += qk;
    }
    if (!(qn > 0.0) || !(qn < INFINITY) || !(bv == bv)) continue;
    // the value of 2 <v, x> - <x, H x> at x = (bv / qn) p, evaluated as WsosCone::prox_lower_bound evaluates it
    const double Lm = std::sqrt(qn), y = bv / Lm, cc = y / Lm;
    const double val = 2.0 * (bv * cc) - cc * qn * cc;
    if (val == val) bounds[c] = val;
    if (val > limit) reject[c] = 1;
  }
  return true;
}

}   // namespace hyp
