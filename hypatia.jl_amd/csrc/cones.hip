// Cone barrier oracles on the device: generic protocol, Nonnegative, PosSemidefTri.
// Reference: /root/reference/src/Cones/Cones.jl, nonnegative.jl, possemideftri.jl (line ranges inline).
#include "cones.hpp"

namespace hyp {

static const double EPS = 2.220446049250313e-16;

// ---------------------------------------------------------------------------------------------
// generic Cone
// ---------------------------------------------------------------------------------------------
void Cone::alloc_common() {
  const size_t b = (size_t)dim * sizeof(double);
  point.alloc(b); dual_point.alloc(b); grad.alloc(b); dder3v.alloc(b); vec1.alloc(b); vec2.alloc(b);
  ctx.zero(point.p, b); ctx.zero(dual_point.p, b); ctx.zero(grad.p, b);
  ctx.zero(dder3v.p, b); ctx.zero(vec1.p, b); ctx.zero(vec2.p, b);
}

void Cone::load_point(const double* d_pt, double scal) {   // Cones.jl:157-166
  ++ctx.cone_epoch;
  if (scal == 1.0) ctx.d2d(point.p, d_pt, (size_t)dim * sizeof(double));
  else dev_scale_copy(ctx, dim, scal, d_pt, point.d());
}
void Cone::load_dual_point(const double* d_pt) {   // :168-171
  ++ctx.cone_epoch;
  ctx.d2d(dual_point.p, d_pt, (size_t)dim * sizeof(double));
  dual_cached = false;
}

double Cone::dot_host(int n, const double* dx, const double* dy) {
  dev_dot(ctx, n, dx, dy, ctx.dscal.d());
  ctx.d2h(ctx.h_pinned, ctx.dscal.p, sizeof(double));
  ctx.sync();
  return ctx.h_pinned[0];
}

bool Cone::check_numerics() {   // Cones.jl:273-290
  const double gtol = sqrt(sqrt(EPS)), Htol = 10 * sqrt(gtol);
  const double* g = get_grad();
  if (fabs(1 + dot_host(dim, g, point.d()) / nu) > gtol * dim) return false;
  if (!inv_hess_ready()) return false;
  inv_hess_prod(vec1.d(), dim, g, dim, 1);
  if (fabs(1 - dot_host(dim, vec1.d(), g) / nu) > Htol * dim) return false;
  return true;
}

double Cone::get_proxsqr(double irtmu, bool) {   // Cones.jl:294-310
  const double negtol = sqrt(EPS);
  const double* g = get_grad();
  // vec1 = irtmu * dual_point + g
  ctx.d2d(vec1.p, g, (size_t)dim * sizeof(double));
  dev_axpby(ctx, dim, irtmu, dual_point.d(), 1.0, vec1.d());
  if (!inv_hess_ready()) return INFINITY;
  inv_hess_prod(vec2.d(), dim, vec1.d(), dim, 1);
  const double prox_sqr = dot_host(dim, vec2.d(), vec1.d());
  if (prox_sqr < -negtol * dim) return INFINITY;
  return fabs(prox_sqr);
}

bool Cone::prox_launch(double irtmu, double* d_out3) {   // the device work of check_numerics + get_proxsqr, no host round trip
  const double* g = get_grad();
  dev_dot(ctx, dim, g, point.d(), d_out3);
  if (!inv_hess_ready()) return false;
  // both inverse-Hessian products in ONE call on two columns [g, v]: every cone's inv_hess_prod! serves all columns with one
  // pass over its factor / its matrices (for the generic cones two triangular sweeps over a dim x dim factor instead of four)
  const size_t vb = (size_t)dim * sizeof(double);
  prox_in.ensure(2 * vb);
  prox_out.ensure(2 * vb);
  double* v0 = prox_in.d();
  double* v1 = prox_in.d() + dim;
  ctx.d2d(v0, g, vb);
  ctx.d2d(v1, g, vb);
  dev_axpby(ctx, dim, irtmu, dual_point.d(), 1.0, v1);
  inv_hess_prod(prox_out.d(), dim, prox_in.d(), dim, 2);
  dev_dot(ctx, dim, prox_out.d(), g, d_out3 + 1);
  dev_dot(ctx, dim, prox_out.d() + dim, v1, d_out3 + 2);
  return true;
}

void Cone::hess_explicit(double* d_out, long ld) {
  DBuf eye((size_t)dim * dim * sizeof(double));
  dev_fill_identity(ctx, dim, eye.d(), dim);
  hess_prod(d_out, ld, eye.d(), dim, dim);
  ctx.sync();
}
void Cone::inv_hess_explicit(double* d_out, long ld) {
  DBuf eye((size_t)dim * dim * sizeof(double));
  dev_fill_identity(ctx, dim, eye.d(), dim);
  inv_hess_prod(d_out, ld, eye.d(), dim, dim);
  ctx.sync();
}

// ---------------------------------------------------------------------------------------------
// Nonnegative (nonnegative.jl:8-145)
// ---------------------------------------------------------------------------------------------
enum { NN_HESS = 0, NN_INVHESS = 1, NN_SQRT = 2, NN_INVSQRT = 3 };
__global__ void nonneg_prod_kernel(int dim, int ncols, int mode, const double* __restrict__ pt, const double* __restrict__ arr,
                                   long lda, double* __restrict__ prod, long ldp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dim) return;
  const double p = pt[i];
  for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
    const double a = arr[(long)j * lda + i];
    double v;
    if (mode == NN_HESS) v = a / p / p;            // nonnegative.jl:88
    else if (mode == NN_INVHESS) v = a * p * p;    // :98
    else if (mode == NN_SQRT) v = a / p;           // :108
    else v = a * p;                                // :118
    prod[(long)j * ldp + i] = v;
  }
}
__global__ void nonneg_grad_kernel(int dim, const double* __restrict__ pt, double* __restrict__ g) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < dim) g[i] = -(1.0 / pt[i]);   // -inv(point), :55
}
__global__ void nonneg_dder3_kernel(int dim, const double* __restrict__ pt, const double* __restrict__ dir, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < dim) {
    const double t = dir[i] / pt[i];
    out[i] = t * t / pt[i];   // abs2(dir / point) / point, :123
  }
}
// out[0] = number of entries <= eps
__global__ __launch_bounds__(1024) void count_le_kernel(int n, const double* __restrict__ x, double thr, double* __restrict__ out) {
  __shared__ int red[16];
  int cnt = 0;
  for (int i = threadIdx.x; i < n; i += 1024) cnt += !(x[i] > thr);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < 16; ++k) t += red[k];
    out[0] = (double)t;
  }
}
// out[0] = max or sum over i of (s_i z_i irtmu - 1)^2   (nonnegative.jl:137-145)
__global__ __launch_bounds__(1024) void nonneg_prox_kernel(int n, const double* __restrict__ s, const double* __restrict__ z, double irtmu,
                                                           int use_max, double* __restrict__ out) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const double t = s[i] * z[i] * irtmu - 1.0;
    const double v = t * t;
    acc = use_max ? fmax(acc, v) : acc + v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(acc, off);
    acc = use_max ? fmax(acc, o) : acc + o;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t = use_max ? fmax(t, red[k]) : t + red[k];
    out[0] = t;
  }
}

NonnegCone::NonnegCone(Ctx& c, int d) : Cone(c, CONE_NONNEG) {
  HYP_REQUIRE(d >= 1, "Nonnegative: dim >= 1");
  dim = d;
  nu = d;
  alloc_common();
}
void NonnegCone::set_initial_point(double* h) {
  for (int i = 0; i < dim; ++i) h[i] = 1.0;
}
static bool all_gt(Ctx& ctx, int n, const double* x, double thr) {
  hipLaunchKernelGGL(count_le_kernel, dim3(1), dim3(1024), 0, ctx.stream, n, x, thr, ctx.dscal.d());
  ctx.d2h(ctx.h_pinned, ctx.dscal.p, sizeof(double));
  ctx.sync();
  return ctx.h_pinned[0] == 0.0;
}
bool NonnegCone::update_feas() {   // :44-49
  is_feas_ = all_gt(ctx, dim, point.d(), EPS);
  feas_updated = true;
  return is_feas_;
}
bool NonnegCone::is_dual_feas() { return all_gt(ctx, dim, dual_point.d(), EPS); }   // :51
void NonnegCone::update_grad() {   // :53-58
  hipLaunchKernelGGL(nonneg_grad_kernel, dim3((dim + 255) / 256), dim3(256), 0, ctx.stream, dim, point.d(), grad.d());
  grad_updated = true;
}
static void nn_launch(Ctx& ctx, int dim, int ncols, int mode, const double* pt, const double* arr, long lda, double* prod, long ldp) {
  if (ncols <= 0) return;
  hipLaunchKernelGGL(nonneg_prod_kernel, dim3((dim + 255) / 256, std::min(ncols, 4096)), dim3(256), 0, ctx.stream, dim, ncols, mode, pt,
                     arr, lda, prod, ldp);
  HYP_CHECK(hipGetLastError());
}
void NonnegCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int nc) { nn_launch(ctx, dim, nc, NN_HESS, point.d(), arr, lda, prod, ldp); }
void NonnegCone::inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int nc) { nn_launch(ctx, dim, nc, NN_INVHESS, point.d(), arr, lda, prod, ldp); }
void NonnegCone::sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int nc) { nn_launch(ctx, dim, nc, NN_SQRT, point.d(), arr, lda, prod, ldp); }
void NonnegCone::inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int nc) { nn_launch(ctx, dim, nc, NN_INVSQRT, point.d(), arr, lda, prod, ldp); }
const double* NonnegCone::dder3(const double* d_dir) {   // :122-125
  hipLaunchKernelGGL(nonneg_dder3_kernel, dim3((dim + 255) / 256), dim3(256), 0, ctx.stream, dim, point.d(), d_dir, dder3v.d());
  return dder3v.d();
}
double NonnegCone::get_proxsqr(double irtmu, bool use_max_prox) {   // :137-145
  hipLaunchKernelGGL(nonneg_prox_kernel, dim3(1), dim3(1024), 0, ctx.stream, dim, point.d(), dual_point.d(), irtmu, use_max_prox ? 1 : 0,
                     ctx.dscal.d());
  ctx.d2h(ctx.h_pinned, ctx.dscal.p, sizeof(double));
  ctx.sync();
  return ctx.h_pinned[0];
}

// ---------------------------------------------------------------------------------------------
// PosSemidefTri (possemideftri.jl:9-207), real symmetric
// ---------------------------------------------------------------------------------------------
static int svec_side_of(int len) {
  int side = (int)((sqrt(1.0 + 8.0 * (double)len)) / 2.0);
  while ((long)side * (side + 1) < 2L * len) ++side;
  while ((long)side * (side + 1) > 2L * len) --side;
  return side;
}

PsdCone::PsdCone(Ctx& c, int d) : Cone(c, CONE_PSD) {
  HYP_REQUIRE(d >= 1, "PosSemidefTri: dim >= 1");
  dim = d;
  side = svec_side_of(d);
  HYP_REQUIRE((long)side * (side + 1) == 2L * d, "PosSemidefTri: dim is not a triangular number");
  nu = side;   // :67
  alloc_common();
  const size_t mb = (size_t)side * side * sizeof(double);
  X.alloc(mb); U.alloc(mb); UT.alloc(mb); Uinv.alloc(mb); UinvT.alloc(mb); Xinv.alloc(mb); tmpmat.alloc(mb); tmpmat2.alloc(mb);
  dinvb.alloc(2 * dinv_elems(side) * sizeof(double));   // [0]: primal factor, [1]: scratch for the dual-feasibility test
  d_info.alloc(64);
}

void PsdCone::set_initial_point(double* h) {   // :69-78: svec(I)
  for (int i = 0; i < dim; ++i) h[i] = 0.0;
  long k = 0;
  for (int i = 1; i <= side; ++i) {
    h[k] = 1.0;
    k += i + 1;
  }
}

static int read_info(Ctx& ctx, const int* d_info) {
  ctx.d2h(ctx.h_info, d_info, sizeof(int));
  ctx.sync();
  return ctx.h_info[0];
}

bool PsdCone::update_feas() {   // :80-90
  const size_t mb = (size_t)side * side * sizeof(double);
  svec_unpack(ctx, side, 1, point.d(), dim, X.d());          // svec_to_smat! (both triangles filled)
  ctx.d2d(U.p, X.p, mb);
  potrf_upper_batched(ctx, side, U.d(), side, 0, 1, nullptr, d_info.i());   // block inverses on demand (ensure_inverses)
  is_feas_ = (read_info(ctx, d_info.i()) == 0);
  feas_updated = true;
  inv_ready = false;
  return is_feas_;
}

// The two feasibility Choleskys of a line-search candidate (smat(point), smat(dual_point)) are independent
// latency-bound chains (2 diagonal-block kernels + panel + update each): run them side by side on the two streams
// and read both LAPACK infos after one synchronisation.
bool PsdCone::prefetch_launch(int slot) {
  if (feas_updated || slot < 0 || 64 + 2 * slot + 1 >= 8192) return false;
  const size_t mb = (size_t)side * side * sizeof(double);
  int* hi = ctx.h_info + 64 + 2 * slot;
  hipEvent_t e0 = ctx.aux_event(2);
  HYP_CHECK(hipEventRecord(e0, ctx.stream));                 // (the loads of point / dual_point were queued on the main stream)
  HYP_CHECK(hipStreamWaitEvent(ctx.stream2, e0, 0));
  {
    StreamSwap on_helper(ctx);
    svec_unpack(ctx, side, 1, dual_point.d(), dim, tmpmat.d());
    potrf_upper_batched(ctx, side, tmpmat.d(), side, 0, 1, nullptr, d_info.i() + 1);
    ctx.d2h(hi + 1, d_info.i() + 1, sizeof(int));
  }
  svec_unpack(ctx, side, 1, point.d(), dim, X.d());
  ctx.d2d(U.p, X.p, mb);
  potrf_upper_batched(ctx, side, U.d(), side, 0, 1, nullptr, d_info.i());
  ctx.d2h(hi, d_info.i(), sizeof(int));
  return true;
}

void PsdCone::prefetch_finish(int slot) {
  const int* hi = ctx.h_info + 64 + 2 * slot;
  is_feas_ = (hi[0] == 0);
  dual_feas_ = (hi[1] == 0);
  feas_updated = true;
  dual_cached = true;
  inv_ready = false;
}

void PsdCone::prefetch_feas() {
  if (!prefetch_launch(0)) return;
  hipEvent_t e1 = ctx.aux_event(3);
  HYP_CHECK(hipEventRecord(e1, ctx.stream2));
  HYP_CHECK(hipStreamWaitEvent(ctx.stream, e1, 0));
  ctx.sync();
  prefetch_finish(0);
}

// sum over the svec entries of (scal * t_i - e_i)^2, e = svec(I): || scal T - I ||_F^2 for T = smat(t) (the sqrt(2) scaling of
// the off-diagonal svec entries makes the svec 2-norm the Frobenius norm); one workgroup
__global__ __launch_bounds__(1024) void psd_prox_direct_kernel(int side, const double* __restrict__ t, double scal, double* __restrict__ out) {
  __shared__ double red[1024];
  const int dim = side * (side + 1) / 2;
  double s = 0.0;
  for (int i = threadIdx.x; i < dim; i += 1024) {
    // svec index i = column j, row r (r <= j) with i = j (j + 1) / 2 + r: the diagonal entries sit at j (j + 3) / 2
    int j = (int)((sqrt(8.0 * (double)i + 1.0) - 1.0) * 0.5);
    while ((long)(j + 1) * (j + 2) / 2 <= i) ++j;
    while ((long)j * (j + 1) / 2 > i) --j;
    const bool diag = (i == j * (j + 3) / 2);
    const double v = scal * t[i] - (diag ? 1.0 : 0.0);
    s = fma(v, v, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// For this cone the proximity value has a form that needs no inverse: with X = U'U, g = -svec(X^-1) and H^-1 v = X V X,
//   <v, H^-1 v> = tr(V X V X) = || U V U' ||_F^2,  V = Z / sqrt(mu) - X^-1  =>  U V U' = U Z U' / sqrt(mu) - I,
// one two-sided product of the dual point with the factor the feasibility test left behind.  The reference's route -- inverse of
// U, X^-1, gradient, two more two-sided products -- gives the same number up to rounding; a candidate for which this one is
// beyond the neighbourhood by more than that (1e-6 relative) is rejected here, every other one goes the reference's way.
bool PsdCone::prox_lower_bound(double irtmu, double limit, double* lb) {
  static const bool on = [] { const char* e = getenv("HYP_PROX_LB"); return !(e && e[0] == '0'); }();
  if (!on || !feas_updated || !is_feas_ || inv_ready || side < 32) return false;
  (void)limit;
  dev_zero_strict_lower(ctx, side, U.d(), side, 1, 0);
  dev_transpose(ctx, side, side, U.d(), side, UT.d(), side, 1, 0, 0);
  two_sided(UT.d(), KR_GE_N, KR_GE_M, vec1.d(), dim, dual_point.d(), dim, 1);     // svec(U Z U')
  double* ds = ctx.dscal.d() + 44;
  hipLaunchKernelGGL(psd_prox_direct_kernel, dim3(1), dim3(1024), 0, ctx.stream, side, vec1.d(), irtmu, ds);
  HYP_CHECK(hipGetLastError());
  ctx.d2h(ctx.h_pinned + 44, ds, sizeof(double));
  ctx.sync();
  const double v = ctx.h_pinned[44];
  if (!(v == v) || !(v < INFINITY)) return false;
  *lb = v / (1.0 + 1e-6);     // (the caller rejects on lb > limit: leave room for the rounding of the two routes)
  return true;
}

bool PsdCone::is_dual_feas() {   // :92-95
  if (dual_cached) return dual_feas_;
  svec_unpack(ctx, side, 1, dual_point.d(), dim, tmpmat.d());
  potrf_upper_batched(ctx, side, tmpmat.d(), side, 0, 1, nullptr, d_info.i() + 1);
  return read_info(ctx, d_info.i() + 1) == 0;
}

void PsdCone::ensure_inverses() {
  if (inv_ready) return;
  dev_zero_strict_lower(ctx, side, U.d(), side, 1, 0);
  potrf_invert_diag_blocks(ctx, side, U.d(), side, 0, 1, dinvb.d());
  trtri_upper_batched(ctx, side, U.d(), side, 0, dinvb.d(), 0, Uinv.d(), side, 0, 1);
  dev_transpose(ctx, side, side, Uinv.d(), side, UinvT.d(), side, 1, 0, 0);
  dev_transpose(ctx, side, side, U.d(), side, UT.d(), side, 1, 0, 0);
  GemmArgs g{};   // Xinv = Uinv * Uinv'  (dpotri of the reference, possemideftri.jl:100 / dense.jl:19-20)
  g.M = side; g.N = side; g.K = side; g.A = Uinv.d(); g.lda = side; g.B = UinvT.d(); g.ldb = side;
  g.C = Xinv.d(); g.ldc = side; g.alpha = 1; g.beta = 0; g.tri = GEMM_FULL; g.krange = KR_GE_M; g.batch = 1;
  gemm(ctx, false, g);
  inv_ready = true;
}

void PsdCone::update_grad() {   // :97-107: grad = -svec(inv(X))
  ensure_inverses();
  svec_pack(ctx, side, 1, Xinv.d(), grad.d(), dim, -1.0);
  grad_updated = true;
}

// W_j = R' V_j R for ncols svec columns.  Two stacked FP64-MFMA GEMMs per chunk of columns:
//   Z_j = V_j R    rows of all V_j stacked into one tall operand (M = nc * side), composite output
//   W_j = R' Z_j   all Z_j side by side (N = nc * side)
static void two_sided_core(Ctx& ctx, int side, int nc, const double* R, int kr2, int kr3, double* ws1, double* ws2) {
  GemmArgs a{};
  a.M = nc * side; a.N = side; a.K = side;
  a.A = ws1; a.lda = side; a.B = R; a.ldb = side; a.C = ws2; a.ldc = side;
  a.cm_blk = side; a.cm_stride = (long)side * side;
  a.alpha = 1; a.beta = 0; a.tri = GEMM_FULL; a.krange = kr2; a.batch = 1;
  gemm(ctx, true, a);
  GemmArgs b{};
  b.M = side; b.N = nc * side; b.K = side;
  b.A = R; b.lda = side; b.B = ws2; b.ldb = side; b.C = ws1; b.ldc = side;
  b.alpha = 1; b.beta = 0; b.tri = GEMM_FULL; b.krange = kr3; b.batch = 1;
  gemm(ctx, true, b);
}

void PsdCone::two_sided(const double* R, int kr2, int kr3, double* prod, long ldp, const double* arr, long lda, int ncols) {
  if (ncols <= 0) return;
  const long s2 = (long)side * side;
  const long ws_cap = 1L << 27;   // doubles per workspace (1 GiB)
  int chunk = (int)std::min<long>(ncols, std::max<long>(1, ws_cap / s2));
  const bool fused = use_fused(ncols);
  if (fused && kr2 == KR_LE_N && psd_two_sided_onchip(ctx, side, ncols, R, 1, arr, lda, prod, ldp)) return;   // (no workspace, no chunks)
  // HYP_TS_CHUNK_MB: the two passes of the fused product run chunk by chunk with the intermediate Z of a chunk held to that
  // many MB (A/B switch: does pass 2 find Z in the 256 MB Infinity Cache?)
  static const long z_mb = [] { const char* e = getenv("HYP_TS_CHUNK_MB"); return e ? atol(e) : 0L; }();
  if (fused && z_mb > 0) chunk = (int)std::max<long>(8, std::min<long>(chunk, (z_mb << 20) / (s2 * (long)sizeof(double))));
  ws1.ensure((size_t)chunk * s2 * sizeof(double));
  ws2.ensure((size_t)chunk * s2 * sizeof(double));
  for (int c0 = 0; c0 < ncols; c0 += chunk) {
    const int nc = std::min(chunk, ncols - c0);
    if (fused) {   // one workgroup per matrix, svec conversions fused (psd_twosided.hip)
      psd_two_sided_fused(ctx, side, nc, R, kr2 == KR_LE_N ? 1 : (kr2 == KR_GE_N ? 2 : 0), arr + (long)c0 * lda, lda,
                          prod + (long)c0 * ldp, ldp, ws1.d());
      continue;
    }
    svec_unpack(ctx, side, nc, arr + (long)c0 * lda, lda, ws1.d());
    two_sided_core(ctx, side, nc, R, kr2, kr3, ws1.d(), ws2.d());
    svec_pack(ctx, side, nc, ws1.d(), prod + (long)c0 * ldp, ldp, 1.0);
  }
}

bool PsdCone::use_fused(int ncols) const {
  static const bool enabled = [] { const char* e = getenv("HYP_PSD_FUSED"); return !(e && e[0] == '0'); }();
  static const int min_cols = [] { const char* e = getenv("HYP_PSD_FUSED_MIN"); return e ? atoi(e) : 1; }();   // (1: also the one- and two-column products of the KKT solves -- two launches instead of unpack + four GEMMs + pack; config 2: directions 4.14 -> 3.49 ms)
  return enabled && ncols >= min_cols && psd_two_sided_fused_ok(side);
}

void PsdCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :126-142  X^-1 V X^-1
  // The reference applies the Cholesky factor four times (rdiv!/ldiv! with fact_mat).  Same here:
  // U^-1 (U^-T V U^-1) U^-T through the factor's inverse, not through the rounded product inv(X) --
  // keeps H consistent with the sqrt form used for the Schur complement (cond(U), not cond(X)).
  ensure_inverses();
  if (ncols <= 0) return;
  const long s2 = (long)side * side;
  const long ws_cap = 1L << 27;
  int chunk = (int)std::min<long>(ncols, std::max<long>(1, ws_cap / s2));
  ws1.ensure((size_t)chunk * s2 * sizeof(double));
  ws2.ensure((size_t)chunk * s2 * sizeof(double));
  const bool fused = use_fused(ncols);
  for (int c0 = 0; c0 < ncols; c0 += chunk) {
    const int nc = std::min(chunk, ncols - c0);
    if (fused) {
      double* pc = prod + (long)c0 * ldp;
      psd_two_sided_fused(ctx, side, nc, Uinv.d(), 1, arr + (long)c0 * lda, lda, pc, ldp, ws1.d());   // U^-T V U^-1
      psd_two_sided_fused(ctx, side, nc, UinvT.d(), 2, pc, ldp, pc, ldp, ws1.d());                     // U^-1 (.) U^-T
      continue;
    }
    svec_unpack(ctx, side, nc, arr + (long)c0 * lda, lda, ws1.d());
    two_sided_core(ctx, side, nc, Uinv.d(), KR_LE_N, KR_LE_M, ws1.d(), ws2.d());    // U^-T V U^-1
    two_sided_core(ctx, side, nc, UinvT.d(), KR_GE_N, KR_GE_M, ws1.d(), ws2.d());   // U^-1 (.) U^-T
    svec_pack(ctx, side, nc, ws1.d(), prod + (long)c0 * ldp, ldp, 1.0);
  }
}
void PsdCone::inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int nc) {   // :144-159  X V X
  two_sided(X.d(), KR_ALL, KR_ALL, prod, ldp, arr, lda, nc);
}
void PsdCone::sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int nc) {   // :161-177  U^-T V U^-1
  ensure_inverses();
  two_sided(Uinv.d(), KR_LE_N, KR_LE_M, prod, ldp, arr, lda, nc);
}
void PsdCone::inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int nc) {   // :179-195  U V U'
  ensure_inverses();
  two_sided(UT.d(), KR_GE_N, KR_GE_M, prod, ldp, arr, lda, nc);
}

const double* PsdCone::dder3(const double* d_dir) {   // :197-207  svec(X^-1 D X^-1 D X^-1)
  ensure_inverses();
  const long s2 = (long)side * side;
  ws1.ensure(s2 * sizeof(double));
  ws2.ensure(s2 * sizeof(double));
  svec_unpack(ctx, side, 1, d_dir, dim, ws1.d());
  two_sided_core(ctx, side, 1, Uinv.d(), KR_LE_N, KR_LE_M, ws1.d(), ws2.d());   // P = U^-T D U^-1 in ws1
  GemmArgs q{};   // Q = P' P
  q.M = side; q.N = side; q.K = side; q.A = ws1.d(); q.lda = side; q.B = ws1.d(); q.ldb = side; q.C = tmpmat2.d(); q.ldc = side;
  q.alpha = 1; q.beta = 0; q.batch = 1;
  gemm(ctx, true, q);
  ctx.d2d(ws1.p, tmpmat2.p, s2 * sizeof(double));
  two_sided_core(ctx, side, 1, UinvT.d(), KR_GE_N, KR_GE_M, ws1.d(), ws2.d());   // U^-1 Q U^-T in ws1
  svec_pack(ctx, side, 1, ws1.d(), dder3v.d(), dim, 1.0);
  return dder3v.d();
}

void PsdCone::dder3_cols(const double* d_dirs, long ldd, int nc, double* d_out, long ldo) {   // :197-207 for nc columns
  ensure_inverses();
  if (nc <= 0) return;
  const long s2 = (long)side * side;
  for (DBuf* b : {&ws1, &ws2, &ws3}) b->ensure((size_t)nc * s2 * sizeof(double));
  svec_unpack(ctx, side, nc, d_dirs, ldd, ws1.d());
  two_sided_core(ctx, side, nc, Uinv.d(), KR_LE_N, KR_LE_M, ws1.d(), ws2.d());   // P_j = U^-T D_j U^-1 in ws1
  GemmArgs q{};   // Q_j = P_j' P_j, one launch for all columns
  q.M = side; q.N = side; q.K = side; q.A = ws1.d(); q.lda = side; q.strideA = s2; q.B = ws1.d(); q.ldb = side; q.strideB = s2;
  q.C = ws3.d(); q.ldc = side; q.strideC = s2;
  q.alpha = 1; q.beta = 0; q.batch = nc;
  gemm(ctx, true, q);
  two_sided_core(ctx, side, nc, UinvT.d(), KR_GE_N, KR_GE_M, ws3.d(), ws2.d());   // U^-1 Q_j U^-T in ws3
  svec_pack(ctx, side, nc, ws3.d(), d_out, ldo, 1.0);
}

}  // namespace hyp
