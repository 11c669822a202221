// EpiNormSpectral{Float64, ComplexF64} on the device through the real embedding (see cones.hpp: CplxEnsCone).
// Reference: src/Cones/epinormspectral.jl:13-294 with R = Complex{T}; the cone vector is (u, W) with the complex d1 x d2
// matrix W held as (re, im) pairs in column-major order (vec_copyto!, src/Cones/arrayutilities.jl:30-60).
//
// phi(a + ib) = [[a, -b], [b, a]] maps W to a real (2 d1) x (2 d2) matrix with phi(W W^H) = phi(W) phi(W)' and
// det phi(M) = |det M|^2, so for the barrier of this Hypatia version, F(u, W) = -logdet(u^2 I - W W^H) + (d1 - 1) log u,
//     F_complex(u, W) = 1/2 F_real(u, phi(W)) - 1/2 log u                (F_real: the real cone of sides 2 d1, 2 d2).
// With E the embedding of the (re, im) pairs (E'E = 2 I) and Ehat = diag(1, E):
//     grad   = 1/2 Ehat' grad_real - e_u / (2 u)
//     hess   = 1/2 Ehat' H_real Ehat + e_u e_u' / (2 u^2)                 =: A + e_u e_u' / (2 u^2)
//     dder3  = 1/2 Ehat' dder3_real(Ehat d) + e_u d_u^2 / (2 u^3)
// and, the range of Ehat being invariant under H_real, A^-1 y = Ehat^+ H_real^-1 Ehat diag(2, I) y, so that the inverse
// Hessian is one application of the real cone's (closed-form) inverse and a Sherman-Morrison correction for the rank-one term:
//     hess^-1 y = x - a x_u / (2 u^2 + a_u),   x = A^-1 y,  a = A^-1 e_u.
// Dual cone (epinormspectral.jl:125-132): u > sum of the singular values of W; those of phi(W) are the same, each twice.
#include "cones.hpp"

namespace hyp {

namespace {

// complex entry (i, j) of the d1 x d2 matrix sits at 1 + 2 (j d1 + i) of the cone vector; embedded vector: [u; vec(phi(W))]
__global__ void cens_embed_kernel(int d1, int d2, int ncols, const double* __restrict__ cvec, long ldc, double* __restrict__ evec, long lde,
                                  double uscale) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nent = (long)d1 * d2;
  if (t >= nent) return;
  const int i = (int)(t % d1), j = (int)(t / d1);
  const long ld = 2L * d1;
  for (int col = blockIdx.y; col < ncols; col += gridDim.y) {
    const double* cv = cvec + (long)col * ldc;
    double* ev = evec + (long)col * lde;
    const double a = cv[1 + 2 * t], b = cv[2 + 2 * t];
    double* w = ev + 1;
    w[(2L * j) * ld + 2 * i] = a;          w[(2L * j + 1) * ld + 2 * i + 1] = a;
    w[(2L * j) * ld + 2 * i + 1] = b;      w[(2L * j + 1) * ld + 2 * i] = -b;
    if (t == 0) ev[0] = uscale * cv[0];
  }
}

// Ehat^+ on the matrix part (mean of the two copies), uscale on u
__global__ void cens_extract_kernel(int d1, int d2, int ncols, const double* __restrict__ evec, long lde, double* __restrict__ cvec, long ldc,
                                    double uscale) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nent = (long)d1 * d2;
  if (t >= nent) return;
  const int i = (int)(t % d1), j = (int)(t / d1);
  const long ld = 2L * d1;
  for (int col = blockIdx.y; col < ncols; col += gridDim.y) {
    const double* w = evec + (long)col * lde + 1;
    double* cv = cvec + (long)col * ldc;
    cv[1 + 2 * t] = 0.5 * (w[(2L * j) * ld + 2 * i] + w[(2L * j + 1) * ld + 2 * i + 1]);
    cv[2 + 2 * t] = 0.5 * (w[(2L * j) * ld + 2 * i + 1] - w[(2L * j + 1) * ld + 2 * i]);
    if (t == 0) cv[0] = uscale * evec[(long)col * lde];
  }
}

// out[0, j] += c1 * src[0, j] + c2 * src[0, j]^2
__global__ void cens_add0_kernel(int ncols, double* __restrict__ out, long ldo, const double* __restrict__ src, long lds, double c1, double c2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncols) return;
  const double s = src[(long)j * lds];
  out[(long)j * ldo] += c1 * s + c2 * s * s;
}

// p[0] = v (add = 0) or p[0] += v: scalars travel as kernel arguments, not through an asynchronous copy from the host stack
__global__ void cens_scalar0_kernel(double* __restrict__ p, double v, int add) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = add ? p[0] + v : v;
}

// out[:, j] = x[:, j] - a * x[0, j] / (twou2 + a[0])
__global__ void cens_sm_kernel(int dim, int ncols, const double* __restrict__ x, long ldx, const double* __restrict__ a, double twou2,
                               double* __restrict__ out, long ldo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dim) return;
  const double den = twou2 + a[0], ai = a[i];
  for (int j = blockIdx.y; j < ncols; j += gridDim.y) out[(long)j * ldo + i] = x[(long)j * ldx + i] - ai * (x[(long)j * ldx] / den);
}

}  // namespace

CplxEnsCone::CplxEnsCone(Ctx& c, int d1_, int d2_, bool use_dual)
    : Cone(c, CONE_EPINORMSPECTRAL_COMPLEX), d1(d1_), d2(d2_), edim(1 + 4L * d1_ * d2_), inner(c, 2 * d1_, 2 * d2_, use_dual) {
  HYP_REQUIRE(d1 >= 1 && d1 <= d2, "EpiNormSpectral (complex): 1 <= d1 <= d2");   // epinormspectral.jl:55
  dim = 1 + 2 * d1 * d2;
  nu = d1 + 1;                                                                      // :97
  use_dual_barrier = use_dual;
  alloc_common();
  ainv.alloc((size_t)dim * sizeof(double));
}

void CplxEnsCone::set_initial_point(double* h) {   // :99-105
  for (int i = 0; i < dim; ++i) h[i] = 0.0;
  h[0] = sqrt(nu);
}

void CplxEnsCone::embed(const double* cvec, long ldc, double* evec, int ncols, double uscale) {
  const long nent = (long)d1 * d2;
  hipLaunchKernelGGL(cens_embed_kernel, dim3((unsigned)((nent + 255) / 256), (unsigned)std::min(ncols, 1024)), dim3(256), 0, ctx.stream, d1, d2, ncols,
                     cvec, ldc, evec, edim, uscale);
  HYP_CHECK(hipGetLastError());
}
void CplxEnsCone::extract(const double* evec, double* cvec, long ldc, int ncols, double uscale) {
  const long nent = (long)d1 * d2;
  hipLaunchKernelGGL(cens_extract_kernel, dim3((unsigned)((nent + 255) / 256), (unsigned)std::min(ncols, 1024)), dim3(256), 0, ctx.stream, d1, d2, ncols,
                     evec, edim, cvec, ldc, uscale);
  HYP_CHECK(hipGetLastError());
}

bool CplxEnsCone::update_feas() {   // :107-123
  embed(point.d(), dim, inner.point.d(), 1, 1.0);
  inner.reset_data();
  ainv_ready = false;
  is_feas_ = inner.is_feas();
  feas_updated = true;
  return is_feas_;
}

bool CplxEnsCone::is_dual_feas() {   // :125-132: u - sum(svdvals(W)) > eps; the embedded matrix has every singular value twice
  ea.ensure((size_t)edim * sizeof(double));
  embed(dual_point.d(), dim, ea.d(), 1, 2.0);
  inner.load_dual_point(ea.d());
  return inner.is_dual_feas();
}

void CplxEnsCone::update_grad() {   // :134-150
  HYP_REQUIRE(feas_updated && is_feas_, "grad: the point is not known to be feasible");
  extract(inner.get_grad(), grad.d(), dim, 1, 0.5);
  const double corr = -0.5 / inner.u;
  dev_axpby_scalar0(corr);
  grad_updated = true;
}

// grad[0] += v (one element; through the stream)
void CplxEnsCone::dev_axpby_scalar0(double v) {
  hipLaunchKernelGGL(cens_scalar0_kernel, dim3(1), dim3(64), 0, ctx.stream, grad.d(), v, 1);
  HYP_CHECK(hipGetLastError());
}

void CplxEnsCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :211-239
  HYP_REQUIRE(feas_updated && is_feas_, "hess_prod: the point is not known to be feasible");
  if (ncols <= 0) return;
  const double u = inner.u;
  const int chunk = (int)std::max<long>(1, std::min<long>(ncols, (1L << 26) / edim));
  ea.ensure((size_t)edim * chunk * sizeof(double));
  eb.ensure((size_t)edim * chunk * sizeof(double));
  for (int j0 = 0; j0 < ncols; j0 += chunk) {
    const int nc = std::min(chunk, ncols - j0);
    embed(arr + (long)j0 * lda, lda, ea.d(), nc, 1.0);
    inner.hess_prod(eb.d(), edim, ea.d(), edim, nc);
    // (arr may alias prod: its u entries are still needed -- they sit in ea)
    extract(eb.d(), prod + (long)j0 * ldp, ldp, nc, 0.5);
    hipLaunchKernelGGL(cens_add0_kernel, dim3((nc + 63) / 64), dim3(64), 0, ctx.stream, nc, prod + (long)j0 * ldp, ldp, ea.d(), edim, 0.5 / (u * u), 0.0);
    HYP_CHECK(hipGetLastError());
  }
}

void CplxEnsCone::apply_ainv(double* xc, long ldx, const double* arr, long lda, int ncols) {   // xc = A^-1 arr (complex coordinates)
  const int chunk = (int)std::max<long>(1, std::min<long>(ncols, (1L << 26) / edim));
  ea.ensure((size_t)edim * chunk * sizeof(double));
  eb.ensure((size_t)edim * chunk * sizeof(double));
  for (int j0 = 0; j0 < ncols; j0 += chunk) {
    const int nc = std::min(chunk, ncols - j0);
    embed(arr + (long)j0 * lda, lda, ea.d(), nc, 2.0);
    inner.inv_hess_prod(eb.d(), edim, ea.d(), edim, nc);
    extract(eb.d(), xc + (long)j0 * ldx, ldx, nc, 1.0);
  }
}

void CplxEnsCone::inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {
  HYP_REQUIRE(feas_updated && is_feas_, "inv_hess_prod: the point is not known to be feasible");
  if (ncols <= 0) return;
  const double u = inner.u;
  if (!ainv_ready) {   // a = A^-1 e_u
    xw.ensure((size_t)dim * sizeof(double));
    ctx.zero(xw.p, (size_t)dim * sizeof(double));
    hipLaunchKernelGGL(cens_scalar0_kernel, dim3(1), dim3(64), 0, ctx.stream, xw.d(), 1.0, 0);
    HYP_CHECK(hipGetLastError());
    apply_ainv(ainv.d(), dim, xw.d(), dim, 1);
    ainv_ready = true;
  }
  xw.ensure((size_t)dim * ncols * sizeof(double));
  apply_ainv(xw.d(), dim, arr, lda, ncols);
  hipLaunchKernelGGL(cens_sm_kernel, dim3((dim + 255) / 256, (unsigned)std::min(ncols, 1024)), dim3(256), 0, ctx.stream, dim, ncols, xw.d(), (long)dim,
                     ainv.d(), 2.0 * u * u, prod, ldp);
  HYP_CHECK(hipGetLastError());
}

const double* CplxEnsCone::dder3(const double* d_dir) {   // :241-294
  HYP_REQUIRE(feas_updated && is_feas_, "dder3: the point is not known to be feasible");
  const double u = inner.u;
  ea.ensure((size_t)edim * sizeof(double));
  embed(d_dir, dim, ea.d(), 1, 1.0);
  extract(inner.dder3(ea.d()), dder3v.d(), dim, 1, 0.5);
  hipLaunchKernelGGL(cens_add0_kernel, dim3(1), dim3(64), 0, ctx.stream, 1, dder3v.d(), (long)dim, ea.d(), edim, 0.0, 0.5 / (u * u * u));
  HYP_CHECK(hipGetLastError());
  return dder3v.d();
}

}  // namespace hyp
