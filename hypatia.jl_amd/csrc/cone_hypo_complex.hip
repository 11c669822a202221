// HypoRootdetTri{Float64, ComplexF64} and HypoPerLogdetTri{Float64, ComplexF64} on the device (see cones.hpp: CplxHypoCone).
// Reference: src/Cones/hyporootdettri.jl:9-324, src/Cones/hypoperlogdettri.jl:9-368 with R = Complex{T}; the matrix part of the
// cone vector is the complex svec of arrayutilities.jl:188-262 (d^2 reals).
//
// phi = the real embedding of twice the side (cone_psd_complex.hip): det phi(W) = det(W)^2.  With z = (u, [v,] w):
//   root-determinant:    det(phi W)^(1/2d) = det(W)^(1/d):
//        F_c(u, W) = -log(det(W)^(1/d) - u) - logdet W          = F_r(u, phi W) + logdet W
//   perspective of logdet:  v logdet(phi(W) / v) = 2 v logdet(W / v):
//        F_c(u, v, W) = -log(v logdet(W / v) - u) - log v - logdet W = F_r(2 u, v, phi W) + logdet W + log 2
// i.e. F_c = F_r o T - F_psd with T = diag(lead_scale, E) and F_psd = -logdet W the complex PosSemidefTri barrier (CplxPsdCone,
// already in complex coordinates).  Gradient, Hessian products and the third-order term are linear in the barrier:
//        grad = T' grad_r - [0; grad_psd],   H v = T' H_r T v - [0; H_psd v_w],   dder3(d) = T' dder3_r(T d) - [0; dder3_psd(d_w)],
// E' being twice CplxPsdCone::extract.  The inverse Hessian has no such decomposition: the explicit Hessian (dim = nlead + d^2) is
// formed from dim products and factored, the generic path of Cones.jl:101-118, 239-251 (the reference has a closed form here;
// same operator, different rounding).  nu = nu_r - d.  Dual cones (hyporootdettri.jl:117-127, hypoperlogdettri.jl:120-131): the
// real test on (2 u, phi W) resp. (u, 2 v, phi W) is twice the complex expression.
#include "cones.hpp"

namespace hyp {

namespace {

// out[l, j] = scale[l] * in[l, j] for the nlead leading entries
__global__ void chypo_lead_kernel(int nlead, int ncols, const double* __restrict__ in, long ldi, double* __restrict__ out, long ldo, double s0, double s1) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncols) return;
  out[(long)j * ldo] = s0 * in[(long)j * ldi];
  if (nlead > 1) out[(long)j * ldo + 1] = s1 * in[(long)j * ldi + 1];
}
// y[i, j] = a * y[i, j] + b * x[i, j]
__global__ void chypo_comb_kernel(int m, int ncols, double a, double* __restrict__ y, long ldy, double b, const double* __restrict__ x, long ldx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  for (int j = blockIdx.y; j < ncols; j += gridDim.y) y[(long)j * ldy + i] = a * y[(long)j * ldy + i] + b * x[(long)j * ldx + i];
}

int side_of_square(int n) { return (int)(sqrt((double)n) + 0.5); }

}  // namespace

CplxHypoCone::CplxHypoCone(Ctx& c, int kind_, int dim_, bool perlog, bool use_dual)
    : GenericHessCone(c, kind_), d(side_of_square(dim_ - (perlog ? 2 : 1))), nlead(perlog ? 2 : 1), psdc(c, side_of_square(dim_ - (perlog ? 2 : 1)) * side_of_square(dim_ - (perlog ? 2 : 1))) {
  HYP_REQUIRE(dim_ > nlead && (long)d * d == dim_ - nlead, "complex hypograph cone: dim = (1 or 2) + side^2");
  dim = dim_;
  cdw = (long)d * d;
  edw = (long)d * (2L * d + 1);
  edim = nlead + edw;
  use_dual_barrier = use_dual;
  if (perlog) {
    inner = new HypoPerLogdetTriCone(c, (int)edim, use_dual);
    lead_scale[0] = 2.0; lead_scale[1] = 1.0;
    dual_scale[0] = 1.0; dual_scale[1] = 2.0;
    nu = d + 2;        // hypoperlogdettri.jl:78
  } else {
    inner = new HypoRootdetTriCone(c, (int)edim, use_dual);
    lead_scale[0] = 1.0; lead_scale[1] = 1.0;
    dual_scale[0] = 2.0; dual_scale[1] = 1.0;
    nu = d + 1;        // hyporootdettri.jl:80
  }
  alloc_common();
  alloc_generic();
}

void CplxHypoCone::set_initial_point(double* h) {   // hyporootdettri.jl:82-99, hypoperlogdettri.jl:80-95 (side d; diagonal of the complex svec)
  for (int i = 0; i < dim; ++i) h[i] = 0.0;
  double w;
  if (nlead == 1) {
    const double dd = d;
    const double c1 = sqrt(5 * dd * dd + 2 * dd + 1);
    const double c2 = h[0] = -sqrt((3 * dd + 1 - c1) / (2 * dd + 2));
    w = -c2 * (dd + 1 + c1) / (2 * dd);
  } else {
    double uvw[3];
    central_ray_hypoperlog(d, uvw);
    h[0] = uvw[0]; h[1] = uvw[1];
    w = uvw[2];
  }
  long k = nlead;
  for (int i = 1; i <= d; ++i) { h[k] = w; k += 2 * i + 1; }
}

void CplxHypoCone::to_inner(const double* cvec, long ldc, double* evec, int ncols, const double* scale) {
  hipLaunchKernelGGL(chypo_lead_kernel, dim3((ncols + 63) / 64), dim3(64), 0, ctx.stream, nlead, ncols, cvec, ldc, evec, edim, scale[0], scale[1]);
  HYP_CHECK(hipGetLastError());
  psdc.embed(cvec + nlead, ldc, evec + nlead, ncols, edim);
}
void CplxHypoCone::from_inner(const double* evec, double* cvec, long ldc, int ncols) {   // T' = diag(lead_scale, 2 extract)
  hipLaunchKernelGGL(chypo_lead_kernel, dim3((ncols + 63) / 64), dim3(64), 0, ctx.stream, nlead, ncols, evec, edim, cvec, ldc, lead_scale[0], lead_scale[1]);
  HYP_CHECK(hipGetLastError());
  psdc.extract(evec + nlead, cvec + nlead, ldc, ncols, edim);
}

bool CplxHypoCone::update_feas() {
  to_inner(point.d(), dim, inner->point.d(), 1, lead_scale);
  inner->reset_data();
  is_feas_ = inner->is_feas();
  if (is_feas_) {
    ctx.d2d(psdc.point.p, point.d() + nlead, (size_t)cdw * sizeof(double));
    psdc.reset_data();
    is_feas_ = psdc.is_feas();
  }
  feas_updated = true;
  return is_feas_;
}

bool CplxHypoCone::is_dual_feas() {
  ea.ensure((size_t)edim * sizeof(double));
  to_inner(dual_point.d(), dim, ea.d(), 1, dual_scale);
  inner->load_dual_point(ea.d());
  return inner->is_dual_feas();
}

void CplxHypoCone::update_grad() {
  HYP_REQUIRE(feas_updated && is_feas_, "grad: the point is not known to be feasible");
  from_inner(inner->get_grad(), grad.d(), dim, 1);
  // W part: E' grad_r - grad_psd = 2 extract(.) - grad_psd
  hipLaunchKernelGGL(chypo_comb_kernel, dim3((unsigned)((cdw + 255) / 256), 1), dim3(256), 0, ctx.stream, (int)cdw, 1, 2.0, grad.d() + nlead, (long)dim, -1.0,
                     psdc.get_grad(), cdw);
  HYP_CHECK(hipGetLastError());
  grad_updated = true;
}

void CplxHypoCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {
  HYP_REQUIRE(feas_updated && is_feas_, "hess_prod: the point is not known to be feasible");
  if (ncols <= 0) return;
  const int chunk = (int)std::max<long>(1, std::min<long>(ncols, (1L << 26) / edim));
  ea.ensure((size_t)edim * chunk * sizeof(double));
  eb.ensure((size_t)edim * chunk * sizeof(double));
  pw.ensure((size_t)cdw * chunk * sizeof(double));
  for (int j0 = 0; j0 < ncols; j0 += chunk) {
    const int nc = std::min(chunk, ncols - j0);
    const double* a = arr + (long)j0 * lda;
    double* o = prod + (long)j0 * ldp;
    to_inner(a, lda, ea.d(), nc, lead_scale);
    psdc.hess_prod(pw.d(), cdw, a + nlead, lda, nc);        // (before prod is written: arr may alias prod)
    inner->hess_prod(eb.d(), edim, ea.d(), edim, nc);
    from_inner(eb.d(), o, ldp, nc);
    hipLaunchKernelGGL(chypo_comb_kernel, dim3((unsigned)((cdw + 255) / 256), (unsigned)std::min(nc, 1024)), dim3(256), 0, ctx.stream, (int)cdw, nc, 2.0,
                       o + nlead, ldp, -1.0, pw.d(), cdw);
    HYP_CHECK(hipGetLastError());
  }
}

void CplxHypoCone::update_hess() {   // explicit Hessian: products with the identity
  ensure_hess_storage(false);
  get_grad();
  dev_fill_identity(ctx, dim, H.d(), dim);
  hess_prod(H.d(), dim, H.d(), dim, dim);
  hess_updated = true;
}

const double* CplxHypoCone::dder3(const double* d_dir) {
  HYP_REQUIRE(feas_updated && is_feas_, "dder3: the point is not known to be feasible");
  ea.ensure((size_t)edim * sizeof(double));
  to_inner(d_dir, dim, ea.d(), 1, lead_scale);
  const double* p3 = psdc.dder3(d_dir + nlead);
  from_inner(inner->dder3(ea.d()), dder3v.d(), dim, 1);
  hipLaunchKernelGGL(chypo_comb_kernel, dim3((unsigned)((cdw + 255) / 256), 1), dim3(256), 0, ctx.stream, (int)cdw, 1, 2.0, dder3v.d() + nlead, (long)dim,
                     -1.0, p3, cdw);
  HYP_CHECK(hipGetLastError());
  return dder3v.d();
}

}  // namespace hyp
