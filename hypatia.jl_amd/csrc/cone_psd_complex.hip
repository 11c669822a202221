// PosSemidefTri{Float64, ComplexF64} on the device through the interleaved real embedding (see cones.hpp: CplxPsdCone).
// Reference: src/Cones/possemideftri.jl:9-207 (R = Complex{T}); src/Cones/arrayutilities.jl:188-210 (smat_to_svec!),
// :240-262 (svec_to_smat!): column by column over the upper triangle, a real diagonal entry, and for i < j the pair
// (re, -im) of sqrt(2) mat[i, j].
#include "cones.hpp"

namespace hyp {

namespace {

__device__ __forceinline__ long eidx(int r, int c) { return (long)c * (c + 1) / 2 + r; }   // upper triangle, r <= c

// one thread per (i <= j) entry of the complex matrix and column: position k of the complex svec = j^2 + 2 i (j^2 + 2 j on the
// diagonal), since column j holds 2 j + 1 reals
__global__ void cpsd_embed_kernel(int s, int ncols, const double* __restrict__ cvec, long ldc, double* __restrict__ evec, long lde) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long npair = (long)s * (s + 1) / 2;
  if (t >= npair) return;
  int j = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((long)(j + 1) * (j + 2) / 2 <= t) ++j;
  while ((long)j * (j + 1) / 2 > t) --j;
  const int i = (int)(t - (long)j * (j + 1) / 2);
  const long k = (long)j * j + 2 * i;
  for (int col = blockIdx.y; col < ncols; col += gridDim.y) {
    const double* cv = cvec + (long)col * ldc;
    double* ev = evec + (long)col * lde;
    if (i == j) {
      const double a = cv[k];
      ev[eidx(2 * i, 2 * i)] = a;
      ev[eidx(2 * i + 1, 2 * i + 1)] = a;
      ev[eidx(2 * i, 2 * i + 1)] = 0.0;
    } else {   // cv[k] = sqrt(2) a_ij, cv[k + 1] = -sqrt(2) b_ij; the embedded off-diagonals carry sqrt(2) too
      const double re = cv[k], mi = cv[k + 1];
      ev[eidx(2 * i, 2 * j)] = re;
      ev[eidx(2 * i + 1, 2 * j + 1)] = re;
      ev[eidx(2 * i, 2 * j + 1)] = mi;        // -b_ij
      ev[eidx(2 * i + 1, 2 * j)] = -mi;       //  b_ij
    }
  }
}

__global__ void cpsd_extract_kernel(int s, int ncols, const double* __restrict__ evec, long lde, double* __restrict__ cvec, long ldc) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long npair = (long)s * (s + 1) / 2;
  if (t >= npair) return;
  int j = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((long)(j + 1) * (j + 2) / 2 <= t) ++j;
  while ((long)j * (j + 1) / 2 > t) --j;
  const int i = (int)(t - (long)j * (j + 1) / 2);
  const long k = (long)j * j + 2 * i;
  for (int col = blockIdx.y; col < ncols; col += gridDim.y) {
    const double* ev = evec + (long)col * lde;
    double* cv = cvec + (long)col * ldc;
    if (i == j) {
      cv[k] = 0.5 * (ev[eidx(2 * i, 2 * i)] + ev[eidx(2 * i + 1, 2 * i + 1)]);
    } else {
      cv[k] = 0.5 * (ev[eidx(2 * i, 2 * j)] + ev[eidx(2 * i + 1, 2 * j + 1)]);
      cv[k + 1] = 0.5 * (ev[eidx(2 * i, 2 * j + 1)] - ev[eidx(2 * i + 1, 2 * j)]);
    }
  }
}

int csvec_side_of(int dim) {
  int s = (int)(sqrt((double)dim) + 0.5);
  return s;
}

}  // namespace

CplxPsdCone::CplxPsdCone(Ctx& c, int d) : Cone(c, CONE_PSD_COMPLEX), side(csvec_side_of(d)), edim((long)csvec_side_of(d) * (2L * csvec_side_of(d) + 1)),
                                          inner(c, (int)((long)csvec_side_of(d) * (2L * csvec_side_of(d) + 1))) {
  HYP_REQUIRE(d >= 1 && (long)side * side == d, "PosSemidefTri (complex): dim must be a square (arrayutilities.jl:103-108)");
  dim = d;
  nu = side;   // possemideftri.jl:67
  alloc_common();
}

void CplxPsdCone::set_initial_point(double* h) {   // :69-78 (2 i + 1 between diagonal entries)
  for (int i = 0; i < dim; ++i) h[i] = 0.0;
  long k = 0;
  for (int i = 1; i <= side; ++i) {
    h[k] = 1.0;
    k += 2 * i + 1;
  }
}

void CplxPsdCone::embed(const double* cvec, long ldc, double* evec, int ncols, long lde) {
  const long npair = (long)side * (side + 1) / 2;
  hipLaunchKernelGGL(cpsd_embed_kernel, dim3((unsigned)((npair + 255) / 256), (unsigned)std::min(ncols, 1024)), dim3(256), 0, ctx.stream, side, ncols, cvec,
                     ldc, evec, lde > 0 ? lde : edim);
  HYP_CHECK(hipGetLastError());
}
void CplxPsdCone::extract(const double* evec, double* cvec, long ldc, int ncols, long lde) {
  const long npair = (long)side * (side + 1) / 2;
  hipLaunchKernelGGL(cpsd_extract_kernel, dim3((unsigned)((npair + 255) / 256), (unsigned)std::min(ncols, 1024)), dim3(256), 0, ctx.stream, side, ncols, evec,
                     lde > 0 ? lde : edim, cvec, ldc);
  HYP_CHECK(hipGetLastError());
}

bool CplxPsdCone::update_feas() {   // :80-90
  embed(point.d(), dim, inner.point.d(), 1);
  inner.reset_data();
  is_feas_ = inner.is_feas();
  feas_updated = true;
  return is_feas_;
}

bool CplxPsdCone::is_dual_feas() {   // :92-95
  ea.ensure((size_t)edim * sizeof(double));
  embed(dual_point.d(), dim, ea.d(), 1);
  inner.load_dual_point(ea.d());
  return inner.is_dual_feas();
}

void CplxPsdCone::update_grad() {   // :97-107: -svec(X^-1)
  HYP_REQUIRE(feas_updated && is_feas_, "grad: the point is not known to be feasible");
  extract(inner.get_grad(), grad.d(), dim, 1);
  grad_updated = true;
}

template <class F>
void CplxPsdCone::through(F f, double* prod, long ldp, const double* arr, long lda, int ncols) {
  HYP_REQUIRE(feas_updated && is_feas_, "product: the point is not known to be feasible");
  // chunks of at most 2^27 doubles of embedded workspace per buffer
  const int chunk = (int)std::max<long>(1, std::min<long>(ncols, (1L << 27) / edim));
  ea.ensure((size_t)edim * chunk * sizeof(double));
  eb.ensure((size_t)edim * chunk * sizeof(double));
  for (int j0 = 0; j0 < ncols; j0 += chunk) {
    const int nc = std::min(chunk, ncols - j0);
    embed(arr + (long)j0 * lda, lda, ea.d(), nc);
    f(eb.d(), ea.d(), nc);
    extract(eb.d(), prod + (long)j0 * ldp, ldp, nc);
  }
}

void CplxPsdCone::hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {           // :126-142
  through([&](double* o, const double* a, int nc) { inner.hess_prod(o, edim, a, edim, nc); }, prod, ldp, arr, lda, ncols);
}
void CplxPsdCone::inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {       // :144-159
  through([&](double* o, const double* a, int nc) { inner.inv_hess_prod(o, edim, a, edim, nc); }, prod, ldp, arr, lda, ncols);
}
void CplxPsdCone::sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {      // :161-177
  through([&](double* o, const double* a, int nc) { inner.sqrt_hess_prod(o, edim, a, edim, nc); }, prod, ldp, arr, lda, ncols);
}
void CplxPsdCone::inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) {  // :179-195
  through([&](double* o, const double* a, int nc) { inner.inv_sqrt_hess_prod(o, edim, a, edim, nc); }, prod, ldp, arr, lda, ncols);
}

const double* CplxPsdCone::dder3(const double* d_dir) {   // :197-207
  HYP_REQUIRE(feas_updated && is_feas_, "dder3: the point is not known to be feasible");
  ea.ensure((size_t)edim * sizeof(double));
  embed(d_dir, dim, ea.d(), 1);
  extract(inner.dder3(ea.d()), dder3v.d(), dim, 1);
  return dder3v.d();
}

}  // namespace hyp
