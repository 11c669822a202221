// WSOSInterpNonnegative{Float64, ComplexF64} on the device (see cones.hpp: CplxWsosCone).
// Reference: src/Cones/wsosinterpnonnegative.jl:15-200 with R = Complex{T}: the cone vector stays REAL (U interpolant values),
// the bases P_k are complex U x L_k (src/PolyUtils/complex.jl:13-72) and Lambda_k = P_k' Diag(pt) P_k is Hermitian (:97-111).
//
// phi(a + ib) = [[a, -b], [b, a]] (block form) is a *-homomorphism, also on rectangular matrices: phi(P') = phi(P)^T,
// phi(P' D P) = phi(P)^T diag(d, d) phi(P), and det phi(L) = det(L)^2 for Hermitian L.  So with E = [I; I] (2U x U)
//        F_c(pt) = -sum_k logdet(Lambda_k) = 1/2 F_r(E pt),   F_r = the REAL cone's barrier for the bases phi(P_k) (2U x 2L_k),
// and everything that is linear in the barrier follows (WsosCone does the work at the duplicated point):
//        grad = 1/2 E' grad_r,   H = 1/2 E' H_r E,   hess_prod_slow(v) = 1/2 E' hps_r(E v),   dder3(d) = 1/2 E' dder3_r(E d),
// feasible iff the embedded point is (Lambda_k > 0 iff phi(Lambda_k) > 0), nu = sum_k L_k.  The explicit Hessian is the fold of
// the four U x U blocks of the real cone's: (M_r o M_r + M_i o M_i)[i, j] = |M[i, j]|^2 for M = P Lambda^-1 P', the reference's
// abs2(UU[i, j]) (:141-146).  The inverse Hessian, the square-root oracles and the proximity test go through the generic
// factored-Hessian path on that U x U matrix (Cones.jl:101-118, 189-259), as for the real cone.
#include "cones.hpp"

namespace hyp {

namespace {

// out[:, j] = [in[:, j]; in[:, j]]   (E)
__global__ void cwsos_dup_kernel(int U, int ncols, const double* __restrict__ in, long ldi, double* __restrict__ out, long ldo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= U) return;
  for (int j = blockIdx.y; j < ncols; j += gridDim.y) {
    const double v = in[(long)j * ldi + i];
    out[(long)j * ldo + i] = v;
    out[(long)j * ldo + U + i] = v;
  }
}
// out[:, j] = 1/2 (in[0:U, j] + in[U:2U, j])   (1/2 E')
__global__ void cwsos_fold_kernel(int U, int ncols, const double* __restrict__ in, long ldi, double* __restrict__ out, long ldo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= U) return;
  for (int j = blockIdx.y; j < ncols; j += gridDim.y) out[(long)j * ldo + i] = 0.5 * (in[(long)j * ldi + i] + in[(long)j * ldi + U + i]);
}
// H = 1/2 E' Hr E: the four U x U blocks of the 2U x 2U matrix
__global__ void cwsos_fold_hess_kernel(int U, const double* __restrict__ Hr, double* __restrict__ H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= U) return;
  const long ld = 2L * U;
  H[(long)j * U + i] = 0.5 * ((Hr[(long)j * ld + i] + Hr[(long)(U + j) * ld + U + i]) + (Hr[(long)(U + j) * ld + i] + Hr[(long)j * ld + U + i]));
}

WsosCone* make_inner(Ctx& c, int U, int K, const int* Ls, const double* const* hPs, bool use_dual) {
  HYP_REQUIRE(U >= 1 && K >= 1, "complex WSOS: sizes");
  std::vector<std::vector<double>> emb(K);
  std::vector<const double*> ptrs(K);
  std::vector<int> L2(K);
  const long U2 = 2L * U;
  for (int k = 0; k < K; ++k) {
    const int L = Ls[k];
    HYP_REQUIRE(L >= 1 && L <= U, "complex WSOS: 1 <= L_k <= U");
    L2[k] = 2 * L;
    emb[k].assign((size_t)U2 * 2 * L, 0.0);
    const double* P = hPs[k];   // U x L complex numbers, (re, im) interleaved, column-major
    for (int l = 0; l < L; ++l)
      for (int i = 0; i < U; ++i) {
        const double re = P[2 * ((long)l * U + i)], im = P[2 * ((long)l * U + i) + 1];
        emb[k][(long)l * U2 + i] = re;            // [[Pr, -Pi],
        emb[k][(long)l * U2 + U + i] = im;        //  [Pi,  Pr]]
        emb[k][(long)(L + l) * U2 + i] = -im;
        emb[k][(long)(L + l) * U2 + U + i] = re;
      }
    ptrs[k] = emb[k].data();
  }
  return new WsosCone(c, (int)U2, K, L2.data(), ptrs.data(), use_dual);
}

}  // namespace

CplxWsosCone::CplxWsosCone(Ctx& c, int U_, int K, const int* Ls, const double* const* hPs, bool use_dual)
    : GenericHessCone(c, CONE_WSOS_COMPLEX), U(U_), inner(make_inner(c, U_, K, Ls, hPs, use_dual)) {
  dim = U;
  use_dual_barrier = !use_dual;   // wsosinterpnonnegative.jl:58
  nu = 0;
  for (int k = 0; k < K; ++k) nu += Ls[k];   // :61 (complex L_k, half the embedded cone's)
  alloc_common();
  alloc_generic();
}

void CplxWsosCone::set_initial_point(double* h) {   // :87
  for (int i = 0; i < dim; ++i) h[i] = 1.0;
}

void CplxWsosCone::dup(const double* in, long ldi, double* out, int ncols) {
  hipLaunchKernelGGL(cwsos_dup_kernel, dim3((U + 255) / 256, (unsigned)std::min(ncols, 1024)), dim3(256), 0, ctx.stream, U, ncols, in, ldi, out, 2L * U);
  HYP_CHECK(hipGetLastError());
}
void CplxWsosCone::fold(const double* in, double* out, long ldo, int ncols) {
  hipLaunchKernelGGL(cwsos_fold_kernel, dim3((U + 255) / 256, (unsigned)std::min(ncols, 1024)), dim3(256), 0, ctx.stream, U, ncols, in, 2L * U, out, ldo);
  HYP_CHECK(hipGetLastError());
}

bool CplxWsosCone::update_feas() {   // :89-117
  dup(point.d(), dim, inner->point.d(), 1);
  inner->reset_data();
  is_feas_ = inner->is_feas();
  feas_updated = true;
  return is_feas_;
}

void CplxWsosCone::update_grad() {   // :119-133
  HYP_REQUIRE(feas_updated && is_feas_, "grad: the point is not known to be feasible");
  fold(inner->get_grad(), grad.d(), dim, 1);
  grad_updated = true;
}

void CplxWsosCone::update_hess() {   // :135-150
  ensure_hess_storage(false);
  get_grad();
  if (!inner->hess_updated) inner->update_hess();
  hipLaunchKernelGGL(cwsos_fold_hess_kernel, dim3((U + 255) / 256, U), dim3(256), 0, ctx.stream, U, inner->H.d(), H.d());
  HYP_CHECK(hipGetLastError());
  hess_updated = true;
}

void CplxWsosCone::hess_prod_slow(double* prod, long ldp, const double* arr, long lda, int ncols) {   // :152-175
  if (!use_hess_prod_slow_updated) update_use_hess_prod_slow();
  if (!use_hess_prod_slow) {
    hess_prod(prod, ldp, arr, lda, ncols);
    return;
  }
  get_grad();
  inner->use_hess_prod_slow = true;   // (the decision is this cone's, Cones.jl:222-231 on the U x U Hessian; the sums are the embedded cone's)
  inner->use_hess_prod_slow_updated = true;
  ea.ensure((size_t)2 * U * sizeof(double));
  eb.ensure((size_t)2 * U * sizeof(double));
  for (int j = 0; j < ncols; ++j) {
    dup(arr + (long)j * lda, lda, ea.d(), 1);
    inner->hess_prod_slow(eb.d(), 2L * U, ea.d(), 2L * U, 1);
    fold(eb.d(), prod + (long)j * ldp, ldp, 1);
  }
}

const double* CplxWsosCone::dder3(const double* d_dir) {   // :177-188
  HYP_REQUIRE(feas_updated && is_feas_, "dder3: the point is not known to be feasible");
  get_grad();
  ea.ensure((size_t)2 * U * sizeof(double));
  dup(d_dir, dim, ea.d(), 1);
  fold(inner->dder3(ea.d()), dder3v.d(), dim, 1);
  return dder3v.d();
}

}  // namespace hyp
