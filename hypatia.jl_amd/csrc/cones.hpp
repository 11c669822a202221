// Device-resident cone objects: the HIP side of Hypatia's `Cone{T}` protocol
// (/root/reference/src/Cones/Cones.jl:27-310).  The lazy-cache flags live here exactly as in the
// reference (feas_updated, grad_updated, hess_updated, hess_fact_updated ...) so that the Julia /
// Python glue only forwards calls.  All array arguments are DEVICE pointers; the C-ABI wrappers in
// capi.hip stage host buffers.
#pragma once
#include "hyp_internal.hpp"

namespace hyp {

enum ConeKind { CONE_NONNEG = 0, CONE_PSD = 1, CONE_EPINORMSPECTRAL = 2, CONE_WSOS = 3 };

struct Cone {
  Ctx& ctx;
  int kind;
  int dim = 0;
  double nu = 0;
  bool use_dual_barrier = false;
  DBuf point, dual_point, grad, dder3v, vec1, vec2;
  bool feas_updated = false, grad_updated = false, hess_updated = false, inv_hess_updated = false,
       hess_fact_updated = false, is_feas_ = false;

  Cone(Ctx& c, int k) : ctx(c), kind(k) {}
  virtual ~Cone() {}
  void alloc_common();

  // Cones.jl:157-171, 185-186
  void load_point(const double* d_pt, double scal);   // point = scal * pt
  void load_dual_point(const double* d_pt);
  virtual void reset_data() { feas_updated = grad_updated = hess_updated = inv_hess_updated = hess_fact_updated = false; }

  // Cones.jl:56, 63, 71
  bool is_feas() { return feas_updated ? is_feas_ : update_feas(); }
  virtual bool is_dual_feas() { return true; }
  const double* get_grad() {
    if (!grad_updated) update_grad();
    return grad.d();
  }
  virtual bool update_feas() = 0;
  virtual void update_grad() = 0;
  virtual void set_initial_point(double* h_out) = 0;   // host output

  // products: prod, arr are (dim x ncols) device matrices with leading dimensions ldp, lda
  virtual void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) = 0;
  virtual void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) = 0;
  virtual bool use_sqrt_hess_oracles(int arr_dim) = 0;
  virtual void sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) = 0;
  virtual void inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) = 0;
  virtual void hess_prod_slow(double* prod, long ldp, const double* arr, long lda, int ncols) { hess_prod(prod, ldp, arr, lda, ncols); }
  virtual const double* dder3(const double* d_dir) = 0;
  virtual bool use_dder3() { return true; }

  // explicit Hessians (tests / sparse solvers only): dim x dim, upper triangle meaningful
  virtual void hess_explicit(double* d_out, long ld);
  virtual void inv_hess_explicit(double* d_out, long ld);

  // Cones.jl:273-310 (generic; Nonnegative overrides get_proxsqr)
  virtual bool check_numerics();
  virtual double get_proxsqr(double irtmu, bool use_max_prox);

  double dot_host(int n, const double* dx, const double* dy);   // synchronous <x,y>
};

struct NonnegCone : Cone {   // src/Cones/nonnegative.jl
  NonnegCone(Ctx& c, int dim);
  bool update_feas() override;
  bool is_dual_feas() override;
  void update_grad() override;
  void set_initial_point(double* h_out) override;
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  bool use_sqrt_hess_oracles(int) override { return true; }
  void sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  void inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  const double* dder3(const double* d_dir) override;
  double get_proxsqr(double irtmu, bool use_max_prox) override;
};

struct PsdCone : Cone {   // src/Cones/possemideftri.jl (real symmetric)
  int side;
  // side x side col-major device matrices
  DBuf X;        // smat(point), both triangles
  DBuf U;        // Cholesky factor, upper (strict lower zero)
  DBuf UT;       // U'
  DBuf Uinv;     // inv(U), upper
  DBuf UinvT;    // inv(U)'
  DBuf Xinv;     // inv(X), both triangles
  DBuf dinvb;    // inverted diagonal blocks from potrf
  DBuf tmpmat, tmpmat2, d_info;
  DBuf ws1, ws2; // batched workspaces (chunk * side^2)
  bool inv_ready = false;   // Uinv / UinvT / Xinv computed for the current point
  PsdCone(Ctx& c, int dim);
  void reset_data() override {
    Cone::reset_data();
    inv_ready = false;
  }
  bool update_feas() override;
  bool is_dual_feas() override;
  void update_grad() override;
  void set_initial_point(double* h_out) override;
  void ensure_inverses();
  // W_j = R' V_j R for every column j; kr_b / kr_a describe R's triangularity (see cones.hip)
  void two_sided(const double* R, int kr_step2, int kr_step3, double* prod, long ldp, const double* arr, long lda, int ncols);
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  bool use_sqrt_hess_oracles(int) override { return true; }
  void sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  void inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  const double* dder3(const double* d_dir) override;
};

}  // namespace hyp
