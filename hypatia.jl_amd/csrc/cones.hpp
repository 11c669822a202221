// Device-resident cone objects: the HIP side of Hypatia's `Cone{T}` protocol
// (/root/reference/src/Cones/Cones.jl:27-310).  The lazy-cache flags live here exactly as in the
// reference (feas_updated, grad_updated, hess_updated, hess_fact_updated ...) so that the Julia /
// Python glue only forwards calls.  All array arguments are DEVICE pointers; the C-ABI wrappers in
// capi.hip stage host buffers.
#pragma once
#include <memory>
#include "hyp_internal.hpp"

namespace hyp {

// psd_twosided.hip: prod[:, j] = svec(R' smat(arr[:, j]) R) with the svec conversions fused (side <= 208)
bool psd_two_sided_fused_ok(int side);
// one workgroup per matrix, the intermediate product in registers (psd_twosided4.hip); false: not applicable, nothing done
bool psd_two_sided_onchip(Ctx& c, int side, int ncols, const double* R, int rstruct, const double* arr, long lda, double* prod, long ldp);
// the same for sides of 3 .. 5 tiles (33 .. 80; config 4): one wavefront per matrix, all of Z in its accumulators (psd_twosided5.hip); arr may alias prod
bool psd_two_sided_wave(Ctx& c, int side, int ncols, const double* R, int rstruct, const double* arr, long lda, double* prod, long ldp);
void psd_two_sided_fused(Ctx& c, int side, int ncols, const double* R, int rstruct /* 0 full, 1 upper, 2 lower */, const double* arr,
                         long lda, double* prod, long ldp, double* zws /* ncols * side^2 */);

enum ConeKind { CONE_NONNEG = 0, CONE_PSD = 1, CONE_EPINORMSPECTRAL = 2, CONE_WSOS = 3, CONE_LMI = 4, CONE_DNN = 5, CONE_HYPOROOTDET = 6, CONE_HYPOPERLOGDET = 7, CONE_WSOSPSD = 8, CONE_PSD_COMPLEX = 9, CONE_EPINORMSPECTRAL_COMPLEX = 10, CONE_HYPOROOTDET_COMPLEX = 11, CONE_HYPOPERLOGDET_COMPLEX = 12, CONE_WSOS_COMPLEX = 13 };

struct Cone {
  Ctx& ctx;
  int kind;
  int dim = 0;
  double nu = 0;
  bool use_dual_barrier = false;
  DBuf point, dual_point, grad, dder3v, vec1, vec2, prox_in, prox_out;
  bool feas_updated = false, grad_updated = false, hess_updated = false, inv_hess_updated = false,
       hess_fact_updated = false, is_feas_ = false;
  // is_dual_feas() answered ahead of time by prefetch_feas() / the batched line-search sweep (PsdCone); cleared by every
  // load_dual_point, so that a caller following the reference's protocol (load_dual_point, then is_dual_feas, with no
  // reset_data in between: Solvers.jl initialize_cone_point) never sees the previous candidate's answer
  bool dual_cached = false, dual_feas_ = false;

  Cone(Ctx& c, int k) : ctx(c), kind(k) {}
  virtual ~Cone() {}
  void alloc_common();

  // Cones.jl:157-171, 185-186
  void load_point(const double* d_pt, double scal);   // point = scal * pt
  void load_dual_point(const double* d_pt);
  virtual void reset_data() { ++ctx.cone_epoch; feas_updated = grad_updated = hess_updated = inv_hess_updated = hess_fact_updated = false; }

  // Cones.jl:56, 63, 71
  bool is_feas() { return feas_updated ? is_feas_ : update_feas(); }
  virtual bool is_dual_feas() { return true; }
  // optional: start everything is_feas() and is_dual_feas() need at once (independent chains on the two streams);
  // the line search calls it right after loading a candidate
  virtual void prefetch_feas() {}
  // the same in two halves, so that a sweep over many cones queues ALL their chains before the one synchronisation:
  // prefetch_launch(slot) queues the work and the read-back of its flags into pinned words 64 + 2 slot (+1);
  // prefetch_finish(slot) (after ctx.sync() and a wait for the helper stream) takes them over.  Default: nothing queued.
  virtual bool prefetch_launch(int slot) { (void)slot; return false; }
  virtual void prefetch_finish(int slot) { (void)slot; }
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
  // true: hess_prod / inv_hess_prod on k columns give every column exactly the sums it gets when it is the only column (the
  // PSD two-sided kernels, the elementwise cones) -- what lets the constant column of update_lhs ride along with the first pair of
  // directions without changing a bit (SysSolver::step_directions).  The generic explicit-Hessian cones multiply several columns
  // with the GEMM and one with the gemv: false.
  virtual bool products_columnwise() const { return false; }
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
  // The three scalar products those two tests consist of -- <g, point>, <H^-1 g, g>, <H^-1 v, v> with v = irtmu dual + g --
  // queued into d_out3 (device) without a synchronisation, so that a sweep over many cones reads them all back at once;
  // false: no usable inverse Hessian (the caller counts the cone as failed).  Cones with their own get_proxsqr opt out.
  virtual bool prox_batchable() { return true; }
  // A cheap LOWER bound of the proximity value <v, H^-1 v>, v = irtmu dual + g, that needs no factorization at this point:
  // by Cauchy-Schwarz in the H inner product <v, H^-1 v> >= <v, w>^2 / <w, H w> for every w.  A candidate whose bound already
  // exceeds the neighbourhood is rejected exactly as the reference rejects it (its value is at least the bound); anything else
  // goes on to the real test.  false: no bound available (default).
  // `limit`: the value beyond which the candidate is rejected -- an implementation may stop refining its bound once it is passed.
  virtual bool prox_lower_bound(double irtmu, double limit, double* lb) { (void)irtmu; (void)limit; (void)lb; return false; }
  // The same idea one step earlier, for cones whose feasibility tests themselves are expensive: called right after a candidate
  // is loaded, before prefetch_feas; true = the candidate is certainly rejected (primal infeasible, or the proximity bound).
  virtual bool early_reject(double irtmu, double bound2) { (void)irtmu; (void)bound2; return false; }
  // Several candidates of the line search at once (a model of ONE large cone whose rejected candidates are chains of short
  // launches: WsosCone, wsos_screen.hip).  screen_max(): candidates per batch the cone can take (0: no screen);
  // screen_ready(): the state the screen needs is there (a Cholesky factor of an earlier Hessian);
  // screen_batch(): h_pts (C x dim, HOST, already scaled as load_point scales them), h_duals (C x dim, host), irtmu[C];
  // reject[c] = 1 iff candidate c is certainly rejected by check_cone_points' cone tests (infeasible, or a rigorous lower bound of
  // its proximity value exceeds `limit`); n_infeas counts the first kind; bounds[c] = the lower bound where one was formed, else
  // a negative number.  false: nothing was evaluated.
  virtual int screen_max() const { return 0; }
  virtual bool screen_ready() { return false; }
  virtual bool screen_batch(int C, const double* h_pts, const double* h_duals, const double* irtmu, double limit, char* reject, int* n_infeas, double* bounds) {
    (void)C; (void)h_pts; (void)h_duals; (void)irtmu; (void)limit; (void)reject; (void)n_infeas; (void)bounds;
    return false;
  }
  bool prox_launch(double irtmu, double* d_out3);
  // false when inv_hess_prod has no usable factorization at this point (generic cones whose explicit
  // Hessian fails both its Cholesky and its Bunch-Kaufman factorization, Cones.jl:239-251: the
  // reference's ldiv! throws a SingularException there; here the trial point is rejected)
  virtual bool inv_hess_ready() { return true; }

  double dot_host(int n, const double* dx, const double* dy);   // synchronous <x,y>
};

struct NonnegCone : Cone {   // src/Cones/nonnegative.jl
  bool products_columnwise() const override { return true; }
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
  bool prox_batchable() override { return false; }
};

struct PsdCone : Cone {   // src/Cones/possemideftri.jl (real symmetric)
  bool products_columnwise() const override { return true; }
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
  DBuf ws3;      // dder3_cols: the squared middle factor of every column
  bool inv_ready = false;   // Uinv / UinvT / Xinv computed for the current point
  // A run of equal cones of one model keeps X, U, U', U^-1, U^-T, X^-1 and the inverted diagonal blocks of all its members
  // in one arena (member g at offset g * side^2), so that the group's inverses are ONE batched launch sequence
  // (SysSolver::group_inverses); the cones hold the arena alive.
  std::shared_ptr<DBuf> group_arena;
  PsdCone(Ctx& c, int dim);
  void reset_data() override {
    Cone::reset_data();
    inv_ready = false;
    dual_cached = false;
  }
  void prefetch_feas() override;
  bool prox_lower_bound(double irtmu, double limit, double* lb) override;
  bool prefetch_launch(int slot) override;
  void prefetch_finish(int slot) override;
  bool update_feas() override;
  bool is_dual_feas() override;
  void update_grad() override;
  void set_initial_point(double* h_out) override;
  void ensure_inverses();
  bool use_fused(int ncols) const;   // psd_twosided.hip path for this side / column count
  // W_j = R' V_j R for every column j; kr_b / kr_a describe R's triangularity (see cones.hip)
  void two_sided(const double* R, int kr_step2, int kr_step3, double* prod, long ldp, const double* arr, long lda, int ncols);
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  bool use_sqrt_hess_oracles(int) override { return true; }
  void sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  void inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  const double* dder3(const double* d_dir) override;
  // dder3 of nc directions at once (columns ldd apart -> out, columns ldo apart): the launches of one column serve all of them
  // (stacked / batched products: every column gets the sums it gets alone)
  void dder3_cols(const double* d_dirs, long ldd, int nc, double* d_out, long ldo);
};

// PosSemidefTri{T, Complex{T}} (src/Cones/possemideftri.jl:9-207 with R = Complex{T}; complex vectorisation of
// src/Cones/arrayutilities.jl:188-210, 240-262): Hermitian positive definite matrices of side s, dim = s^2.
// Device form: the INTERLEAVED real embedding phi(a + i b) = [[a, -b], [b, a]] (rows 2 i, 2 i + 1; columns 2 j, 2 j + 1) maps
// Hermitian matrices of side s onto real symmetric matrices of side 2 s, multiplicatively (phi(X Y) = phi(X) phi(Y),
// phi(X^H) = phi(X)') -- and, because the 2 x 2 diagonal blocks of phi(U) are diagonal, the complex Cholesky factor onto the
// REAL Cholesky factor of phi(X).  Every oracle of the complex cone is therefore the real PosSemidefTri oracle of side 2 s on
// embedded vectors (same square root included), and the complex svec <-> embedded real svec maps are signed copies without
// rescaling (both carry sqrt(2) on off-diagonals).  Twice the arithmetic of native complex kernels, all of it on the tuned
// real path (MFMA two-sided products, blocked Cholesky).
struct CplxPsdCone : Cone {
  int side;            // complex side s
  long edim;           // s (2 s + 1): svec length of the embedded matrix
  PsdCone inner;       // real PosSemidefTri of side 2 s
  DBuf ea, eb;         // embedded column workspaces
  CplxPsdCone(Ctx& c, int dim);
  void reset_data() override {
    Cone::reset_data();
    inner.reset_data();
  }
  bool update_feas() override;
  bool is_dual_feas() override;
  void update_grad() override;
  void set_initial_point(double* h_out) override;
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  bool use_sqrt_hess_oracles(int) override { return true; }   // possemideftri.jl:54
  void sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  void inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  const double* dder3(const double* d_dir) override;
  void embed(const double* cvec, long ldc, double* evec, int ncols, long lde = 0);     // E: complex svec columns -> embedded real svec columns (stride lde, default edim)
  void extract(const double* evec, double* cvec, long ldc, int ncols, long lde = 0);   // the projection back (mean of the two copies)
  template <class F> void through(F f, double* prod, long ldp, const double* arr, long lda, int ncols);
};

// Cones whose Hessian is formed explicitly and factored (the generic fallbacks of Cones.jl:101-118,
// 189-259): WSOSInterpNonnegative and EpiNormSpectral in this Hypatia version.
struct GenericHessCone : Cone {
  DBuf H;          // dim x dim explicit Hessian, BOTH triangles (symmetrised after update_hess)
  DBuf Hfact;      // Cholesky factor of H (upper), strict lower zeroed; or the unit factor of the Bunch-Kaufman fallback
  DBuf Hdinv, Hinfo, trsm_work, tmpd, tmpd2;
  bool hess_fact_ok = false;   // issuccess(hess_fact)
  bool hess_fact_bk = false;   // hess_fact is the Bunch-Kaufman fallback of a failed Cholesky (posdef_fact_copy!(.., false), Cones.jl:247)
  BKFact Hbk;
  TriSolvePlan Hplan;          // super-block plan for one-vector solves with a large factor (built on first use per factorization)
  bool use_hess_prod_slow = false, use_hess_prod_slow_updated = false;
  GenericHessCone(Ctx& c, int kind) : Cone(c, kind) { Hplan.owner_class = 1; }
  void alloc_generic();
  void ensure_hess_storage(bool with_fact);   // explicit Hessian (and its factor) on first use
  void reset_data() override {
    Cone::reset_data();
    use_hess_prod_slow = use_hess_prod_slow_updated = false;
  }
  virtual void update_hess() = 0;                 // fills H (both triangles), sets hess_updated
  bool update_hess_fact();                        // Cones.jl:239-251
  bool inv_hess_ready() override { return update_hess_fact(); }
  void update_use_hess_prod_slow();               // Cones.jl:222-231
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;        // :101-105
  void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;    // :113-118
  bool use_sqrt_hess_oracles(int arr_dim) override;                                                // :189-195
  void sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;   // :198-206
  void inv_sqrt_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;   // :209-218
  void hess_explicit(double* d_out, long ld) override;
};

struct WsosCone : GenericHessCone {   // src/Cones/wsosinterpnonnegative.jl (real)
  int U, K;
  std::vector<int> Ls;
  std::vector<DBuf> P, PT, SP, Lam, LamDinv, LFLP, LFLPT, LU, LL;   // per k
  DBuf tmpUU, infos, gparts, trsm_work2, LamArena;
  bool lam_dinv_ready = false;   // LamDinv holds the inverted diagonal blocks of the CURRENT factors Lam (formed with the gradient)
  WsosCone(Ctx& c, int U, int K, const int* Ls, const double* const* hPs, bool use_dual);
  bool update_feas() override;
  void update_grad() override;
  void update_hess() override;
  void set_initial_point(double* h_out) override;
  void hess_prod_slow(double* prod, long ldp, const double* arr, long lda, int ncols) override;   // :152-175
  const double* dder3(const double* d_dir) override;                                               // :177-188
  void partial_lambda(int k, const double* d_dir);                                                 // :190-200 -> LU[k]
  void lambda_of(int k, const double* d_dir);                                                      // its first half -> LL[k] (symmetric)
  bool prox_lower_bound(double irtmu, double limit, double* lb) override;
  void gram_norms(const double* d_dir, double* d_out);     // d_out[k] = || LFLP_k diag(dir) LFLP_k' ||_F^2 (LL[k] left in place), both streams
  void hess_vec_from_LL(double* d_out);                    // H dir from the LL[k] of the last gram_norms(dir): sum_k diag(LFLP_k' LL_k LFLP_k)
  DBuf lbP, lbHP, lbR;
  // candidate screen (wsos_screen.hip)
  DBuf scrSP, scrLF, scrLFT, scrLam, scrLL, scrDinv, scrVec, scrInfo;
  int screen_max() const override;
  bool screen_ready() override;
  int screen_splitk(int L, int batch) const;   // K slices of a batch of L x L x U Gram products
  bool screen_batch(int C, const double* h_pts, const double* h_duals, const double* irtmu, double limit, char* reject, int* n_infeas, double* bounds) override;
};

struct LmiCone : GenericHessCone {   // src/Cones/linmatrixineq.jl (real dense symmetric members; complex Hermitian members embedded)
  int side;
  // Complex Hermitian members (side s) are held as their real embeddings phi(a + ib) = [[a, -b], [b, a]] of side 2 s: phi is a
  // ring homomorphism with det phi(M) = |det M|^2, so -logdet(sum w_i A_i) = bscale * (-logdet(sum w_i phi(A_i))) with
  // bscale = 1/2 -- the same function of w, hence every oracle of the complex cone is bscale times the real one's on the
  // embedded members (gradient, Hessian, third-order term), and nu = s.
  double bscale = 1.0;
  DBuf Amat;      // side^2 x dim: column i = A_i (col-major side x side)
  DBuf sumA, fact, fdinv, Rinv, T, Mmat, dirmat, Zm, infos, Jm;   // Mmat: side^2 x dim, column i = L^-1 A_i L^-T
  LmiCone(Ctx& c, int dim, int side, const double* hAs, bool use_dual, bool complex_members = false);
  bool update_feas() override;                                                                      // :87-96
  void update_grad() override;                                                                      // :98-109
  void update_hess() override;                                                                      // :111-123
  void set_initial_point(double* h_out) override;                                                   // :74-81
  void hess_prod_slow(double* prod, long ldp, const double* arr, long lda, int ncols) override;    // :125-144
  const double* dder3(const double* d_dir) override;                                                // :146-159
};

struct DnnCone : GenericHessCone {   // src/Cones/doublynonnegativetri.jl: PSD and entrywise nonnegative, svec format
  PsdCone psd;     // the -logdet part: same point, same kernels
  DBuf cnt;
  DnnCone(Ctx& c, int dim, bool use_dual);
  void reset_data() override {
    GenericHessCone::reset_data();
    psd.reset_data();
  }
  bool update_feas() override;                                                                      // :130-143
  void update_grad() override;                                                                      // :145-156
  void update_hess() override;                                                                      // :158-171
  void set_initial_point(double* h_out) override;                                                   // :71-128
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;         // :173-192
  const double* dder3(const double* d_dir) override;                                                // :194-205
};

struct HypoRootdetTriCone : GenericHessCone {   // src/Cones/hyporootdettri.jl (real): (u, w), u <= det(smat(w))^(1/d)
  int d;
  PsdCone psd;      // the W part of the point: Cholesky, W^-1, the two-sided products
  PsdCone psdd;     // the W part of the DUAL point (is_dual_feas needs its Cholesky and log-determinant)
  double di = 0, u = 0, phi = 0, zeta = 0, phizidi = 0;
  DBuf Wi_vec, dots, tmpw, ld;
  HypoRootdetTriCone(Ctx& c, int dim, bool use_dual);
  void reset_data() override {
    GenericHessCone::reset_data();
    psd.reset_data();
  }
  bool update_feas() override;                                                                      // :101-115
  bool is_dual_feas() override;                                                                     // :117-127
  void update_grad() override;                                                                      // :129-141
  void update_hess() override;                                                                      // :143-170
  void set_initial_point(double* h_out) override;                                                   // :82-99
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;         // :172-203
  void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;     // :235-272
  bool inv_hess_ready() override { return true; }                                                   // closed form
  const double* dder3(const double* d_dir) override;                                                // :274-324
  double logdet_of(PsdCone& k);
};

struct HypoPerLogdetTriCone : GenericHessCone {   // src/Cones/hypoperlogdettri.jl (real): (u, v, w), u <= v logdet(smat(w) / v)
  int d;
  PsdCone psd;      // the W part of the point
  PsdCone psdd;     // the W part of the dual point
  double u = 0, v = 0, phi = 0, zeta = 0;
  DBuf Wi_vec, dots, tmpw, ld;
  HypoPerLogdetTriCone(Ctx& c, int dim, bool use_dual);
  void reset_data() override {
    GenericHessCone::reset_data();
    psd.reset_data();
  }
  bool update_feas() override;                                                                      // :97-118
  bool is_dual_feas() override;                                                                     // :120-131
  void update_grad() override;                                                                      // :133-150
  void update_hess() override;                                                                      // :152-193
  void set_initial_point(double* h_out) override;                                                   // :80-95
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;         // :195-236
  void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;     // :271-316
  bool inv_hess_ready() override { return true; }
  const double* dder3(const double* d_dir) override;                                                // :318-368
  double logdet_of(PsdCone& k);
};

struct WsosPsdCone : GenericHessCone {   // src/Cones/wsosinterppossemideftri.jl: R x R matrices of interpolant-basis polynomials
  int R, U, K, nblk;            // nblk = R (R + 1) / 2 svec blocks of length U
  std::vector<int> Ls;
  std::vector<DBuf> P, SP, Lam, LamDinv, FLP, Mk, Tk, LRUR;   // per basis k
  DBuf PLiP, tU, infos;
  WsosPsdCone(Ctx& c, int R, int U, int K, const int* Ls, const double* const* hPs, bool use_dual);
  bool update_feas() override;                                                                      // :110-140
  void update_grad() override;                                                                      // :142-186
  void update_hess() override;                                                                      // :188-236
  void set_initial_point(double* h_out) override;                                                   // :100-108
  void hess_prod_slow(double* prod, long ldp, const double* arr, long lda, int ncols) override;    // :238-247
  const double* dder3(const double* d_dir) override;                                                // :249-252
  void block_matrix(int k, const double* d_vec, double* M);                                         // :122-130, 300-307 (upper blocks)
  void partial_prod(double* prod, long ldp, const double* arr, long lda, int ncols, bool use_symm_prod);   // :288-321
};

struct EpiNormSpectralCone : GenericHessCone {   // src/Cones/epinormspectral.jl (real)
  int d1, d2;
  bool hess_aux_updated = false;
  double u = 0, Huu = 0, trZi2 = 0, g0_host = 0;
  DBuf W, WT, Z, Zfact, Zdinv, Zi, tau, HuW, WtauI, Zitau, Zinfo;
  DBuf t12a, t12b, t12c, t12d, t11, t22, t22b, wsA, wsB, wsC;
  EpiNormSpectralCone(Ctx& c, int d1, int d2, bool use_dual);
  void reset_data() override {
    GenericHessCone::reset_data();
    hess_aux_updated = false;
    svd_updated = false;
  }
  bool update_feas() override;
  bool is_dual_feas() override;
  void update_grad() override;
  void update_hess_aux();
  void update_hess() override;
  void set_initial_point(double* h_out) override;
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;   // :211-239
  const double* dder3(const double* d_dir) override;                                          // :241-294
  void zsolve(double* X, long ldx, int nrhs);    // X <- Z^-1 X
  // one-workgroup forms of update_feas / update_grad + update_hess_aux / hess_prod / closed_inv_apply where the matrices fit one
  // CU's LDS (d1 <= 64; HYP_ENS_FUSED, cone_epinormspectral.hip)
  bool fused();
  bool fused_checked = false, fused_ok = false;
  DBuf frec;                  // device record of the fused kernels: u, info, tr(Zi), <Zi, Zi>, g0, Huu
  // d_decide_u: device address of the epigraph variable the caller compares the norm with (dual feasibility test): the sweeps of the
  // decomposition stop as soon as rigorous bounds put the norm on one side of it (HYP_ENS_DUAL_DECIDE, cone_epinormspectral.hip)
  double nuclear_norm(const double* d_mat /* d1 x d2 col-major */, const double* d_decide_u = nullptr);
  void nuclear_norm_launch(const double* d_mat, double* d_out, const double* d_decide_u = nullptr);
  void prefetch_feas() override;
  bool early_reject(double irtmu, double bound2) override;
  void closed_inv_apply(double u_used, double* prod, long ldp, const double* arr, long lda, int ncols);
  double u_svd = 0;           // the epigraph variable of the point Usvd / sig / V1 belong to
  // Closed-form inverse Hessian (NOT in this Hypatia version, whose inv_hess_prod! is the generic explicit-Hessian
  // Cholesky of Cones.jl:113-118): with W = U S V1' the Hessian of epinormspectral.jl:211-239 decouples in the rotated
  // coordinates U' A [V1 V2] into 2 x 2 blocks over the index pairs (i, j), (j, i), a diagonal scaling on the V2 part and an
  // arrow system coupling u with the diagonal -- see cone_epinormspectral.hip.  HYP_ENS_CLOSED_INV=0 restores the generic path.
  bool closed_inv = true, svd_updated = false, svd_ok = false;
  bool svd_prev_ok = false;   // Usvd holds the rotations of an earlier decomposition without zero singular values: warm start
  int svd_warm_count = 0;
  DBuf Jdual;                 // the same for the nuclear norm of the dual point (is_dual_feas)
  bool dual_prev_ok = false;
  int dual_warm_count = 0;
  DBuf Usvd, V1, V1T, sig, Bj, Jm, cw1, cw2, cw3, cw4, cw5;
  void reset_svd() { svd_updated = false; }
  bool update_svd();
  void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  bool inv_hess_ready() override;
};

// HypoRootdetTri / HypoPerLogdetTri with a complex Hermitian matrix part (cone_hypo_complex.hip): with phi the real embedding
// of twice the side, logdet(phi(W)) = 2 logdet(W), so the complex barrier is the real cone's on the embedded point (leading
// scalars rescaled) MINUS the complex PosSemidefTri barrier -logdet(W), which the real one counts twice.
void central_ray_hypoperlog(int d, double* uvw);   // cones_generic.hip (hypoperlogdettri.jl:80-95)
struct CplxHypoCone : GenericHessCone {
  int d, nlead;              // complex side; leading scalars: 1 (u) for the root-determinant, 2 (u, v) for the perspective of logdet
  double lead_scale[2];      // primal map of the leading scalars into the real cone
  double dual_scale[2];      // and of the dual point's
  long cdw, edw, edim;       // d^2, d (2 d + 1), nlead + edw
  Cone* inner;               // real cone of side 2 d (owned)
  CplxPsdCone psdc;          // -logdet(W) in complex svec coordinates
  DBuf ea, eb, pw;
  CplxHypoCone(Ctx& c, int kind, int dim, bool perlog, bool use_dual);
  ~CplxHypoCone() override { delete inner; }
  void reset_data() override {
    GenericHessCone::reset_data();
    inner->reset_data();
    psdc.reset_data();
  }
  bool update_feas() override;
  bool is_dual_feas() override;
  void update_grad() override;
  void update_hess() override;
  void set_initial_point(double* h_out) override;
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  const double* dder3(const double* d_dir) override;
  void to_inner(const double* cvec, long ldc, double* evec, int ncols, const double* scale);      // T
  void from_inner(const double* evec, double* cvec, long ldc, int ncols);                          // T'
};

// WSOSInterpNonnegative{Float64, ComplexF64}: the real cone at the duplicated point over the embedded bases, barrier halved
// (cone_wsos_complex.hip).
struct CplxWsosCone : GenericHessCone {
  int U;
  WsosCone* inner;     // bases phi(P_k), 2U x 2L_k (owned)
  DBuf ea, eb;
  CplxWsosCone(Ctx& c, int U, int K, const int* Ls, const double* const* hPs /* complex, (re, im) interleaved */, bool use_dual);
  ~CplxWsosCone() override { delete inner; }
  void reset_data() override {
    GenericHessCone::reset_data();
    inner->reset_data();
  }
  bool update_feas() override;
  void update_grad() override;
  void update_hess() override;
  void set_initial_point(double* h_out) override;
  void hess_prod_slow(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  const double* dder3(const double* d_dir) override;
  void dup(const double* in, long ldi, double* out, int ncols);    // E
  void fold(const double* in, double* out, long ldo, int ncols);   // 1/2 E'
};

// EpiNormSpectral{Float64, ComplexF64}: the real cone of twice the sides on the embedded matrix, barrier halved, plus the
// univariate term the halving leaves over (cone_ens_complex.hip).
struct CplxEnsCone : Cone {
  int d1, d2;          // complex sides
  long edim;           // 1 + 4 d1 d2: dimension of the embedded real cone
  EpiNormSpectralCone inner;
  DBuf ea, eb, xw, ainv;   // embedded workspaces; A^-1 columns in complex coordinates; a = A^-1 e_u
  bool ainv_ready = false;
  CplxEnsCone(Ctx& c, int d1, int d2, bool use_dual);
  void reset_data() override {
    Cone::reset_data();
    inner.reset_data();
    ainv_ready = false;
  }
  bool update_feas() override;
  bool is_dual_feas() override;
  void update_grad() override;
  void set_initial_point(double* h_out) override;
  void hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  void inv_hess_prod(double* prod, long ldp, const double* arr, long lda, int ncols) override;
  bool use_sqrt_hess_oracles(int) override { return false; }   // (no factored Hessian: the assembly takes the hess_prod! branch, qrchol.jl:240-246)
  void sqrt_hess_prod(double*, long, const double*, long, int) override { HYP_REQUIRE(false, "EpiNormSpectral (complex): no square-root oracle"); }
  void inv_sqrt_hess_prod(double*, long, const double*, long, int) override { HYP_REQUIRE(false, "EpiNormSpectral (complex): no square-root oracle"); }
  const double* dder3(const double* d_dir) override;
  bool inv_hess_ready() override { return inner.inv_hess_ready(); }
  void embed(const double* cvec, long ldc, double* evec, int ncols, double uscale);
  void extract(const double* evec, double* cvec, long ldc, int ncols, double uscale);
  void apply_ainv(double* xc, long ldx, const double* arr, long lda, int ncols);
  void dev_axpby_scalar0(double v);
};

}  // namespace hyp
