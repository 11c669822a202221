// Device-resident QRCholDenseSystemSolver (/root/reference/src/Solvers/systemsolvers/qrchol.jl:104-257).
#pragma once
#include <chrono>
#include "cones.hpp"

namespace hyp {

// a <- a - b in place and *d_out = max |a_i| (NaN if any), one launch
void dev_sub_absmax(Ctx& c, int n, double* a, const double* b, double* d_out);
// nr columns ld apart in one launch (d_out[r]); work: nr x 33 doubles, zero when first used
void dev_sub_absmax_cols(Ctx& c, int n, int nr, double* a, const double* b, long ld, double* d_out, double* work);

struct SysSolver {
  Ctx& ctx;
  int n, p, q, nmp;
  std::vector<Cone*> cones;
  std::vector<int> offs;        // cone k occupies rows [offs[k], offs[k+1]) of z / s
  // matrices (col-major, device)
  DBuf G;       // q x n   model.G (after preprocessing); aliases GQ2 storage when p == 0
  DBuf GQ1;     // q x p   (p > 0 only)
  DBuf GQ2s;    // q x nmp (own storage only when p > 0)
  DBuf Qm;      // n x n   Ap_Q (p > 0 only)
  DBuf Rinv;    // p x p   inverse of Ap_R (upper), computed once at load
  DBuf HGQ2;    // q x nmp
  DBuf lhs;     // nmp x nmp (upper)
  DBuf lhs_tri; // its packed upper triangle: the payload of the Schur all-reduce
  DBuf lhs_fact, dinv, d_info;
  TriSolvePlan tri;   // super-block inverses of lhs_fact for the one-RHS solves
  // vectors
  DBuf QpbxGHbz, Gx, HGx, GQ1x, HGQ1x, tmpn, sol, rhs, tmpq;
  std::vector<int> use_sqrt;
  bool fact_ok = false;
  // runs of >= 4 consecutive PosSemidefTri cones of equal side (config 4: 64 x side 80): group-owned storage, batched inverses
  struct PsdRun { int k0, count, side; double *X, *U, *UT, *Uinv, *UinvT, *Xinv, *dinvb, *point, *dual; };
  std::vector<PsdRun> psd_runs;
  void make_psd_runs();
  void group_inverses();   // ensure_inverses() of every run whose members are all feasible and not yet inverted, batched
  // hess_prod! of ALL members of the run that starts at cone k on ncols <= 3 columns (prod / arr point at cone k's rows; the
  // members' rows follow): per column one unpack, four batched GEMMs, one pack.  Returns the number of cones done (0: cone k
  // does not start a usable run -- the caller takes the per-cone path).
  int run_hess_prod(size_t k, double* prod, long ldp, const double* arr, long lda, int ncols);
  DBuf run_ws1, run_ws2, run_ws3, run_g, run_v, run_h, run_info;
  const PsdRun* whole_model_run();                                   // the run, if one run of primal-barrier cones is the whole model and still owns the members' storage
  void run_grad(const PsdRun& r, double* d_out);                     // -svec(X^-1) of every member (possemideftri.jl:97-107)
  void run_dder3(const PsdRun& r, const double* d_dir, double* d_out);   // svec(X^-1 D X^-1 D X^-1) of every member (:197-207)
  // the three scalar products of check_numerics / get_proxsqr (Cones.jl:273-310) of every member of run r, 3 per member
  void run_prox_launch(const PsdRun& r, double irtmu, double* d_out);
  DBuf prox_scal;         // 3 scalars per cone of the batched proximity test of check_cone_points
  BKFact bk;              // the factorization after a failed Cholesky (posdef_fact_copy!, dense.jl:194-215)
  bool use_bk = false;    // lhs_fact holds U of P lhs P' = U' D U instead of the Cholesky factor
  DBuf bk_work;

  SysSolver(Ctx& c, int n_, int p_, int q_, const std::vector<Cone*>& cs);
  ~SysSolver();
  SysSolver(const SysSolver&) = delete;
  SysSolver& operator=(const SysSolver&) = delete;
  const double* GQ2() const { return p == 0 ? G.d() : GQ2s.d(); }
  // host pointers; GQ1/GQ2/Q/R may be null when p == 0 (then GQ2 = G, Q = I)
  void load(const double* hG, const double* hGQ1, const double* hGQ2, const double* hQ, const double* hR);
  void block_hess_prod_vec(double* d_out, const double* d_in);                 // qrchol.jl:87-98 on a q-vector
  void update_lhs_fact(int* info, int* used_fallback);                         // qrchol.jl:201-257
  void assemble_lhs();                                                         //   :214-246 (Schur sum over this process's cones)
  void factor_lhs(int* info, int* used_fallback);                              //   :249-250
  void prelaunch_sqrt_hess();                                                  //   HYP_SHP_PRELAUNCH: the next update_lhs's cone products queued when the search accepts
  bool shp_prelaunched = false;                                                //   ... HGQ2 holds them, for the cone states of epoch shp_epoch
  unsigned long shp_epoch = 0;
  int shp_wasted = 0;                                                          //   ... consecutive prelaunches nobody used (2: off for this model)
  long shp_used = 0, shp_unused = 0;
  int chol_split_point(int n, int K);                                          //   HYP_CHOL_SPLIT: columns of the leading block factored under the product (0: none)
  int chol_split_n1 = 0;                                                       //   ... pending between assemble_lhs and factor_lhs_begin
  long chol_split_count = 0;
  bool split_unjoined = false;                                                 //   ... the leading factorization may still run on its lane
  hipEvent_t split_ev_ready = nullptr, split_ev_done = nullptr;
  hipEvent_t ov_ev_fork = nullptr;                                             //   overlapped exchange: the row groups' lanes start behind this
  void factor_lhs_begin();                                                     //   ... queued: Cholesky attempt, info read-back, solve plan
  void factor_lhs_end(int* info, int* used_fallback, bool times_later = false);   //   ... after a synchronisation: info, fall-back chain
  hipEvent_t plan_ev_fork = nullptr, plan_ev_done = nullptr;
  bool plan_join_pending = false;   // the solve plan is being built on a lane of its own (HYP_PLAN_LANE): its first user waits
  void join_plan();
  void factor_lhs_times();                                                     //   ... the phases' HIP-event times into ctx.kstat (times_later: the caller's job, off the critical path)
  void tri_solves(double* d_x);                                                // both triangular solves of the potrs
  void potrs(double* d_x);                                                     // x <- lhs^-1 x with the current factor (:66-69)
  void solve3(double* d_sol, const double* d_rhs);                             // qrchol.jl:39-85

  // ---- multi-GPU (cone sharding): this solver then holds only ITS rank's cones and rows of G (z / s vectors are the
  // local rows, x-space vectors are replicated).  At the exchange points -- the Schur sum, G' z, the scalar products
  // over z, the residual norm, the line-search flags -- the payload is copied into a device staging buffer owned by
  // the host framework (a torch tensor) and `comm_fn` all-reduces it in place over RCCL.  Unset: single GPU.
  typedef int (*CommFn)(void* user, long count, int op);   // op 0 sum, 1 max, 2 min; returns 0 on success
  CommFn comm_fn = nullptr;
  void* comm_user = nullptr;
  double* comm_stage = nullptr;
  long comm_cap = 0;
  // The same exchanges through RCCL inside the library (hyp_sys_set_comm_rccl): ncclAllReduce in place on the library's own
  // stream -- no staging copy, no host callback, and no host synchronisation for device payloads (stream order).
  void* rccl_comm = nullptr;     // ncclComm_t
  long comm_calls = 0;           // exchanges issued since creation (hyp_sys_comm_stats)
  double comm_doubles = 0;
  // K-panel sharding of ONE model over several GPUs (third bullet: configs 2 / 3, a single cone): model, cones,
  // point and every solve are REPLICATED on all ranks; only the Schur product lhs = HGQ2' HGQ2 is cut along its K dimension
  // (rank r sums rows [q r / N, q (r + 1) / N) of HGQ2) and the partial n x n matrices are all-reduced.  Everything a rank
  // computes afterwards is bitwise the same on every rank (the all-reduce hands every rank the same sum).
  int ks_rank = 0, ks_world = 1;
  bool dist() const { return (comm_fn != nullptr || rccl_comm != nullptr) && ks_world <= 1; }
  void allreduce_lhs();   // the n x n exchange of either sharding mode
  // site: where in the iteration the exchange is issued (comm_hist, hyp_sys_comm_hist): 0 Schur sum, 1 solve G'z, 2 solve h'z,
  // 3 residual (G'z alone, or fused with its scalars), 4 residual h'z, 5 residual norm, 6 constant column h'z, 7 / 8 candidate
  // screen, 9 / 10 line-search trial (sums / closing), 11 residual_products, 12 host-requested, 13 screen agreement, 15 other
  void allreduce_dev(double* d_buf, long count, int op, int site = 15);
  DBuf ar_dev;   // device staging of allreduce_host for payloads beyond the context's 32 scalar slots
  void allreduce_host(double* h_buf, int count, int op, int site = 15);
  long comm_hist[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // Round 5: device time of the exchanges by site (hyp_sys_comm_times), so that an N-GPU run says where a shortfall comes from.
  // RCCL in the library: a HIP-event pair around every collective on the library's stream (resolved lazily: comm_times_flush);
  // callback transport: host clock around the callback.  Slot 14: the Schur exchange INCLUDING its triangle pack / unpack kernels.
  double comm_ms[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  struct CommEv { hipEvent_t a, b; int site; };
  std::vector<CommEv> comm_ev_pending;
  std::vector<hipEvent_t> comm_ev_free;
  hipEvent_t comm_event();
  void comm_time_begin(int site, hipEvent_t* a, hipStream_t st = nullptr);   // (st: the stream the exchange is queued on; default the library's main stream)
  void comm_time_end(int site, hipEvent_t a, hipStream_t st = nullptr);
  // Round 5 (HYP_DIST_OVERLAP = number of row groups, 0 = off): the Schur product emitted in row groups of equal area, each group's
  // part of the upper triangle all-reduced on the helper stream while the next group is being multiplied (RCCL in the library only,
  // every cone through its square-root product).  Returns false if the conditions do not hold (the caller takes the one-launch path).
  bool assemble_lhs_overlapped(long kr0, long kr1, int groups);
  DBuf ov_tri;
  std::vector<hipEvent_t> ov_events;
  void comm_times_flush();
  // Round 4: ONE exchange for a device payload and the scalars that used to follow it in collectives of their own.  The buffer
  // is [payload (npay) | scalars to be summed (nsum) | one slot per rank and scalar to be max-ed (world x nmax)]; a rank fills only
  // its own slots, so a SUM all-reduce returns every rank's value and the maximum (NaN wins) is taken locally -- sums and maxima
  // travel in the same ncclAllReduce.  Needs the communicator's layout (hyp_sys_set_comm_layout, or the RCCL communicator's).
  int comm_rank_ = 0, comm_world_ = 0;
  bool fused_ok() const { static const bool on = [] { const char* e = getenv("HYP_DIST_FUSED"); return !(e && e[0] == '0'); }(); return on && comm_world_ > 0; }
  struct FusedTail { int nsum = 0, nmax = 0; const double* sum_src[8]; const double* max_src[8]; };
  // d_buf must have room for npay + nsum + world * nmax doubles; h_out receives nsum sums, then nmax maxima
  void allreduce_fused(double* d_buf, long npay, const FusedTail& t, double* h_out, int site);
  // the same exchange left on the device: the summed tail stays behind the payload ([sums | world x nmax slots]) for a kernel of the caller's
  void allreduce_fused_dev(double* d_buf, long npay, const FusedTail& t, int site);
  double screen_sz_[18];      // sharded: the screen's all-reduced <z, s> and failure flag of each candidate ...
  double screen_szfail_[18];
  int screen_pass_g_ = -1;                 // ... and which of them check_cone_points is being asked about (-1: none)
  // the products of calc_convergence_params / calc_mu (Solvers.jl:418-483) on THIS process's rows of z, s and G, the sums over
  // ranks taken here: Gtz (n, summed) = G' z; Gx_s (q, these rows) = G x + s; dots = {h' z, z' s} (summed)
  DBuf rp_x, rp_z, rp_s, rp_t, rp_g;
  DBuf rp_loc;   // residual_products2: the two local maxima in front of their exchange
  // round 6: the same products of the NEXT iterate, queued when the line search accepts a candidate (see syssolver.hip)
  void prefetch_residual_products(int mode, double alpha, const double* d_cand, const double* h_cand);
  bool rp_pre_valid = false;
  double* rp_pre_host = nullptr;   // pinned: [G' z (n); h' z; z' s | G x + s (q) | x (n)]
  size_t rp_pre_host_n = 0;
  hipEvent_t rp_pre_ev = nullptr;
  std::vector<double> rp_pre_cand;
  long rp_pre_hits = 0, rp_pre_misses = 0;
  void residual_products2(const double* h_x, const double* h_z, const double* h_s, double tau, double* h_Gtz, double* h_Gx_s, double* h_dots,
                          double* h_norms);
  void residual_products(const double* h_x, const double* h_z, const double* h_s, double* h_Gtz, double* h_Gx_s, double* h_dots);

  // ---- device-resident direction solves (systemsolvers/common.jl:15-182): the 6x6 system of one
  // stepper direction, reduced 6 -> 4 -> 3 on the device, with the reference's iterative refinement.
  // Vectors use the Point layout [x(n); y(p); z(q); tau; s(q); kap]; tau / kap travel as host scalars.
  DBuf mc, mb, mh, mA;                  // model.c, b, h, A (p x n) after preprocessing
  DBuf v_rhs, v_dir, v_res, v_tmp;      // Point-layout work vectors
  DBuf sub_rhs, sub_sol, sol_const;     // 3x3 subsystem vectors [x; y; z]
  bool model_loaded = false;
  double dot_const = 0.0;               // dot_obj(model, sol_const)
  DBuf Gx_dir;                          // G * (x of the direction solve_system just produced), reused by its residual
  bool Gx_dir_valid = false;
  int dimv() const { return n + p + q + 1 + q + 1; }
  void load_model(const double* hc, const double* hb, const double* hh, const double* hA);
  void sgemv(bool trans, int m, int n_, double alpha, const double* A, long lda, const double* x, double beta, double* y);
  void update_const();                                                          // qrchol.jl:191-197
  void update_const_pre();                                                      // the same in two halves around the triangular solves
  void update_const_post();
  struct Scal { double tau, kap; };
  Scal solve_system(double* d_sol, const double* d_rhs, Scal rhs, double mu, double taubar);   // common.jl:129-182
  Scal apply_lhs(double* d_res, const double* d_dir, Scal dir, double mu, double taubar);      // common.jl:79-121
  // line-search acceptance test of one candidate (search.jl:74-138) for all cones in one call: cand = host
  // [z(q); tau; s(q); kap].  Cones are visited in order and the first failure stops the sweep, exactly as the
  // reference's loop; n_loaded = number of cones whose points were (re)loaded.
  DBuf cand_d;
  bool check_cone_points(const double* h_ztsk, double min_prox, double prox_bound, bool use_max_prox, double nup1, double* prox_out,
                         int* n_loaded, double* irtmu_out);
  // Side-by-side screening of the remaining candidates of the schedule walk (a model of ONE primal-barrier PosSemidefTri cone,
  // single process): the two feasibility factorizations and the inverse-free proximity value of up to SCREEN_MAX candidates
  // in one batched launch sequence and one read-back.  rej[g] = 1 only where check_cone_points would certainly reject
  // candidate g (scalar tests, a failed factorization, or the proximity bound missed by more than the rounding of the two
  // routes); every other candidate still goes through check_cone_points, which alone accepts.
  // The candidates are formed ON THE DEVICE when the point and the four directions are still there from step_directions
  // (search_alpha(..., resident = true)); otherwise on the host and uploaded.
  static constexpr int SCREEN_MAX = 18;   // the reference's whole schedule (search.jl:41-43)
  DBuf screen_buf, screen_info;
  long screen_count = 0, screen_rejected = 0;   // statistics: screens run, candidates they rejected
  int gprev_acc_ = -1;   // schedule index the previous search_alpha accepted (-1: none): where the one-cone screen's batches stop
  long screen_checked = 0, screen_mismatch = 0; // HYP_WSOS_SCREEN_CHECK=1: rejected verdicts compared with the sequential test / disagreements
  bool cand_scalars(const double* h, double min_prox, double prox_bound, double nup1, double* irtmu) const;
  bool screen_usable() const { return screen_mode() != 0; }
  int screen_agreed = -1;    // sharded: the minimum of the ranks' screen_mode(), agreed once per model (-1: not yet)
  int screen_kmax() const;   // candidates per screening batch the buffers admit for this model (<= SCREEN_MAX; < 2: no screen)
  int screen_mode() const;   // 0: no screen; 1: one PosSemidefTri cone (single process); 2: equal PosSemidefTri cones are the whole (local) model
  void screen_candidates_run(const double* d_cands, int K, const double* tau, const double* kap, double min_prox, double prox_bound,
                             double nup1, bool use_max_prox, char* rej);
  bool screen_survivor = false;   // set around the check_cone_points call of a candidate the screen has passed: its proximity lower bound is not evaluated again
  // d_cands: K candidates [z; tau; s; kap] of length 2 q + 2 on the device (their tau / kap slots are not read: tau, kap)
  void screen_candidates(const double* d_cands, int K, const double* tau, const double* kap, double min_prox, double prox_bound, double nup1,
                         char* rej);
  // what step_directions left on the device: the point, the four directions (s_dirs: cent, pred, centadj, predadj) and the
  // tau / kap entries of the five vectors (which travel on the host)
  bool s_resident = false;
  int s_resident_q = -1;   // the q the resident vectors were formed with
  double s_tk[5][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
  double residual(double* res, const double* dir, const double* rhs, Scal rs, Scal dcur, Scal& rsc, double mu, double taubar);
  double residual_fused(double* res, const double* dir, const double* rhs, Scal rs, Scal dcur, Scal& rsc, double mu, double taubar);
  DBuf rf_buf;   // residual_fused: [G' z (n) | the exchange's tail]
  double refine(double* rhs, double* dir, double* res, double* tmp, Scal rs, Scal& dsc, Scal rsc, double res_norm, double mu, double taubar,
                int max_ref_steps, double res_norm_cutoff, double min_impr_tol, int* n_solves);
  // ---- two right-hand sides at once (directions_multi.hip): the stepper's (cent, pred) and (centadj, predadj)
  // pairs are independent, and every pass over G / the factor / the cone matrices serves both columns
  DBuf m_rhs, m_dir, m_res, m_subr, m_subs, m_t, m_Gx, m_HGx, m_Gxd;
  void solve3_multi(double* sol, const double* rhs, int nr, double* x_third = nullptr);   // p == 0 only; columns n + q apart; x_third: a third lhs-space right-hand side through the same triangular sweeps
  void get_directions2(double* h_dirs, const double* h_rhss, double mu, double taubar, int max_ref_steps, double res_norm_cutoff,
                       double min_impr_tol, double* res_norms, int* n_solves);
  // search_alpha (search.jl:46-69) with the candidate of update_stepper_points (combined.jl:124-170) formed here:
  // walks alpha_sched from index `start`, returns the index of the first accepted step (or -1); all vectors are
  // host `ztsk` views (length 2 q + 2): the current point and the four stepper directions
  int search_alpha(const double* pt, const double* d_cent, const double* d_pred, const double* d_centadj, const double* d_predadj,
                   bool unadj_only, bool cent_only, const double* sched, int nsched, int start, double min_prox, double prox_bound,
                   bool use_max_prox, double nup1, double* cand_out, double* prox_out, int* n_trials, int* n_loaded, double* irtmu_out,
                   bool resident = false);   // resident: the five vectors are those step_directions left on the device (host pointers unused)
  // ---- the direction phase of CombinedStepper.step (steppers/combined.jl:60-95) in one call: update_lhs, the four
  // right-hand sides of steppers/common.jl:7-118 built on the device, and the two paired solves.  h_point = current
  // Point vector; h_res = [x_residual(n); y_residual(p); z_residual(q)] and tau_residual from calc_convergence_params;
  // h_dirs receives dir_cent, dir_pred, dir_centadj, dir_predadj (4 Point vectors).  p = 0 only (callers fall back).
  DBuf s_point, s_resid, s_dots, s_dirs;
  double last_update_lhs_s = 0.0;   // wall seconds of the update_lhs part of the last step_directions call
  void build_rhs_pair(int stage, double* rhs2, const double* d_point, double mu, double tau, double kap, double tau_residual,
                      const double* d_dirs2, const double* dir_tau2, double* rs_flat /* 2 x (tau, kap) */, bool resident = false);
  void step_directions(const double* h_point, const double* h_res, double tau_residual, double mu, int max_ref_steps,
                       double res_norm_cutoff, double min_impr_tol, double* h_dirs, double* res_norms, int* n_solves, int* use_sqrt_out,
                       int* info, int* used_fallback, double* h_sol_const);
  void pair_solve_device(double* rhs2, const Scal* rs, double mu, double taubar, int max_ref_steps, double res_norm_cutoff,
                         double min_impr_tol, Scal* dsc, double* res_norms, int* n_solves, bool with_const = false, bool joint_const = false);
  // Round 6: the same in pieces whose scalars may stay ON THE DEVICE (resident = true): the tau / kap of a solve are formed by a
  // one-thread kernel from the scalar products (the host's operations in the host's order) and read by the kernels behind it from
  // device memory, so that a paired solve with its residual is queued without a single host round trip; the host reads the
  // scalar block once per pair (pair_finish), decides about refinement, and refines with the SAME column routines (one
  // round trip per refinement step; two columns that both need a step share its passes over G and the factor).
  // d_sc: [0, 6) solve dots (c'x, h'z per column), [8, 12) residual dots, [12, 14) residual maxima, [16, 20) tau / kap of the
  // direction per column, [20] dot_const, [24, 28) tau / kap of the last solve_system per column; mirrored to ctx.h_sc().
  DBuf d_sc;
  enum { SC_SOLVE = 0, SC_RESD = 8, SC_AMAX = 12, SC_DSC = 16, SC_DOTC = 20, SC_HZ = 21, SC_CSC = 24, SC_INFO = 30, SC_SEQ = 31, SC_N = 32, SC_WORK = 64, SC_TOTAL = 64 + 3 * 33 + 5 };
  unsigned long sc_seq = 0;   // sequence number of the last solve queued with device scalars (stamped into d_sc[SC_SEQ] by its tau kernel)
  void wait_scalars();        // host: until the pinned mirror carries that solve's stamp (HYP_DIR_POLL=0: a stream synchronisation)
  void ensure_d_sc();
  bool dirs_resident() const;   // HYP_DIR_RESIDENT (default on): single process, p = 0
  // solve_system (common.jl:129-182) for nr columns: rhs -> sol (Point layout, columns dimv() apart); base != null: the
  // direction's scalars become base - (the solve's) (a refinement step's correction), else the solve's own
  void cols_solve(double* sol, const double* rhs, int nr, const Scal* rs, double mu, double taubar, bool with_const, bool joint_const,
                  bool resident, bool both, const Scal* base, Scal* dsc_host);
  // apply_lhs (common.jl:79-121) of nr directions minus their right-hand sides; scalar products and maxima left in d_sc.
  // fresh: G dir.x / G' dir.z are formed here (else they are the ones cols_solve left in m_Gxd / m_t)
  void cols_residual(double* res, const double* dir, const double* rhs, int nr, const Scal* dsc_host, bool resident, bool fresh, bool both);
  void cols_read_scalars(bool resident);   // queue the copy of d_sc to its pinned mirror
  // the host's part behind a synchronisation: scalars of the nr directions and residuals out of the mirror
  void cols_finish(int nr, const Scal* rs, double mu, double taubar, bool resident, Scal* dsc, Scal* rsc, double* res_norms);
  void refine_cols(double* rhs, double* dir, double* res, const Scal* rs, Scal* dsc, Scal* rsc, double* res_norms, double mu, double taubar,
                   int max_ref_steps, double res_norm_cutoff, double min_impr_tol, int* n_solves, bool resident);
  void pair_enqueue(double* rhs2, const Scal* rs, double mu, double taubar, int max_ref_steps, bool with_const, bool joint_const,
                    bool resident, Scal* dsc_host, bool read_scalars = true);
  void pair_finish(double* rhs2, const Scal* rs, double mu, double taubar, int max_ref_steps, double res_norm_cutoff, double min_impr_tol,
                   bool with_const, bool joint_const, bool resident, Scal* dsc, double* res_norms, int* n_solves);
  static bool const3_on();    // HYP_CONST_COL3 (default on)
  static bool tri3_on();      // HYP_CONST_TRI3 (default on)
  static bool getenv_on(const char* name);
  bool dirs_x_only = false;   // hyp_sys_set_direction_rows: step_directions hands the host only the x rows and tau / kap of the directions
  hipEvent_t up_ev0 = nullptr, up_ev1 = nullptr;   // ordering of the late upload of the point / residuals on the helper stream
  hipEvent_t dirs_copied_ev = nullptr;   // the first pair's directions have reached the pinned staging
  std::chrono::steady_clock::time_point t_rest0;
  double last_rest_update_lhs_s = 0.0;
  void step_directions_rest(bool resident, double tau, double kap, double tau_residual, double mu, int max_ref_steps, double res_norm_cutoff,
                            double min_impr_tol, double* h_dirs, double* res_norms, int* n_solves, double* h_sol_const, double* hs_const,
                            double* hs_dirs, bool first_pair_done, Scal* d01, double* rn01);
  // returns res_norm; dir / rhs are HOST Point vectors (common.jl:15-76)
  double get_directions(double* h_dir, const double* h_rhs, double mu, double taubar, int max_ref_steps, double res_norm_cutoff,
                        double min_impr_tol, int* n_solves);
};

// Device-resident SymIndefDenseSystemSolver (symindef.jl:203-271): see symindef.hip
struct SymIndefSys {
  Ctx& ctx;
  int n, p, q, npq;
  std::vector<Cone*> cones;
  std::vector<int> offs;
  DBuf lhs;      // npq x npq, upper triangle: [0 A' G'; . 0 0; . . -M]
  DBuf fact, dinv, xb;
  BKFact bk;
  TriSolvePlan tri;
  bool fact_ok = false;
  SymIndefSys(Ctx& c, int n_, int p_, int q_, const std::vector<Cone*>& cs);
  void load(const double* hA, const double* hG);                 // :222-240 (host pointers, col-major p x n and q x n)
  void update_lhs(int* info, int* used_fallback);                // :242-262 without the constant-column solve
  void solve3(double* h_sol, const double* h_rhs);               // :264-271
};

}  // namespace hyp
