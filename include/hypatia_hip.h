/* C-ABI of libhypatia_hip.so: the MI355X-native replacement for Hypatia.jl's per-iteration hot path.
 *
 * What binds to this: a Julia `ccall` glue that defines new subtypes of
 * `Hypatia.Cones.Cone{Float64}` and `Hypatia.Solvers.QRCholSystemSolver{Float64}`; in this repo the
 * Python mirror `hypatia.jl_amd` (ctypes) and the tests.  Citations are file:line in the reference
 * tree (chriscoey/Hypatia.jl v0.5.1).
 *
 * Conventions
 *   - every function returns int: 0 = ok, < 0 = bad argument / HIP error (see hyp_last_error),
 *     numerical outcomes (LAPACK-style info, booleans) come back through out-pointers;
 *   - all arrays are HOST pointers to Float64, column-major; the library copies what it keeps and
 *     never returns pointers to its own memory; leading dimensions are passed explicitly;
 *   - calls are synchronous; one caller thread per context; no C++ exception crosses the boundary;
 *   - handles are opaque and freed only by the matching destroy.
 */
#ifndef HYPATIA_HIP_H
#define HYPATIA_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hyp_ctx hyp_ctx;
typedef struct hyp_cone hyp_cone;
typedef struct hyp_sys hyp_sys;

/* ---- context -------------------------------------------------------------------------------- */
int hyp_ctx_create(int device, hyp_ctx** out);
int hyp_ctx_destroy(hyp_ctx* ctx);
int hyp_ctx_synchronize(hyp_ctx* ctx);   /* hipDeviceSynchronize on the context's device */
const char* hyp_last_error(hyp_ctx* ctx);
int hyp_device_count(int* out);
/* timers in the order of Solvers.jl:86-96 (rescale, initx, inity, unproc, loadsys, upsys, upfact,
 * uprhs, getdir, search); the library fills upsys/upfact, the caller the rest */
int hyp_get_timers(hyp_ctx* ctx, double* out10);
int hyp_reset_timers(hyp_ctx* ctx);
/* HIP-event timings (ms, accumulated since the last reset) of the update_lhs phases measured on the
 * library stream: out8 = [sqrt-Hessian products, Schur syrk, Cholesky, #update_lhs_fact, #syrk launches, #Hessian factorizations of
 * generic cones (Cones.jl:239-251), #Bunch-Kaufman factorizations (dense.jl:164-165), #gradients of generic-Hessian cones] */
int hyp_get_kernel_stats(hyp_ctx* ctx, double* out8);
/* fall-backs behind a failed Cholesky since the context was created (posdef_fact_copy!'s second link, dense.jl:194-215, symm_fact!
 * :164-165): out3 = [hybrid factorizations (Cholesky block steps kept in front of the rook-pivoted trailing block), calls in which the
 * growth guard refused kept steps, plain rook-pivoted factorizations from column 0] */
int hyp_ctx_bk_stats(hyp_ctx* ctx, long long* out3);
/* solve plans of triangular factors built since the context was created (the super-block inverses behind every ldiv! on a large
 * factor: qrchol.jl:66-69, Cones.jl:113-118) and how many of them run ONE step of refinement against the factor instead of two
 * because the inverses' measured quality makes the second step void (HYP_TRSV_ADAPT, docs/NUMERICS.md): out2 = [plans, one-step plans] */
int hyp_ctx_plan_stats(hyp_ctx* ctx, long long* out2);

/* ---- cone lifecycle: constructors of src/Cones/nonnegative.jl:27-33, possemideftri.jl:36-46 --- */
int hyp_cone_create_nonnegative(hyp_ctx* ctx, int dim, hyp_cone** out);
int hyp_cone_create_possemideftri(hyp_ctx* ctx, int dim, hyp_cone** out);
/* Cones.PosSemidefTri{Float64, ComplexF64}(dim) (possemideftri.jl:9-46 with R = Complex{T}): Hermitian matrices of side s in the
 * complex svec format of arrayutilities.jl:188-210, dim = s^2; nu = s */
int hyp_cone_create_possemideftri_complex(hyp_ctx* ctx, int dim, hyp_cone** out);
/* Cones.EpiNormSpectral{Float64,Float64}(d1, d2; use_dual) (epinormspectral.jl:53-66): dim = 1 + d1*d2 */
int hyp_cone_create_epinormspectral(hyp_ctx* ctx, int d1, int d2, int use_dual, hyp_cone** out);
/* Cones.EpiNormSpectral{Float64, ComplexF64}(d1, d2; use_dual): W complex d1 x d2, the cone vector is (u, W) with W as (re, im)
 * pairs in column-major order (vec_copyto!, arrayutilities.jl:30-60): dim = 1 + 2 d1 d2, nu = d1 + 1 */
int hyp_cone_create_epinormspectral_complex(hyp_ctx* ctx, int d1, int d2, int use_dual, hyp_cone** out);
/* Cones.WSOSInterpNonnegative{Float64,Float64}(U, Ps; use_dual) (wsosinterpnonnegative.jl:49-63):
 * Ps[k] is U x Ls[k], column-major; the matrices are copied to the device */
int hyp_cone_create_wsosinterpnonnegative(hyp_ctx* ctx, int U, int K, const int* Ls, const double* const* Ps, int use_dual, hyp_cone** out);
/* Cones.WSOSInterpNonnegative{Float64,ComplexF64}(U, Ps; use_dual) (wsosinterpnonnegative.jl:15, 49-63 with R = Complex{T}; bases
 * from src/PolyUtils/complex.jl:13-72): Ps[k] is a U x Ls[k] matrix of complex numbers, (re, im) interleaved = Matrix{ComplexF64},
 * column-major; the cone vector stays real (dim = U); nu = sum_k Ls[k] */
int hyp_cone_create_wsosinterpnonnegative_complex(hyp_ctx* ctx, int U, int K, const int* Ls, const double* const* Ps, int use_dual, hyp_cone** out);
/* Cones.LinMatrixIneq{Float64}(As; use_dual) (linmatrixineq.jl:36-65), real dense symmetric members: As holds the
 * dim matrices one after the other, each side x side column-major (A_1 positive definite, dim <= side (side + 1) / 2);
 * copied to the device.  nu = side. */
int hyp_cone_create_linmatrixineq(hyp_ctx* ctx, int dim, int side, const double* As, int use_dual, hyp_cone** out);
/* The same with complex Hermitian members (linmatrixineq.jl:44-53 accepts any Hermitian A_i): As holds the dim matrices one
 * after the other, each side x side complex numbers (re, im interleaved = Matrix{ComplexF64}), column-major.  The cone vector
 * stays real (one weight per member); nu = side. */
int hyp_cone_create_linmatrixineq_complex(hyp_ctx* ctx, int dim, int side, const double* As, int use_dual, hyp_cone** out);
/* Cones.DoublyNonnegativeTri{Float64}(dim; use_dual) (doublynonnegativetri.jl:38-52): svec format, dim = side (side + 1) / 2;
 * nu = dim */
int hyp_cone_create_doublynonnegativetri(hyp_ctx* ctx, int dim, int use_dual, hyp_cone** out);
/* Cones.HypoRootdetTri{Float64, Float64}(dim; use_dual) (hyporootdettri.jl:44-59): (u, svec(W)), dim = 1 + side (side + 1) / 2;
 * nu = 1 + side */
int hyp_cone_create_hyporootdettri(hyp_ctx* ctx, int dim, int use_dual, hyp_cone** out);
/* Cones.HypoPerLogdetTri{Float64, Float64}(dim; use_dual) (hypoperlogdettri.jl:43-58): (u, v, svec(W)),
 * dim = 2 + side (side + 1) / 2; nu = 2 + side */
int hyp_cone_create_hypoperlogdettri(hyp_ctx* ctx, int dim, int use_dual, hyp_cone** out);
/* The same two cones over complex Hermitian matrices, Cones.HypoRootdetTri{Float64, ComplexF64}(dim) and
 * Cones.HypoPerLogdetTri{Float64, ComplexF64}(dim): the matrix part is the complex svec (side^2 reals), dim = 1 + side^2 resp.
 * 2 + side^2; nu = 1 + side resp. 2 + side */
int hyp_cone_create_hyporootdettri_complex(hyp_ctx* ctx, int dim, int use_dual, hyp_cone** out);
int hyp_cone_create_hypoperlogdettri_complex(hyp_ctx* ctx, int dim, int use_dual, hyp_cone** out);
/* Cones.WSOSInterpPosSemidefTri{Float64}(R, U, Ps; use_dual) (wsosinterppossemideftri.jl:46-69): R x R symmetric matrices of
 * polynomials given by U interpolant values each (svec order by blocks of length U), dim = U R (R + 1) / 2; Ps[k] is U x Ls[k],
 * column-major, copied to the device; nu = R sum_k Ls[k] */
int hyp_cone_create_wsosinterppossemideftri(hyp_ctx* ctx, int R, int U, int K, const int* Ls, const double* const* Ps, int use_dual,
                                            hyp_cone** out);
int hyp_cone_destroy(hyp_cone* cone);
int hyp_cone_dimension(hyp_cone* cone, int* out);            /* Cones.jl:34 */
int hyp_cone_get_nu(hyp_cone* cone, double* out);            /* Cones.jl:41 */
int hyp_cone_use_dual_barrier(hyp_cone* cone, int* out);     /* Cones.jl:138 */

/* ---- cone state (Cones.jl:140-171, 185-186) -------------------------------------------------- */
int hyp_cone_set_initial_point(hyp_cone* cone, double* out_dim);
int hyp_cone_load_point(hyp_cone* cone, const double* point, double scal);   /* cone.point = scal * point */
int hyp_cone_load_dual_point(hyp_cone* cone, const double* point);
int hyp_cone_reset_data(hyp_cone* cone);
int hyp_cone_get_point(hyp_cone* cone, double* out_dim);        /* mirrors the field cone.point */
int hyp_cone_get_dual_point(hyp_cone* cone, double* out_dim);

/* ---- cone oracles (Cones.jl:56-134, 189-237, 273-310) ----------------------------------------- */
int hyp_cone_is_feas(hyp_cone* cone, int* out);
int hyp_cone_is_dual_feas(hyp_cone* cone, int* out);
int hyp_cone_grad(hyp_cone* cone, double* out_dim);
/* prod, arr: dim x ncols, leading dimensions ldp, lda (SubArray views of HGQ2, qrchol.jl:162-165) */
int hyp_cone_hess_prod(hyp_cone* cone, double* prod, int ldp, const double* arr, int lda, int ncols);
int hyp_cone_inv_hess_prod(hyp_cone* cone, double* prod, int ldp, const double* arr, int lda, int ncols);
int hyp_cone_hess_prod_slow(hyp_cone* cone, double* prod, int ldp, const double* arr, int lda, int ncols);
int hyp_cone_use_sqrt_hess_oracles(hyp_cone* cone, int arr_dim, int* out);
int hyp_cone_sqrt_hess_prod(hyp_cone* cone, double* prod, int ldp, const double* arr, int lda, int ncols);
int hyp_cone_inv_sqrt_hess_prod(hyp_cone* cone, double* prod, int ldp, const double* arr, int lda, int ncols);
int hyp_cone_dder3(hyp_cone* cone, const double* dir, double* out_dim);
/* update_use_hess_prod_slow (Cones.jl:222-231) and the field cone.use_hess_prod_slow (test/cone.jl:89-95);
 * no-ops for cones without the slow path */
int hyp_cone_update_use_hess_prod_slow(hyp_cone* cone, int* out);
int hyp_cone_set_use_hess_prod_slow(hyp_cone* cone, int value);
int hyp_cone_check_numerics(hyp_cone* cone, int* out);
int hyp_cone_get_proxsqr(hyp_cone* cone, double irtmu, int use_max_prox, double* out);
int hyp_cone_hess(hyp_cone* cone, double* out_dimxdim);        /* explicit, tests / sparse solvers only */
int hyp_cone_inv_hess(hyp_cone* cone, double* out_dimxdim);

/* ---- QRCholDenseSystemSolver (systemsolvers/qrchol.jl:104-257) -------------------------------- */
/* cones[k] occupies rows sum(dim[0..k-1]) .. of z / s (Models.jl:54-66 cone_idxs) */
int hyp_sys_create(hyp_ctx* ctx, int n, int p, int q, hyp_cone* const* cones, int ncones, hyp_sys** out);
int hyp_sys_destroy(hyp_sys* sys);

/* ---- SymIndefDenseSystemSolver (src/Solvers/systemsolvers/symindef.jl:203-271): the 3x3 symmetric indefinite form
 * [0 A' G'; A 0 0; G 0 -M], M_k = inv_hess(cone_k) for a primal-barrier cone and hess(cone_k) for a dual-barrier one (at
 * the cones' currently loaded, scaled points: :247-252), factored by Bunch-Kaufman with rook pivoting (symm_fact_copy!,
 * src/linearalgebra/dense.jl:170-184).  No preprocessing of A is needed.  The Julia subtype keeps setup_rhs3
 * (symindef.jl:33-56) and the shared 6 -> 4 -> 3 reductions on the host and calls: */
typedef struct hyp_symindef hyp_symindef;
int hyp_symindef_create(hyp_ctx* ctx, int n, int p, int q, hyp_cone* const* cones, int ncones, hyp_symindef** out);
int hyp_symindef_destroy(hyp_symindef* sys);
/* load (:222-240): A is p x n (NULL when p == 0), G is q x n, both column-major; copied to the device */
int hyp_symindef_load(hyp_symindef* sys, const double* A, const double* G);
/* update_lhs (:242-257) without the constant-column solve: z-blocks from the cones, then symm_fact_copy!.
 * used_fallback = 1 when the first factorization met an exactly singular pivot and increase_diag! was applied;
 * info = 0 <=> issuccess(fact) */
int hyp_symindef_update_lhs(hyp_symindef* sys, int* info, int* used_fallback);
/* solve_subsystem3 (:264-271): sol.vec = fact \ rhs.vec, vectors [x(n); y(p); z(q)] */
int hyp_symindef_solve3(hyp_symindef* sys, double* sol_vec, const double* rhs_vec);
/* y = alpha * op(G) x + beta * y on the device-resident G (trans != 0: op(G) = G'), for the residuals the driver computes */
int hyp_symindef_mul_G(hyp_symindef* sys, int trans, double alpha, const double* x, double beta, double* y);
int hyp_symindef_get_lhs(hyp_symindef* sys, double* out_npqxnpq);   /* upper triangle meaningful (tests) */

/* ---- QRCholDenseSystemSolver, continued ------------------------------------------------------- */
/* load (qrchol.jl:138-179): G = model.G (q x n).  When p == 0 pass NULL for GQ1, GQ2, Q, R (GQ2 = G,
 * Ap_Q = I).  Otherwise Q = Ap_Q (n x n), R = Ap_R (p x p), and either GQ1 = (G*Ap_Q)[:, 1:p], GQ2 = (G*Ap_Q)[:, p+1:n]
 * or NULL for both: the product G*Ap_Q of qrchol.jl:154 is then formed on the device. */
int hyp_sys_load(hyp_sys* sys, const double* G, const double* GQ1, const double* GQ2, const double* Q, const double* R);
/* update_lhs_fact (qrchol.jl:201-257): Schur assembly + posdef_fact_copy! (src/linearalgebra/dense.jl:194-215).
 * use_sqrt_out[ncones] receives use_sqrt_hess_cones.  used_fallback names the link of the chain that produced the
 * factorization: 0 Cholesky, 1 Bunch-Kaufman with rook pivoting (symm_fact!, dense.jl:164-165) after a failed
 * Cholesky, 2 increase_diag! (dense.jl:106-113) + Bunch-Kaufman after that found an exactly singular pivot.
 * info = 0, or LAPACK's info of the last link (issuccess(fact) = info == 0). */
int hyp_sys_update_lhs_fact(hyp_sys* sys, int* use_sqrt_out, int* info, int* used_fallback);
/* the two halves of update_lhs_fact, for the multi-GPU path: each process assembles the Schur sum over
 * ITS cones (qrchol.jl:214-246), the partial n x n matrices are summed across processes (RCCL
 * all-reduce on a caller-owned device buffer, or through the host), then every process factors
 * (qrchol.jl:249-250).  *_dev take DEVICE pointers to (n-p)^2 doubles. */
int hyp_sys_assemble_lhs(hyp_sys* sys, int* use_sqrt_out);
int hyp_sys_factor_lhs(hyp_sys* sys, int* info, int* used_fallback);
int hyp_sys_lhs_export_dev(hyp_sys* sys, void* dst_device);
int hyp_sys_lhs_import_dev(hyp_sys* sys, const void* src_device);
int hyp_sys_set_lhs(hyp_sys* sys, const double* in_nmpxnmp);
/* x <- lhs^-1 x with the current factorization (ldiv!(x_sub2, fact, Q2div), qrchol.jl:66-69); n - p entries */
int hyp_sys_potrs(hyp_sys* sys, double* x);
/* solve_subsystem3 (qrchol.jl:39-85): vectors of length n + p + q laid out [x; y; z] */
int hyp_sys_solve3(hyp_sys* sys, double* sol_vec, const double* rhs_vec);
/* block_hess_prod!.(out_k, in_k, cones) on a q-vector (qrchol.jl:87-98, 191-195) */
int hyp_sys_block_hess_prod(hyp_sys* sys, double* out_q, const double* in_q);
/* y = alpha * op(G) x + beta * y with the device-resident model.G (qrchol.jl:52,73; common.jl:91,94,144;
 * Solvers.jl:432,450).  trans = 0: x has n entries, y has q; trans = 1: x has q, y has n. */
int hyp_sys_mul_G(hyp_sys* sys, int trans, double alpha, const double* x, double beta, double* y);
/* ---- device-resident direction solves: Solvers.get_directions (systemsolvers/common.jl:15-76) with
 * solve_system / solve_subsystem4 (common.jl:129-182), setup_rhs3 (qrchol.jl:16-37) and the residual
 * apply_lhs (common.jl:79-121) all on the GPU; only the right-hand side goes in and the direction
 * comes out.  Vectors use Hypatia's Point layout [x(n); y(p); z(q); tau; s(q); kap] (point.jl:5-54). */
/* The products of Solvers.calc_convergence_params and calc_mu (src/Solvers/Solvers.jl:418-483: G' z, G x + s, h' z, z' s) in
 * one call on the resident G.  On a cone-sharded solver (hyp_sys_set_comm / hyp_sys_set_comm_rccl) z, s and out_Gx_s are THIS
 * process's rows and the sums over ranks are taken inside (ONE all-reduce of n + 2 doubles): out_Gtz (n) = G' z summed,
 * out_Gx_s (q) = G x + s on these rows, out_dots2 = {h' z, z' s} summed -- no q-vector ever leaves a rank.  Needs
 * hyp_sys_load_model (for h). */
int hyp_sys_residual_products(hyp_sys* sys, const double* x, const double* z, const double* s, double* out_Gtz, double* out_Gx_s, double* out_dots2);
/* The same plus the two residual norms of Solvers.jl:447-457 over ALL ranks' rows -- out_norms2 = {max |G x + s|, max |G x + s - h tau|}
 * -- carried by the SAME all-reduce (a slot per rank behind the sums; needs hyp_sys_set_comm_layout or an RCCL communicator, else a
 * second small exchange): a cone-sharded host then needs no collective of its own per iteration. */
int hyp_sys_residual_products2(hyp_sys* sys, const double* x, const double* z, const double* s, double tau, double* out_Gtz, double* out_Gx_s,
                               double* out_dots2, double* out_norms2);
/* In-place all-reduce of up to 32 host doubles over the solver's communicator (op 0 sum, 1 max, 2 min; a no-op on a single-GPU
 * solver): the residual norms of Solvers.jl:425-483 on a cone-sharded solver */
int hyp_sys_allreduce_host(hyp_sys* sys, double* buf, int count, int op);
/* model.c (n), model.b (p), model.h (q), model.A (p x n col-major, NULL when p = 0) after preprocessing */
int hyp_sys_load_model(hyp_sys* sys, const double* c, const double* b, const double* h, const double* A);
/* update_lhs (qrchol.jl:181-199): update_lhs_fact + sol_const = solve_subsystem3([-c; b; H h]); the
 * constant solution stays on the device (sol_const_out, n + p + q entries, may be NULL) */
int hyp_sys_update_lhs(hyp_sys* sys, int* use_sqrt_hess_cones_out, int* info, int* used_fallback, double* sol_const_out);
/* get_directions(stepper, solver): dir <- K^-1 rhs with up to max_ref_steps refinement steps; mu = solver.mu,
 * tau = solver.point.tau[]; res_norm_cutoff / min_impr_tol as in common.jl:15-20; n_solves counts solve_system calls */
int hyp_sys_get_directions(hyp_sys* sys, double* dir_vec, const double* rhs_vec, double mu, double tau, int max_ref_steps,
                           double res_norm_cutoff, double min_impr_tol, double* res_norm, int* n_solves);
/* the same for TWO independent right-hand sides at once (the stepper's (cent, pred) and (centadj, predadj)
 * pairs, steppers/combined.jl:60-95): dir_vecs / rhs_vecs hold two Point vectors back to back; every pass over
 * G, the factor and the cone matrices serves both.  res_norms[2]; n_solves counts solve_system calls (2 + refinements). */
int hyp_sys_get_directions2(hyp_sys* sys, double* dir_vecs, const double* rhs_vecs, double mu, double tau, int max_ref_steps,
                            double res_norm_cutoff, double min_impr_tol, double* res_norms, int* n_solves);
/* check_cone_points (steppers/search.jl:74-138) for one line-search candidate, all cones in one call:
 * cand_ztsk = [z(q); tau; s(q); kap] (the `ztsk` view of the candidate Point); min_prox / prox_bound /
 * use_max_prox / nup1 are the StepSearcher fields (search.jl:8-39).  accept = the function's Bool; prox =
 * searcher.prox on acceptance; n_loaded = number of leading cones whose point / dual_point were reloaded
 * (scaled by irtmu) before the sweep stopped -- the host mirrors of exactly those cones must follow. */
int hyp_sys_check_cone_points(hyp_sys* sys, const double* cand_ztsk, double min_prox, double prox_bound, int use_max_prox, double nup1,
                              int* accept, double* prox, int* n_loaded, double* irtmu);
/* The direction phase of step(::CombinedStepper) (steppers/combined.jl:60-95) in one call, p = 0 (the default reduce = true
 * path): update_lhs, the right-hand sides update_rhs_cent / _pred / _centadj / _predadj (steppers/common.jl:7-118) built on the
 * device, and the two paired solves.  point_vec = solver.point.vec; residuals = [x_residual(n); y_residual(p); z_residual(q)] and
 * tau_residual as left by calc_convergence_params (Solvers.jl:425-483); dir_vecs4 receives dir_cent, dir_pred, dir_centadj,
 * dir_predadj (four Point vectors back to back), res_norms4 their residual norms; info != 0: the factorization failed. */
int hyp_sys_step_directions(hyp_sys* sys, const double* point_vec, const double* residuals, double tau_residual, double mu, int max_ref_steps,
                            double res_norm_cutoff, double min_impr_tol, double* dir_vecs4, double* res_norms4, int* n_solves,
                            int* use_sqrt_hess_cones_out, int* info, int* used_fallback, double* sol_const_out /* n + p + q or NULL */);
/* Multi-GPU (one process per GPU, cones sharded): the hyp_sys of a rank is created over ITS cones and its rows of G only;
 * z / s vectors passed to the calls below are the local rows, x-space vectors are replicated.  At every exchange point
 * (the n x n Schur sum once per iteration; G' z, the scalar products over z, the residual norm and the line-search flags
 * per solve / trial) the library copies the payload into `device_staging` (a device buffer owned by the caller's
 * communication framework, >= n*n doubles) and calls allreduce(user, count, op) -- op 0 sum, 1 max, 2 min -- which must
 * all-reduce the first `count` doubles of that buffer IN PLACE over all ranks (RCCL) and return 0.  Every rank issues the
 * same sequence of calls.  allreduce = NULL restores single-GPU behaviour.  (qrchol.jl:219-246, search.jl:118-134) */
int hyp_sys_set_comm(hyp_sys* sys, int (*allreduce)(void* user, long count, int op), void* user, void* device_staging, long capacity_doubles);
/* The same exchanges through RCCL INSIDE the library (one process per GPU, xGMI): the library owns the communicator and
 * issues ncclAllReduce(sum | max | min, double) in place on its own stream -- no staging copy, no host callback, no host
 * synchronisation for device payloads.  Rank 0 creates the 128-byte id (ncclUniqueId) and distributes it out of band
 * (torch.distributed / MPI / a file); every rank then joins with hyp_comm_init_rank on its context's device. */
typedef struct hyp_comm hyp_comm;
int hyp_comm_unique_id(char* out128);
int hyp_comm_init_rank(hyp_ctx* ctx, int nranks, int rank, const char* id128, hyp_comm** out);
int hyp_comm_destroy(hyp_comm* comm);
/* in-place all-reduce of count doubles at a DEVICE pointer (op 0 sum, 1 max, 2 min); returns when it is complete */
int hyp_comm_allreduce(hyp_comm* comm, void* device_buf, long count, int op);
/* route the exchange points of hyp_sys_set_comm's description through the communicator (NULL: back to single GPU / callback) */
int hyp_sys_set_comm_rccl(hyp_sys* sys, hyp_comm* comm);
/* rank / world of the communicator behind hyp_sys_set_comm's callback (hyp_sys_set_comm_rccl takes them from its communicator).  With
 * the layout known, an n-vector exchange carries the scalars that accompany it -- sums in shared slots, maxima in a slot per rank --
 * in ONE all-reduce instead of three (per solve: G' z with h' z and the residual norm; common.jl:79-121, qrchol.jl:39-85).
 * world = 0 (default): layout unknown, every reduction is a collective of its own. */
int hyp_sys_set_comm_layout(hyp_sys* sys, int rank, int world);
/* exchanges issued since creation by place in the iteration: out16[0] Schur sum, [1] solve G' z, [2] solve h' z, [3] residual (fused),
 * [4] residual h' z, [5] residual norm, [6] constant column, [7] [8] candidate screen, [9] [10] line-search trial, [11] residual
 * products, [12] host-requested, [13] screen agreement, [15] other */
int hyp_sys_comm_hist(hyp_sys* sys, long long* out16);
/* time spent in those exchanges since creation, milliseconds, by the same places (out16[i] belongs to hyp_sys_comm_hist's out16[i]):
 * RCCL inside the library: HIP events on the library's stream around every ncclAllReduce; callback transport: host clock around the
 * callback.  out16[14] = the Schur exchange of qrchol.jl:219-246's sum INCLUDING the pack / unpack of its upper triangle (out16[0] is
 * its collective alone).  Synchronises with the pending exchanges. */
int hyp_sys_comm_times(hyp_sys* sys, double* out16);
/* K-panel sharding of ONE replicated model (a single cone: configs[1] / [2]): with a communicator (or callback) installed
 * and world > 1, hyp_sys_update_lhs / _assemble_lhs sum only rows [q rank / world, q (rank + 1) / world) of the sqrt-Hessian
 * product into the Schur matrix (the K dimension of outer_prod!, qrchol.jl:234) and all-reduce the n x n result; model,
 * cones, points and solves stay replicated, so no other exchange exists.  world = 1 switches it off. */
int hyp_sys_set_kshard(hyp_sys* sys, int rank, int world);
/* out2 = {exchanges issued by this solver since its creation, doubles moved by them} */
int hyp_sys_comm_stats(hyp_sys* sys, double* out2);
/* Which rows of the four directions hyp_sys_step_directions copies into dir_vecs4: 0 (default) the whole Point vectors; 1 only
 * the x rows and tau / kap of each direction -- what update_stepper_points (steppers/combined.jl:124-170) still needs on the host when
 * the line search runs on the directions the call left on the device (hyp_sys_search_alpha_resident hands back the accepted
 * candidate's z / tau / s / kap rows); the z / s rows of the caller's block are then left as they are.  At q = 207 360 the whole
 * block is 13 MB over PCIe and through a pageable copy per iteration. */
int hyp_sys_set_direction_rows(hyp_sys* sys, int x_rows_only);
/* wall seconds the update_lhs part (solver.time_upsys) took inside the last hyp_sys_step_directions call */
int hyp_sys_last_update_lhs_seconds(hyp_sys* sys, double* out);
/* Measurement helper: HIP-event time (ms, averaged over reps back-to-back launches) of the four passes over the resident
 * G (q x n) that a KKT solve is made of (qrchol.jl:51-53, 71-73): out4 = {G' X on 2 columns, G X on 2 columns, G' x, G x}.
 * Algorithmic bytes of each pass: q*n*8. */
int hyp_sys_bench_gemv(hyp_sys* sys, int reps, double* ms_out4);
/* search_alpha (steppers/search.jl:46-69) for one stepper mode: forms each candidate exactly as update_stepper_points
 * (steppers/combined.jl:124-170) does -- all vectors are `ztsk` views [z(q); tau; s(q); kap] of the current point and of the
 * four directions -- and runs check_cone_points on it, from alpha_sched[start] on.  accepted_index = 0-based index of the first
 * accepted step or -1; cand_ztsk (caller buffer) holds the last candidate tried; n_loaded / irtmu as in
 * hyp_sys_check_cone_points (host mirrors of the first n_loaded cones follow the last candidate). */
int hyp_sys_search_alpha(hyp_sys* sys, const double* point_ztsk, const double* dir_cent, const double* dir_pred, const double* dir_centadj,
                         const double* dir_predadj, int unadj_only, int cent_only, const double* alpha_sched, int nsched, int start,
                         double min_prox, double prox_bound, int use_max_prox, double nup1, double* cand_ztsk, int* accepted_index,
                         double* prox, int* n_trials, int* n_loaded, double* irtmu);
/* hyp_sys_search_alpha for the point and the four directions of the LAST hyp_sys_step_directions call, which are still on the
 * device: nothing of length q is uploaded, the candidates of the whole schedule are formed there (combined.jl:124-170, the host
 * loop's operations in the host loop's order) and screened side by side; only a candidate that survives the screen comes back for
 * the sequential acceptance test.  For the caller who has not touched the vectors since that call (steppers/combined.jl:60-118
 * does not).  Error unless hyp_sys_search_screen_stats reports usable (sharded: on EVERY rank -- agree on it first, all ranks
 * must make the same call) and a step_directions call preceded.  Sharded: the vectors are this rank's rows, as for
 * hyp_sys_search_alpha. */
int hyp_sys_search_alpha_resident(hyp_sys* sys, int unadj_only, int cent_only, const double* alpha_sched, int nsched, int start, double min_prox,
                                  double prox_bound, int use_max_prox, double nup1, double* cand_ztsk, int* accepted_index, double* prox,
                                  int* n_trials, int* n_loaded, double* irtmu);
/* The side-by-side candidate screen inside hyp_sys_search_alpha[_resident]: the rejecting tests of search.jl:86-116 /
 * possemideftri.jl:80-95 / Cones.jl:294-310 for all remaining candidates of the schedule at once; acceptance stays with the
 * sequential test.  Applies to a model whose cones -- on a sharded solver: this rank's cones, on every rank -- are all
 * primal-barrier PosSemidefTri cones of one size (one cone or many: batch = candidates x cones; sharded: two all-reduces of a few
 * numbers per candidate make the verdicts the same on all ranks).
 * usable = 1 where it applies to this handle (0 also with HYP_SEARCH_SCREEN=0), screens run so far and candidates they rejected. */
int hyp_sys_search_screen_stats(hyp_sys* sys, int* usable, long long* screens, long long* rejected);
int hyp_sys_get_lhs(hyp_sys* sys, double* out_nmpxnmp);        /* upper triangle meaningful (tests) */

/* ---- dense kernels exposed for parity tests and micro-benchmarks ------------------------------- */
/* C = alpha * op(A) * B + beta * C (col-major; transa: A is K x M; upper != 0: only col >= row) */
int hyp_dense_gemm(hyp_ctx* ctx, int transa, int upper, int M, int N, int K, double alpha, const double* A, int lda,
                   const double* B, int ldb, double beta, double* C, int ldc);
/* C = A'A, upper triangle only (dsyrk 'U','T'; A is K x N): the Schur-assembly kernel path incl. split-K */
int hyp_dense_syrk(hyp_ctx* ctx, int N, int K, const double* A, int lda, double* C, int ldc);
/* in-place upper Cholesky (dpotrf 'U'); info as LAPACK */
int hyp_dense_potrf(hyp_ctx* ctx, int n, double* A, int lda, int* info);
/* dposv 'U': A (upper triangle read) is overwritten by its Cholesky factor U, x (in: b) by A^-1 b */
int hyp_dense_posv(hyp_ctx* ctx, int n, double* A, int lda, double* x, int* info);
/* the same with nrhs right-hand sides (X: n x nrhs, leading dimension ldx, overwritten by the solution): the blocked multi-column
 * triangular sweeps behind every cone's ldiv! on a factor (Cones.jl:113-118, wsosinterpnonnegative.jl:124) -- diagonal blocks
 * through their inverses + two refinement steps against the factor, i.e. dtrsm's backward error (dense.jl:164-200) */
int hyp_dense_posv_multi(hyp_ctx* ctx, int n, double* A, int lda, double* X, int nrhs, int ldx, int* info);
/* Symmetric indefinite solve through the rook-pivoted factorization P A P' = U' D U (the reference's
 * bunchkaufman!(Symmetric(A, :U), true, check = false) + ldiv!: symm_fact!, src/linearalgebra/dense.jl:164-165).
 * A: upper triangle in, U (unit upper, diagonal explicit) out.  perm / blk / d / e (length n, each may be NULL)
 * describe the factorization: perm[i] = original index in position i; blk[i] = 0 for a 1x1 pivot d[i], 1 / 2 for
 * the two rows of a 2x2 pivot [[d[i], e[i]], [e[i], d[i+1]]].  x (n x nrhs, ld ldx) is overwritten with A^-1 x.
 * info = 0, or the 1-based position of the first exactly singular pivot (LAPACK dsytrf_rook). */
int hyp_dense_sysv_rook(hyp_ctx* ctx, int n, double* A, int lda, double* x, int nrhs, int ldx, int* info, int* perm, int* blk,
                        double* d, double* e);
/* posdef_fact_copy! + ldiv! on a host matrix (dense.jl:194-215): Cholesky, and behind a failed one the symmetric indefinite
 * factorization -- the Cholesky steps in front of the failing pivot's 128-column block kept WHERE their pivots and rows pass the growth guard
 * (pivot^2 >= n eps max|a_ii|, row entries^2 <= 16 max|a_ii|: what an unpivoted elimination needs to stay bounded), rook pivoting
 * (dsytrf_rook's rule) on the trailing block only; bk_start = the column it started from (0: the whole matrix, also with
 * HYP_BK_HYBRID=0 or when the first block step already fails the guard) */
int hyp_dense_posdef_solve(hyp_ctx* ctx, int n, double* A, int lda, double* x, int nrhs, int ldx, int* info, int* used_fallback, int* bk_start);
/* Column-pivoted Householder QR on the device with LAPACK dgeqp3's semantics (= Julia's qr!(AG, ColumnNorm()) of
 * find_initial_x, src/Solvers/process.jl:64-178; the rank decision of :373-382 reads the diagonal of R).  A is m x n column-major
 * (host, copied); rhs (m entries, may be NULL) rides along as an extra column: after the factorization it holds Q' rhs.
 * hyp_qrcp_get: jpvt (n, 0-based: column jpvt[i] of A is column i of A P), R (min(m, n) x n column-major with leading dimension
 * min(m, n), upper triangle meaningful), rdiag (min(m, n)), qtb (m) -- any of them may be NULL.
 * hyp_qrcp_apply_q: vec (m, host, in place) <- Q' vec (trans != 0) or Q vec. */
typedef struct hyp_qrcp hyp_qrcp;
int hyp_qrcp_factor(hyp_ctx* ctx, int m, int n, const double* A, int lda, const double* rhs, hyp_qrcp** out);
int hyp_qrcp_get(hyp_qrcp* q, int* jpvt, double* R, double* rdiag, double* qtb);
int hyp_qrcp_apply_q(hyp_qrcp* q, int trans, double* vec);
int hyp_qrcp_destroy(hyp_qrcp* q);
/* Set-up helper for find_initial_x (src/Solvers/process.jl:64-178): least squares x = argmin ||A x - b|| for a tall, dense,
 * well-conditioned A (m x n col-major, host pointers) by the Cholesky factorization of A'A on the device with one step of
 * corrected semi-normal equations; rcond_est ~ sigma_min(A) / sigma_max(A) from power iterations, info = dpotrf's.  The
 * reference uses a column-pivoted QR there (also to detect dependent columns): the caller takes this x only when info = 0
 * and rcond_est is far from the rank-decision threshold, and otherwise runs the reference's pivoted QR on the host. */
int hyp_dense_lstsq_normal(hyp_ctx* ctx, int m, int n, const double* A, int lda, const double* b, double* x, double* rcond_est, int* info);
/* y = alpha * op(A) x + beta * y (A is m x n col-major; trans != 0: op(A) = A') */
int hyp_dense_gemv(hyp_ctx* ctx, int trans, int m, int n, double alpha, const double* A, int lda, const double* x, double beta,
                   double* y);
/* Both products of one matrix in ONE pass over it (test hook of the kernel behind apply_lhs' residual, systemsolvers/common.jl:79-121,
 * and calc_convergence_params, Solvers.jl:425-483): Yn[:, r] = A Xn[:, r] + beta_n Yn[:, r] (Xn n x nr, Yn m x nr) and
 * Yt[:, r] = A' Xt[:, r] + beta_t Yt[:, r] (Xt m x nr, Yt n x nr), nr = 1 or 2, all column-major and dense.  used_fused = 1 when the
 * one-pass kernel applies (m >= 1024, n >= 64, lda % 4 == 0), 0 when the two one-sided products ran instead. */
int hyp_dense_gemv_both(hyp_ctx* ctx, int m, int n, int nr, const double* A, int lda, const double* Xn, double beta_n, double* Yn,
                        const double* Xt, double beta_t, double* Yt, int* used_fused);
/* Measurement helper: HIP-event time (ms, mean of reps) of the blocked upper Cholesky of an n x n positive definite matrix
 * resident in HBM (posdef_fact_copy!'s first link, src/linearalgebra/dense.jl:194-200). */
int hyp_bench_potrf(hyp_ctx* ctx, int n, int reps, double* ms_out);
/* Measurement helper for the one-, two- and three-right-hand-side potrs of qrchol.jl:68 on an n x n factor resident in HBM: HIP-event
 * times (ms, mean of reps) of [0] building the super-block solve plan, [1] U'^-1 then U^-1 on one vector, [2] on two vectors, [3] on
 * three.  x_out (6 n doubles, may be NULL): the solutions of the last repetition (one vector, the two, the three), for A/B comparisons. */
int hyp_bench_trsv(hyp_ctx* ctx, int n, int reps, double* ms_out4, double* x_out);
/* time `reps` launches of the syrk C = A'A (A is K x N) with HIP events on the library stream; ms per launch */
int hyp_bench_syrk(hyp_ctx* ctx, int N, int K, int reps, double* ms_out);

#ifdef __cplusplus
}
#endif
#endif
