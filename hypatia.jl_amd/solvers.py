"""Host-side mirror of the callers of the hot path: `Solvers.Solver`, `CombinedStepper`, `StepSearcher`,
`Point`, preprocessing -- so the MI355X path can be driven end to end without Julia.

In production these stay in Julia (Hypatia itself); only the `Cone` and `SystemSolver` subtypes are
replaced.  This mirror follows the reference call for call:
  /root/reference/src/Solvers/Solvers.jl:62-240, 245-416, 418-548
  /root/reference/src/Solvers/point.jl:5-54
  /root/reference/src/Solvers/process.jl:13-60, 64-178, 182-365, 385-458
  /root/reference/src/Solvers/search.jl:8-138
  /root/reference/src/Solvers/steppers/common.jl:7-118, steppers/combined.jl:5-189
  /root/reference/src/Solvers/systemsolvers/common.jl:15-121
Every product with the constraint matrix G goes through the system solver's device-resident copy
(`syssolver.mul_G`); cone oracles and the KKT solves are C-ABI calls.  Nothing here imports `oracle/`.
"""
import os
import time

import numpy as np
import scipy.linalg as sla

from .models import Model
from .systemsolvers import QRCholDenseSystemSolver

EPS = np.finfo(np.float64).eps


def _cap(sysv, what):
    """which device-resident routines a system solver offers (single-GPU solver: all; sharded solver: the fused step
    and the schedule walk, on its local rows)"""
    if not getattr(sysv, "native_directions", False):
        return False
    caps = getattr(sysv, "native_caps", None)
    return True if caps is None else (what in caps)


def _env_on(name):
    """experiment switches: set and not "0" """
    v = os.environ.get(name)
    return v is not None and v not in ("", "0")

try:   # keep the host BLAS pool small inside the iteration loop: the driver's numpy calls are tiny, and a
    # large pool of spinning OpenBLAS workers starves the HIP runtime's own threads (measured: 3x slower)
    from threadpoolctl import ThreadpoolController as _ThreadpoolController
except Exception:   # pragma: no cover
    _ThreadpoolController = None
_controller = None


class _blas_limit:
    """context manager: cap the host BLAS pools while the IPM loop runs.  The controller (a scan of the loaded
    shared libraries, ~0.4 ms) is built once; re-entering with the limit already in force is free."""
    _depth = 0

    def __init__(self, n=int(os.environ.get("HYP_HOST_BLAS_THREADS", "1"))):
        self.n = n
        self.cm = None

    def __enter__(self):
        global _controller
        if _ThreadpoolController is not None and _blas_limit._depth == 0:
            if _controller is None:
                _controller = _ThreadpoolController()
            self.cm = _controller.limit(limits=self.n, user_api="blas")
            self.cm.__enter__()
        _blas_limit._depth += 1
        return self

    def __exit__(self, *a):
        _blas_limit._depth -= 1
        if self.cm is not None:
            self.cm.__exit__(*a)
            self.cm = None
        return False


# status codes (Solvers.jl:34-49)
STATUSES = ("NotLoaded", "Loaded", "SolveCalled", "Optimal", "PrimalInfeasible", "DualInfeasible",
            "IllPosed", "PrimalInconsistent", "DualInconsistent", "SlowProgress", "IterationLimit",
            "TimeLimit", "NumericalFailure", "UnknownStatus")


class Point:
    """point.jl:5-54: flat vector [x(n); y(p); z(q); tau; s(q); kap] with views."""

    def __init__(self, model=None, dims=None, storage=None):
        if model is not None:
            n, p, q = model.n, model.p, model.q
        else:
            n, p, q = dims
        self.n, self.p, self.q = n, p, q
        self.tau_idx = n + p + q
        if storage is None:
            self.vec = np.zeros(self.tau_idx + q + 2)
        else:   # a caller-owned contiguous row (the fused device call writes the four directions straight into these)
            assert storage.shape == (self.tau_idx + q + 2,) and storage.flags.c_contiguous
            self.vec = storage
        v = self.vec
        self.x = v[:n]
        self.y = v[n:n + p]
        self.z = v[n + p:n + p + q]
        self.s = v[self.tau_idx + 1:self.tau_idx + 1 + q]
        self.ztsk = v[n + p:]
        if model is not None:
            self.z_views = [self.z[idx] for idx in model.cone_idxs]
            self.s_views = [self.s[idx] for idx in model.cone_idxs]
            self.dual_views = [self.s_views[k] if c.use_dual_barrier() else self.z_views[k]
                               for k, c in enumerate(model.cones)]
            self.primal_views = [self.z_views[k] if c.use_dual_barrier() else self.s_views[k]
                                 for k, c in enumerate(model.cones)]

    @property
    def tau(self):
        return self.vec[self.tau_idx]

    @tau.setter
    def tau(self, v):
        self.vec[self.tau_idx] = v

    @property
    def kap(self):
        return self.vec[-1]

    @kap.setter
    def kap(self, v):
        self.vec[-1] = v


# ==============================================================================================
# directions (systemsolvers/common.jl:15-121)
# ==============================================================================================
def apply_lhs(stepper, solver):   # common.jl:79-121
    model = solver.model
    dir, res = stepper.dir, stepper.temp
    tau_dir, kap_dir = dir.tau, dir.kap
    sys = solver.syssolver
    res.x[:] = model.c * tau_dir
    sys.mul_G(True, dir.z, alpha=1.0, beta=1.0, y=res.x)
    res.z[:] = model.h * tau_dir - dir.s
    sys.mul_G(False, dir.x, alpha=-1.0, beta=1.0, y=res.z)
    res.tau = -(model.c @ dir.x) - model.h @ dir.z - kap_dir
    if model.p != 0:
        res.x += model.A.T @ dir.y
        res.y[:] = model.b * tau_dir - model.A @ dir.x
        res.tau = res.tau - model.b @ dir.y
    hooks = getattr(model, "dist_hooks", None)
    if hooks is not None:      # cone-sharded model: owners compute concurrently, one all-reduce
        hooks.apply_lhs_cones(res, dir)
    else:
        for k, cone_k in enumerate(model.cones):
            s_res_k = res.s_views[k]
            cone_k.hess_prod_slow(s_res_k, dir.primal_views[k])
            s_res_k += dir.dual_views[k]
    tau = solver.point.tau
    res.kap = solver.mu / tau * tau_dir / tau + kap_dir
    return res


def get_directions(stepper, solver, min_impr_tol=0.5):   # common.jl:15-76
    rhs, dir, res = stepper.rhs, stepper.dir, stepper.temp
    dir_temp = stepper.dir_temp
    syssolver = solver.syssolver
    res_norm_cutoff = solver.res_norm_cutoff
    max_ref_steps = solver.max_ref_steps

    if _cap(syssolver, "single"):   # the whole routine on the device (hyp_sys_get_directions)
        res_norm, ns = syssolver.get_directions_native(solver, dir, rhs, min_impr_tol)
        solver.n_solves += ns
        assert not np.isnan(res_norm)
        if max_ref_steps > 0:
            solver.worst_dir_res = max(solver.worst_dir_res, res_norm)
        return dir

    syssolver.solve_system(solver, dir, rhs)
    solver.n_solves += 1
    if max_ref_steps == 0:
        return dir

    dir_temp[:] = dir.vec
    apply_lhs(stepper, solver)
    res.vec -= rhs.vec
    res_norm = np.max(np.abs(res.vec))

    if res_norm > res_norm_cutoff:
        is_prev_slow = False
        prev_res_norm = res_norm
        for _ in range(max_ref_steps):
            syssolver.solve_system(solver, dir, res)
            solver.n_solves += 1
            dir.vec[:] = dir_temp - dir.vec
            apply_lhs(stepper, solver)
            res.vec -= rhs.vec
            res_norm_new = np.max(np.abs(res.vec))
            if res_norm_new >= res_norm:
                dir.vec[:] = dir_temp
                break
            dir_temp[:] = dir.vec
            res_norm = res_norm_new
            if res_norm < res_norm_cutoff:
                break
            is_curr_slow = res_norm > min_impr_tol * prev_res_norm
            if is_prev_slow and is_curr_slow:
                break
            prev_res_norm = res_norm
            is_prev_slow = is_curr_slow

    assert not np.isnan(res_norm)
    solver.worst_dir_res = max(solver.worst_dir_res, res_norm)
    return dir


# ==============================================================================================
# RHS builders (steppers/common.jl:7-118)
# ==============================================================================================
def update_rhs_pred(solver, rhs):   # :7-24
    rhs.x[:] = solver.x_residual
    rhs.y[:] = solver.y_residual
    rhs.z[:] = solver.z_residual
    rhs.tau = solver.tau_residual
    for s_k, d_k in zip(rhs.s_views, solver.point.dual_views):
        s_k[:] = -d_k
    rhs.kap = -solver.point.kap
    return rhs


def update_rhs_predadj(solver, rhs, dir):   # :27-60
    hooks = getattr(solver.model, "dist_hooks", None)
    if hooks is not None:
        return hooks.update_rhs_adj(solver, rhs, dir, "pred")
    rhs.vec[:] = 0
    rteps = np.sqrt(EPS)
    irtrtmu = 1.0 / np.sqrt(np.sqrt(solver.mu))
    for k, cone_k in enumerate(solver.model.cones):
        if not cone_k.use_dder3():
            continue
        H_prim_dir_k = cone_k.vec1
        prim_k_scal = cone_k.vec2
        prim_dir_k = dir.primal_views[k]
        prim_k_scal[:] = irtrtmu * prim_dir_k
        cone_k.hess_prod_slow(H_prim_dir_k, prim_dir_k)
        dder3_k = cone_k.dder3(prim_k_scal)
        dot1 = dder3_k @ cone_k.point
        dot2 = irtrtmu * (prim_k_scal @ H_prim_dir_k)
        dder3_viol = abs(dot1 - dot2) / (rteps + abs(dot2))
        if dder3_viol < 1e-4:
            rhs.s_views[k][:] = H_prim_dir_k + dder3_k
    taubar = solver.point.tau
    tau_dir_tau = dir.tau / taubar
    rhs.kap = tau_dir_tau * solver.mu / taubar * (1 + tau_dir_tau)
    return rhs


def update_rhs_cent(solver, rhs):   # :63-84
    hooks = getattr(solver.model, "dist_hooks", None)
    if hooks is not None:
        return hooks.update_rhs_cent(solver, rhs)
    rhs.x[:] = 0
    rhs.y[:] = 0
    rhs.z[:] = 0
    rhs.tau = 0
    rtmu = np.sqrt(solver.mu)
    for k, cone_k in enumerate(solver.model.cones):
        duals_k = solver.point.dual_views[k]
        grad_k = cone_k.get_grad()
        rhs.s_views[k][:] = -duals_k - rtmu * grad_k
    rhs.kap = -solver.point.kap + solver.mu / solver.point.tau
    return rhs


def update_rhs_centadj(solver, rhs, dir):   # :87-118
    hooks = getattr(solver.model, "dist_hooks", None)
    if hooks is not None:
        return hooks.update_rhs_adj(solver, rhs, dir, "cent")
    rhs.vec[:] = 0
    rteps = np.sqrt(EPS)
    irtrtmu = 1.0 / np.sqrt(np.sqrt(solver.mu))
    for k, cone_k in enumerate(solver.model.cones):
        if not cone_k.use_dder3():
            continue
        H_prim_dir_k_scal = cone_k.vec1
        prim_k_scal = cone_k.vec2
        prim_dir_k = dir.primal_views[k]
        prim_k_scal[:] = irtrtmu * prim_dir_k
        cone_k.hess_prod_slow(H_prim_dir_k_scal, prim_k_scal)
        dder3_k = cone_k.dder3(prim_k_scal)
        dot1 = dder3_k @ cone_k.point
        dot2 = prim_k_scal @ H_prim_dir_k_scal
        dder3_viol = abs(dot1 - dot2) / (rteps + abs(dot2))
        if dder3_viol < 1e-4:
            rhs.s_views[k][:] = dder3_k
    taubar = solver.point.tau
    tau_dir_tau = dir.tau / taubar
    rhs.kap = tau_dir_tau * solver.mu / taubar * tau_dir_tau
    return rhs


# ==============================================================================================
# line search (search.jl)
# ==============================================================================================
DEFAULT_ALPHA_SCHED = [0.9999, 0.999, 0.99, 0.97, 0.95, 0.9, 0.85, 0.8, 0.7, 0.6, 0.5,
                       0.3, 0.1, 0.05, 0.01, 0.005, 0.001, 0.0005]   # search.jl:41-43


class StepSearcher:   # search.jl:8-39
    def __init__(self, model, min_prox=0.01, prox_bound=0.99, use_max_prox=True, alpha_sched=None):
        self.min_prox = min_prox
        self.prox_bound = prox_bound
        self.use_max_prox = use_max_prox
        self.alpha_sched = list(DEFAULT_ALPHA_SCHED if alpha_sched is None else alpha_sched)
        self.szk = np.zeros(len(model.cones))
        self.nup1 = model.nu + 1
        self.prev_sched = 0
        self.prox = 0.0
        self.n_trials = 0


def search_alpha(point, model, stepper, sched=None):   # search.jl:46-69
    searcher = stepper.searcher
    if sched is None:
        sched = stepper.start_sched(searcher)
    sysv = getattr(stepper, "syssolver", None)
    if sysv is not None and _cap(sysv, "search"):
        return sysv.search_alpha_native(model, point, stepper, sched)   # the whole schedule walk in one device call
    while sched <= len(searcher.alpha_sched):
        alpha = searcher.alpha_sched[sched - 1]
        stepper.update_stepper_points(alpha, point, True)
        searcher.n_trials += 1
        if check_cone_points(model, stepper):
            searcher.prev_sched = sched
            return alpha
        sched += 1
    searcher.prev_sched = sched
    return 0.0


def check_cone_points(model, stepper):   # search.jl:74-138
    searcher = stepper.searcher
    cand = stepper.temp
    sysv = getattr(stepper, "syssolver", None)
    if sysv is not None and _cap(sysv, "check"):   # one C-ABI call for the whole test
        return sysv.check_cone_points_native(model, cand, searcher)
    szk = searcher.szk
    cones = model.cones
    min_prox = searcher.min_prox
    use_max_prox = searcher.use_max_prox
    proxsqr_bound = searcher.prox_bound ** 2

    taukap = cand.tau * cand.kap
    if min(cand.tau, cand.kap, taukap) < EPS:
        return False
    for k in range(len(cones)):
        szk[k] = cand.primal_views[k] @ cand.dual_views[k]
        if szk[k] < EPS:
            return False
    mu = (np.sum(szk) + taukap) / searcher.nup1
    if mu < EPS:
        return False
    taukap_rel = taukap / mu
    if taukap_rel < min_prox:
        return False
    taukap_proxsqr = (taukap_rel - 1) ** 2
    if taukap_proxsqr > proxsqr_bound:
        return False
    for k in range(len(cones)):
        nu_k = cones[k].get_nu()
        sz_rel_k = szk[k] / (mu * nu_k)
        if sz_rel_k < min_prox or nu_k * (sz_rel_k - 1) ** 2 > proxsqr_bound:
            return False

    # (the reference visits cones in order of last measured oracle time: affects order only)
    irtmu = 1.0 / np.sqrt(mu)
    hooks = getattr(model, "dist_hooks", None)
    if hooks is not None:      # cone-sharded model: every owner tests its cones at once, two scalar all-reduces
        ok, agg = hooks.check_cones(cand, irtmu, use_max_prox, taukap_proxsqr, proxsqr_bound)
        if ok:
            searcher.prox = np.sqrt(agg)
        return ok
    agg_proxsqr = taukap_proxsqr
    for k in range(len(cones)):
        cone_k = cones[k]
        cone_k.load_point(cand.primal_views[k], irtmu)
        cone_k.load_dual_point(cand.dual_views[k])
        cone_k.reset_data()
        in_prox_k = False
        if cone_k.is_feas() and cone_k.is_dual_feas() and cone_k.check_numerics():
            proxsqr_k = cone_k.get_proxsqr(irtmu, use_max_prox)
            agg_proxsqr = max(agg_proxsqr, proxsqr_k) if use_max_prox else agg_proxsqr + proxsqr_k
            in_prox_k = agg_proxsqr < proxsqr_bound
        if not in_prox_k:
            return False
    searcher.prox = np.sqrt(agg_proxsqr)
    return True


# ==============================================================================================
# CombinedStepper (steppers/combined.jl)
# ==============================================================================================
class CombinedStepper:
    def __init__(self, shift_sched=0, **searcher_options):
        self.shift_sched = shift_sched
        self.searcher_options = searcher_options

    def load(self, solver):   # :35-51
        model = solver.model
        self.prev_alpha = 1.0
        self.rhs = Point(model)
        self.dir = Point(model)
        self.temp = Point(model)
        self.dirs4 = np.zeros((4, self.rhs.vec.shape[0]))  # dir_cent, dir_pred, dir_centadj, dir_predadj: one block the fused call fills
        self.dir_cent = Point(model, storage=self.dirs4[0])
        self.dir_pred = Point(model, storage=self.dirs4[1])
        self.dir_centadj = Point(model, storage=self.dirs4[2])
        self.dir_predadj = Point(model, storage=self.dirs4[3])
        self.dir_temp = np.zeros(self.rhs.vec.shape[0])
        self.rhs2 = np.zeros((2, self.rhs.vec.shape[0]))   # the two right-hand sides / directions of a paired solve
        self.dir2 = np.zeros((2, self.rhs.vec.shape[0]))
        self.searcher = StepSearcher(model, **self.searcher_options)
        self.unadj_only = self.cent_only = False
        self.syssolver = solver.syssolver
        return self

    def step(self, solver):   # :53-120
        point, model = solver.point, solver.model
        rhs, dir = self.rhs, self.dir
        T = time.perf_counter

        sysv = solver.syssolver
        fused = (_cap(sysv, "fused") and model.p == 0 and not _env_on("HYP_NO_PAIR") and not _env_on("HYP_NO_FUSED_STEP"))
        assert fused or not getattr(sysv, "row_local", False), "the row-local sharded driver needs the fused step (distributed.py)"
        if fused:   # update_lhs + the four right-hand sides + the two paired solves: one device call
            lib_t0 = T()
            ok = sysv.step_directions_native(solver, self)
            dt = T() - lib_t0
            up = sysv.last_update_lhs_seconds() if hasattr(sysv, "last_update_lhs_seconds") else 0.0
            solver.time_upsys += up
            solver.time_getdir += dt - up
            if not ok:
                return self._factorization_failed(solver)
        if not fused:
            t0 = T(); solver.syssolver.update_lhs(solver); solver.time_upsys += T() - t0
            if getattr(sysv, "last_info", 0) != 0:
                return self._factorization_failed(solver)

        if fused:
            pass
        elif _cap(sysv, "pair") and not _env_on("HYP_NO_PAIR"):
            # (cent, pred) and (centadj, predadj) are independent pairs: each pair is one device call in which
            # every pass over G, the factor and the cone matrices serves both right-hand sides
            r2, d2 = self.rhs2, self.dir2
            t0 = T()
            update_rhs_cent(solver, rhs); r2[0] = rhs.vec
            update_rhs_pred(solver, rhs); r2[1] = rhs.vec
            solver.time_uprhs += T() - t0
            t0 = T(); self._pair(solver, self.dir_cent, self.dir_pred); solver.time_getdir += T() - t0
            t0 = T()
            update_rhs_centadj(solver, rhs, self.dir_cent); r2[0] = rhs.vec
            update_rhs_predadj(solver, rhs, self.dir_pred); r2[1] = rhs.vec
            solver.time_uprhs += T() - t0
            t0 = T(); self._pair(solver, self.dir_centadj, self.dir_predadj); solver.time_getdir += T() - t0
        else:
            t0 = T(); update_rhs_cent(solver, rhs); solver.time_uprhs += T() - t0
            t0 = T(); get_directions(self, solver); solver.time_getdir += T() - t0
            self.dir_cent.vec[:] = dir.vec
            t0 = T(); update_rhs_centadj(solver, rhs, dir); solver.time_uprhs += T() - t0
            t0 = T(); get_directions(self, solver); solver.time_getdir += T() - t0
            self.dir_centadj.vec[:] = dir.vec

            t0 = T(); update_rhs_pred(solver, rhs); solver.time_uprhs += T() - t0
            t0 = T(); get_directions(self, solver); solver.time_getdir += T() - t0
            self.dir_pred.vec[:] = dir.vec
            t0 = T(); update_rhs_predadj(solver, rhs, dir); solver.time_uprhs += T() - t0
            t0 = T(); get_directions(self, solver); solver.time_getdir += T() - t0
            self.dir_predadj.vec[:] = dir.vec

        self.unadj_only = self.cent_only = False
        try:
            t0 = T(); alpha = search_alpha(point, model, self); solver.time_search += T() - t0
            if alpha == 0:
                self.unadj_only = True
                t0 = T(); alpha = search_alpha(point, model, self); solver.time_search += T() - t0
                if alpha == 0:
                    self.cent_only = True
                    self.unadj_only = False
                    t0 = T(); alpha = search_alpha(point, model, self); solver.time_search += T() - t0
                    if alpha == 0:
                        self.unadj_only = True
                        t0 = T(); alpha = search_alpha(point, model, self); solver.time_search += T() - t0
                        if alpha == 0:
                            solver.status = "NumericalFailure"
                            self.prev_alpha = alpha
                            return False
        finally:
            if fused:
                sysv._dirs_resident = False   # (the point moves below: the device copies of this step's vectors are stale)
        sysv = solver.syssolver
        if getattr(sysv, "row_local", False):
            # cone-sharded solver: this rank's rows of z / s (and tau, kap) are the accepted candidate the library formed and
            # loaded its cones with; x is replicated; the other ranks' rows are never read on this rank
            self.update_stepper_points_x(alpha, point)
            sysv.accept_candidate(point)
            self.prev_alpha = alpha
            return True
        if _cap(sysv, "search") and getattr(sysv, "cand_in_temp", True):
            # the z / tau / s / kap rows ARE the accepted candidate the cones were loaded with (formed natively, same operations):
            # only the x / y rows are formed here
            self.update_stepper_points_x(alpha, point)
            point.ztsk[:] = self.temp.ztsk
        else:
            self.update_stepper_points(alpha, point, False)
        self.prev_alpha = alpha
        return True

    def _factorization_failed(self, solver):
        """Every link of posdef_fact_copy! failed (Cholesky, Bunch-Kaufman, shifted Bunch-Kaufman: dense.jl:194-215).  The
        reference warns (qrchol.jl:253-255) and carries on with the unusable factorization: its directions are not finite, no
        step of the schedule passes check_cone_points and step() ends in NumericalFailure (combined.jl:97-117).  The device keeps
        no failed factorization to solve with, so the same outcome is reported directly -- without re-running the assembly."""
        solver.status = "NumericalFailure"
        self.prev_alpha = 0.0
        return False

    def _pair(self, solver, dir_a, dir_b):
        (ra, rb), ns = solver.syssolver.get_directions2_native(solver, self.dir2, self.rhs2)
        dir_a.vec[:] = self.dir2[0]
        dir_b.vec[:] = self.dir2[1]
        solver.n_solves += ns
        assert not (np.isnan(ra) or np.isnan(rb))
        if solver.max_ref_steps > 0:
            solver.worst_dir_res = max(solver.worst_dir_res, ra, rb)

    def update_stepper_points_x(self, alpha, point):   # the x / y entries of :124-170 (same operation order per entry)
        k = point.n + point.p
        cand = point.vec[:k]
        sel = lambda pt: pt.vec[:k]
        dir_cent, dir_pred = sel(self.dir_cent), sel(self.dir_pred)
        if self.unadj_only:
            if self.cent_only:
                cand += alpha * dir_cent
            else:
                cand += alpha * dir_pred + (1 - alpha) * dir_cent
        else:
            dir_centadj = sel(self.dir_centadj)
            alpha_sqr = alpha ** 2
            if self.cent_only:
                cand += alpha * dir_cent + alpha_sqr * dir_centadj
            else:
                dir_predadj = sel(self.dir_predadj)
                alpha_m1 = 1 - alpha
                alpha_m1sqr = alpha_m1 ** 2
                cand += (alpha * dir_pred + alpha_sqr * dir_predadj + alpha_m1 * dir_cent
                         + alpha_m1sqr * dir_centadj)

    def update_stepper_points(self, alpha, point, ztsk_only):   # :124-170
        if ztsk_only:
            cand = self.temp.ztsk
            cand[:] = point.ztsk
            sel = lambda pt: pt.ztsk
        else:
            cand = point.vec
            sel = lambda pt: pt.vec
        dir_cent, dir_pred = sel(self.dir_cent), sel(self.dir_pred)
        if self.unadj_only:
            if self.cent_only:
                cand += alpha * dir_cent
            else:
                cand += alpha * dir_pred + (1 - alpha) * dir_cent
        else:
            dir_centadj = sel(self.dir_centadj)
            alpha_sqr = alpha ** 2
            if self.cent_only:
                cand += alpha * dir_cent + alpha_sqr * dir_centadj
            else:
                dir_predadj = sel(self.dir_predadj)
                alpha_m1 = 1 - alpha
                alpha_m1sqr = alpha_m1 ** 2
                cand += (alpha * dir_pred + alpha_sqr * dir_predadj + alpha_m1 * dir_cent
                         + alpha_m1sqr * dir_centadj)

    def start_sched(self, searcher):   # :172-175
        if self.shift_sched <= 0:
            return 1
        return max(1, searcher.prev_sched - self.shift_sched)

    def step_name(self):
        if self.cent_only:
            return "cent" if self.unadj_only else "ce-a"
        return "comb" if self.unadj_only else "co-a"


# ==============================================================================================
# Solver (Solvers.jl)
# ==============================================================================================
class Solver:
    def __init__(self, verbose=False, iter_limit=1000, time_limit=np.inf, tol_rel_opt=None, tol_abs_opt=None,
                 tol_feas=None, tol_infeas=None, tol_illposed=None, default_tol_power=None,
                 default_tol_relax=None, tol_slow=1e-3, preprocess=True, reduce=True, rescale=True,
                 init_use_indirect=False, init_tol_qr=1000 * EPS, stepper=None, syssolver=None):   # Solvers.jl:162-240
        if reduce:
            assert preprocess
        if default_tol_power is None:
            default_tol_power = 0.5
        loose = EPS ** default_tol_power
        tight = EPS ** (1.5 * default_tol_power)
        if default_tol_relax is not None:
            loose *= default_tol_relax
            tight *= default_tol_relax
        self.verbose = verbose
        self.iter_limit = iter_limit
        self.time_limit = time_limit
        self.tol_rel_opt = loose if tol_rel_opt is None else tol_rel_opt
        self.tol_abs_opt = tight if tol_abs_opt is None else tol_abs_opt
        self.tol_feas = loose if tol_feas is None else tol_feas
        self.tol_infeas = tight if tol_infeas is None else tol_infeas
        self.tol_illposed = tight / 100 if tol_illposed is None else tol_illposed
        self.tol_slow = tol_slow
        self.preprocess = preprocess
        self.reduce = reduce
        self.rescale = rescale
        self.init_tol_qr = init_tol_qr
        self.init_use_indirect = init_use_indirect   # Solvers.jl:176: LSQR initial x instead of the pivoted QR of [A; G]
        self.stepper = stepper if stepper is not None else CombinedStepper()
        self.syssolver = syssolver if syssolver is not None else QRCholDenseSystemSolver()
        self.status = "NotLoaded"
        self.iter_callback = None
        self._setup_only = False

    def load(self, model):   # Solvers.jl:566-571
        self.orig_model = model
        self.status = "Loaded"
        return self

    # ------------------------------------------------------------------------------------------
    def solve(self):   # Solvers.jl:245-416
        assert self.status == "Loaded"
        self.status = "SolveCalled"
        start_time = time.perf_counter()
        self.num_iters = 0
        self.n_solves = 0
        for f in ("rescale", "initx", "inity", "unproc", "loadsys", "upsys", "upfact", "uprhs", "getdir", "search"):
            setattr(self, "time_" + f, 0.0)
        self.res_norm_cutoff = 0.0
        self.max_ref_steps = 5
        nan = float("nan")
        self.x_norm_res_t = self.y_norm_res_t = self.z_norm_res_t = nan
        self.x_norm_res = self.y_norm_res = self.z_norm_res = nan
        self.primal_obj_t = self.dual_obj_t = self.primal_obj = self.dual_obj = self.gap = nan
        self.x_feas = self.y_feas = self.z_feas = self.tau_feas = nan

        om = self.orig_model
        self.result = Point(om)
        model = self.model = om.copy()
        init_z, init_s = initialize_cone_point(om)

        t0 = time.perf_counter()
        self.used_rescaling = rescale_data(self)
        self.time_rescale = time.perf_counter() - t0

        self.Ap_Q = None
        self.Ap_R = np.zeros((0, 0))
        if self.reduce:
            t0 = time.perf_counter(); init_y = find_initial_y(self, init_z, True); self.time_inity = time.perf_counter() - t0
            t0 = time.perf_counter(); init_x = find_initial_x(self, init_s); self.time_initx = time.perf_counter() - t0
        else:
            t0 = time.perf_counter(); init_x = find_initial_x(self, init_s); self.time_initx = time.perf_counter() - t0
            t0 = time.perf_counter(); init_y = find_initial_y(self, init_z, False); self.time_inity = time.perf_counter() - t0

        if self.status == "SolveCalled":
            model = self.model
            point = self.point = Point(model)
            point.x[:] = init_x
            point.y[:] = init_y
            point.z[:] = init_z
            point.s[:] = init_s
            point.tau = 1.0
            point.kap = 1.0
            self.calc_mu()
            for k, cone in enumerate(model.cones):
                cone.load_point(point.primal_views[k])
                cone.load_dual_point(point.dual_views[k])

            self.x_residual = np.zeros(model.n)
            self.y_residual = np.zeros(model.p)
            self.z_residual = np.zeros(model.q)
            self.tau_residual = 0.0
            self.x_conv_tol = 1.0 / (1 + _norm_inf(model.c))
            self.y_conv_tol = 1.0 / (1 + _norm_inf(model.b))
            self.z_conv_tol = 1.0 / (1 + _norm_inf(model.h))
            self.prev_is_slow = self.prev2_is_slow = False
            self.worst_dir_res = 0.0

            stepper = self.stepper
            stepper.load(self)
            t0 = time.perf_counter(); self.syssolver.load(self); self.time_loadsys = time.perf_counter() - t0
            if self.verbose:
                self.print_header()

            self._start_time = start_time
            self._initial_point_vec = point.vec.copy()
            if self._setup_only:
                return self
            self.iter_start_time = time.perf_counter()
            with _blas_limit():
                while self.iterate():
                    pass
            self.iter_time = time.perf_counter() - self.iter_start_time

            t0 = time.perf_counter(); postprocess(self); self.time_unproc = time.perf_counter() - t0

        self.solve_time = time.perf_counter() - start_time
        if self.verbose:
            print(f"\nstatus is {self.status} after {self.num_iters} iterations and {self.solve_time:.3f} seconds\n")
        return self

    def setup(self):
        """everything of `solve` up to (not including) the iteration loop: preprocessing, initial point,
        stepper and system-solver load.  Then drive with `iterate()` (used by bench.py)."""
        self._setup_only = True
        try:
            self.solve()
        finally:
            self._setup_only = False
        return self

    def iterate(self):
        """one pass of the `while true` body of Solvers.solve (Solvers.jl:340-398); False when it stops."""
        with _blas_limit():
            return self._iterate()

    def _iterate(self):
        point, stepper = self.point, self.stepper
        improv = self.calc_convergence_params()
        if self.verbose:
            self.print_iteration()
        if self.iter_callback is not None:
            self.iter_callback(self)
        if self.check_convergence():
            return False
        if self.num_iters == self.iter_limit:
            self.status = "IterationLimit"
            return False
        timed_out = time.perf_counter() - self._start_time >= self.time_limit
        agree = getattr(self.syssolver, "agree_any", None)
        if agree is not None and np.isfinite(self.time_limit):
            timed_out = agree(timed_out)   # multi-GPU: a wall-clock decision is taken by all ranks together or by none
        if timed_out:
            self.status = "TimeLimit"
            return False
        if improv < self.tol_slow:
            if self.prev_is_slow and self.prev2_is_slow:
                self.status = "SlowProgress"
                return False
            self.prev2_is_slow = self.prev_is_slow
            self.prev_is_slow = True
        else:
            self.prev2_is_slow = self.prev_is_slow
            self.prev_is_slow = False

        self.res_norm_cutoff = 1e-4 * max(self.x_norm_res, self.y_norm_res, self.z_norm_res, self.tau_feas)
        self.worst_dir_res = 0.0

        if not stepper.step(self):
            return False
        self.calc_mu()
        if min(point.tau, point.kap, self.mu) <= 0:
            self.status = "NumericalFailure"
            return False
        self.num_iters += 1
        return True

    def reset_iterate(self):
        """put the solver back at its initial iterate (bench.py: keep stepping after convergence)."""
        model, point = self.model, self.point
        point.vec[:] = self._initial_point_vec
        self.calc_mu()
        for k, cone in enumerate(model.cones):
            cone.reset_data()
            cone.load_point(point.primal_views[k])
            cone.load_dual_point(point.dual_views[k])
            assert cone.is_feas()
            cone.get_grad()
        self.status = "SolveCalled"
        self.x_feas = self.y_feas = self.z_feas = self.tau_feas = float("nan")
        self.prev_is_slow = self.prev2_is_slow = False
        self.stepper.prev_alpha = 1.0
        self.stepper.searcher.prev_sched = 0

    def calc_mu(self):   # :418-423
        pt = self.point
        sysv = self.syssolver
        if getattr(sysv, "row_local", False):
            # cone-sharded solver: z / s hold this rank's rows; one device call forms G'z, G x + s, h'z and z's for the point
            # (sums over ranks inside the library) -- calc_convergence_params of the next pass reads the same record
            rp = self._row_products = sysv.residual_products(pt)
            self.mu = (rp["zs"] + pt.tau * pt.kap) / (self.model.nu + 1)
            return self.mu
        self.mu = (pt.z @ pt.s + pt.tau * pt.kap) / (self.model.nu + 1)
        return self.mu

    def _calc_convergence_params_row_local(self):   # :425-483 on a cone-sharded solver
        model, point, sysv = self.model, self.point, self.syssolver
        tau = point.tau
        rp = getattr(self, "_row_products", None)
        if rp is None or rp["version"] != sysv.point_version(point):
            rp = sysv.residual_products(point)
        self._row_products = None
        rows = sysv.rsl
        xr = rp["Gtz"]
        self.x_norm_res_t = _norm_inf(xr)
        xr = xr + model.c * tau
        self.x_norm_res = _norm_inf(xr) / tau
        self.x_residual[:] = -xr
        x_feas = self.x_norm_res * self.x_conv_tol
        self.y_norm_res_t = self.y_norm_res = 0.0
        y_feas = 0.0
        zr = rp["Gx_s"]                        # this rank's rows of G x + s
        zr = zr - model.h[rows] * tau
        self.z_residual[rows] = zr
        # (the two norms over ALL ranks' rows came back with the sums of the same exchange: no collective of the host's own here)
        self.z_norm_res_t = float(rp["zn_t"])
        self.z_norm_res = float(rp["zn"]) / tau
        z_feas = self.z_norm_res * self.z_conv_tol
        self.primal_obj_t = model.c @ point.x
        self.dual_obj_t = -rp["hz"]
        self.tau_residual = self.primal_obj_t - self.dual_obj_t + point.kap
        tau_feas = abs(self.tau_residual)
        improv = 0.0
        for curr, prev in ((x_feas, self.x_feas), (y_feas, self.y_feas), (z_feas, self.z_feas), (tau_feas, self.tau_feas)):
            if np.isnan(prev) or np.isnan(curr):
                continue
            improv = max(improv, (prev - curr) / (abs(prev) + EPS))
        self.x_feas, self.y_feas, self.z_feas, self.tau_feas = x_feas, y_feas, z_feas, tau_feas
        self.primal_obj = self.primal_obj_t / tau + model.obj_offset
        self.dual_obj = self.dual_obj_t / tau + model.obj_offset
        self.gap = rp["zs"]
        return improv

    def calc_convergence_params(self):   # :425-483
        if getattr(self.syssolver, "row_local", False):
            return self._calc_convergence_params_row_local()
        model, point = self.model, self.point
        tau = point.tau
        rp = self.syssolver.residual_products(point) if getattr(self.syssolver, "one_pass_residual_products", False) else None
        xr = rp["Gtz"] if rp is not None else self.syssolver.mul_G(True, point.z)
        if model.p:
            xr = xr + model.A.T @ point.y
        self.x_norm_res_t = _norm_inf(xr)
        xr = xr + model.c * tau
        self.x_norm_res = _norm_inf(xr) / tau
        self.x_residual[:] = -xr
        x_feas = self.x_norm_res * self.x_conv_tol

        yr = model.A @ point.x if model.p else np.zeros(0)
        self.y_norm_res_t = _norm_inf(yr)
        yr = yr - model.b * tau
        self.y_norm_res = _norm_inf(yr) / tau
        self.y_residual[:] = yr
        y_feas = self.y_norm_res * self.y_conv_tol

        zr = rp["Gx_s"] if rp is not None else self.syssolver.mul_G(False, point.x) + point.s
        self.z_norm_res_t = _norm_inf(zr)
        zr = zr - model.h * tau
        self.z_norm_res = _norm_inf(zr) / tau
        self.z_residual[:] = zr
        z_feas = self.z_norm_res * self.z_conv_tol

        self.primal_obj_t = model.c @ point.x
        self.dual_obj_t = -(model.b @ point.y) - model.h @ point.z
        self.tau_residual = self.primal_obj_t - self.dual_obj_t + point.kap
        tau_feas = abs(self.tau_residual)

        improv = 0.0
        for curr, prev in ((x_feas, self.x_feas), (y_feas, self.y_feas), (z_feas, self.z_feas), (tau_feas, self.tau_feas)):
            if np.isnan(prev) or np.isnan(curr):
                continue
            improv = max(improv, (prev - curr) / (abs(prev) + EPS))
        self.x_feas, self.y_feas, self.z_feas, self.tau_feas = x_feas, y_feas, z_feas, tau_feas
        self.primal_obj = self.primal_obj_t / tau + model.obj_offset
        self.dual_obj = self.dual_obj_t / tau + model.obj_offset
        self.gap = point.z @ point.s
        return improv

    def check_convergence(self):   # :485-528
        tau = self.point.tau
        p_t, d_t = self.primal_obj_t, self.dual_obj_t
        is_feas = max(self.x_feas, self.y_feas, self.z_feas) <= self.tol_feas
        is_abs_opt = self.gap <= self.tol_abs_opt
        is_rel_opt = min(self.gap / tau, abs(p_t - d_t)) <= self.tol_rel_opt * max(tau, min(abs(p_t), abs(d_t)))
        if is_feas and (is_abs_opt or is_rel_opt):
            self.status = "Optimal"
            return True
        if d_t > EPS and self.x_norm_res_t <= self.tol_infeas * d_t:
            self.status = "PrimalInfeasible"
            self.primal_obj, self.dual_obj = p_t, d_t
            return True
        if p_t < -EPS and max(self.y_norm_res_t, self.z_norm_res_t) <= self.tol_infeas * -p_t:
            self.status = "DualInfeasible"
            self.primal_obj, self.dual_obj = p_t, d_t
            return True
        if self.mu <= self.tol_illposed and tau <= self.tol_illposed * min(1.0, self.point.kap):
            self.status = "IllPosed"
            return True
        return False

    # getters (Solvers.jl:550-564)
    def get_status(self): return self.status
    def get_num_iters(self): return self.num_iters
    def get_primal_obj(self): return self.primal_obj
    def get_dual_obj(self): return self.dual_obj
    def get_s(self): return self.result.s.copy()
    def get_z(self): return self.result.z.copy()
    def get_x(self): return self.result.x.copy()
    def get_y(self): return self.result.y.copy()
    def get_tau(self): return self.point.tau
    def get_kappa(self): return self.point.kap
    def get_mu(self): return self.mu

    def print_header(self):   # :587-600
        print(f"\n{'iter':>5} {'p_obj':>12} {'d_obj':>12} |{'abs_gap':>9} {'x_feas':>9} {'y_feas':>9} {'z_feas':>9} "
              f"|{'tau':>9} {'kap':>9} {'mu':>9} |{'dir_res':>8} {'prox':>8} {'step':>5} {'alpha':>9}")

    def print_iteration(self):   # :603-619
        s = (f"{self.num_iters:5d} {self.primal_obj:12.4e} {self.dual_obj:12.4e} |{self.gap:9.2e} {self.x_feas:9.2e} "
             f"{self.y_feas:9.2e} {self.z_feas:9.2e} |{self.point.tau:9.2e} {self.point.kap:9.2e} {self.mu:9.2e} |")
        if self.num_iters:
            s += f"{self.worst_dir_res:8.1e} {self.stepper.searcher.prox:8.1e} {self.stepper.step_name():>5} {self.stepper.prev_alpha:9.2e}"
        print(s, flush=True)


def _norm_inf(v):
    return float(np.max(np.abs(v))) if v.shape[0] else 0.0


def initialize_cone_point(model):   # Solvers.jl:530-548
    init_z = np.zeros(model.q)
    init_s = np.zeros(model.q)
    for cone, idxs in zip(model.cones, model.cone_idxs):
        cone.setup_data()
        primal_k = (init_z if cone.use_dual_barrier() else init_s)[idxs]
        dual_k = (init_s if cone.use_dual_barrier() else init_z)[idxs]
        cone.set_initial_point(primal_k)
        cone.load_point(primal_k)
        assert cone.is_feas()
        g = cone.get_grad()
        dual_k[:] = -g
        cone.load_dual_point(dual_k)
        assert cone.is_dual_feas()
    return init_z, init_s


# ==============================================================================================
# process.jl
# ==============================================================================================
def rescale_data(solver):   # process.jl:13-60
    if not solver.rescale:
        return False
    model = solver.model
    if getattr(model, "comm", None) is not None:   # cone-sharded model: maxima of G reduced over ranks
        from .distributed import rescale_data_dist
        return rescale_data_dist(solver)
    c, A, b, G, h = model.c, model.A, model.b, model.G, model.h
    minval = np.sqrt(EPS)

    def maxabsmin(v):
        return max(minval, float(np.max(np.abs(v)))) if v.size else minval

    c_scale = np.array([np.sqrt(max(abs(c[j]), maxabsmin(A[:, j]), maxabsmin(G[:, j]))) for j in range(model.n)])
    b_scale = np.array([np.sqrt(max(abs(b[i]), maxabsmin(A[i, :]))) for i in range(model.p)])
    h_scale = np.ones(model.q)
    for k, cone in enumerate(model.cones):
        idxs = model.cone_idxs[k]
        if getattr(cone, "is_nonnegative", False):
            for i in range(idxs.start, idxs.stop):
                h_scale[i] = np.sqrt(max(abs(h[i]), maxabsmin(G[i, :])))
        else:
            h_scale[idxs] = np.sqrt(max(maxabsmin(h[idxs]), maxabsmin(G[idxs, :])))
    solver.c_scale, solver.b_scale, solver.h_scale = c_scale, b_scale, h_scale
    model.c = c / c_scale
    model.A = (A / c_scale[None, :]) / b_scale[:, None] if model.p else A / c_scale[None, :]
    model.G = (G / c_scale[None, :]) / h_scale[:, None]
    model.b = b / b_scale
    model.h = h / h_scale
    return True


def get_rank_est(R, tol):   # process.jl:373-382
    d = np.abs(np.diagonal(R))
    return int(np.sum(d > tol))


def find_initial_x(solver, init_s):   # process.jl:64-178
    if solver.status != "SolveCalled":
        return np.zeros(0)
    model = solver.model
    if getattr(model, "comm", None) is not None:   # cone-sharded model: matrix-free least squares (LSQR)
        from .distributed import find_initial_x_dist
        return find_initial_x_dist(solver, init_s)
    n, p, q = model.n, model.p, model.q
    if n == 0:
        solver.x_keep_idxs = np.zeros(0, dtype=int)
        return np.zeros(0)
    A, G = model.A, model.G
    solver.x_keep_idxs = np.arange(n)
    rhs = np.concatenate([model.b, model.h - init_s])
    if solver.init_use_indirect:   # process.jl:81-95 (IterativeSolvers.lsqr on [A; G])
        from .distributed import lsqr

        class _AG:
            class model:   # lsqr only needs .model.n, .mul, .mul_t
                pass

            def mul(self, x):
                return np.concatenate([A @ x, G @ x]) if p else G @ x

            def mul_t(self, z):
                return (A.T @ z[:p] + G.T @ z[p:]) if p else G.T @ z
        op = _AG()
        op.model.n = n
        return lsqr(op, rhs)
    AG = G.copy() if p == 0 else np.vstack([A, G])
    # Large [A; G]: the reference's own algorithm on the device -- column-pivoted Householder QR with dgeqp3's semantics
    # (hyp_qrcp_factor: pivots, R and the rank decision as LAPACK's, which is what Julia's qr!(AG, ColumnNorm()) calls); the
    # right-hand side rides along as an extra column, so Q' rhs comes with the factorization.  HYP_INITX_DEVICE=0: host LAPACK;
    # HYP_INITX_DEVICE=normal: the conditioning-gated normal-equations shortcut of round 1 (a different algorithm, opt-in).
    mode = os.environ.get("HYP_INITX_DEVICE", "1")
    if (p + q) * n * n >= 2e9 and mode not in ("0", "normal"):
        init_x = _find_initial_x_device_qr(solver, model, AG, rhs)
        if init_x is not None:
            return init_x
    # Large, clearly full-rank [A; G]: the least-squares x from the device (Cholesky of AG'AG + one corrected
    # semi-normal-equations step) instead of the host's pivoted QR, which is 2 (p + q) n^2 flops (11 s of 12 at config 2).
    # Taken only when the estimated sigma_min / sigma_max is far above the rank-decision threshold of get_rank_est, where
    # the pivoted QR would report full rank too and return the same x up to rounding; otherwise the reference's path below.
    if (p + q) * n * n >= 2e10 and solver.preprocess and mode == "normal":
        from . import _lib as L
        import ctypes
        AGf = np.asfortranarray(AG)
        xs, rc, info = np.zeros(n), ctypes.c_double(0.0), ctypes.c_int(-1)
        L.check(L.lib().hyp_dense_lstsq_normal(L.ctx(), p + q, n, AGf.ctypes.data_as(ctypes.c_void_p), p + q, L.vec_ptr(np.ascontiguousarray(rhs)),
                                               L.vec_ptr(xs), ctypes.byref(rc), ctypes.byref(info)), "hyp_dense_lstsq_normal")
        if info.value == 0 and rc.value > 1e-3:
            return xs
    Qf, R, piv = sla.qr(AG, mode="economic", pivoting=True, overwrite_a=True)   # Q: (p+q) x n
    AG_rank = get_rank_est(R, solver.init_tol_qr)

    if (not solver.preprocess) or AG_rank == n:
        # init_x = AG_fact \ rhs
        qtb = Qf.T @ rhs
        r = min(AG_rank, n)
        xs = np.zeros(n)
        xs[:r] = sla.solve_triangular(R[:r, :r], qtb[:r], lower=False)
        init_x = np.zeros(n)
        init_x[piv] = xs
        return init_x

    x_keep_idxs = piv[:AG_rank]
    AG_R = R[:AG_rank, :AG_rank]
    c_sub = model.c[x_keep_idxs]
    yz_sub = Qf[:, :AG_rank] @ sla.solve_triangular(AG_R, c_sub, trans="T", lower=False)
    residual = _norm_inf(A.T @ yz_sub[:p] + G.T @ yz_sub[p:] - model.c)
    if residual > solver.init_tol_qr:
        solver.status = "DualInconsistent"
        return np.zeros(0)
    model.c = c_sub
    model.A = A[:, x_keep_idxs]
    model.G = np.ascontiguousarray(G[:, x_keep_idxs])
    model.n = AG_rank
    solver.x_keep_idxs = x_keep_idxs
    temp = Qf.T @ np.concatenate([model.b, model.h - init_s])
    init_x = sla.solve_triangular(AG_R, temp[:model.n], lower=False)
    return init_x


class DeviceQRCP:
    """column-pivoted QR of a tall dense matrix on the device (hyp_qrcp_*): LAPACK dgeqp3's pivots / R / reflectors"""

    def __init__(self, M, rhs=None):
        from . import _lib as L
        import ctypes
        self.L, self.ct = L, ctypes
        Mf = np.asfortranarray(M, dtype=np.float64)
        self.m, self.n = Mf.shape
        h = ctypes.c_void_p()
        rp = L.vec_ptr(np.ascontiguousarray(rhs, dtype=np.float64)) if rhs is not None else None
        L.check(L.lib().hyp_qrcp_factor(L.ctx(), self.m, self.n, Mf.ctypes.data_as(ctypes.c_void_p), self.m, rp, ctypes.byref(h)), "hyp_qrcp_factor")
        self._h = h
        self.has_rhs = rhs is not None

    def get(self, want_R=True):
        L, ct = self.L, self.ct
        r = min(self.m, self.n)
        piv = np.zeros(self.n, dtype=np.int32)
        R = np.zeros((r, self.n), order="F") if want_R else None
        rdiag = np.zeros(r)
        qtb = np.zeros(self.m) if self.has_rhs else None
        L.check(L.lib().hyp_qrcp_get(self._h, piv.ctypes.data_as(ct.c_void_p), R.ctypes.data_as(ct.c_void_p) if want_R else None,
                                     L.vec_ptr(rdiag), L.vec_ptr(qtb) if qtb is not None else None), "hyp_qrcp_get")
        return piv.astype(int), (np.triu(R) if want_R else None), rdiag, qtb

    def apply_q(self, vec, trans):
        v = np.ascontiguousarray(vec, dtype=np.float64).copy()
        self.L.check(self.L.lib().hyp_qrcp_apply_q(self._h, int(bool(trans)), self.L.vec_ptr(v)), "hyp_qrcp_apply_q")
        return v

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self.L._lib is not None:
                self.L._lib.hyp_qrcp_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _find_initial_x_device_qr(solver, model, AG, rhs):   # process.jl:64-178 with the factorization on the device
    n, p = model.n, model.p
    A, G = model.A, model.G
    fact = DeviceQRCP(AG, rhs)
    piv, R, rdiag, qtb = fact.get(want_R=True)
    AG_rank = int(np.sum(np.abs(rdiag) > solver.init_tol_qr))          # get_rank_est (process.jl:373-382)
    if (not solver.preprocess) or AG_rank == n:
        r = min(AG_rank, n)
        xs = np.zeros(n)
        xs[:r] = sla.solve_triangular(R[:r, :r], qtb[:r], lower=False)
        init_x = np.zeros(n)
        init_x[piv] = xs
        return init_x
    x_keep_idxs = piv[:AG_rank]
    AG_R = R[:AG_rank, :AG_rank]
    c_sub = model.c[x_keep_idxs]
    w = np.zeros(AG.shape[0])
    w[:AG_rank] = sla.solve_triangular(AG_R, c_sub, trans="T", lower=False)
    yz_sub = fact.apply_q(w, trans=False)                                 # Q[:, 1:rank] (R1' \ c_sub)
    residual = _norm_inf(A.T @ yz_sub[:p] + G.T @ yz_sub[p:] - model.c)
    if residual > solver.init_tol_qr:
        solver.status = "DualInconsistent"
        return np.zeros(0)
    model.c = c_sub
    model.A = A[:, x_keep_idxs]
    model.G = np.ascontiguousarray(G[:, x_keep_idxs])
    model.n = AG_rank
    solver.x_keep_idxs = x_keep_idxs
    return sla.solve_triangular(AG_R, qtb[:AG_rank], lower=False)


def find_initial_y(solver, init_z, reduce):   # process.jl:182-365
    if solver.status != "SolveCalled":
        return np.zeros(0)
    model = solver.model
    p = model.p
    if p == 0:
        solver.y_keep_idxs = np.zeros(0, dtype=int)
        solver.Ap_R = np.zeros((0, 0))
        solver.Ap_Q = None   # I
        return np.zeros(0)
    n, q = model.n, model.q
    A = model.A
    solver.y_keep_idxs = np.arange(p)

    Qf, R, piv = sla.qr(A.T.copy(), mode="full", pivoting=True)
    Ap_rank = get_rank_est(R, solver.init_tol_qr)

    if (not reduce) and (not solver.preprocess):
        rhs = -model.c - model.G.T @ init_z
        qtb = Qf.T @ rhs
        ys = np.zeros(p)
        ys[:Ap_rank] = sla.solve_triangular(R[:Ap_rank, :Ap_rank], qtb[:Ap_rank], lower=False)
        init_y = np.zeros(p)
        init_y[piv] = ys
        return init_y

    Ap_R = R[:Ap_rank, :Ap_rank]
    y_keep_idxs = piv[:Ap_rank]
    Ap_Q = Qf
    b_sub = model.b[y_keep_idxs]
    if Ap_rank < p:
        x_sub = np.zeros(n)
        x_sub[:Ap_rank] = sla.solve_triangular(Ap_R, b_sub, trans="T", lower=False)
        x_sub = Ap_Q @ x_sub
        residual = _norm_inf(A @ x_sub - model.b)
        if residual > solver.init_tol_qr:
            solver.status = "PrimalInconsistent"
            return np.zeros(0)

    if reduce:
        solver.reduce_row_piv_inv = np.zeros(0, dtype=int)
        cQ = model.c @ Ap_Q
        cQ1 = solver.reduce_cQ1 = cQ[:Ap_rank]
        cQ2 = cQ[Ap_rank:]
        model.c = cQ2
        model.n = cQ2.shape[0]
        Rpib0 = solver.reduce_Rpib0 = sla.solve_triangular(Ap_R, b_sub, trans="T", lower=False)
        model.obj_offset += cQ1 @ Rpib0
        GQ = model.G @ Ap_Q
        GQ1 = solver.reduce_GQ1 = GQ[:, :Ap_rank]
        GQ2 = GQ[:, Ap_rank:]
        model.h = model.h - GQ1 @ Rpib0
        model.G = np.ascontiguousarray(GQ2)
        model.p = 0
        model.A = np.zeros((0, model.n))
        model.b = np.zeros(0)
        solver.reduce_Ap_R = Ap_R
        solver.reduce_Ap_Q = Ap_Q
        solver.reduce_y_keep_idxs = y_keep_idxs
        solver.Ap_R = np.zeros((0, 0))
        solver.Ap_Q = None
        return np.zeros(0)

    temp = Ap_Q.T @ (model.c + model.G.T @ init_z)
    init_y = -temp[:Ap_rank]
    init_y = sla.solve_triangular(Ap_R, init_y, lower=False)
    model.A = A[y_keep_idxs, :]
    model.b = b_sub
    model.p = Ap_rank
    solver.y_keep_idxs = y_keep_idxs
    solver.Ap_R = Ap_R
    solver.Ap_Q = Ap_Q
    return init_y


def postprocess(solver):   # process.jl:385-458
    point, result = solver.point, solver.result
    if getattr(solver.syssolver, "row_local", False):
        solver.syssolver.gather_rows(point)   # cone-sharded solver: every rank gets all rows of z and s, once, for the result
    om = solver.orig_model
    if solver.status in ("PrimalInfeasible", "DualInfeasible"):
        tau = 1.0
    else:
        tau = point.tau
        if tau <= 0:
            result.vec[:] = np.nan
            return
    result.s[:] = point.s / tau
    result.z[:] = point.z / tau

    if solver.preprocess and om.n != 0 and not np.any(np.isnan(point.x)):
        if solver.reduce and om.p != 0:
            xa = np.zeros(om.n - solver.reduce_Rpib0.shape[0])
            xa[solver.x_keep_idxs] = point.x / tau
            if solver.status in ("PrimalInfeasible", "DualInfeasible"):
                Rpib0 = np.zeros(solver.reduce_Rpib0.shape[0])
            else:
                Rpib0 = solver.reduce_Rpib0
            xb = solver.reduce_Ap_Q @ np.concatenate([Rpib0, xa])
            result.x[:] = xb
        else:
            result.x[solver.x_keep_idxs] = point.x / tau
    else:
        result.x[:] = point.x / tau

    if solver.preprocess and om.p != 0 and not np.any(np.isnan(point.y)):
        if solver.reduce:
            ya = solver.reduce_GQ1.T @ result.z
            if solver.status not in ("PrimalInfeasible", "DualInfeasible"):
                ya = ya + solver.reduce_cQ1
            k = solver.reduce_y_keep_idxs.shape[0]
            ya[:k] = sla.solve_triangular(solver.reduce_Ap_R, ya[:k], lower=False)
            result.y[solver.reduce_y_keep_idxs] = -ya
        else:
            result.y[solver.y_keep_idxs] = point.y / tau
    else:
        result.y[:] = point.y / tau

    if solver.used_rescaling:
        result.s *= solver.h_scale
        result.z /= solver.h_scale
        result.y /= solver.b_scale
        result.x /= solver.c_scale
